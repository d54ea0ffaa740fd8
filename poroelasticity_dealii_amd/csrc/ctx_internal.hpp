// Declarations shared by the host-logic translation units of the HIP back end (ctx*.hip); not part of the C-ABI.
#pragma once
#include <rccl/rccl.h>
#include <functional>
#include <string>
#include <vector>
#include "common.hpp"

namespace poro {
namespace ctx_detail {

extern thread_local std::string g_err;
inline int ipow(int b, int e) { int r = 1; while (e--) r *= b; return r; }

// ---- RCCL, resolved at run time so single-GPU use has no dependency on it (ctx_comm.hip) ----------------------------------------
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  void load();
};
extern Rccl g_rccl;
#define PORO_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw poro::Error(std::string(#x) + " -> " + poro::ctx_detail::g_rccl.GetErrorString(r_)); } while (0)

// ---- timing: HIP events on the launch stream around kernel families, drawn from a per-context pool (ctx_comm.hip) ---------------
hipEvent_t event_get(poro_ctx *c);
struct Timed {
  poro_ctx *c; Timer *t = nullptr; hipEvent_t a = nullptr, b = nullptr;
  Timed(poro_ctx *c_, const char *name);
  ~Timed();
};
void timers_collect(poro_ctx *c);
void isolate_sampled_dispatch(poro_ctx *c);
// a start / stop event pair that is returned to the context's pool on every exit path
struct EventPair {
  poro_ctx *c; hipEvent_t e0, e1;
  explicit EventPair(poro_ctx *c_) : c(c_), e0(event_get(c_)), e1(event_get(c_)) {}
  ~EventPair() { c->event_pool.push_back(e0); c->event_pool.push_back(e1); }
  EventPair(const EventPair &) = delete; EventPair &operator=(const EventPair &) = delete;
};
// device -> host scalars through the pinned mailbox (no copy engine, no stream synchronisation)
void post_and_wait(poro_ctx *c, const double *dev_src, int n, const PcgScalars *sc = nullptr);

// ---- communication (ctx_comm.hip) ---------------------------------------------------------------------------------------------------
void setup_general_partition(poro_ctx *c, const poro_desc *d);
void exchange_planes(poro_ctx *c, const double *send_lo, const double *send_hi, int64_t plane);
void exchange_add(poro_ctx *c, double *v, int64_t n, int64_t plane);
void allreduce_sum(poro_ctx *c, double *dev, int n);
int64_t owned(poro_ctx *c, int64_t n, int64_t plane);
AsmArgs asm_args(poro_ctx *c);
MfArgs mf_args(poro_ctx *c);
void mf_operator(poro_ctx *c, const double *x, double *y, bool constrained);
double *vec(poro_ctx *c, int which);
int64_t vec_len(poro_ctx *c, int which);
bool is_u_vec(int which);
bool apply_A_u(poro_ctx *c, const double *x, double *y, int mode, double *dot_partials = nullptr, bool fix_rows = true, const PcgScalars *pcg_state = nullptr,
               bool exchange = true /* false: leave the rank's partial product (the caller folds constrained rows first) */);

// ---- Krylov drivers (ctx_pcg.hip) -----------------------------------------------------------------------------------------------------
typedef std::function<bool(const double *, double *, double *)> ApplyFn;
int pcg(poro_ctx *c, const ApplyFn &apply, int64_t n, int64_t plane, double *x, const double *b, const DiagVec &diag, double *g, double *d, double *h, const poro_solver_opts *opts,
        poro_solve_info *info, const ApplyFn *precond = nullptr, int *its_hint = nullptr, bool precond_gated = false, const FdmOct *oct = nullptr);
double dot_host(poro_ctx *c, const double *a, const double *b, int64_t n);
int pcg_ssor(poro_ctx *c, CsrDev &A, const double *val, double *x, const double *b, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info);
int pcg_ilu0(poro_ctx *c, CsrDev &A, const double *val, DevBuf<double> &lu, bool &valid, double *x, const double *b, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info);
double estimate_lmax_u(poro_ctx *c, const ApplyFn &apply, const DiagVec &dj);

// ---- fast-diagonalisation preconditioners (ctx_prec.hip) ------------------------------------------------------------------------------
bool fdm_p_supported(poro_ctx *c);
void build_fdm_p(poro_ctx *c);
void alltoall_blocks(poro_ctx *c, double *send, double *recv, int64_t blk, bool self_in_place = false /* the caller has already put its own block into recv */);
void setup_two_level(poro_ctx *c, const poro_desc *d);                // uploads P and its transpose (poro_desc.coarse)
bool two_level_supported(poro_ctx *c);
bool two_level_supported_p(poro_ctx *c);
void two_level_precondition_p(poro_ctx *c, double a, double kappa, const double *dinv, const double *g, double *z, double omega);   // z = omega D^-1 g + P (a M_H + kappa K_H)^-1 P^T g
void two_level_precondition_u(poro_ctx *c, const double *g, double *z, double omega);   // z = omega D^-1 g + P B_H^-1 P^T g
void fdm_precondition_u_slab(poro_ctx *c, const double *g_quadrant, double *z_quadrant, const PcgScalars *gate);
void fdm_precondition_p(poro_ctx *c, double a, const double k[3], const double *g, double *z);
void analyse_fdm_u(poro_ctx *c);
void build_fdm_u(poro_ctx *c);
void fdm_precondition_u(poro_ctx *c, const double *g, double *z);

// ---- set-up (ctx_setup.hip) -------------------------------------------------------------------------------------------------------------
void setup(poro_ctx *c, const poro_desc *d);
void sync_source_vector(poro_ctx *c);

}  // namespace ctx_detail
}  // namespace poro
