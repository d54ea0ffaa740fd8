// extern "C" entry points of include/poroel_hip.h (the drop-in boundary); the host logic behind them lives in ctx_setup / ctx_comm / ctx_pcg / ctx_prec.hip.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <unordered_map>
#include "common.hpp"
#include "ctx_internal.hpp"

using namespace poro;
using namespace poro::ctx_detail;

namespace {
template <class F> int guarded(F &&f) {
  try { return f(); }
  catch (const std::exception &e) { g_err = e.what(); return -1; }
}

}  // namespace

// ======================================= extern "C" ================================================================
extern "C" {

const char *poro_last_error(void) { return g_err.c_str(); }
int poro_abi_version(void) { return PORO_ABI_VERSION; }

int poro_ctx_create(const poro_desc *desc, int device, int operator_mode, poro_ctx **out) {
  return guarded([&] {
    if (!desc || !out) throw Error("null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw Error("no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= ndev) throw Error("device index out of range");
    PORO_HIP(hipSetDevice(device));
    std::unique_ptr<poro_ctx> c(new poro_ctx());
    c->device = device; c->operator_mode = operator_mode;
    PORO_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    kron_prepare_device();   // function attributes are per device: opt in to the large dynamic LDS on THIS one
    { void *mbp = nullptr; PORO_HIP(hipHostMalloc(&mbp, sizeof(Mailbox), hipHostMallocDefault)); std::memset(mbp, 0, sizeof(Mailbox)); c->mailbox = static_cast<Mailbox *>(mbp); }
    setup(c.get(), desc);
    if (desc->coarse.enabled) {
      // two-level preconditioner: the underlying uniform box becomes a context of its own on this context's stream
      if (!desc->coarse.box_problem || !desc->coarse.ptr || !desc->coarse.node || !desc->coarse.weight) throw Error("poro_desc.coarse: box_problem / ptr / node / weight missing");
      if (!desc->coarse.box_problem->box.enabled || desc->coarse.box_problem->coarse.enabled) throw Error("poro_desc.coarse.box_problem must be a uniform box (box.enabled) without a coarse space of its own");
      if (desc->coarse.box_problem->dim != desc->dim || desc->coarse.box_problem->degree_u != desc->degree_u) throw Error("poro_desc.coarse.box_problem: dimension / degree differ");
      if (!c->interleaved_u || c->comm.multi()) throw Error("poro_desc.coarse needs node-interleaved displacement dofs on one rank");
      poro_ctx *box = nullptr;
      if (poro_ctx_create(desc->coarse.box_problem, device, PORO_OP_MATRIX_FREE, &box) != 0) throw Error(std::string("poro_desc.coarse.box_problem: ") + poro_last_error());
      PORO_HIP(hipStreamSynchronize(box->stream)); (void)hipStreamDestroy(box->stream); box->stream = c->stream; box->borrowed_stream = true;
      c->two_level.box = box;
      setup_two_level(c.get(), desc);
    }
    *out = c.release();
    return 0;
  });
}

void poro_ctx_destroy(poro_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  timers_collect(c);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  c->event_pool.clear();
  if (c->comm.nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)c->comm.nccl_comm);
  if (c->two_level.box) { poro_ctx_destroy(c->two_level.box); c->two_level.box = nullptr; }
  if (c->stream && !c->borrowed_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int poro_comm_unique_id(void *id128) {
  return guarded([&] { g_rccl.load(); ncclUniqueId id; PORO_NCCL(g_rccl.GetUniqueId(&id)); static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId"); std::memcpy(id128, &id, 128); return 0; });
}
int poro_ctx_comm_init_rccl(poro_ctx *c, const void *id128) {
  return guarded([&] {
    g_rccl.load(); PORO_HIP(hipSetDevice(c->device));
    ncclUniqueId id; std::memcpy(&id, id128, 128); ncclComm_t comm;
    PORO_NCCL(g_rccl.CommInitRank(&comm, c->comm.part.n_ranks, id, c->comm.part.rank));
    c->comm.nccl_comm = comm;
    // self-test of the data plane on the compute stream: an all-reduce of a known value and one grouped neighbour exchange
    // (to itself when there is a single rank), so a broken RCCL set-up fails here with a message instead of inside a solve
    const int nr = c->comm.part.n_ranks, rk = c->comm.part.rank;
    DevBuf<double> t; t.alloc(4);
    double h[4] = {1.0 + rk, 2.0, 100.0 + rk, -1.0};
    PORO_HIP(hipMemcpyAsync(t.p, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    PORO_NCCL(g_rccl.AllReduce(t.p, t.p, 2, ncclFloat64, ncclSum, comm, c->stream));
    const int up = (rk + 1) % nr, down = (rk + nr - 1) % nr;
    PORO_NCCL(g_rccl.GroupStart());
    PORO_NCCL(g_rccl.Send(t.p + 2, 1, ncclFloat64, up, comm, c->stream));
    PORO_NCCL(g_rccl.Recv(t.p + 3, 1, ncclFloat64, down, comm, c->stream));
    PORO_NCCL(g_rccl.GroupEnd());
    PORO_HIP(hipMemcpyAsync(h, t.p, sizeof(h), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
    if (h[0] != 0.5 * nr * (nr + 1) || h[1] != 2.0 * nr || h[3] != 100.0 + down) throw Error("RCCL self-test failed (all-reduce / send-recv returned wrong data)");
    // the two exchange patterns of the solver, on one double each: the bidirectional neighbour exchange of exchange_add and the
    // all-to-all of the partitioned fast diagonalisation
    {
      DevBuf<double> sb, rb; sb.alloc(nr + 2); rb.alloc(nr + 2);
      std::vector<double> hs(nr + 2), hr(nr + 2, -1.0);
      for (int q = 0; q < nr; ++q) hs[q] = 1000.0 * rk + q;
      hs[nr] = 7.0 + rk; hs[nr + 1] = 9.0 + rk;
      PORO_HIP(hipMemcpyAsync(sb.p, hs.data(), (nr + 2) * sizeof(double), hipMemcpyHostToDevice, c->stream));
      PORO_HIP(hipMemcpyAsync(rb.p, hr.data(), (nr + 2) * sizeof(double), hipMemcpyHostToDevice, c->stream));
      const bool has_up = rk + 1 < nr, has_dn = rk > 0;
      PORO_NCCL(g_rccl.GroupStart());
      if (has_up) { PORO_NCCL(g_rccl.Send(sb.p + nr, 1, ncclFloat64, rk + 1, comm, c->stream)); PORO_NCCL(g_rccl.Recv(rb.p + nr, 1, ncclFloat64, rk + 1, comm, c->stream)); }
      if (has_dn) { PORO_NCCL(g_rccl.Send(sb.p + nr + 1, 1, ncclFloat64, rk - 1, comm, c->stream)); PORO_NCCL(g_rccl.Recv(rb.p + nr + 1, 1, ncclFloat64, rk - 1, comm, c->stream)); }
      PORO_NCCL(g_rccl.GroupEnd());
      PORO_NCCL(g_rccl.GroupStart());
      for (int q = 0; q < nr; ++q) if (q != rk) { PORO_NCCL(g_rccl.Send(sb.p + q, 1, ncclFloat64, q, comm, c->stream)); PORO_NCCL(g_rccl.Recv(rb.p + q, 1, ncclFloat64, q, comm, c->stream)); }
      PORO_NCCL(g_rccl.GroupEnd());
      PORO_HIP(hipMemcpyAsync(hr.data(), rb.p, (nr + 2) * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
      bool ok = true;
      if (has_up && hr[nr] != 9.0 + (rk + 1)) ok = false;          // the upper neighbour's "down" message
      if (has_dn && hr[nr + 1] != 7.0 + (rk - 1)) ok = false;      // the lower neighbour's "up" message
      for (int q = 0; q < nr; ++q) if (q != rk && hr[q] != 1000.0 * q + rk) ok = false;
      if (!ok) throw Error("RCCL self-test failed (neighbour exchange / all-to-all returned wrong data)");
    }
    return 0;
  });
}
int poro_ctx_comm_init_callbacks(poro_ctx *c, poro_allreduce_fn ar, poro_sendrecv_fn sr, void *user) { c->comm.ar = ar; c->comm.sr = sr; c->comm.user = user; return 0; }

int poro_ctx_synchronize(poro_ctx *c) { return guarded([&] { PORO_HIP(hipSetDevice(c->device)); PORO_HIP(hipStreamSynchronize(c->stream)); return 0; }); }

int poro_vec_set(poro_ctx *c, int which, const double *host, int64_t n) {
  return guarded([&] { PORO_HIP(hipSetDevice(c->device)); if (vec_len(c, which) != n) throw Error("vector length mismatch"); PORO_HIP(hipMemcpyAsync(vec(c, which), host, n * sizeof(double), hipMemcpyHostToDevice, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream)); return 0; });
}
int poro_vec_get(poro_ctx *c, int which, double *host, int64_t n) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device)); if (vec_len(c, which) != n) throw Error("vector length mismatch");
    if (which == PORO_VEC_SOURCE_P) sync_source_vector(c);
    if (which == PORO_VEC_DIAG_U) la_copy(c->stream, vec(c, which), c->diag_u.p ? c->diag_u.p : c->diag_u_local.p, n);
    PORO_HIP(hipMemcpyAsync(host, vec(c, which), n * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream)); return 0;
  });
}
int poro_vec_fill(poro_ctx *c, int which, double v) { return guarded([&] { PORO_HIP(hipSetDevice(c->device)); la_fill(c->stream, vec(c, which), v, vec_len(c, which)); return 0; }); }
int poro_vec_copy(poro_ctx *c, int dst, int src) {
  return guarded([&] { PORO_HIP(hipSetDevice(c->device)); if (vec_len(c, dst) != vec_len(c, src)) throw Error("vector length mismatch"); la_copy(c->stream, vec(c, dst), vec(c, src), vec_len(c, dst)); return 0; });
}
int poro_vec_axpy(poro_ctx *c, int y, double a, int x) {
  return guarded([&] { PORO_HIP(hipSetDevice(c->device)); if (vec_len(c, y) != vec_len(c, x)) throw Error("vector length mismatch"); la_axpy(c->stream, vec(c, y), a, vec(c, x), vec_len(c, y)); return 0; });
}
int poro_vec_norm(poro_ctx *c, int which, double *l2, double *linf) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    const int64_t n = vec_len(c, which), plane = is_u_vec(which) ? c->comm.part.plane_u : c->comm.part.plane_p;
    la_norm_partials(c->stream, vec(c, which), owned(c, n, plane), c->partials.p, c->partials.p + kMaxPartials);
    la_reduce_finish(c->stream, c->partials.p, 2, c->red.p, 2);
    allreduce_sum(c, c->red.p, 1);   // linf stays rank-local under a partition (reporting only, PoroelasticityFSS.h:387-389)
    post_and_wait(c, c->red.p, 2);
    if (l2) *l2 = std::sqrt(c->mailbox->vals[0]); if (linf) *linf = c->mailbox->vals[1]; return 0;
  });
}

int poro_state_save(poro_ctx *c) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    std::vector<double *> dst; std::vector<const double *> src; std::vector<int64_t> len;
    for (auto &kv : c->vec) { DevBuf<double> &d = c->vec_saved[kv.first]; if (d.n != kv.second.n) d.alloc(kv.second.n); dst.push_back(d.p); src.push_back(kv.second.p); len.push_back((int64_t)kv.second.n); }
    la_copy_many(c->stream, (int)dst.size(), dst.data(), src.data(), len.data());
    PORO_HIP(hipStreamSynchronize(c->stream)); return 0;
  });
}
int poro_state_restore(poro_ctx *c) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (c->vec_saved.empty()) throw Error("state_restore without state_save");
    std::vector<double *> dst; std::vector<const double *> src; std::vector<int64_t> len;
    for (auto &kv : c->vec_saved) { dst.push_back(c->vec.at(kv.first).p); src.push_back(kv.second.p); len.push_back((int64_t)kv.second.n); }
    la_copy_many(c->stream, (int)dst.size(), dst.data(), src.data(), len.data());      // (one launch instead of two dozen)
    // the solves that follow repeat earlier ones: forget the iteration-count history, so that a measurement of a repeated step cannot profit from a perfect prediction
    for (int *h : {c->pcg_hint_u, c->pcg_hint_fdm_u, c->pcg_hint_cheb_u, c->pcg_hint_p, c->pcg_hint_proj}) h[0] = h[1] = 0;
    return 0;
  });
}

int poro_disp_assemble_system(poro_ctx *c, int rebuild_matrix) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream; const AsmArgs a = asm_args(c);
    if (rebuild_matrix || !c->matrix_built) {
      Timed tm(c, "assemble_u_matrix");
      c->lift_u.zero(s); c->neumann_u.zero(s);
      if (c->operator_mode == PORO_OP_CSR && c->box_asm) {
        // uniform box: one element matrix, every CSR entry written once by its owner (kernels_box.hip); the lifting -(A_full g) comes
        // from the unconstrained structured operator
        if (!c->Ke.p) c->Ke.alloc((size_t)c->dpc_u * c->dpc_u);
        asm_u_element_matrix(s, a, 0, c->Ke.p);
        box_asm_u_matrix(s, c->dim, c->k_u, c->box, c->Ke.p, c->Au, c->dir_mask.p, c->Au_val.p);
        if (!c->box_asm_checked && c->Au.nnz <= 60000000 && !std::getenv("PORO_DIAG_SKIP_SELFCHECK")) {   // once, against the coloured per-cell assembly
          DevBuf<double> ref, lift_ref; ref.alloc(c->Au.nnz); ref.zero(s); lift_ref.alloc(c->n_u); lift_ref.zero(s);
          for (size_t k = 0; k + 1 < c->color_off.size(); ++k)
            asm_u_matrix(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], c->Au.rp.p, c->Au.col.p, ref.p, lift_ref.p);
          la_axpy(s, ref.p, -1.0, c->Au_val.p, c->Au.nnz);
          la_norm_partials(s, ref.p, c->Au.nnz, c->partials.p, c->partials.p + kMaxPartials); la_norm_partials(s, c->Au_val.p, c->Au.nnz, c->partials.p + 2 * kMaxPartials, c->partials.p + 3 * kMaxPartials);
          la_reduce_finish(s, c->partials.p, 4, c->red.p, 2 | 8);
          double h[4]; PORO_HIP(hipMemcpyAsync(h, c->red.p, sizeof(h), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s));
          if (!(h[1] <= 1e-12 * h[3])) throw Error("structured CSR assembly disagrees with the per-cell assembly: max diff " + std::to_string(h[1]) + " vs max " + std::to_string(h[3]));
        }
        c->box_asm_checked = true;
        mf_operator(c, c->dir_val.p, c->wh_u.p, false);
        la_fill(s, c->lift_u.p, 0.0, c->n_u); la_axpy(s, c->lift_u.p, -1.0, c->wh_u.p, c->n_u);
        la_csr_diag(s, c->Au, c->Au_val.p, c->diag_u_local.p);
      } else if (c->operator_mode == PORO_OP_CSR) {
        c->Au_val.zero(s);
        for (size_t k = 0; k + 1 < c->color_off.size(); ++k)
          asm_u_matrix(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], c->Au.rp.p, c->Au.col.p, c->Au_val.p, c->lift_u.p);
        la_csr_diag(s, c->Au, c->Au_val.p, c->diag_u_local.p);
      } else if (!c->box.enabled) {
        // general mesh, matrix-free: the diagonal and the lifting -(A_full g) come from the same quadrature-level cell loop
        mfg_apply(s, a, c->color_cells.p, c->color_off, c->n_u, nullptr, c->diag_u_local.p, false, 1);
        mfg_apply(s, a, c->color_cells.p, c->color_off, c->n_u, c->dir_val.p, c->wh_u.p, false, 0);
        la_fill(s, c->lift_u.p, 0.0, c->n_u); la_axpy(s, c->lift_u.p, -1.0, c->wh_u.p, c->n_u);
      } else {
        asm_u_element_matrix(s, a, 0, c->Ke.p);
        mf_diag(s, mf_args(c), c->diag_u_local.p);
        // lifting: -(A_full g) on the free rows, through the unconstrained operator
        mf_operator(c, c->dir_val.p, c->wh_u.p, false);
        if (c->mf_variant == 1 && kron_supported(c->dim, c->k_u) && !std::getenv("PORO_DIAG_SKIP_SELFCHECK")) {
          // self-check of the sum-factorised operator against the element-matrix gather on a synthetic vector (guards the
          // FE-table / numbering assumptions of the structured path); both unconstrained
          std::vector<double> hx(c->n_u); for (int64_t i = 0; i < c->n_u; ++i) hx[i] = std::sin(0.37 * (double)i);
          DevBuf<double> tx, t1, t2; tx.upload(hx); t1.alloc(c->n_u); t2.alloc(c->n_u);
          kron_apply(s, mf_args(c), tx.p, t1.p, false, c->n_cus); mf_apply(s, mf_args(c), tx.p, t2.p, false);
          la_axpy(s, t1.p, -1.0, t2.p, c->n_u);
          la_norm_partials(s, t1.p, c->n_u, c->partials.p, c->partials.p + kMaxPartials); la_norm_partials(s, t2.p, c->n_u, c->partials.p + 2 * kMaxPartials, c->partials.p + 3 * kMaxPartials);
          la_reduce_finish(s, c->partials.p, 4, c->red.p, 2 | 8);
          double h[4]; PORO_HIP(hipMemcpyAsync(h, c->red.p, sizeof(h), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s));
          if (!(h[1] <= 1e-11 * h[3])) throw Error("sum-factorised operator disagrees with the element-matrix operator: max diff " + std::to_string(h[1]) + " vs max " + std::to_string(h[3]));
        }
        la_fill(s, c->lift_u.p, 0.0, c->n_u); la_axpy(s, c->lift_u.p, -1.0, c->wh_u.p, c->n_u);
      }
      asm_u_neumann(s, a, c->n_bfaces, c->bface_cell.p, c->bface_local.p, c->bface_id.p, c->n_neumann, c->neu_label.p, c->neu_comp.p, c->neu_val.p, c->neumann_u.p);
      if (!c->diag_u.p) c->diag_u.alloc(c->n_u);
      la_copy(s, c->diag_u.p, c->diag_u_local.p, c->n_u);
      exchange_add(c, c->diag_u.p, c->n_u, c->comm.part.plane_u);
      // dictionary form of the Jacobi diagonal: on a uniform box only a few dozen distinct per-node triples exist, so the PCG kernels
      // can read one class byte per node and a tiny table instead of 8 bytes per dof.  Built by de-duplicating the actual values.
      c->diag_u_cls.release(); c->diag_u_tab.release();
      if (!c->dinv_u.p) c->dinv_u.alloc(c->n_u);
      la_reciprocal(s, c->dinv_u.p, c->diag_u.p, c->n_u);
      la_mask_zero(s, c->dinv_u.p, c->cons_u.inert.p, c->n_u);                   // zero reciprocal = inert (Dirichlet or hanging) dof (DiagVec)
      if (c->operator_mode == PORO_OP_MATRIX_FREE && c->box.enabled) {
        std::vector<double> hd(c->n_u);
        PORO_HIP(hipMemcpyAsync(hd.data(), c->dinv_u.p, c->n_u * sizeof(double), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s));
        const int nc = c->dim; const int64_t nnode = c->n_u / nc;
        struct KeyHash { size_t operator()(const std::array<double, 3> &k) const { uint64_t h = 1469598103934665603ull; for (double v : k) { uint64_t b; std::memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; h ^= h >> 29; } return (size_t)h; } };
        std::unordered_map<std::array<double, 3>, int, KeyHash> dict; std::vector<uint8_t> cls(nnode); std::vector<double> tab; bool ok = true;
        for (int64_t nd = 0; nd < nnode && ok; ++nd) {
          std::array<double, 3> key{0, 0, 0}; for (int k = 0; k < nc; ++k) key[k] = hd[nd * nc + k];
          auto it = dict.find(key);
          if (it == dict.end()) { if (dict.size() >= 255) { ok = false; break; } it = dict.emplace(key, (int)dict.size()).first; for (int k = 0; k < nc; ++k) tab.push_back(key[k]); }
          cls[nd] = (uint8_t)it->second;
        }
        if (ok) { c->diag_u_cls.upload(cls); c->diag_u_tab.upload(tab); }
      }
      c->matrix_built = true; c->ilu_u_valid = false; c->cheb_lmax = 0;
    }
    {
      Timed tm(c, "assemble_u_rhs");
      double *rhs = vec(c, PORO_VEC_RHS_U);
      if (c->box_asm) box_rhs_u(s, c->dim, c->box_cpl, c->mat.biot_alpha, vec(c, PORO_VEC_P), c->lift_u.p, c->neumann_u.p, c->dir_mask.p, rhs);
      else {
        la_fill(s, rhs, 0.0, c->n_u);                                            // rhs_vector = 0 (:204)
        for (size_t k = 0; k + 1 < c->color_off.size(); ++k)
          asm_u_rhs(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], vec(c, PORO_VEC_P), rhs);
        la_rhs_u_finish(s, rhs, c->lift_u.p, c->neumann_u.p, c->dir_mask.p, c->n_u);
      }
    }
    if (c->cons_u.n) {
      // condensed right-hand side C^T (b - A x_inh), x_inh = the constraints' inhomogeneities (distribute_local_to_global, :280-286).  On a partition everything here is
      // the rank's PARTIAL vector: rows are folded into their masters first, the interface sums come last (a rank may hold a master without holding the constrained dof)
      double *rhs = vec(c, PORO_VEC_RHS_U);
      if (c->cons_u.any_inhom) {
        la_fill(s, c->wd_u.p, 0.0, c->n_u); la_cons_expand(s, c->cons_u, c->wd_u.p, true);
        if (c->operator_mode == PORO_OP_CSR) la_csr_spmv(s, c->Au, c->Au_val.p, c->wd_u.p, c->wh_u.p);
        else mf_operator(c, c->wd_u.p, c->wh_u.p, true);
        la_mask_zero(s, c->wh_u.p, c->dir_mask.p, c->n_u);
        la_axpy(s, rhs, -1.0, c->wh_u.p, c->n_u);
      }
      la_cons_reduce(s, c->cons_u, rhs);
    }
    exchange_add(c, vec(c, PORO_VEC_RHS_U), c->n_u, c->comm.part.plane_u);
    // stream-ordered: the right-hand side is consumed by kernels of the same stream (a caller that wants the host to wait calls poro_ctx_synchronize)
    return 0;
  });
}

int poro_supports_preconditioner(poro_ctx *c, int32_t which_system, int32_t prec) {
  if (!c) return 0;
  if (prec == PORO_PREC_NONE || prec == PORO_PREC_JACOBI) return 1;
  if (prec == PORO_PREC_TWO_LEVEL) return which_system == 0 ? two_level_supported(c) : (two_level_supported_p(c) && !c->n_pdir);
  if (which_system == 0 && c->cons_u.n) return prec == PORO_PREC_CHEBYSHEV;      // condensed operators exist at operator level only: Jacobi, the polynomial built on it, the two-level form above
  if (which_system == 1 && (c->cons_p.n || c->n_pdir)) return 0;
  if (prec == PORO_PREC_SSOR || prec == PORO_PREC_ILU0) return !c->comm.multi() && (which_system == 1 || c->operator_mode == PORO_OP_CSR);
  if (prec == PORO_PREC_CHEBYSHEV) return which_system == 0;
  if (prec == PORO_PREC_FDM && which_system == 1) return fdm_p_supported(c);
  if (prec == PORO_PREC_FDM) { analyse_fdm_u(c); return c->fdm_u_state == 1; }
  return 0;
}
int poro_disp_solve(poro_ctx *c, const poro_solver_opts *opts, poro_solve_info *info) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (!c->matrix_built) throw Error("disp_solve before disp_assemble_system");
    const int mode = c->operator_mode;
    if (c->cons_u.n && opts->preconditioner != PORO_PREC_JACOBI && opts->preconditioner != PORO_PREC_NONE && opts->preconditioner != PORO_PREC_CHEBYSHEV && opts->preconditioner != PORO_PREC_TWO_LEVEL)
      throw Error("meshes with constraint lists: PORO_PREC_JACOBI / CHEBYSHEV / TWO_LEVEL / NONE only (the operator is condensed on the fly)");
    if (opts->preconditioner == PORO_PREC_ILU0) {
      if (mode != PORO_OP_CSR) throw Error("PORO_PREC_ILU0 needs the assembled CSR operator");
      const int rc = pcg_ilu0(c, c->Au, c->Au_val.p, c->ilu_u, c->ilu_u_valid, vec(c, PORO_VEC_U), vec(c, PORO_VEC_RHS_U), c->wg_u.p, c->wd_u.p, c->wh_u.p, opts, info);
      la_set_constrained(c->stream, vec(c, PORO_VEC_U), c->dir_mask.p, c->dir_val.p, c->n_u);
      PORO_HIP(hipStreamSynchronize(c->stream));
      return rc;
    }
    if (opts->preconditioner == PORO_PREC_SSOR) {
      if (mode != PORO_OP_CSR) throw Error("PORO_PREC_SSOR needs the assembled CSR operator");
      const int rc = pcg_ssor(c, c->Au, c->Au_val.p, vec(c, PORO_VEC_U), vec(c, PORO_VEC_RHS_U), c->wg_u.p, c->wd_u.p, c->wh_u.p, opts, info);
      la_set_constrained(c->stream, vec(c, PORO_VEC_U), c->dir_mask.p, c->dir_val.p, c->n_u);
      PORO_HIP(hipStreamSynchronize(c->stream));
      return rc;
    }
    const std::function<bool(const double *, double *, double *)> apply = [&](const double *x, double *y, double *dp) {
      if (!c->cons_u.n) return apply_A_u(c, x, y, mode, dp, false, dp ? c->scal.p : nullptr);
      // C^T A C: the search direction's hanging entries follow their masters, the product's hanging rows fold into the masters' rows
      la_cons_expand(c->stream, c->cons_u, const_cast<double *>(x), false);
      apply_A_u(c, x, y, mode, nullptr, false, nullptr, false);                 // the rank's partial product ...
      la_cons_reduce(c->stream, c->cons_u, y);                                   // ... folded ...
      exchange_add(c, y, c->n_u, c->comm.part.plane_u);                          // ... then summed over the interface
      return false;
    };
    if (opts->preconditioner == PORO_PREC_CHEBYSHEV) {
      // z = q(D^-1 A) D^-1 g with the Chebyshev polynomial q of degree m for the interval [lambda_max / ratio, lambda_max]: m operator applications
      // without dot products; on 3D boxes (one rank) the recurrence runs inside the structured operator kernel
      int m = opts->poly_degree > 0 ? opts->poly_degree : 6;
      DiagVec dj; dj.full = c->dinv_u.p; dj.ncomp = c->dim; dj.inert = c->cons_u.inert.p;
      if (c->diag_u_cls.p) { dj.cls = c->diag_u_cls.p; dj.tab = c->diag_u_tab.p; }
      // lambda_max(D^-1 A): on a uniform box all cells share one element matrix and lambda_max <= lambda_max(diag(K_e)^-1 K_e) holds rigorously
      // (x^T A x = sum_e x_e^T K_e x_e <= mu sum_e x_e^T diag(K_e) x_e = mu x^T D x) but is loose (3.8 against 2.5 for Q2 hexahedra), so the working
      // value is the Lanczos estimate (+5 %) capped by it.  Only EVEN degrees are used: should an eigenvalue still exceed the assumed bound, it meets
      // T_{m+1} outside [-1, 1], and q(lambda) lambda stays positive (the preconditioner SPD) exactly when m + 1 is odd
      const bool have_bound = c->box.enabled && c->Ke.p && !c->cons_u.n;
      if (m & 1) ++m;
      if (!(c->cheb_lmax > 0)) {
        if (have_bound) {
          std::vector<double> ke((size_t)c->dpc_u * c->dpc_u);
          PORO_HIP(hipMemcpyAsync(ke.data(), c->Ke.p, ke.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
          c->cheb_lmax = std::min(jacobi_scaled_lambda_max(c->dpc_u, ke), estimate_lmax_u(c, apply, dj));   // the element bound is rigorous but loose
        } else c->cheb_lmax = estimate_lmax_u(c, apply, dj); if (std::getenv("PORO_CHEB_VERBOSE")) std::fprintf(stderr, "[poro] lambda_max(D^-1 A_u) ~ %.6f\n", c->cheb_lmax); }
      // `omega` doubles as the interval ratio; anything below 4 (the SSOR relaxation 1.2 a caller may have left there, 0) means "default"
      double ratio = opts->omega;
      if (!(ratio >= 4.0)) {   // default: a few times lambda_min, which scales with h^2 (calibrated on box runs of 8^3 .. 72^3 cells)
        if (!(c->cheb_ratio_default > 0)) {
          // from GLOBAL mesh sizes, so that every rank of a partitioned run builds the same polynomial (rank-local sizes gave uneven slabs different roots on
          // either side of a shared plane): the cell layers of the partitioned direction (slabs) / the cell count (general partitions) are summed over the ranks
          double h[2] = {(double)c->box.n[c->dim - 1], (double)c->n_cells};
          if (c->comm.multi()) {
            PORO_HIP(hipMemcpyAsync(c->red.p, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
            allreduce_sum(c, c->red.p, 2);
            PORO_HIP(hipMemcpyAsync(h, c->red.p, sizeof(h), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
          }
          double nmax = 1;
          if (c->box.enabled) { for (int k = 0; k < c->dim; ++k) nmax = std::max(nmax, k == c->dim - 1 ? h[0] : (double)c->box.n[k]); }
          else nmax = std::round(std::pow(h[1], 1.0 / c->dim));
          c->cheb_ratio_default = std::min(400.0, std::max(10.0, (c->k_u == 2 ? 0.2 : 0.05) * nmax * nmax));
        }
        ratio = c->cheb_ratio_default;
      }
      // root form: the residual polynomial of degree m + 1 is prod_i (1 - lambda / r_i) with the roots r_i of the Chebyshev polynomial shifted to
      // [lambda_max / ratio, lambda_max]; z_1 = D^-1 g / r_0, z_{j+1} = z_j + D^-1 (g - A z_j) / r_j.  Same polynomial as the three-term recurrence
      // (identical CG iteration counts in the prototype for every ordering at these degrees) with ONE extra stream per step (g) instead of two;
      // the roots are taken alternately from both ends so that no run of small roots inflates the intermediate iterates
      const double lmax = c->cheb_lmax, lmin = lmax / ratio, theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
      std::vector<double> roots;
      { std::vector<double> r(m + 1); for (int i = 0; i <= m; ++i) r[i] = theta - delta * std::cos(3.14159265358979323846 * (2 * i + 1) / (2.0 * (m + 1)));
        int lo = 0, hi = m; while (lo <= hi) { roots.push_back(r[hi--]); if (lo <= hi) roots.push_back(r[lo++]); } }
      if (!c->cheb_z.p) { c->cheb_z.alloc(c->n_u); c->cheb_z.zero(c->stream); c->cheb_t.alloc(c->n_u); c->cheb_t.zero(c->stream); }
      if (!c->wz_u.p) { c->wz_u.alloc(c->n_u); c->wz_u.zero(c->stream); }
      const bool fusable = mode == PORO_OP_MATRIX_FREE && c->mf_variant == 1 && c->box.enabled && kron_supported(c->dim, c->k_u) && c->diag_u_cls.p && !c->cons_u.n && !std::getenv("PORO_CHEB_UNFUSED");
      const bool fuse = fusable && !c->comm.multi();
      // slab partitions (3D): the fused kernel runs on every rank with its LOCAL partial product; on the two shared node planes it also leaves the raw partial, the neighbours
      // swap those planes and a plane-sized kernel redoes the update there with the complete sum (the same two numbers on both ranks: bitwise equal copies)
      const bool fuse_multi = fusable && c->comm.multi() && !c->comm.general && c->dim == 3;
      if (fuse_multi && c->cheb_side_lo.n < (size_t)c->comm.part.plane_u) { c->cheb_side_lo.alloc(c->comm.part.plane_u); c->cheb_side_hi.alloc(c->comm.part.plane_u); }
      const int64_t n_own = owned(c, c->n_u, c->comm.part.plane_u);
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *gz_partials) {
        Timed tm(c, "precondition_u_chebyshev");
        hipStream_t s = c->stream;
        double *X[2] = {(m % 2 == 0) ? z : c->cheb_z.p, (m % 2 == 0) ? c->cheb_z.p : z};   // z_{j+1} lands in X[j & 1]; the last one (j = m) in z
        if (!gz_partials && !c->cheb_z1_ready) la_cheb_first(s, X[0], g, dj, 1.0 / roots[0], c->n_u);   // inside the iteration z_1 = D^-1 g / r_0 was stored by the residual update (DiagVec::z1_out)
        c->cheb_z1_ready = false;
        bool dot_done = false;
        // (the device-side "solve finished" flag may only gate launches inside the iteration: before pcg_scalars_start it still holds the previous solve's state)
        const PcgScalars *pstate = gz_partials ? c->scal.p : nullptr;
        for (int j = 1; j <= m; ++j) {
          const double omega = 1.0 / roots[j];
          double *zj = X[(j - 1) & 1], *zn = X[j & 1];
          const bool last = j == m;
          if (fuse_multi) {
            const poro_partition &pt = c->comm.part; const int64_t plane = pt.plane_u;
            KronCheb kc; kc.g = g; kc.znew = zn; kc.omega = omega; kc.cls = c->diag_u_cls.p; kc.tab = c->diag_u_tab.p;
            kc.side_lo = pt.has_lower ? c->cheb_side_lo.p : nullptr; kc.side_hi = pt.has_upper ? c->cheb_side_hi.p : nullptr;
            if (c->timing && c->timers["apply_u_chebyshev_fused"].sample(c->timing_stride)) { isolate_sampled_dispatch(c); Timer &t = c->timers["apply_u_chebyshev_fused"]; hipEvent_t e0 = event_get(c), e1 = event_get(c);
                                                                                           (void)kron_apply(s, mf_args(c), zj, nullptr, true, c->n_cus, nullptr, e0, e1, nullptr, &kc); t.pending.emplace_back(e0, e1); t.launches++; }
            else (void)kron_apply(s, mf_args(c), zj, nullptr, true, c->n_cus, nullptr, nullptr, nullptr, nullptr, &kc);
            if (pt.has_lower || pt.has_upper) {
              { Timed te(c, "halo_exchange"); exchange_planes(c, c->cheb_side_lo.p, c->cheb_side_hi.p, plane); }
              la_cheb_fix_planes(s, zn, zj, g, kc.side_lo, c->comm.recv_lo.p, kc.side_hi, c->comm.recv_hi.p, dj, omega, c->n_u, plane);
            }
            if (last && gz_partials) { la_dot_partials(s, g, zn, n_own, gz_partials); dot_done = true; }
          } else if (fuse) {
            KronCheb kc; kc.g = g; kc.znew = zn; kc.omega = omega; kc.cls = c->diag_u_cls.p; kc.tab = c->diag_u_tab.p;
            double *dp = (last && gz_partials) ? gz_partials : nullptr;
            if (dp) PORO_HIP(hipMemsetAsync(dp, 0, kMaxPartials * sizeof(double), s));
            int slots;
            if (c->timing && c->timers["apply_u_chebyshev_fused"].sample(c->timing_stride)) { isolate_sampled_dispatch(c); Timer &t = c->timers["apply_u_chebyshev_fused"]; hipEvent_t e0 = event_get(c), e1 = event_get(c);
                             slots = kron_apply(s, mf_args(c), zj, nullptr, true, c->n_cus, dp, e0, e1, pstate, &kc); t.pending.emplace_back(e0, e1); t.launches++; }
            else slots = kron_apply(s, mf_args(c), zj, nullptr, true, c->n_cus, dp, nullptr, nullptr, pstate, &kc);
            if (dp && slots > 0) dot_done = true;
          } else {
            apply(zj, c->cheb_t.p, nullptr);
            la_cheb_step(s, zn, zj, g, c->cheb_t.p, dj, omega, c->n_u, n_own, (last && gz_partials) ? gz_partials : nullptr);
            if (last && gz_partials) dot_done = true;
          }
          ++c->cheb_applies;
        }
        return dot_done;
      };
      DiagVec dz = dj; dz.z = c->wz_u.p;
      dz.z1_out = (m % 2 == 0) ? c->wz_u.p : c->cheb_z.p; dz.z1_scale = 1.0 / roots[0];
      const int64_t applies0 = c->cheb_applies;
      const int rc = pcg(c, apply, c->n_u, c->comm.part.plane_u, vec(c, PORO_VEC_U), vec(c, PORO_VEC_RHS_U), dz, c->wg_u.p, c->wd_u.p, c->wh_u.p, opts, info, &P, c->pcg_hint_cheb_u, fuse);
      // useful operator applications: one per CG iteration + the initial residual, and m per preconditioner call (one call per iteration + the first direction)
      if (info) info->operator_applications = (int64_t)info->iterations + 1 + (int64_t)m * (info->iterations + 1);
      (void)applies0;
      la_set_constrained(c->stream, vec(c, PORO_VEC_U), c->dir_mask.p, c->dir_val.p, c->n_u);
      la_cons_expand(c->stream, c->cons_u, vec(c, PORO_VEC_U), true);
      return rc;   // (stream-ordered: pcg() returned after the finishing iteration, `distribute` follows in the stream)
    }
    if (opts->preconditioner == PORO_PREC_FDM) {
      // z = blockdiag(A_cc)^-1 g by fast diagonalisation: the same device-controlled SolverCG recurrence with an explicit preconditioner vector
      build_fdm_u(c);
      const FdmOct *oct = c->fdm_oct.built ? &c->fdm_oct : nullptr;
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *in_iteration) {
        // g, z in octant form: three contiguous sweeps; inside the iteration the launches are gated on the device-side "solve finished" flag (before
        // pcg_scalars_start it still holds the previous solve's state)
        if (oct && oct->slab.on) fdm_precondition_u_slab(c, g, z, in_iteration ? c->scal.p : nullptr);
        else if (oct && oct->planar) { Timed tm(c, "precondition_u_fdm"); fdmo_apply_planar(c->stream, *oct, g, z, in_iteration ? c->scal.p : nullptr); }
        else if (oct) {
          Timed tm(c, "precondition_u_fdm");
          if (c->timing && c->timers["fdm_u_pass1"].sample(c->timing_stride)) {     // the three transform dispatches individually (per-kernel roofline of the bench)
            c->timers["fdm_u_pass2"].enqueued++; c->timers["fdm_u_pass3"].enqueued++;
            isolate_sampled_dispatch(c);
            hipEvent_t ev[6]; for (auto &e : ev) e = event_get(c);
            fdmo_apply(c->stream, *oct, g, z, c->fdm_oct.t.p, in_iteration ? c->scal.p : nullptr, ev);
            const char *names[3] = {"fdm_u_pass1", "fdm_u_pass2", "fdm_u_pass3"};
            for (int k = 0; k < 3; ++k) { Timer &t = c->timers[names[k]]; t.pending.emplace_back(ev[2 * k], ev[2 * k + 1]); t.launches++; }
          } else fdmo_apply(c->stream, *oct, g, z, c->fdm_oct.t.p, in_iteration ? c->scal.p : nullptr);
        }
        else fdm_precondition_u(c, g, z);
        return false; };
      DiagVec dz; dz.full = c->dinv_u.p; dz.ncomp = c->dim; dz.inert = c->dir_mask.p; dz.z = c->wz_u.p;
      const int rc = pcg(c, apply, c->n_u, c->comm.part.plane_u, vec(c, PORO_VEC_U), vec(c, PORO_VEC_RHS_U), dz, c->wg_u.p, c->wd_u.p, c->wh_u.p, opts, info, &P, c->pcg_hint_fdm_u, oct != nullptr /* every launch of an iteration is gated: overshooting is cheap */, oct);
      la_set_constrained(c->stream, vec(c, PORO_VEC_U), c->dir_mask.p, c->dir_val.p, c->n_u);
      return rc;
    }
    if (opts->preconditioner == PORO_PREC_TWO_LEVEL) {
      // z = omega D^-1 g + P B_H^-1 P^T g: Jacobi on this mesh + the block fast diagonalisation of the underlying uniform box; SolverCG's recurrence with an explicit preconditioner vector
      if (!two_level_supported(c)) throw Error("PORO_PREC_TWO_LEVEL needs poro_desc.coarse (a refinement of a uniform box whose Dirichlet conditions cover whole faces)");
      const double om = opts->omega > 0 ? opts->omega : 1.0;
      if (!c->wz_u.p) { c->wz_u.alloc(c->n_u); c->wz_u.zero(c->stream); }
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *) { two_level_precondition_u(c, g, z, om); return false; };
      DiagVec dz; dz.full = c->dinv_u.p; dz.ncomp = c->dim; dz.inert = c->cons_u.inert.p; dz.z = c->wz_u.p;
      const int rc = pcg(c, apply, c->n_u, c->comm.part.plane_u, vec(c, PORO_VEC_U), vec(c, PORO_VEC_RHS_U), dz, c->wg_u.p, c->wd_u.p, c->wh_u.p, opts, info, &P, c->pcg_hint_u);
      la_set_constrained(c->stream, vec(c, PORO_VEC_U), c->dir_mask.p, c->dir_val.p, c->n_u);
      la_cons_expand(c->stream, c->cons_u, vec(c, PORO_VEC_U), true);
      return rc;
    }
    DiagVec dv; dv.full = c->dinv_u.p; dv.ncomp = c->dim; dv.inert = c->cons_u.inert.p;
    if (c->diag_u_cls.p) { dv.cls = c->diag_u_cls.p; dv.tab = c->diag_u_tab.p; }
    const int rc = pcg(c, apply, c->n_u, c->comm.part.plane_u, vec(c, PORO_VEC_U), vec(c, PORO_VEC_RHS_U), dv, c->wg_u.p, c->wd_u.p, c->wh_u.p, opts, info, nullptr, c->pcg_hint_u);
    la_set_constrained(c->stream, vec(c, PORO_VEC_U), c->dir_mask.p, c->dir_val.p, c->n_u);   // constraints.distribute (:306)
    la_cons_expand(c->stream, c->cons_u, vec(c, PORO_VEC_U), true);
    return rc;
  });
}

int poro_pres_assemble_residual(poro_ctx *c, double dt, double *l2) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream; double *R = vec(c, PORO_VEC_RESIDUAL_P);
    {
      Timed tm(c, "pressure_residual");
      la_pressure_tmp(s, c->tmp_p.p, vec(c, PORO_VEC_EPSV), vec(c, PORO_VEC_EPSV0), vec(c, PORO_VEC_P), vec(c, PORO_VEC_P_OLD), c->mat.biot_alpha / dt, 1. / c->mat.biot_M / dt, c->n_p);
      if (c->operator_mode == PORO_OP_MATRIX_FREE && c->box.enabled) p_residual_stencil(s, c->dim, c->box, c->mat.k_over_mu, c->tmp_p.p, vec(c, PORO_VEC_P), c->src_local.p, R);
      else la_csr_residual(s, c->Ap, c->Mp.p, c->Kp.p, c->mat.k_over_mu, c->tmp_p.p, vec(c, PORO_VEC_P), c->src_local.p, R);
    }
    la_cons_reduce(s, c->cons_p, R);                                              // constraints.condense(residual) (:153); partial rows first, interface sums after
    exchange_add(c, R, c->n_p, c->comm.part.plane_p);
    if (c->n_pdir) la_mask_zero(s, R, c->pdir_mask.p, c->n_p);                    // prescribed-pressure rows are not part of the Newton system
    la_dot_partials(s, R, R, owned(c, c->n_p, c->comm.part.plane_p), c->partials.p);
    la_reduce_finish(s, c->partials.p, 1, c->red.p, 0);
    allreduce_sum(c, c->red.p, 1);
    post_and_wait(c, c->red.p, 1);
    if (l2) *l2 = std::sqrt(c->mailbox->vals[0]);
    return 0;
  });
}

int poro_pres_apply_boundary_values(poro_ctx *c) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (c->n_pdir) la_set_constrained(c->stream, vec(c, PORO_VEC_P), c->pdir_mask.p, c->pdir_val.p, c->n_p);
    return 0;
  });
}

int poro_pres_assemble_jacobian(poro_ctx *c, double dt) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    // J = M/(M_b dt) + (k/mu) K depends on dt only (the reference recomputes it every pressure iteration): same dt, same matrix
    if (c->jac_dt == dt) return 0;
    Timed tm(c, "pressure_jacobian");
    la_jacobian(c->stream, c->Jp.p, c->Mp.p, c->Kp.p, 1. / c->mat.biot_M / dt, c->mat.k_over_mu, c->Ap.nnz);
    la_csr_diag(c->stream, c->Ap, c->Jp.p, c->diag_J.p);
    exchange_add(c, c->diag_J.p, c->n_p, c->comm.part.plane_p);
    if (!c->dinv_J.p) c->dinv_J.alloc(c->n_p);
    la_reciprocal(c->stream, c->dinv_J.p, c->diag_J.p, c->n_p);
    if (c->cons_p.n || c->n_pdir) la_mask_zero(c->stream, c->dinv_J.p, c->cons_p.inert.p, c->n_p);
    c->ilu_J_valid = false;
    c->jac_dt = dt;
    return 0;
  });
}

int poro_pres_solve(poro_ctx *c, const poro_solver_opts *opts, poro_solve_info *info) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (c->jac_dt < 0) throw Error("pres_solve before pres_assemble_jacobian");
    if ((c->cons_p.n || c->n_pdir) && opts->preconditioner != PORO_PREC_JACOBI && opts->preconditioner != PORO_PREC_NONE && !(opts->preconditioner == PORO_PREC_TWO_LEVEL && !c->n_pdir)) throw Error("meshes with hanging-node constraints or prescribed pressures: PORO_PREC_JACOBI / NONE (hanging nodes: also TWO_LEVEL) only");
    if (opts->preconditioner == PORO_PREC_ILU0) {
      return pcg_ilu0(c, c->Ap, c->Jp.p, c->ilu_J, c->ilu_J_valid, vec(c, PORO_VEC_DP), vec(c, PORO_VEC_RESIDUAL_P), c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
    }
    if (opts->preconditioner == PORO_PREC_SSOR) return pcg_ssor(c, c->Ap, c->Jp.p, vec(c, PORO_VEC_DP), vec(c, PORO_VEC_RESIDUAL_P), c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
    const bool stencil = c->operator_mode == PORO_OP_MATRIX_FREE && c->box.enabled;   // uniform box: J is a constant-coefficient stencil
    const double ja = 1. / c->mat.biot_M / c->jac_dt, jk = c->mat.k_over_mu;
    auto apply = [&](const double *x, double *y, double *) {
      la_cons_expand(c->stream, c->cons_p, const_cast<double *>(x), false);      // condensed Jacobian C^T J C (:168)
      if (stencil) { Timed tm(c, "apply_p_stencil"); p_stencil_apply(c->stream, c->dim, c->box, ja, jk, x, y); }
      else { Timed tm(c, "apply_p_csr"); la_csr_spmv(c->stream, c->Ap, c->Jp.p, x, y); }
      la_cons_reduce(c->stream, c->cons_p, y);
      exchange_add(c, y, c->n_p, c->comm.part.plane_p); return false;
    };
    if (opts->preconditioner == PORO_PREC_FDM) {
      build_fdm_p(c);
      const double kk[3] = {jk, jk, jk};
      if (!c->wz_p.p) c->wz_p.alloc(c->n_p);
      // one rank, uniform box (2D or 3D): the fast diagonalisation is the exact inverse of J, so the update is computed directly and its residual checked against
      // the reference's stopping rule (:175) with one poll; info->iterations = 0 marks a directly solved system.  A failed check falls through to CG with that update as start
      static const bool iterative = std::getenv("PORO_PRES_ITERATIVE") != nullptr;
      if (!iterative && opts->stop_rule == PORO_STOP_RHS && stencil) {            // (slab partitions too: the distributed fast diagonalisation is the same exact inverse)
        hipStream_t s = c->stream; double *x = vec(c, PORO_VEC_DP); const double *b = vec(c, PORO_VEC_RESIDUAL_P); const double *y = c->wh_p.p;
        const auto t0 = std::chrono::steady_clock::now();
        fdm_precondition_p(c, ja, kk, b, x);
        { Timed tm(c, "apply_p_stencil"); p_stencil_apply(s, c->dim, c->box, ja, jk, x, c->wh_p.p); }
        exchange_add(c, c->wh_p.p, c->n_p, c->comm.part.plane_p);
        la_residual_norms_many(s, 1, &y, &b, owned(c, c->n_p, c->comm.part.plane_p), c->partials.p);
        pcg_scalars_sum(s, c->partials.p, 2, c->red.p);
        allreduce_sum(c, c->red.p, 2);
        post_and_wait(c, c->red.p, 2);
        const double res = std::sqrt(c->mailbox->vals[0]), bn = std::sqrt(c->mailbox->vals[1]);
        if (res <= std::max(opts->abs_tol, opts->rel_tol * bn)) {
          if (info) { *info = poro_solve_info{}; info->iterations = 0; info->converged = 1; info->initial_residual = bn; info->final_residual = res; info->operator_applications = 1;
                      info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
          return 0;
        }
      }
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *) { fdm_precondition_p(c, ja, kk, g, z); return false; };
      DiagVec dz; dz.full = c->dinv_J.p; dz.z = c->wz_p.p;
      return pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_DP), vec(c, PORO_VEC_RESIDUAL_P), dz, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, &P, c->pcg_hint_p);
    }
    if (opts->preconditioner == PORO_PREC_TWO_LEVEL) {
      if (!two_level_supported_p(c)) throw Error("PORO_PREC_TWO_LEVEL (pressure): needs poro_desc.coarse with the pressure interpolation (ptr_p / node_p / weight_p)");
      if (!c->wz_p.p) c->wz_p.alloc(c->n_p);
      const double om = opts->omega > 0 ? opts->omega : 1.0;
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *) { two_level_precondition_p(c, ja, jk, c->dinv_J.p, g, z, om); return false; };
      DiagVec dz; dz.full = c->dinv_J.p; dz.z = c->wz_p.p; dz.inert = c->cons_p.n ? c->cons_p.inert.p : nullptr;
      const int rc = pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_DP), vec(c, PORO_VEC_RESIDUAL_P), dz, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, &P, c->pcg_hint_p);
      la_cons_expand(c->stream, c->cons_p, vec(c, PORO_VEC_DP), true);
      return rc;
    }
    DiagVec dv; dv.full = c->dinv_J.p; dv.inert = (c->cons_p.n || c->n_pdir) ? c->cons_p.inert.p : nullptr;
    const int rc = pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_DP), vec(c, PORO_VEC_RESIDUAL_P), dv, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, nullptr, c->pcg_hint_p);
    la_cons_expand(c->stream, c->cons_p, vec(c, PORO_VEC_DP), true);              // constraints.distribute(solution_update) (:180)
    return rc;
  });
}

int poro_pres_update_volumetric_strain(poro_ctx *c) {
  return guarded([&] { PORO_HIP(hipSetDevice(c->device)); la_axpy(c->stream, vec(c, PORO_VEC_EPSV), c->mat.biot_alpha / c->mat.bulk_K, vec(c, PORO_VEC_DP), c->n_p); return 0; });
}

int poro_proj_assemble_matrix(poro_ctx *c) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    la_csr_diag(c->stream, c->Ap, c->Mp.p, c->diag_M.p);                       // projection_matrix = mass_matrix (StrainProjector.h:104)
    exchange_add(c, c->diag_M.p, c->n_p, c->comm.part.plane_p);
    if (!c->dinv_M.p) c->dinv_M.alloc(c->n_p);
    la_reciprocal(c->stream, c->dinv_M.p, c->diag_M.p, c->n_p);
    if (c->cons_p.n) la_mask_zero(c->stream, c->dinv_M.p, c->cons_p.inert.p, c->n_p);
    c->projection_matrix_ready = true; return 0;
  });
}

int poro_proj_assemble_rhs(poro_ctx *c, const int32_t *tensor_components, int32_t n_comp) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    const int dim = c->dim; if (n_comp < 0 || n_comp > 6) throw Error("n_comp out of range");
    static const int m2[4] = {0, 1, 1, 2}, m3[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};   // TensorIndexer.h:24-31
    double *rhs[6];
    for (int k = 0; k < n_comp; ++k) {
      if (tensor_components[k] < 0 || tensor_components[k] >= dim * dim) throw Error("tensor component out of range");
      const int e = dim == 2 ? m2[tensor_components[k]] : m3[tensor_components[k]];
      rhs[k] = vec(c, PORO_VEC_PROJ_RHS0 + e); if (!c->box_asm) la_fill(c->stream, rhs[k], 0.0, c->n_p);   // :146-147
    }
    {
      Timed tm(c, "projection_rhs");
      if (c->box_asm) box_proj_rhs(c->stream, c->dim, c->box_cpl, vec(c, PORO_VEC_U), n_comp, tensor_components, rhs);
      else {
        const AsmArgs a = asm_args(c);
        for (size_t k = 0; k + 1 < c->color_off.size(); ++k)
          asm_proj_rhs(c->stream, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], vec(c, PORO_VEC_U), n_comp, tensor_components, rhs);
      }
    }
    for (int k = 0; k < n_comp; ++k) { la_cons_reduce(c->stream, c->cons_p, rhs[k]); exchange_add(c, rhs[k], c->n_p, c->comm.part.plane_p); }   // StrainProjector.h:191-194 (partial rows folded, then the interface sums)
    return 0;
  });
}

int poro_proj_solve(poro_ctx *c, int32_t entry, const poro_solver_opts *opts, poro_solve_info *info) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (!c->projection_matrix_ready) throw Error("proj_solve before proj_assemble_matrix");
    if (entry < 0 || entry >= c->dim * (c->dim + 1) / 2) throw Error("rhs_entry out of range");
    if (c->cons_p.n && opts->preconditioner != PORO_PREC_JACOBI && opts->preconditioner != PORO_PREC_NONE && opts->preconditioner != PORO_PREC_TWO_LEVEL) throw Error("meshes with hanging-node constraints: PORO_PREC_JACOBI / TWO_LEVEL / NONE only");
    if (opts->preconditioner == PORO_PREC_ILU0) {
      return pcg_ilu0(c, c->Ap, c->Mp.p, c->ilu_M, c->ilu_M_valid, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
    }
    if (opts->preconditioner == PORO_PREC_SSOR) return pcg_ssor(c, c->Ap, c->Mp.p, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
    const bool stencil = c->operator_mode == PORO_OP_MATRIX_FREE && c->box.enabled;
    auto apply = [&](const double *x, double *y, double *) {
      la_cons_expand(c->stream, c->cons_p, const_cast<double *>(x), false);      // condensed projection matrix (StrainProjector.h:104-105)
      if (stencil) { Timed tm(c, "apply_p_stencil"); p_stencil_apply(c->stream, c->dim, c->box, 1.0, 0.0, x, y); }
      else { Timed tm(c, "apply_p_csr"); la_csr_spmv(c->stream, c->Ap, c->Mp.p, x, y); }
      la_cons_reduce(c->stream, c->cons_p, y);
      exchange_add(c, y, c->n_p, c->comm.part.plane_p); return false;
    };
    if (opts->preconditioner == PORO_PREC_FDM) {
      build_fdm_p(c);
      const double kk[3] = {0, 0, 0};
      if (!c->wz_p.p) c->wz_p.alloc(c->n_p);
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *) { fdm_precondition_p(c, 1.0, kk, g, z); return false; };
      DiagVec dz; dz.full = c->dinv_M.p; dz.z = c->wz_p.p;
      return pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), dz, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, &P, c->pcg_hint_proj);
    }
    if (opts->preconditioner == PORO_PREC_TWO_LEVEL) {
      if (!two_level_supported_p(c)) throw Error("PORO_PREC_TWO_LEVEL (projection): needs poro_desc.coarse with the pressure interpolation");
      if (!c->wz_p.p) c->wz_p.alloc(c->n_p);
      const double om = opts->omega > 0 ? opts->omega : 1.0;
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *) { two_level_precondition_p(c, 1.0, 0.0, c->dinv_M.p, g, z, om); return false; };
      DiagVec dz; dz.full = c->dinv_M.p; dz.z = c->wz_p.p; dz.inert = c->cons_p.n ? c->cons_p.inert.p : nullptr;
      const int rc = pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), dz, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, &P, c->pcg_hint_proj);
      la_cons_expand(c->stream, c->cons_p, vec(c, PORO_VEC_STRAIN0 + entry), true);
      return rc;
    }
    DiagVec dv; dv.full = c->dinv_M.p; dv.inert = c->cons_p.n ? c->cons_p.inert.p : nullptr;
    const int rc = pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), dv, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, nullptr, c->pcg_hint_proj);
    la_cons_expand(c->stream, c->cons_p, vec(c, PORO_VEC_STRAIN0 + entry), true);   // constraints.distribute (StrainProjector.h:216)
    return rc;
  });
}

// Several projection systems at once (the three normal strains of a time step, PoroelasticityFSS.h:153-164).  Where the fast diagonalisation is the EXACT inverse of the
// projection mass matrix (uniform box or tensor grid, one rank, no constraint lists) the systems are solved directly: x_e = M^-1 b_e for all of them in one set of three
// launches, then ||M x_e - b_e|| is checked against the stopping rule of the reference's CG (SolverControl, StrainProjector.h:209) and one poll returns all norms.
// Otherwise - or if a check fails - every entry goes through poro_proj_solve.  info[e].iterations = 0 marks a directly solved entry.
int poro_proj_solve_many(poro_ctx *c, const int32_t *entries, int32_t n_entries, const poro_solver_opts *opts, poro_solve_info *info) {
  if (!c || !entries || !opts || n_entries < 0) { g_err = "null argument"; return -1; }
  int done_direct = 0;
  const int rc0 = guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (!c->projection_matrix_ready) throw Error("proj_solve before proj_assemble_matrix");
    for (int e = 0; e < n_entries; ++e) if (entries[e] < 0 || entries[e] >= c->dim * (c->dim + 1) / 2) throw Error("rhs_entry out of range");
    static const bool iterative = std::getenv("PORO_PROJ_ITERATIVE") != nullptr;
    const bool stencil = c->operator_mode == PORO_OP_MATRIX_FREE && c->box.enabled;
    if (iterative || opts->preconditioner != PORO_PREC_FDM || opts->stop_rule != PORO_STOP_RHS || c->cons_p.n || !stencil || n_entries < 1 || n_entries > 3 || !fdm_p_supported(c)) return 0;
    build_fdm_p(c);
    const bool batched = !c->comm.multi() && c->fdm_p_fused.built && !c->fdm_p_fused.slab.on;      // one rank, 3D, lines of <= 128 vertices: all right-hand sides in one set of launches
    hipStream_t s = c->stream; const double *b[3]; double *x[3]; const double *y[3];
    if (c->proj_y.n < (size_t)3 * c->n_p) c->proj_y.alloc((size_t)3 * c->n_p);
    for (int e = 0; e < n_entries; ++e) { b[e] = vec(c, PORO_VEC_PROJ_RHS0 + entries[e]); x[e] = vec(c, PORO_VEC_STRAIN0 + entries[e]); y[e] = c->proj_y.p + (size_t)e * c->n_p; }
    const auto t0 = std::chrono::steady_clock::now();
    if (batched) { Timed tm(c, "precondition_p_fdm"); fdmo_scalar_apply_many(s, c->fdm_p_fused, 1.0, 0.0, n_entries, b, x); }
    else { const double kk[3] = {0, 0, 0}; for (int e = 0; e < n_entries; ++e) fdm_precondition_p(c, 1.0, kk, b[e], x[e]); }
    for (int e = 0; e < n_entries; ++e) { { Timed tm(c, "apply_p_stencil"); p_stencil_apply(s, c->dim, c->box, 1.0, 0.0, x[e], const_cast<double *>(y[e])); } exchange_add(c, const_cast<double *>(y[e]), c->n_p, c->comm.part.plane_p); }
    la_residual_norms_many(s, n_entries, y, b, owned(c, c->n_p, c->comm.part.plane_p), c->partials.p);
    pcg_scalars_sum(s, c->partials.p, 2 * n_entries, c->red.p);
    allreduce_sum(c, c->red.p, 2 * n_entries);
    post_and_wait(c, c->red.p, 2 * n_entries);
    bool all = true;
    for (int e = 0; e < n_entries; ++e) {
      const double res = std::sqrt(c->mailbox->vals[2 * e]), bn = std::sqrt(c->mailbox->vals[2 * e + 1]);
      const bool ok = res <= std::max(opts->abs_tol, opts->rel_tol * bn);
      all = all && ok;
      if (info) { info[e] = poro_solve_info{}; info[e].iterations = 0; info[e].converged = ok ? 1 : 0; info[e].initial_residual = bn; info[e].final_residual = res; info[e].operator_applications = 1;
                  info[e].seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n_entries; }
    }
    done_direct = all ? 1 : 0;     // (a failed check leaves x_e = M^-1 b_e as the warm start of the iterative solve below)
    return 0;
  });
  if (rc0 != 0) return rc0;
  if (done_direct) return 0;
  int worst = 0;
  for (int e = 0; e < n_entries; ++e) { const int rc = poro_proj_solve(c, entries[e], opts, info ? info + e : nullptr); if (rc < 0) return rc; worst = std::max(worst, rc); }
  return worst;
}

int poro_get_volumetric_strain(poro_ctx *c) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    const int dim = c->dim; const double *sp[3];
    static const int m2[4] = {0, 1, 1, 2}, m3[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    for (int a = 0; a < dim; ++a) sp[a] = vec(c, PORO_VEC_STRAIN0 + (dim == 2 ? m2[a * dim + a] : m3[a * dim + a]));
    la_sum_strains(c->stream, vec(c, PORO_VEC_EPSV), sp, dim, c->n_p); return 0;
  });
}

int poro_get_effective_stresses(poro_ctx *c) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    const int ne = c->dim * (c->dim + 1) / 2; const double *e[6]; double *g[6];
    for (int k = 0; k < ne; ++k) { e[k] = vec(c, PORO_VEC_STRAIN0 + k); g[k] = vec(c, PORO_VEC_STRESS0 + k); }
    la_effective_stress(c->stream, e, g, c->dim, c->mat.lame_lambda, c->mat.shear_G, c->n_p);
    PORO_HIP(hipStreamSynchronize(c->stream)); return 0;
  });
}

static void csr_of(poro_ctx *c, int which, CsrDev **A, double **val) {
  switch (which) {
    case PORO_MAT_A_U: if (c->operator_mode != PORO_OP_CSR) throw Error("A_u is matrix-free in this context"); *A = &c->Au; *val = c->Au_val.p; return;
    case PORO_MAT_MASS_P: *A = &c->Ap; *val = c->Mp.p; return;
    case PORO_MAT_LAPLACE_P: *A = &c->Ap; *val = c->Kp.p; return;
    case PORO_MAT_JACOBIAN_P: *A = &c->Ap; *val = c->Jp.p; return;
  }
  throw Error("unknown matrix id");
}
int poro_export_csr_size(poro_ctx *c, int which, int64_t *n_rows, int64_t *nnz) {
  return guarded([&] { CsrDev *A; double *v; csr_of(c, which, &A, &v); *n_rows = A->n; *nnz = A->nnz; return 0; });
}
int poro_export_csr(poro_ctx *c, int which, int64_t *row_ptr, int32_t *col, double *val) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device)); CsrDev *A; double *v; csr_of(c, which, &A, &v);
    PORO_HIP(hipStreamSynchronize(c->stream));
    PORO_HIP(hipMemcpy(row_ptr, A->rp.p, (A->n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost)); PORO_HIP(hipMemcpy(col, A->col.p, A->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    PORO_HIP(hipMemcpy(val, v, A->nnz * sizeof(double), hipMemcpyDeviceToHost)); return 0;
  });
}

int poro_apply_operator(poro_ctx *c, int which, const double *x_host, double *y_host) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device)); hipStream_t s = c->stream;
    const bool isu = which == PORO_MAT_A_U; const int64_t n = isu ? c->n_u : c->n_p;
    DevBuf<double> x, y; x.upload(x_host, n); y.alloc(n);
    if (isu) { if (!c->matrix_built) throw Error("apply A_u before disp_assemble_system"); apply_A_u(c, x.p, y.p, c->operator_mode); }
    else { CsrDev *A; double *v; csr_of(c, which, &A, &v); la_csr_spmv(s, *A, v, x.p, y.p); exchange_add(c, y.p, n, c->comm.part.plane_p); }
    PORO_HIP(hipMemcpyAsync(y_host, y.p, n * sizeof(double), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s)); return 0;
  });
}

int poro_apply_preconditioner_u(poro_ctx *c, int32_t preconditioner, const double *g_host, double *z_host, int32_t reps, double *seconds_per_apply) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device)); hipStream_t s = c->stream;
    if (!c->matrix_built) throw Error("apply_preconditioner_u before disp_assemble_system");
    DevBuf<double> g, z; g.upload(g_host, c->n_u); z.alloc(c->n_u); z.zero(s);
    if (preconditioner == PORO_PREC_FDM) {
      build_fdm_u(c);
      FdmOct &O = c->fdm_oct;
      // octant form where the solver uses it: butterflies outside (H, H'), the three transform passes in between - the timed part, as inside PCG
      auto once = [&]() { if (O.built && O.slab.on) fdm_precondition_u_slab(c, O.g.p, O.z.p, nullptr); else if (O.built && O.planar) fdmo_apply_planar(s, O, O.g.p, O.z.p); else if (O.built) fdmo_apply(s, O, O.g.p, O.z.p, O.t.p); else fdm_precondition_u(c, g.p, z.p); };
      if (O.built) fdmo_from_nodal(s, O, g.p, O.g.p);
      once();
      if (O.built) fdmo_to_nodal(s, O, O.z.p, z.p);
      if (reps > 0 && seconds_per_apply) {
        EventPair ev(c); PORO_HIP(hipEventRecord(ev.e0, s));
        for (int k = 0; k < reps; ++k) once();
        PORO_HIP(hipEventRecord(ev.e1, s)); PORO_HIP(hipEventSynchronize(ev.e1));
        float ms = 0; PORO_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1)); *seconds_per_apply = ms * 1e-3 / reps;
      }
    } else if (preconditioner == PORO_PREC_JACOBI) {
      PORO_HIP(hipMemcpyAsync(z.p, g.p, c->n_u * sizeof(double), hipMemcpyDeviceToDevice, s));
      la_pointwise_mul(s, z.p, c->dinv_u.p, c->n_u);
    } else throw Error("apply_preconditioner_u: PORO_PREC_JACOBI or PORO_PREC_FDM");
    PORO_HIP(hipMemcpyAsync(z_host, z.p, c->n_u * sizeof(double), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s)); return 0;
  });
}

int poro_bench_operator(poro_ctx *c, int which, int operator_mode, int reps, double *seconds_per_apply) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device)); hipStream_t s = c->stream;
    if (which != PORO_MAT_A_U) throw Error("bench_operator: only A_u");
    if (!c->matrix_built) throw Error("bench_operator before disp_assemble_system");
    if (operator_mode != c->operator_mode) throw Error("bench_operator: context was created with the other operator mode");
    std::vector<double> hx(c->n_u); for (int64_t i = 0; i < c->n_u; ++i) hx[i] = std::sin(0.37 * (double)i);   // SURVEY 8d synthetic vector
    DevBuf<double> x, y; x.upload(hx); y.alloc(c->n_u);
    for (int k = 0; k < 3; ++k) apply_A_u(c, x.p, y.p, operator_mode);
    EventPair ev(c); const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    PORO_HIP(hipEventRecord(e0, s));
    for (int k = 0; k < reps; ++k) apply_A_u(c, x.p, y.p, operator_mode);
    PORO_HIP(hipEventRecord(e1, s)); PORO_HIP(hipEventSynchronize(e1));
    float ms = 0; PORO_HIP(hipEventElapsedTime(&ms, e0, e1));
    *seconds_per_apply = ms * 1e-3 / reps; return 0;
  });
}

int poro_timers_reset(poro_ctx *c) { return guarded([&] { PORO_HIP(hipSetDevice(c->device)); timers_collect(c); c->timers.clear(); c->timing = true; c->timing_stride = 1; return 0; }); }
int poro_timers_enable(poro_ctx *c, int on) { return guarded([&] { PORO_HIP(hipSetDevice(c->device)); timers_collect(c); c->timing = on != 0; c->timing_stride = on > 1 ? on : 1; return 0; }); }
int poro_timers_get(poro_ctx *c, const char *name, double *seconds, int64_t *launches) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device)); timers_collect(c);
    auto it = c->timers.find(name);
    // sampled families: the measured time is scaled to all launches of the family (mean sampled duration x launches enqueued)
    const bool have = it != c->timers.end() && it->second.launches > 0;
    if (seconds) *seconds = have ? it->second.seconds * (double)std::max(it->second.enqueued, it->second.launches) / (double)it->second.launches : 0.0;
    if (launches) *launches = it == c->timers.end() ? 0 : std::max(it->second.enqueued, it->second.launches);
    return 0;
  });
}

}  // extern "C"
