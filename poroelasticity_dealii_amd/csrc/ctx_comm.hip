// Plumbing shared by the host logic: RCCL (resolved at run time), kernel timing, the host mailbox, slab / general-partition exchanges and all-reduces, the operator application.
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

namespace poro {
namespace ctx_detail {
// ---- RCCL, resolved at run time so single-GPU use has no dependency on it ---------------------------------------
thread_local std::string g_err;

Rccl g_rccl;
void Rccl::load() {
  if (lib) return;
  lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) throw Error(std::string("cannot load librccl: ") + dlerror());
  auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) throw Error(std::string("librccl lacks ") + n); return p; };
  GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId"); CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
  CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy"); AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
  Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv"); GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
  GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
}


// ---- timing -----------------------------------------------------------------------------------------------------
// HIP events on the launch stream around every kernel family, drawn from a pool so a timed launch costs two
// hipEventRecord calls; elapsed times are read back in bulk by timers_collect.  The pool belongs to the context (its device, its host thread).
hipEvent_t event_get(poro_ctx *c) {
  if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
  hipEvent_t e; PORO_HIP(hipEventCreate(&e)); return e;
}
Timed::Timed(poro_ctx *c_, const char *name) : c(c_) {
  if (!c->timing) return;
  Timer *tt = &c->timers[name];
  if (!tt->sample(c->timing_stride)) return;
  t = tt;
  a = event_get(c); b = event_get(c); (void)hipEventRecord(a, c->stream);
}
Timed::~Timed() { if (!t) return; (void)hipEventRecord(b, c->stream); t->pending.emplace_back(a, b); t->launches++; }
void timers_collect(poro_ctx *c) {
  (void)hipStreamSynchronize(c->stream);
  for (auto &kv : c->timers) {
    for (auto &p : kv.second.pending) { float ms = 0; (void)hipEventElapsedTime(&ms, p.first, p.second); kv.second.seconds += ms * 1e-3; c->event_pool.push_back(p.first); c->event_pool.push_back(p.second); }
    kv.second.pending.clear();
  }
}
// before a dispatch that carries start / stop events in a stream whose other dispatches carry none: a marker that drains the stream, so that the bracket holds the kernel
// alone (otherwise its first workgroups share the chip with the tail of the previous kernel and the bracket reads a few microseconds long)
void isolate_sampled_dispatch(poro_ctx *c) {
  if (c->timing_stride <= 1) return;
  hipEvent_t m = event_get(c); (void)hipEventRecord(m, c->stream); c->event_pool.push_back(m);
}


// ---- device -> host scalars without a copy engine or a stream synchronisation (Mailbox, common.hpp) ------------------------------------------
// enqueue the publishing kernel behind everything that is in the stream, then spin on the sequence number in pinned host memory
void post_and_wait(poro_ctx *c, const double *dev_src, int n, const PcgScalars *sc) {
  if (n > 16) throw Error("post_and_wait: at most 16 scalars");
  const unsigned long long want = ++c->mb_seq;
  la_post(c->stream, c->mailbox, want, dev_src, n, sc);
  PORO_HIP(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now(); unsigned spins = 0;
  while (__atomic_load_n(const_cast<unsigned long long *>(&c->mailbox->seq), __ATOMIC_ACQUIRE) != want) {
    __builtin_ia32_pause();
    if ((++spins & 0xfffff) == 0) {   // every ~million spins: is the device still alive?  (a faulted kernel would otherwise leave the host spinning for ever)
      const hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) throw Error(std::string("device failed while the host waited for its answer: ") + hipGetErrorString(q));
      if (q == hipSuccess && __atomic_load_n(const_cast<unsigned long long *>(&c->mailbox->seq), __ATOMIC_ACQUIRE) != want) throw Error("mailbox: the stream drained without publishing the expected sequence number");
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 600.0) throw Error("mailbox: no answer from the device within 600 s");
    }
  }
}

// ---- communication: sum the neighbour's partial rows on the shared node planes; all-reduce scalars -------------------
// general partition: per-neighbour interface lists (poro_partition.shared_*).  One pack kernel, one grouped exchange with all neighbours, one kernel
// that sums own + received partial rows in ascending rank order.
void build_interface(poro_ctx *c, IfcDev &I, const int64_t *ptr, const int32_t *dof, int64_t n, int64_t n_owned) {
  const poro_partition &pt = c->comm.part; const int nn = pt.n_neighbours;
  if (!ptr || (ptr[nn] > 0 && !dof)) throw Error("poro_partition: interface lists missing");
  if (n_owned < 0 || n_owned > n) throw Error("poro_partition: n_owned out of range");
  I.n_owned = n_owned; I.ptr.assign(ptr, ptr + nn + 1); I.m_send = ptr[nn];
  if (ptr[0] != 0) throw Error("poro_partition: shared_ptr must start at 0");
  std::vector<std::vector<int32_t>> src(n);          // per local dof: sources in ascending rank order
  std::vector<char> own_in(n, 0);
  for (int k = 0; k < nn; ++k) {
    if (ptr[k + 1] < ptr[k]) throw Error("poro_partition: shared_ptr not monotone");
    const bool self_before = pt.rank < pt.neighbour_rank[k];
    for (int64_t j = ptr[k]; j < ptr[k + 1]; ++j) {
      const int32_t i = dof[j]; if (i < 0 || i >= n) throw Error("poro_partition: shared dof out of range");
      if (self_before && !own_in[i]) { src[i].push_back(-1); own_in[i] = 1; }
      src[i].push_back((int32_t)j);
    }
  }
  std::vector<int32_t> sh_dof, sh_src; std::vector<int64_t> sh_ptr{0};
  for (int64_t i = 0; i < n; ++i) if (!src[i].empty()) {
    if (!own_in[i]) src[i].push_back(-1);
    sh_dof.push_back((int32_t)i); sh_src.insert(sh_src.end(), src[i].begin(), src[i].end()); sh_ptr.push_back((int64_t)sh_src.size());
  }
  I.m_shared = (int64_t)sh_dof.size();
  if (I.m_send) { I.dof.upload(std::vector<int32_t>(dof, dof + I.m_send)); I.send.alloc(I.m_send); I.recv.alloc(I.m_send); I.hsend.resize(I.m_send); I.hrecv.resize(I.m_send); }
  if (I.m_shared) { I.sh_dof.upload(sh_dof); I.sh_src.upload(sh_src); I.sh_ptr.upload(sh_ptr); }
}
void setup_general_partition(poro_ctx *c, const poro_desc *d) {
  Comm &cm = c->comm; const poro_partition &pt = cm.part;
  if (pt.n_neighbours <= 0) return;
  if (pt.n_ranks < 2) throw Error("poro_partition: neighbours on a single rank");
  if (!pt.neighbour_rank) throw Error("poro_partition: neighbour_rank missing");
  for (int k = 0; k < pt.n_neighbours; ++k) {
    const int q = pt.neighbour_rank[k];
    if (q < 0 || q >= pt.n_ranks || q == pt.rank || (k && q <= pt.neighbour_rank[k - 1])) throw Error("poro_partition: neighbour_rank must be ascending, in range and without the own rank");
  }
  cm.general = true; cm.neighbours.assign(pt.neighbour_rank, pt.neighbour_rank + pt.n_neighbours);
  build_interface(c, cm.ifc_u, pt.shared_ptr_u, pt.shared_dof_u, d->n_dofs_u, pt.n_owned_u);
  build_interface(c, cm.ifc_p, pt.shared_ptr_p, pt.shared_dof_p, d->n_dofs_p, pt.n_owned_p);
  if (c->dim > 1 && pt.n_owned_u % c->dim) throw Error("poro_partition: n_owned_u must hold whole displacement nodes");
}
void exchange_add_general(poro_ctx *c, double *v, int64_t n) {
  Comm &cm = c->comm; IfcDev &I = n == c->n_u ? cm.ifc_u : cm.ifc_p;
  if (!I.m_send) return;
  const int nn = (int)cm.neighbours.size();
  la_ifc_pack(c->stream, I, v);
  if (cm.nccl_comm) {
    ncclComm_t comm = (ncclComm_t)cm.nccl_comm;
    PORO_NCCL(g_rccl.GroupStart());
    for (int k = 0; k < nn; ++k) { const int64_t m = I.ptr[k + 1] - I.ptr[k]; if (!m) continue;
      PORO_NCCL(g_rccl.Send(I.send.p + I.ptr[k], m, ncclFloat64, cm.neighbours[k], comm, c->stream)); PORO_NCCL(g_rccl.Recv(I.recv.p + I.ptr[k], m, ncclFloat64, cm.neighbours[k], comm, c->stream)); }
    PORO_NCCL(g_rccl.GroupEnd());
  } else if (cm.sr) {
    PORO_HIP(hipMemcpyAsync(I.hsend.data(), I.send.p, I.m_send * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < nn; ++k) { const int64_t m = I.ptr[k + 1] - I.ptr[k]; if (m) cm.sr(I.hsend.data() + I.ptr[k], I.hrecv.data() + I.ptr[k], m, cm.neighbours[k], cm.user); }
    PORO_HIP(hipMemcpyAsync(I.recv.p, I.hrecv.data(), I.m_send * sizeof(double), hipMemcpyHostToDevice, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
  } else throw Error("partitioned context without a communicator (call poro_ctx_comm_init_* first)");
  la_ifc_sum(c->stream, I, v);
}
// slab partitions: send the planes `send_lo` / `send_hi` to the lower / upper neighbour, receive theirs into comm.recv_lo / recv_hi (one grouped exchange)
void exchange_planes(poro_ctx *c, const double *send_lo, const double *send_hi, int64_t plane) {
  Comm &cm = c->comm;
  if (cm.recv_lo.n < (size_t)plane) { cm.recv_lo.alloc(plane); cm.recv_hi.alloc(plane); }
  if (cm.nccl_comm) {
    ncclComm_t comm = (ncclComm_t)cm.nccl_comm;
    PORO_NCCL(g_rccl.GroupStart());
    if (cm.part.has_upper) { PORO_NCCL(g_rccl.Send(send_hi, plane, ncclFloat64, cm.part.rank + 1, comm, c->stream)); PORO_NCCL(g_rccl.Recv(cm.recv_hi.p, plane, ncclFloat64, cm.part.rank + 1, comm, c->stream)); }
    if (cm.part.has_lower) { PORO_NCCL(g_rccl.Send(send_lo, plane, ncclFloat64, cm.part.rank - 1, comm, c->stream)); PORO_NCCL(g_rccl.Recv(cm.recv_lo.p, plane, ncclFloat64, cm.part.rank - 1, comm, c->stream)); }
    PORO_NCCL(g_rccl.GroupEnd());
  } else if (cm.sr) {
    cm.hsend.resize(plane); cm.hrecv.resize(plane);
    auto one = [&](const double *dev_send, double *dev_recv, int peer) {
      PORO_HIP(hipMemcpyAsync(cm.hsend.data(), dev_send, plane * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      PORO_HIP(hipStreamSynchronize(c->stream));
      cm.sr(cm.hsend.data(), cm.hrecv.data(), plane, peer, cm.user);
      PORO_HIP(hipMemcpyAsync(dev_recv, cm.hrecv.data(), plane * sizeof(double), hipMemcpyHostToDevice, c->stream));
      PORO_HIP(hipStreamSynchronize(c->stream));
    };
    if (cm.part.has_upper) one(send_hi, cm.recv_hi.p, cm.part.rank + 1);
    if (cm.part.has_lower) one(send_lo, cm.recv_lo.p, cm.part.rank - 1);
  } else throw Error("partitioned context without a communicator (call poro_ctx_comm_init_* first)");
}
void exchange_add(poro_ctx *c, double *v, int64_t n, int64_t plane) {
  Comm &cm = c->comm;
  if (!cm.multi()) return;
  Timed tm(c, "halo_exchange");
  if (cm.general) { exchange_add_general(c, v, n); return; }
  exchange_planes(c, v, v + n - plane, plane);
  la_add_two_ranges(c->stream, cm.part.has_upper ? v + n - plane : nullptr, cm.recv_hi.p, cm.part.has_lower ? v : nullptr, cm.recv_lo.p, plane);
}
void allreduce_sum(poro_ctx *c, double *dev, int n) {
  Comm &cm = c->comm;
  if (!cm.multi()) return;
  Timed tm(c, "allreduce");
  if (cm.nccl_comm) PORO_NCCL(g_rccl.AllReduce(dev, dev, n, ncclFloat64, ncclSum, (ncclComm_t)cm.nccl_comm, c->stream));
  else if (cm.ar) {
    double h[kScalarSlots];
    PORO_HIP(hipMemcpyAsync(h, dev, n * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
    cm.ar(h, n, cm.user);
    PORO_HIP(hipMemcpyAsync(dev, h, n * sizeof(double), hipMemcpyHostToDevice, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
  } else throw Error("partitioned context without a communicator");
}
int64_t owned(poro_ctx *c, int64_t n, int64_t plane) {
  if (c->comm.general) return n == c->n_u ? c->comm.ifc_u.n_owned : c->comm.ifc_p.n_owned;
  return (c->comm.multi() && c->comm.part.has_upper) ? n - plane : n;
}

AsmArgs asm_args(poro_ctx *c) {
  AsmArgs a{};
  a.dim = c->dim; a.k_u = c->k_u; a.ns_u = c->ns_u; a.ns_p = c->ns_p; a.nv = c->nv; a.dpc_u = c->dpc_u; a.fe = c->fe;
  a.cell_dofs_u = c->cell_dofs_u.p; a.cell_dofs_p = c->cell_dofs_p.p; a.cell_X = c->cell_X.p; a.cell_geo = c->cell_geo.p; a.dir_mask = c->dir_mask.p; a.dir_val = c->dir_val.p; a.mat = c->mat;
  a.interleaved_u = c->interleaved_u;
  return a;
}
MfArgs mf_args(poro_ctx *c) {
  MfArgs a{}; a.dim = c->dim; a.k_u = c->k_u; a.box = c->box; a.Ke = c->Ke.p; a.mask = c->dir_mask.p; a.diag_local = c->diag_u_local.p;
  a.lam = c->mat.lame_lambda; a.G = c->mat.shear_G; a.mask_anywhere = c->mask_anywhere;
  a.nodemask = c->node_mask.p; a.dirichlet_dofs = c->dir_dofs.p; a.n_dirichlet = (int64_t)c->dir_dofs.n; return a;
}
// y = A_u x without forming A_u: sum-factorised sweeps where available, element-matrix gather otherwise
void mf_operator(poro_ctx *c, const double *x, double *y, bool constrained) {
  if (!c->box.enabled) {   // general mesh: quadrature-level cell loop; the Dirichlet rows from the constraint list as for the structured kernels
    mfg_apply(c->stream, asm_args(c), c->color_cells.p, c->color_off, c->n_u, x, y, constrained, 0);
    if (constrained) kron_fix_constrained(c->stream, mf_args(c), x, y, nullptr, 0);
    return;
  }
  if (c->mf_variant == 1 && kron_supported(c->dim, c->k_u)) { const int slots = kron_apply(c->stream, mf_args(c), x, y, constrained, c->n_cus); if (constrained) kron_fix_constrained(c->stream, mf_args(c), x, y, nullptr, std::abs(slots)); }
  else mf_apply(c->stream, mf_args(c), x, y, constrained);
}

double *vec(poro_ctx *c, int which) {
  auto it = c->vec.find(which);
  if (it == c->vec.end()) throw Error("unknown vector id " + std::to_string(which));
  return it->second.p;
}
int64_t vec_len(poro_ctx *c, int which) { return (int64_t)c->vec.at(which).n; }
bool is_u_vec(int which) { return which == PORO_VEC_U || which == PORO_VEC_RHS_U || which == PORO_VEC_DIAG_U; }

// y = A_u x (+ interface exchange).  dot_partials != null asks for the block partials of x.y; returns true when they were produced
// by the operator kernel itself (fused), false when the caller still has to launch the dot kernel.
bool apply_A_u(poro_ctx *c, const double *x, double *y, int mode, double *dot_partials, bool fix_rows, const PcgScalars *pcg_state, bool exchange) {
  bool fused = false;
  if (mode == PORO_OP_MATRIX_FREE && c->box.enabled && c->mf_variant == 1 && kron_supported(c->dim, c->k_u)) {
    int slots;
    if (c->timing && c->timers["apply_u_matrix_free"].sample(c->timing_stride)) {   // events attached to the dispatch itself: the kernel's own duration, without the gaps to its neighbours in the stream
      isolate_sampled_dispatch(c);
      Timer &t = c->timers["apply_u_matrix_free"]; hipEvent_t e0 = event_get(c), e1 = event_get(c);
      slots = kron_apply(c->stream, mf_args(c), x, y, true, c->n_cus, dot_partials, e0, e1, pcg_state);
      t.pending.emplace_back(e0, e1); t.launches++;
    } else slots = kron_apply(c->stream, mf_args(c), x, y, true, c->n_cus, dot_partials, nullptr, nullptr, pcg_state);
    // inside PCG the Dirichlet rows are inert (zero residual and direction), so what the structured kernel leaves there is never read
    fused = dot_partials != nullptr && slots > 0;   // slots < 0: too many workgroups for the partial slots, the kernel ran without the fused x.y
    if (fix_rows) { Timed tm(c, "apply_u_dirichlet_rows"); kron_fix_constrained(c->stream, mf_args(c), x, y, fused ? dot_partials : nullptr, slots > 0 ? slots : -slots); }
  } else if (mode == PORO_OP_MATRIX_FREE && !c->box.enabled) {
    Timed tm(c, "apply_u_matrix_free");
    mfg_apply(c->stream, asm_args(c), c->color_cells.p, c->color_off, c->n_u, x, y, true, 0);
    if (fix_rows) kron_fix_constrained(c->stream, mf_args(c), x, y, nullptr, 0);
  } else {
    Timed tm(c, mode == PORO_OP_MATRIX_FREE ? "apply_u_matrix_free" : "apply_u_csr");
    if (mode == PORO_OP_MATRIX_FREE) { mf_apply(c->stream, mf_args(c), x, y, true, dot_partials); fused = dot_partials != nullptr; }
    else la_csr_spmv(c->stream, c->Au, c->Au_val.p, x, y);
  }
  if (exchange) exchange_add(c, y, c->n_u, c->comm.part.plane_u);
  return fused;
}

}  // namespace ctx_detail
}  // namespace poro
