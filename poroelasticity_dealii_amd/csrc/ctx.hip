// Host logic of the HIP back end: context set-up (setup_dofs of the three solvers), the device-resident
// PCG loop, slab-partition communication (RCCL over xGMI, or host-staged callbacks for tests) and the
// extern "C" entry points of include/poroel_hip.h.
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

using namespace poro;

namespace {

thread_local std::string g_err;

int ipow(int b, int e) { int r = 1; while (e--) r *= b; return r; }

template <class F> void parallel_for(int64_t n, F &&f) {
  unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 32) nt = 32;
  if (n < 20000 || nt == 1) { f(0, n); return; }
  std::vector<std::thread> th; const int64_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) { const int64_t b = t * chunk, e = std::min<int64_t>(n, b + chunk); if (b < e) th.emplace_back([=, &f] { f(b, e); }); }
  for (auto &t : th) t.join();
}

// DoFTools::make_sparsity_pattern(keep_constrained_dofs = true): every dof couples with all dofs of its cells
void build_pattern(int64_t n, int64_t n_cells, int dpc, const int32_t *cell_dofs, std::vector<int64_t> &rp, std::vector<int32_t> &col, std::vector<int64_t> &diag) {
  std::vector<int64_t> cnt(n + 1, 0);
  for (int64_t i = 0; i < n_cells * dpc; ++i) cnt[cell_dofs[i] + 1]++;
  for (int64_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
  std::vector<int32_t> adj(cnt[n]);
  { std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int64_t c = 0; c < n_cells; ++c) for (int i = 0; i < dpc; ++i) adj[pos[cell_dofs[c * dpc + i]]++] = (int32_t)c; }
  rp.assign(n + 1, 0);
  auto row_cols = [&](int64_t r, std::vector<int32_t> &row) {
    row.clear();
    for (int64_t a = cnt[r]; a < cnt[r + 1]; ++a) { const int32_t *cd = cell_dofs + (int64_t)adj[a] * dpc; row.insert(row.end(), cd, cd + dpc); }
    std::sort(row.begin(), row.end()); row.erase(std::unique(row.begin(), row.end()), row.end());
  };
  parallel_for(n, [&](int64_t b, int64_t e) { std::vector<int32_t> row; for (int64_t r = b; r < e; ++r) { row_cols(r, row); rp[r + 1] = (int64_t)row.size(); } });
  for (int64_t r = 0; r < n; ++r) rp[r + 1] += rp[r];
  col.resize(rp[n]); diag.resize(n);
  parallel_for(n, [&](int64_t b, int64_t e) {
    std::vector<int32_t> row;
    for (int64_t r = b; r < e; ++r) {
      row_cols(r, row); std::copy(row.begin(), row.end(), col.begin() + rp[r]);
      diag[r] = rp[r] + (std::lower_bound(row.begin(), row.end(), (int32_t)r) - row.begin());
    }
  });
}

void upload_csr(CsrDev &A, int64_t n, const std::vector<int64_t> &rp, const std::vector<int32_t> &col, const std::vector<int64_t> &diag) {
  A.n = n; A.nnz = (int64_t)col.size(); A.rp.upload(rp); A.col.upload(col); A.diag_pos.upload(diag);
  const double avg = n ? (double)A.nnz / n : 1; int L = 2;
  while (L < 64 && L * 4 < avg) L *= 2;
  A.lanes_per_row = L;
}

// greedy colouring: cells of one colour share no vertex, hence no dof
void colour_cells(int64_t n_cells, int64_t n_vertices, int nv, const int32_t *cv, std::vector<int32_t> &cells_sorted, std::vector<int64_t> &off) {
  std::vector<int64_t> vp(n_vertices + 1, 0);
  for (int64_t i = 0; i < n_cells * nv; ++i) vp[cv[i] + 1]++;
  for (int64_t i = 0; i < n_vertices; ++i) vp[i + 1] += vp[i];
  std::vector<int32_t> vc(vp[n_vertices]);
  { std::vector<int64_t> pos(vp.begin(), vp.end() - 1); for (int64_t c = 0; c < n_cells; ++c) for (int v = 0; v < nv; ++v) vc[pos[cv[c * nv + v]]++] = (int32_t)c; }
  std::vector<int> colour(n_cells, -1); int ncol = 0;
  for (int64_t c = 0; c < n_cells; ++c) {
    uint64_t used = 0;
    for (int v = 0; v < nv; ++v) { const int32_t vx = cv[c * nv + v]; for (int64_t a = vp[vx]; a < vp[vx + 1]; ++a) { const int k = colour[vc[a]]; if (k >= 0) used |= (1ull << k); } }
    int k = 0; while (used & (1ull << k)) ++k;
    if (k >= 63) throw Error("colouring needs more than 63 colours");
    colour[c] = k; ncol = std::max(ncol, k + 1);
  }
  off.assign(ncol + 1, 0);
  for (int64_t c = 0; c < n_cells; ++c) off[colour[c] + 1]++;
  for (int k = 0; k < ncol; ++k) off[k + 1] += off[k];
  cells_sorted.resize(n_cells);
  { std::vector<int64_t> pos(off.begin(), off.end() - 1); for (int64_t c = 0; c < n_cells; ++c) cells_sorted[pos[colour[c]]++] = (int32_t)c; }
}

// ---- RCCL, resolved at run time so single-GPU use has no dependency on it ---------------------------------------
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
  void load() {
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
} g_rccl;
#define PORO_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) throw Error(std::string(#x) + " -> " + g_rccl.GetErrorString(r_)); } while (0)

// ---- timing -----------------------------------------------------------------------------------------------------
// HIP events on the launch stream around every kernel family, drawn from a pool so a timed launch costs two
// hipEventRecord calls; elapsed times are read back in bulk by timers_collect.  The pool belongs to the context (its device, its host thread).
hipEvent_t event_get(poro_ctx *c) {
  if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
  hipEvent_t e; PORO_HIP(hipEventCreate(&e)); return e;
}
struct Timed {
  poro_ctx *c; Timer *t = nullptr; hipEvent_t a = nullptr, b = nullptr;
  Timed(poro_ctx *c_, const char *name) : c(c_) {
    if (!c->timing) return;
    Timer *tt = &c->timers[name];
    if (!tt->sample(c->timing_stride)) return;
    t = tt;
    a = event_get(c); b = event_get(c); (void)hipEventRecord(a, c->stream);
  }
  ~Timed() { if (!t) return; (void)hipEventRecord(b, c->stream); t->pending.emplace_back(a, b); t->launches++; }
};
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
// a start / stop event pair that is returned to the context's pool on every exit path
struct EventPair {
  poro_ctx *c; hipEvent_t e0, e1;
  explicit EventPair(poro_ctx *c_) : c(c_), e0(event_get(c_)), e1(event_get(c_)) {}
  ~EventPair() { c->event_pool.push_back(e0); c->event_pool.push_back(e1); }
  EventPair(const EventPair &) = delete; EventPair &operator=(const EventPair &) = delete;
};

// ---- device -> host scalars without a copy engine or a stream synchronisation (Mailbox, common.hpp) ------------------------------------------
// enqueue the publishing kernel behind everything that is in the stream, then spin on the sequence number in pinned host memory
void post_and_wait(poro_ctx *c, const double *dev_src, int n, const PcgScalars *sc = nullptr) {
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
  a.cell_dofs_u = c->cell_dofs_u.p; a.cell_dofs_p = c->cell_dofs_p.p; a.cell_X = c->cell_X.p; a.dir_mask = c->dir_mask.p; a.dir_val = c->dir_val.p; a.mat = c->mat;
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
bool apply_A_u(poro_ctx *c, const double *x, double *y, int mode, double *dot_partials = nullptr, bool fix_rows = true, const PcgScalars *pcg_state = nullptr) {
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
  exchange_add(c, y, c->n_u, c->comm.part.plane_u);
  return fused;
}

// ---- single-reduction PCG for partitioned runs (Chronopoulos & Gear) ------------------------------------------------------------------
// Same Krylov space, same stopping test and same iteration count as SolverCG's recurrence in exact arithmetic, rearranged so that an iteration
// costs ONE all-reduce: z = P^-1 g, w = A z, then {g.z, w.z, g.g} in one reduction, then d = -z + beta d, s = -w + beta s (= A d), x += alpha d,
// g += alpha s.  Per iteration: 1 operator application (1 grouped neighbour exchange) + the exchanges inside P^-1 + 1 all-reduce of 4 doubles.
// The price is two more vector passes than pcg(), which is why single-rank runs keep the three-kernel recurrence.
int pcg_single_reduction(poro_ctx *c, const std::function<bool(const double *, double *, double *)> &apply, int64_t n, int64_t plane, double *x, const double *b,
                         const DiagVec &diag, double *g, double *d, double *sv, const poro_solver_opts *opts, poro_solve_info *info,
                         const std::function<bool(const double *, double *, double *)> *precond, int *its_hint) {
  hipStream_t s = c->stream;
  const int which = n == c->n_u ? 0 : 1;
  if (c->cg1_w[which].n < (size_t)n) { c->cg1_w[which].alloc(n); c->cg1_z[which].alloc(n); }
  if (!c->cg1_state.p) c->cg1_state.alloc(1);
  double *w = c->cg1_w[which].p, *z = (precond && diag.z) ? const_cast<double *>(diag.z) : c->cg1_z[which].p;
  if (precond && !diag.z) throw Error("pcg: explicit preconditioner without a z vector");
  const int64_t n_own = owned(c, n, plane);
  const bool jacobi = opts->preconditioner == PORO_PREC_JACOBI;
  Cg1State *st = c->cg1_state.p; double *part = c->partials.p, *red = c->red.p;
  EventPair ev(c); PORO_HIP(hipEventRecord(ev.e0, s));
  apply(x, w, nullptr);
  pcg_init_residual(s, g, w, b, diag.inert, n);        // g = A x - b, zero on the inert dofs
  la_fill(s, d, 0.0, n); la_fill(s, sv, 0.0, n);
  DiagVec upd = diag; if (jacobi && !precond) { upd.z1_out = z; upd.z1_scale = 1.0; }   // Jacobi: the update kernel also leaves z = D^-1 g_new for the next iteration (in place of z)
  Cg1State hs{};
  int expect = 0, enq = 0;
  if (its_hint && its_hint[0] > 0) { expect = its_hint[1] > 0 ? 2 * its_hint[0] - its_hint[1] : its_hint[0]; expect = std::min(std::max(expect, its_hint[0] / 2), its_hint[0]); }   // (never above the last count: nothing here is gated)
  auto next_batch = [&](int done_its) { const int left = expect - 4 - done_its; return left >= 4 ? std::min(32, left) : 2; };
  int batch = expect > 0 ? next_batch(0) : 1;
  while (true) {
    for (int k = 0; k < batch; ++k) {
      c->cheb_z1_ready = precond && enq > 0 && diag.z1_out != nullptr;      // stored by the previous cg1_update
      if (precond) (void)(*precond)(g, z, nullptr);
      else if (jacobi) { if (enq == 0) la_cheb_first(s, z, g, diag, 1.0, n); }   // z = D^-1 g (zero on the inert dofs); after the first iteration the update kernel stores it with the new residual
      else la_copy(s, z, g, n);
      apply(z, w, nullptr);
      cg1_dots(s, g, z, w, enq == 0 ? b : nullptr, n_own, part);
      pcg_scalars_sum(s, part, 4, red);
      allreduce_sum(c, red, 4);
      cg1_scalars(s, st, red, enq == 0 ? 1 : 0, opts->abs_tol, opts->rel_tol, opts->max_iter, opts->stop_rule);
      cg1_update(s, st, d, sv, x, g, z, w, upd, n);
      ++enq;
    }
    PORO_HIP(hipMemcpyAsync(&hs, st, sizeof(hs), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s));
    if (hs.done) break;
    // (the preconditioner and operator launches of an iteration are not gated by the device-side `done` flag: without a hint poll at least every 8 iterations)
    if (expect > 0) batch = next_batch(enq); else if (batch < 8) batch *= 2;
  }
  if (its_hint) { its_hint[1] = its_hint[0]; its_hint[0] = hs.it; }
  PORO_HIP(hipEventRecord(ev.e1, s)); PORO_HIP(hipEventSynchronize(ev.e1));
  float ms = 0; PORO_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  if (info) { info->iterations = hs.it; info->converged = hs.converged; info->initial_residual = hs.res0; info->final_residual = hs.res; info->seconds = ms * 1e-3;
              info->operator_applications = hs.it + 2; }   // initial residual + one per iteration + the one that found the converged residual
  return hs.converged ? 0 : 1;
}

// ---- PCG with device-side control: SolverCG<>::solve restated, Jacobi instead of SSOR ---------------
// apply(x, y, dot_partials) as apply_A_u.  The vector kernels compute alpha / beta / the stopping test in their prologues: from the
// block partials (single rank, 3 launches per iteration incl. the operator) or from the all-reduced scalars (partitioned).
// precond != null: explicit preconditioner z = P^-1 g (a sequence of launches on the stream, e.g. the fast diagonalisation) written into
// diag.z between the two update kernels; the scalars stay on the device exactly as in the Jacobi case.  precond(g, z, gz_partials) returns true
// when it has already left the block partials of g . z (over the owned rows) in gz_partials.
int pcg(poro_ctx *c, const std::function<bool(const double *, double *, double *)> &apply, int64_t n, int64_t plane, double *x, const double *b,
        const DiagVec &diag, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info,
        const std::function<bool(const double *, double *, double *)> *precond = nullptr, int *its_hint = nullptr, bool precond_gated = false,
        const FdmOct *oct = nullptr /* single rank, explicit preconditioner: the residual and z = P^-1 g live in octant form (kernels_fdmo.hip), `g` is unused */) {
  static const bool two_reductions = std::getenv("PORO_TWO_REDUCTION_CG") != nullptr;    // A/B hook: the three-kernel recurrence on partitioned runs too
  if (c->comm.multi() && !two_reductions) return pcg_single_reduction(c, apply, n, plane, x, b, diag, g, d, h, opts, info, precond, its_hint);
  hipStream_t s = c->stream;
  const int prec = opts->preconditioner == PORO_PREC_JACOBI ? 1 : 0;
  double *zbuf = const_cast<double *>(diag.z);
  if (oct) { if (!precond || c->comm.multi()) throw Error("pcg: the octant form needs an explicit preconditioner on one rank"); g = oct->g.p; zbuf = oct->z.p; }
  if (precond && !zbuf) throw Error("pcg: explicit preconditioner without a z vector");
  const int64_t n_own = owned(c, n, plane);
  const bool multi = c->comm.multi();
  double *part = c->partials.p, *red = c->red.p; PcgScalars *sc = c->scal.p;
  double *part_dh = part + 3 * (size_t)kMaxPartials;      // slots of the fused / separate d.h partials
  const auto t_start = std::chrono::steady_clock::now();
  int64_t applies = 0;
  // g = A x - b ; d = -P^-1 g ; gh = g.P^-1 g
  apply(x, h, nullptr); ++applies;
  if (oct) fdmo_init_residual(s, *oct, g, h, b, diag.inert); else pcg_init_residual(s, g, h, b, diag.inert, n);
  la_dot_partials(s, b, b, n_own, part);
  if (precond) (void)(*precond)(g, zbuf, nullptr);
  if (oct) fdmo_first_direction(s, *oct, d, g, zbuf, part + kMaxPartials); else pcg_first_direction(s, d, g, diag, prec, n, n_own, part + kMaxPartials);
  pcg_scalars_sum(s, part, 3, red);
  allreduce_sum(c, red, 3);
  pcg_scalars_start(s, sc, red, opts->abs_tol, opts->rel_tol, opts->max_iter, opts->stop_rule);
  PORO_HIP(hipMemsetAsync(part_dh, 0, kMaxPartials * sizeof(double), s));
  PcgScalars hs{};
  int it = 0;
  // Iterations are enqueued in batches, THEN the device-side state is polled (a host round trip idles the GPU for ~50 us).  Launches behind the finishing
  // iteration are no-ops (the vector kernels, the structured operator and the fused Chebyshev kernels test the device-side flag; ~1 us each), so where
  // everything is gated an overshoot is cheaper than a poll; an ungated explicit preconditioner (fast diagonalisation) is not, so its batches stop short.
  // Expected iteration count: linear extrapolation of the last two solves of this system (a transient's warm-started counts drift steadily).
  int expect = 0;
  if (its_hint && its_hint[0] > 0) { expect = its_hint[1] > 0 ? 2 * its_hint[0] - its_hint[1] : its_hint[0]; expect = std::max(expect, its_hint[0] / 2); }
  const bool cheap_overshoot = !precond || precond_gated;
  // the extrapolation must not run away after an atypical solve (a warm restart that took 3 iterations, followed by a real step): never expect more than a quarter
  // above the last count, and nothing above it where an overshoot is expensive
  if (expect > 0) expect = std::min(expect, cheap_overshoot ? its_hint[0] + std::max(2, its_hint[0] / 4) : its_hint[0]);
  int batch = expect > 0 ? (cheap_overshoot ? std::min(expect, 256) : std::max(1, expect - 1)) : (precond && !cheap_overshoot ? 1 : 4);   // no history: a poll (~15 us through the mailbox) every 4 iterations
  while (true) {
    for (int k = 0; k < batch; ++k) {
      ++it;
      // operator (+ fused or separate d.h partials).  A fused dot runs over ALL local rows of the pre-exchange partial product, which
      // sums to the global d.Ad over the ranks; the separate kernel sees the exchanged h and therefore skips the upper shared plane.
      if (!apply(d, h, part_dh)) pcg_dot_dh(s, sc, d, h, n_own, part_dh);
      ++applies;
      if (multi) { pcg_scalars_sum(s, part_dh, 1, red); allreduce_sum(c, red, 1); }
      if (oct) fdmo_update_g(s, *oct, sc, (it - 1) & 1, g, h, diag.inert, part_dh, part);
      else pcg_update_g_fused(s, sc, (it - 1) & 1, g, h, diag, prec, n, n_own, part_dh, multi ? red : nullptr, part);
      if (precond && !(*precond)(g, zbuf, part + kMaxPartials)) la_dot_partials(s, g, zbuf, oct ? oct->n_oct : n_own, part + kMaxPartials, precond_gated ? sc : nullptr);
      if (multi) { pcg_scalars_sum(s, part, 2, red + 1); allreduce_sum(c, red + 1, 2); }
      if (oct) fdmo_update_d(s, *oct, sc, (it - 1) & 1, it, x, d, zbuf, part);
      else pcg_update_d_fused(s, sc, (it - 1) & 1, it, x, d, g, diag, prec, n, part, multi ? red + 1 : nullptr);
    }
    post_and_wait(c, nullptr, 0, sc); hs = c->mailbox->sc;
    if (hs.done || hs.finishing) break;
    if (expect > 0) batch = cheap_overshoot ? 3 : 1;
    else if (cheap_overshoot && precond) batch = 4;          // (an explicit preconditioner: a no-op iteration still costs ~8 launches)
    else if (batch < 32) batch *= 2;
  }
  if (its_hint) { its_hint[1] = its_hint[0]; its_hint[0] = hs.it; }
  // (the last poll returned after the finishing iteration: the solve is complete on the device; wall time of the solve on the host clock)
  if (info) { info->iterations = hs.it; info->converged = hs.converged; info->initial_residual = hs.res0; info->final_residual = hs.res;
              info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
              info->operator_applications = hs.it + 1;   // initial residual + one per iteration (launches enqueued behind the finishing iteration are no-ops and are not counted)
              (void)applies; }
  return hs.converged ? 0 : 1;
}

// ---- PreconditionSSOR fidelity mode: SolverCG with the reference's SSOR(omega) in natural row order ---------------------------
void build_ssor_levels(poro_ctx *c, CsrDev &A) {
  if (A.ssor.built) return;
  std::vector<int64_t> rp(A.n + 1); std::vector<int32_t> col(A.nnz);
  PORO_HIP(hipMemcpy(rp.data(), A.rp.p, (A.n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost)); PORO_HIP(hipMemcpy(col.data(), A.col.p, A.nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  auto levels = [&](bool fwd, DevBuf<int32_t> &rows_dev, std::vector<int64_t> &off) {
    std::vector<int32_t> lvl(A.n, 0); int maxl = 0;
    if (fwd) for (int64_t r = 0; r < A.n; ++r) { int l = 0; for (int64_t j = rp[r]; j < rp[r + 1] && col[j] < r; ++j) l = std::max(l, lvl[col[j]] + 1); lvl[r] = l; maxl = std::max(maxl, l); }
    else for (int64_t r = A.n - 1; r >= 0; --r) { int l = 0; for (int64_t j = rp[r + 1] - 1; j >= rp[r] && col[j] > r; --j) l = std::max(l, lvl[col[j]] + 1); lvl[r] = l; maxl = std::max(maxl, l); }
    off.assign(maxl + 2, 0);
    for (int64_t r = 0; r < A.n; ++r) off[lvl[r] + 1]++;
    for (int l = 0; l <= maxl; ++l) off[l + 1] += off[l];
    std::vector<int32_t> rows(A.n); std::vector<int64_t> pos(off.begin(), off.end() - 1);
    for (int64_t r = 0; r < A.n; ++r) rows[pos[lvl[r]]++] = (int32_t)r;
    rows_dev.upload(rows);
  };
  levels(true, A.ssor.fwd_rows, A.ssor.fwd_off); levels(false, A.ssor.bwd_rows, A.ssor.bwd_off);
  A.ssor.built = true;
}
// global dot product on the host; n = rows this rank owns (the upper shared plane belongs to the neighbour)
double dot_host(poro_ctx *c, const double *a, const double *b, int64_t n) {
  la_dot_partials(c->stream, a, b, n, c->partials.p); la_reduce_finish(c->stream, c->partials.p, 1, c->red.p, 0);
  allreduce_sum(c, c->red.p, 1);
  post_and_wait(c, c->red.p, 1);
  return c->mailbox->vals[0];
}
// SolverCG<>::solve with an explicit preconditioner z = P^-1 g, host-driven scalars.  Used where an application of P^-1 is many
// launches anyway (SSOR sweeps) or where only a handful of iterations happen (fast diagonalisation).  Partitioned runs: `apply` and
// `precond` return vectors that are consistent on the shared planes; dots run over the `n_own` owned rows and are all-reduced.
int pcg_host(poro_ctx *c, int64_t n, int64_t n_own, const std::function<void(const double *, double *)> &apply, const std::function<void(const double *, double *)> &precond,
             double *x, const double *b, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info) {
  hipStream_t s = c->stream;
  const auto t0 = std::chrono::steady_clock::now();
  int64_t applies = 0; int it = 0, conv = 0;
  apply(x, g); ++applies;
  la_axpy(s, g, -1.0, b, n);                                     // g = A x - b
  double res = std::sqrt(dot_host(c, g, g, n_own)); const double res0 = res;
  const double tol = std::max(opts->abs_tol, opts->rel_tol * (opts->stop_rule == PORO_STOP_REDUCTION ? res0 : std::sqrt(dot_host(c, b, b, n_own))));
  if (res <= tol) conv = 1;
  else {
    precond(g, h);
    la_fill(s, d, 0.0, n); la_axpy(s, d, -1.0, h, n);          // d = -h
    double gh = dot_host(c, g, h, n_own);
    while (true) {
      ++it;
      apply(d, h); ++applies;
      const double alpha = gh / dot_host(c, d, h, n_own);
      la_axpy(s, g, alpha, h, n); la_axpy(s, x, alpha, d, n);
      res = std::sqrt(dot_host(c, g, g, n_own));
      if (res <= tol) { conv = 1; break; }
      if (it >= opts->max_iter) break;
      precond(g, h);
      const double beta_old = gh; gh = dot_host(c, g, h, n_own);
      la_xpby(s, d, gh / beta_old, -1.0, h, n);                   // d = beta d - h
    }
  }
  PORO_HIP(hipStreamSynchronize(s));
  if (info) { info->iterations = it; info->converged = conv; info->initial_residual = res0; info->final_residual = res; info->operator_applications = applies;
              info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  return conv ? 0 : 1;
}
int pcg_ssor(poro_ctx *c, CsrDev &A, const double *val, double *x, const double *b, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info) {
  if (c->comm.multi()) throw Error("PORO_PREC_SSOR is a single-rank fidelity mode (the sweeps are order dependent)");
  build_ssor_levels(c, A);
  const double om = opts->omega > 0 ? opts->omega : 1.0;
  return pcg_host(c, A.n, A.n, [&](const double *v, double *y) { la_csr_spmv(c->stream, A, val, v, y); },
                  [&](const double *gg, double *z) { la_ssor_apply(c->stream, A, val, A.ssor, om, gg, z); }, x, b, g, d, h, opts, info);
}

// ---- ILU(0): factorisation and solves on the device, both level-scheduled in the natural row order (la_ilu0_factor, la_ilu_apply) -----------------
void ilu0_factor(poro_ctx *c, const CsrDev &A, const double *val, DevBuf<double> &lu_dev) {
  std::vector<int64_t> rp(A.n + 1);
  PORO_HIP(hipMemcpy(rp.data(), A.rp.p, (A.n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  int64_t longest = 0; for (int64_t i = 0; i < A.n; ++i) longest = std::max(longest, rp[i + 1] - rp[i]);
  if (longest > 512) throw Error("ILU(0): rows longer than 512 entries are not supported by the device factorisation");
  if (lu_dev.n < (size_t)A.nnz) lu_dev.alloc(A.nnz);
  DevBuf<int> flag; flag.alloc(1); flag.zero(c->stream);
  PORO_HIP(hipMemcpyAsync(lu_dev.p, val, A.nnz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  la_ilu0_factor(c->stream, A, A.ssor, lu_dev.p, flag.p);
  int h = 0; PORO_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
  if (h) throw Error("ILU(0): zero pivot in row " + std::to_string(h - 1));
}
int pcg_ilu0(poro_ctx *c, CsrDev &A, const double *val, DevBuf<double> &lu, bool &valid, double *x, const double *b, double *g, double *d, double *h,
             const poro_solver_opts *opts, poro_solve_info *info) {
  if (c->comm.multi()) throw Error("PORO_PREC_ILU0 is implemented for one rank (the factorisation is sequential in the row order)");
  build_ssor_levels(c, A);
  if (!valid) { ilu0_factor(c, A, val, lu); valid = true; }
  return pcg_host(c, A.n, A.n, [&](const double *v, double *y) { la_csr_spmv(c->stream, A, val, v, y); },
                  [&](const double *gg, double *z) { la_ilu_apply(c->stream, A, lu.p, A.ssor, gg, z); }, x, b, g, d, h, opts, info);
}

// ---- fast diagonalisation of the Q1 box operators (kernels_fdm.hip) -------------------------------------------------------------
bool fdm_p_supported(poro_ctx *c) {
  if (!c->box.enabled) return false;
  if (c->comm.multi() && !(c->comm.nccl_comm || (c->comm.ar && c->comm.sr))) return false;
  return true;
}
static void upload_dir(FdmDir &D, int n_cells, double h) {
  std::vector<double> S, lam; q1_eig(n_cells, h, S, lam);
  const int n = n_cells + 1;
  std::vector<double> St((size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) St[(size_t)j * n + i] = S[(size_t)i * n + j];
  D.n = n; D.S.upload(S); D.St.upload(St); D.lam.upload(lam);
}
void build_fdm_p(poro_ctx *c) {
  if (c->fdm_p.built) return;
  if (!fdm_p_supported(c)) throw Error("PORO_PREC_FDM needs a uniform box (poro_desc.box.enabled) and, when partitioned, an initialised communicator");
  c->fdm_p.dim = c->dim;
  for (int d = 0; d < c->dim; ++d) upload_dir(c->fdm_p.dir[d], c->box.n[d], c->box.h[d]);   // local slab; the last direction is replaced below when partitioned
  c->fdm_t1.alloc(c->n_p); c->fdm_t2.alloc(c->n_p);
  if (c->comm.multi()) {
    FdmDist &F = c->fdm_dist; const int N = std::max(1, c->comm.part.n_ranks), r = c->comm.part.rank, last = c->dim - 1;
    F.n_ranks = N; F.rank = r;
    // every rank learns all slab thicknesses through the existing all-reduce
    std::vector<double> lay(N, 0.0); lay[r] = c->box.n[last];
    DevBuf<double> tmp; tmp.upload(lay);
    for (int base = 0; base < N; base += kScalarSlots) {
      const int m = std::min(kScalarSlots, N - base);
      PORO_HIP(hipMemcpyAsync(c->red.p, tmp.p + base, m * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      allreduce_sum(c, c->red.p, m);
      PORO_HIP(hipMemcpyAsync(lay.data() + base, c->red.p, m * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
    }
    F.layers.resize(N); F.off.resize(N); int acc = 0;
    for (int q = 0; q < N; ++q) { F.layers[q] = (int)std::lround(lay[q]); F.off[q] = acc; acc += F.layers[q]; }
    F.ng = acc + 1;
    F.ncol_total = 1; for (int d = 0; d < last; ++d) F.ncol_total *= c->box.n[d] + 1;
    F.C = (F.ncol_total + N - 1) / N;
    F.max_own = 0; F.max_nl = 0;
    for (int q = 0; q < N; ++q) { F.max_own = std::max(F.max_own, F.layers[q] + (q == N - 1 ? 1 : 0)); F.max_nl = std::max(F.max_nl, F.layers[q] + 1); }
    upload_dir(F.last, acc, c->box.h[last]);
    const size_t blk = (size_t)std::max(F.max_own, F.max_nl) * F.C;
    F.sendbuf.alloc(blk * N); F.recvbuf.alloc(blk * N); F.tz1.alloc((size_t)F.ng * F.C); F.tz2.alloc((size_t)F.ng * F.C);
    F.built = true;
  }
  c->fdm_p.built = true;
}
// every rank sends block q of `send` (blk doubles) to rank q and receives block q of `recv` from it
void alltoall_blocks(poro_ctx *c, double *send, double *recv, int64_t blk) {
  Comm &cm = c->comm; const int N = cm.part.n_ranks, r = cm.part.rank;
  Timed tm(c, "alltoall");
  PORO_HIP(hipMemcpyAsync(recv + (size_t)r * blk, send + (size_t)r * blk, blk * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  if (N <= 1) return;
  if (cm.nccl_comm) {
    ncclComm_t comm = (ncclComm_t)cm.nccl_comm;
    PORO_NCCL(g_rccl.GroupStart());
    for (int q = 0; q < N; ++q) if (q != r) {
      PORO_NCCL(g_rccl.Send(send + (size_t)q * blk, blk, ncclFloat64, q, comm, c->stream));
      PORO_NCCL(g_rccl.Recv(recv + (size_t)q * blk, blk, ncclFloat64, q, comm, c->stream));
    }
    PORO_NCCL(g_rccl.GroupEnd());
  } else if (cm.sr) {
    FdmDist &F = c->fdm_dist; if ((int64_t)F.hsend.size() < blk) { F.hsend.resize(blk); F.hrecv.resize(blk); }
    for (int step = 0; step < N; ++step) {                       // pairwise schedule: at step s rank r meets (s - r) mod N, which meets r
      const int q = ((step - r) % N + N) % N;
      if (q == r) continue;
      PORO_HIP(hipMemcpyAsync(F.hsend.data(), send + (size_t)q * blk, blk * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
      cm.sr(F.hsend.data(), F.hrecv.data(), blk, q, cm.user);
      PORO_HIP(hipMemcpyAsync(recv + (size_t)q * blk, F.hrecv.data(), blk * sizeof(double), hipMemcpyHostToDevice, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
    }
  } else throw Error("partitioned context without a communicator");
}
// z = (a M + sum_d k_d K_d)^-1 g for the Q1 space of the (global) box
void fdm_precondition_p(poro_ctx *c, double a, const double k[3], const double *g, double *z) {
  Timed tm(c, "precondition_p_fdm");
  hipStream_t s = c->stream;
  if (!c->comm.multi()) { fdm_apply(s, c->fdm_p, a, k, g, z, c->fdm_t1.p, c->fdm_t2.p); return; }
  FdmDist &F = c->fdm_dist; const FdmScalar &L = c->fdm_p;
  const int dim = c->dim, N = F.n_ranks, r = F.rank, last = dim - 1;
  const int n0 = L.dir[0].n, nl = c->box.n[last] + 1;             // local planes incl. the shared ones
  const int64_t SIp = F.ncol_total;
  double *t1 = c->fdm_t1.p, *t2 = c->fdm_t2.p;
  // leading directions: local (the shared planes are transformed by both owners)
  const double *cur = g;
  if (dim == 3) { fdm_transform(s, L.dir[0].St.p, n0, 1, (int64_t)L.dir[1].n * nl, g, t1, nullptr); fdm_transform(s, L.dir[1].St.p, L.dir[1].n, n0, nl, t1, t2, nullptr); cur = t2; }
  else { fdm_transform(s, L.dir[0].St.p, n0, 1, nl, g, t1, nullptr); cur = t1; }
  // gather whole lines of the last direction for this rank's column group
  const int own_r = F.layers[r] + (r == N - 1 ? 1 : 0);
  const int64_t blk1 = (int64_t)F.max_own * F.C;
  auto ncols_of = [&](int q) { return std::max<int64_t>(0, std::min<int64_t>(F.C, F.ncol_total - (int64_t)q * F.C)); };
  for (int q = 0; q < N; ++q) fdm_window(s, F.sendbuf.p + (size_t)q * blk1, cur, true, own_r, F.max_own, F.C, ncols_of(q), SIp, (int64_t)q * F.C, 0);
  alltoall_blocks(c, F.sendbuf.p, F.recvbuf.p, blk1);
  PORO_HIP(hipMemsetAsync(F.tz1.p, 0, (size_t)F.ng * F.C * sizeof(double), s));
  for (int q = 0; q < N; ++q) fdm_window(s, F.tz1.p, F.recvbuf.p + (size_t)q * blk1, false, F.layers[q] + (q == N - 1 ? 1 : 0), F.max_own, F.C, F.C, F.C, 0, F.off[q]);
  FdmScale sc{}; sc.a = a; sc.ncol = F.C; sc.col0 = (int64_t)r * F.C; sc.col_total = F.ncol_total;
  for (int d = 0; d < 3; ++d) { sc.lam[d] = d < dim ? (d == last ? F.last.lam.p : L.dir[d].lam.p) : nullptr; sc.k[d] = d < dim ? k[d] : 0.0; sc.n[d] = d < dim ? (d == last ? F.ng : L.dir[d].n) : 1; }
  fdm_transform(s, F.last.St.p, F.ng, F.C, 1, F.tz1.p, F.tz2.p, &sc);
  fdm_transform(s, F.last.S.p, F.ng, F.C, 1, F.tz2.p, F.tz1.p, nullptr);
  // scatter back: every rank gets all of its planes (shared ones included) of every column group
  const int64_t blk2 = (int64_t)F.max_nl * F.C;
  for (int q = 0; q < N; ++q) fdm_window(s, F.sendbuf.p + (size_t)q * blk2, F.tz1.p, true, F.layers[q] + 1, F.max_nl, F.C, F.C, F.C, 0, F.off[q]);
  alltoall_blocks(c, F.sendbuf.p, F.recvbuf.p, blk2);
  double *back = dim == 3 ? t2 : t1;
  for (int q = 0; q < N; ++q) fdm_window(s, back, F.recvbuf.p + (size_t)q * blk2, false, nl, F.max_nl, F.C, ncols_of(q), SIp, (int64_t)q * F.C, 0);
  if (dim == 3) { fdm_transform(s, L.dir[1].S.p, L.dir[1].n, n0, nl, t2, t1, nullptr); fdm_transform(s, L.dir[0].S.p, n0, 1, (int64_t)L.dir[1].n * nl, t1, z, nullptr); }
  else fdm_transform(s, L.dir[0].S.p, n0, 1, nl, t1, z, nullptr);
}

// ---- block fast diagonalisation of the displacement system (kernels_fdmu.hip) ---------------------------------------------------------
// Usable when the box is node-interleaved and, per component, the Dirichlet dofs are exactly a union of whole faces (then the 1D matrices
// of that component just lose their end nodes) with at least one face each (otherwise the block is singular).
void analyse_fdm_u(poro_ctx *c) {
  if (c->fdm_u_state != 0) return;
  c->fdm_u_state = -1;
  const bool multi = c->comm.multi();
  if (multi && !(c->comm.nccl_comm || (c->comm.ar && c->comm.sr))) { c->fdm_u_state = 0; c->fdm_u_why = "partitioned context without a communicator yet"; return; }
  const int dim = c->dim, last = dim - 1; const int64_t nn[3] = {c->box.nn[0], c->box.nn[1], dim == 3 ? c->box.nn[2] : 1};
  std::string why;
  FdmU &F = c->fdm_u;
  if (!c->box.enabled || !c->interleaved_u) why = "needs a uniform box with node-interleaved displacement dofs";
  else {
    for (int d = 0; d < dim; ++d) if (nn[d] > 4096) why = "more than 4096 nodes per grid line";
  }
  if (why.empty()) {
    const std::vector<uint8_t> &nm = c->h_node_mask;
    auto node = [&](int64_t i, int64_t j, int64_t k) { return (k * nn[1] + j) * nn[0] + i; };
    // a face of the partitioned direction is a physical boundary only at the first / last rank
    auto physical = [&](int d, int side) { return !(multi && d == last && (side == 0 ? c->comm.part.has_lower : c->comm.part.has_upper)); };
    for (int comp = 0; comp < dim && why.empty(); ++comp) {
      for (int d = 0; d < dim; ++d) for (int side = 0; side < 2; ++side) {
        bool all = physical(d, side);
        const int64_t fixed = side ? nn[d] - 1 : 0;
        const int d1 = (d + 1) % 3, d2 = (d + 2) % 3;
        for (int64_t a = 0; a < nn[d1] && all; ++a) for (int64_t b = 0; b < nn[d2]; ++b) {
          int64_t ix[3]; ix[d] = fixed; ix[d1] = a; ix[d2] = b;
          if (!(nm[node(ix[0], ix[1], ix[2])] >> comp & 1)) { all = false; break; }
        }
        F.fix[comp][d][side] = all ? 1 : 0;
      }
      for (int64_t k = 0; k < nn[2] && why.empty(); ++k) for (int64_t j = 0; j < nn[1] && why.empty(); ++j) for (int64_t i = 0; i < nn[0]; ++i) {
        const int64_t ix[3] = {i, j, k}; bool on = false;
        for (int d = 0; d < dim; ++d) on = on || (ix[d] == 0 && F.fix[comp][d][0]) || (ix[d] == nn[d] - 1 && F.fix[comp][d][1]);
        if (on != (bool)(nm[node(i, j, k)] >> comp & 1)) { why = "Dirichlet dofs are not a union of whole faces per component"; break; }
      }
    }
  }
  // ranks agree on the verdict and on the face flags (the end faces of the partitioned direction live on the first / last rank only)
  if (multi) {
    double h[kScalarSlots] = {0}; int m = 0;
    for (int comp = 0; comp < 3; ++comp) for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) h[m++] = (comp < dim && d < dim) ? F.fix[comp][d][side] : 0;
    h[m++] = why.empty() ? 0.0 : 1.0;
    PORO_HIP(hipMemcpyAsync(c->red.p, h, m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    allreduce_sum(c, c->red.p, m);
    PORO_HIP(hipMemcpyAsync(h, c->red.p, m * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
    m = 0;
    for (int comp = 0; comp < 3; ++comp) for (int d = 0; d < 3; ++d) for (int side = 0; side < 2; ++side) { if (comp < dim && d < dim) F.fix[comp][d][side] = h[m] > 0.5 ? 1 : 0; ++m; }
    if (h[m] > 0.5 && why.empty()) why = "another rank's Dirichlet dofs are not face-separable";
  }
  if (why.empty()) for (int comp = 0; comp < dim; ++comp) {
    bool any = false;
    for (int d = 0; d < dim; ++d) any = any || F.fix[comp][d][0] || F.fix[comp][d][1];
    if (!any) why = "a displacement component without a constrained face (singular block)";
  }
  c->fdm_u_why = why;
  c->fdm_u_state = why.empty() ? 1 : -1;
}
void build_fdm_u(poro_ctx *c) {
  FdmU &F = c->fdm_u;
  if (F.built) return;
  analyse_fdm_u(c);
  if (c->fdm_u_state != 1) throw Error("PORO_PREC_FDM (displacement): " + c->fdm_u_why);
  const int dim = c->dim, last = dim - 1, ku = c->k_u;
  const bool multi = c->comm.multi();
  F.dim = dim; F.single = !multi && std::getenv("PORO_FDMU_SINGLE") != nullptr;   // fp32 transforms (experimental switch, one rank)
  for (int d = 0; d < 3; ++d) F.nn[d] = d < dim ? c->box.nn[d] : 1;
  const double l2g = c->mat.lame_lambda + 2 * c->mat.shear_G, G = c->mat.shear_G;
  for (int comp = 0; comp < dim; ++comp) for (int d = 0; d < dim; ++d) F.coef[comp][d] = d == comp ? l2g : G;
  int n_cells_last = c->box.n[last];
  if (multi) {
    // every rank learns all slab thicknesses through the existing all-reduce
    const int N = std::max(1, c->comm.part.n_ranks), r = c->comm.part.rank;
    F.dist = true; F.n_ranks = N; F.rank = r;
    std::vector<double> lay(N, 0.0); lay[r] = c->box.n[last];
    DevBuf<double> tmp; tmp.upload(lay);
    for (int base = 0; base < N; base += kScalarSlots) {
      const int m = std::min(kScalarSlots, N - base);
      PORO_HIP(hipMemcpyAsync(c->red.p, tmp.p + base, m * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      allreduce_sum(c, c->red.p, m);
      PORO_HIP(hipMemcpyAsync(lay.data() + base, c->red.p, m * sizeof(double), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
    }
    F.layers.resize(N); F.off.resize(N); int acc = 0;
    for (int q = 0; q < N; ++q) { F.layers[q] = (int)std::lround(lay[q]); F.off[q] = ku * acc; acc += F.layers[q]; }
    n_cells_last = acc; F.ng = ku * acc + 1;
    if (F.ng > 4096) throw Error("PORO_PREC_FDM (displacement): more than 4096 nodes per global grid line");
    F.ncol_total = 1; for (int d = 0; d < last; ++d) F.ncol_total *= F.nn[d];
    F.C = (F.ncol_total + N - 1) / N;
    F.max_own = 0; F.max_nl = 0;
    for (int q = 0; q < N; ++q) { F.max_own = std::max(F.max_own, ku * F.layers[q] + (q == N - 1 ? 1 : 0)); F.max_nl = std::max(F.max_nl, ku * F.layers[q] + 1); }
    const size_t blk = (size_t)dim * std::max(F.max_own, F.max_nl) * F.C;
    F.sendbuf.alloc(blk * N); F.recvbuf.alloc(blk * N); F.tz1.alloc((size_t)dim * F.ng * F.C); F.tz2.alloc((size_t)dim * F.ng * F.C);
    F.sendbuf.zero(c->stream); F.recvbuf.zero(c->stream); F.tz1.zero(c->stream); F.tz2.zero(c->stream);
  }
  // octant form (kernels_fdmo.hip): one rank, 3D, every direction mirror-symmetric for every component, half lines of at most 80 entries
  bool oct_ok = !multi && !F.single && !std::getenv("PORO_FDMU_NO_OCT");
  { int nn3[3] = {F.nn[0], F.nn[1], F.nn[2]}; oct_ok = oct_ok && fdmo_usable(dim, nn3);
    for (int d = 0; d < dim && oct_ok; ++d) for (int comp = 0; comp < dim; ++comp) oct_ok = oct_ok && F.fix[comp][d][0] == F.fix[comp][d][1];
    if (oct_ok) fdmo_init(c->fdm_oct, nn3, F.coef, c->stream); }
  // eigenpairs per (direction, end conditions); components with the same end conditions share the host work.  A direction takes the even / odd
  // form (half the MFMA work) when every component has the same condition at both ends there - all components of a pass share one kernel
  for (int d = 0; d < dim; ++d) {
    std::vector<double> S[4], lam[4]; bool have[4] = {false, false, false, false};
    const bool global_dir = multi && d == last;
    const int ncell = global_dir ? n_cells_last : c->box.n[d], nnode = ku * ncell + 1;
    bool allow_split = true;
    for (int comp = 0; comp < dim; ++comp) allow_split = allow_split && F.fix[comp][d][0] == F.fix[comp][d][1];
    for (int attempt = 0; attempt < 2; ++attempt) {
      bool all_split = true;
      for (int comp = 0; comp < dim; ++comp) {
        const int key = F.fix[comp][d][0] * 2 + F.fix[comp][d][1];
        if (!have[key]) { fdmu_eig_1d(ku, ncell, c->box.h[d], F.fix[comp][d][0], F.fix[comp][d][1], S[key], lam[key]); have[key] = true; }
        FdmuDir &D = global_dir ? F.last_global[comp] : F.dir[comp][d];
        fdmu_upload_dir(D, S[key], lam[key], nnode, global_dir ? false : F.single, allow_split);
        all_split = all_split && D.split;
        if (oct_ok && attempt == 0) oct_ok = fdmo_upload_dir(c->fdm_oct, comp, d, S[key], lam[key], nnode);
      }
      if (!allow_split || all_split) break;
      allow_split = false;                       // the numerical symmetry check failed for some component: the whole direction in the full form
    }
  }
  c->fdmu_t1.alloc(c->n_u); c->fdmu_t2.alloc(c->n_u); c->fdmu_t1.zero(c->stream); c->fdmu_t2.zero(c->stream);   // (only finite values ever live in the scratch arrays)
  if (!c->wz_u.p) { c->wz_u.alloc(c->n_u); c->wz_u.zero(c->stream); }
  if (oct_ok) fdmo_finalize(c->fdm_oct);
  c->fdm_oct.built = oct_ok;
  F.built = true;
}
void alltoall_blocks(poro_ctx *c, double *send, double *recv, int64_t blk);
void fdm_precondition_u(poro_ctx *c, const double *g, double *z) {
  Timed tm(c, "precondition_u_fdm");
  hipStream_t s = c->stream; FdmU &F = c->fdm_u;
  if (!F.dist) { fdmu_apply(s, F, g, z, c->fdmu_t1.p, c->fdmu_t2.p, 2); return; }
  // leading directions locally (the shared planes are transformed by both owners), then whole lines of the partitioned direction for this
  // rank's column group: gather by an all-to-all, fused forward / scale / backward pass with the GLOBAL 1D eigenvectors, scatter back
  const int dim = F.dim, N = F.n_ranks, r = F.rank, last = dim - 1, ku = c->k_u;
  const int nl = F.nn[last];
  fdmu_apply(s, F, g, z, c->fdmu_t1.p, c->fdmu_t2.p, 0);
  double *cur = dim == 3 ? c->fdmu_t2.p : c->fdmu_t1.p;
  auto ncols_of = [&](int q) { return std::max<int64_t>(0, std::min<int64_t>(F.C, F.ncol_total - (int64_t)q * F.C)); };
  auto own_of = [&](int q) { return ku * F.layers[q] + (q == N - 1 ? 1 : 0); };
  const int64_t blk1 = (int64_t)dim * F.max_own * F.C;
  for (int q = 0; q < N; ++q) fdmu_window(s, F.sendbuf.p + (size_t)q * blk1, cur, true, dim, own_of(r), F.max_own, F.C, ncols_of(q), F.ncol_total, nl, (int64_t)q * F.C, 0);
  alltoall_blocks(c, F.sendbuf.p, F.recvbuf.p, blk1);
  PORO_HIP(hipMemsetAsync(F.tz1.p, 0, (size_t)dim * F.ng * F.C * sizeof(double), s));
  for (int q = 0; q < N; ++q) fdmu_window(s, F.tz1.p, F.recvbuf.p + (size_t)q * blk1, false, dim, own_of(q), F.max_own, F.C, F.C, F.C, F.ng, 0, F.off[q]);
  fdmu_lines(s, F, F.last_global, F.C, (int64_t)r * F.C, ncols_of(r), F.tz1.p, F.tz2.p);
  const int64_t blk2 = (int64_t)dim * F.max_nl * F.C;
  for (int q = 0; q < N; ++q) fdmu_window(s, F.sendbuf.p + (size_t)q * blk2, F.tz2.p, true, dim, ku * F.layers[q] + 1, F.max_nl, F.C, F.C, F.C, F.ng, 0, F.off[q]);
  alltoall_blocks(c, F.sendbuf.p, F.recvbuf.p, blk2);
  for (int q = 0; q < N; ++q) fdmu_window(s, cur, F.recvbuf.p + (size_t)q * blk2, false, dim, nl, F.max_nl, F.C, ncols_of(q), F.ncol_total, nl, (int64_t)q * F.C, 0);
  fdmu_apply(s, F, g, z, c->fdmu_t1.p, c->fdmu_t2.p, 1);
}

// poro_constraints -> device lists (+ transposed lists for the gather form of C^T y); `fixed` = byte mask of the Dirichlet dofs of the same space (or null)
void upload_constraints(ConsDev &C, const poro_constraints &h, int64_t n_dofs, const std::vector<uint8_t> *fixed, const char *what) {
  C.n = h.n;
  std::vector<uint8_t> inert(n_dofs, 0);
  if (fixed) inert = *fixed;
  if (h.n < 0) throw Error(std::string(what) + ": negative constraint count");
  if (h.n == 0) { if (fixed) C.inert.upload(inert); return; }
  if (!h.dof || !h.ptr || !h.inhomogeneity) throw Error(std::string(what) + ": null constraint arrays");
  std::vector<uint8_t> hanging(n_dofs, 0);
  for (int64_t i = 0; i < h.n; ++i) {
    const int32_t dof = h.dof[i];
    if (dof < 0 || dof >= n_dofs) throw Error(std::string(what) + ": constrained dof out of range");
    if (hanging[dof]) throw Error(std::string(what) + ": dof constrained twice");
    if (fixed && (*fixed)[dof]) throw Error(std::string(what) + ": dof is both in the Dirichlet list and in the constraint list");
    hanging[dof] = 1;
  }
  const int64_t nm = h.ptr[h.n];
  if (h.ptr[0] != 0 || nm < 0 || (nm && (!h.master || !h.weight))) throw Error(std::string(what) + ": bad constraint offsets");
  std::map<int32_t, std::vector<std::pair<int32_t, double>>> tr;
  for (int64_t i = 0; i < h.n; ++i) {
    if (h.ptr[i + 1] < h.ptr[i]) throw Error(std::string(what) + ": constraint offsets not ascending");
    for (int64_t k = h.ptr[i]; k < h.ptr[i + 1]; ++k) {
      const int32_t m = h.master[k];
      if (m < 0 || m >= n_dofs) throw Error(std::string(what) + ": master dof out of range");
      if (hanging[m] || (fixed && (*fixed)[m])) throw Error(std::string(what) + ": constraints are not closed (a master is itself constrained)");
      tr[m].emplace_back(h.dof[i], h.weight[k]);
    }
    if (h.inhomogeneity[i] != 0.0) C.any_inhom = true;
  }
  C.dof.upload(h.dof, h.n); C.ptr.upload(h.ptr, h.n + 1); C.inhom.upload(h.inhomogeneity, h.n);
  if (nm) { C.master.upload(h.master, nm); C.weight.upload(h.weight, nm); }
  std::vector<int32_t> tm, td; std::vector<int64_t> tp{0}; std::vector<double> tw;
  for (auto &kv : tr) { tm.push_back(kv.first); for (auto &e : kv.second) { td.push_back(e.first); tw.push_back(e.second); } tp.push_back((int64_t)td.size()); }
  C.n_masters = (int64_t)tm.size();
  if (C.n_masters) { C.t_master.upload(tm); C.t_dof.upload(td); C.t_ptr.upload(tp); C.t_weight.upload(tw); }
  for (int64_t i = 0; i < n_dofs; ++i) inert[i] = inert[i] | hanging[i];
  C.inert.upload(inert);
}

void setup(poro_ctx *c, const poro_desc *d) {
  if (d->abi_version != PORO_ABI_VERSION) throw Error("poro_desc.abi_version mismatch");
  if (d->dim != 2 && d->dim != 3) throw Error("dim must be 2 or 3");
  if (d->degree_u != 1 && d->degree_u != 2) throw Error("degree_u must be 1 or 2");
  if (d->degree_p != 1) throw Error("degree_p must be 1 (PoroElasticPressureSolver.h:20)");
  c->dim = d->dim; c->k_u = d->degree_u; c->nv = 1 << d->dim; c->ns_u = ipow(c->k_u + 1, c->dim); c->ns_p = c->nv; c->dpc_u = c->ns_u * c->dim; c->dpc_p = c->ns_p;
  c->n_cells = d->n_cells; c->n_u = d->n_dofs_u; c->n_p = d->n_dofs_p; c->mat = d->mat; c->comm.part = d->part;
  if (c->comm.part.n_ranks < 1) { c->comm.part.n_ranks = 1; c->comm.part.rank = 0; }
  c->comm.force_multi = std::getenv("PORO_FORCE_PARTITIONED_PATH") != nullptr;   // test hook: run the partitioned code path on one rank
  const poro_fe_tables &f = d->fe;
  if (f.nq_u != ipow(c->k_u + 1, c->dim) || f.nq_p != c->nv || f.ns_u != c->ns_u || f.ns_p != c->ns_p || f.nq_f != ipow(c->k_u + 1, c->dim - 1)) throw Error("poro_fe_tables sizes do not match dim / degree");
  if (c->n_cells <= 0 || c->n_u <= 0 || c->n_p <= 0) throw Error("empty mesh");
  for (int64_t i = 0; i < c->n_cells * c->dpc_u; ++i) if (d->cell_dofs_u[i] < 0 || d->cell_dofs_u[i] >= c->n_u) throw Error("cell_dofs_u out of range");
  for (int64_t i = 0; i < c->n_cells * c->dpc_p; ++i) if (d->cell_dofs_p[i] < 0 || d->cell_dofs_p[i] >= c->n_p) throw Error("cell_dofs_p out of range");
  for (int64_t i = 0; i < c->n_cells * c->nv; ++i) if (d->cell_vertices[i] < 0 || d->cell_vertices[i] >= d->n_vertices) throw Error("cell_vertices out of range");
  { bool inter = true;     // node-interleaved displacement numbering?
    for (int64_t i = 0; i < c->n_cells * c->ns_u && inter; ++i) { const int32_t b = d->cell_dofs_u[i * c->dim]; if (b % c->dim) inter = false; for (int k = 1; k < c->dim && inter; ++k) if (d->cell_dofs_u[i * c->dim + k] != b + k) inter = false; }
    c->interleaved_u = inter ? 1 : 0; }
  if (d->part.n_neighbours > 0 && d->box.enabled) throw Error("a general partition (poro_partition.n_neighbours > 0) carries no box tag: pieces are not boxes");
  if (!d->box.enabled && d->part.n_ranks > 1 && d->part.n_neighbours <= 0) throw Error("a partitioned general mesh needs the interface lists of poro_partition (n_neighbours > 0)");
  setup_general_partition(c, d);
  if (d->box.enabled) {
    // the lexicographic numbering the structured kernels assume must be the caller's numbering (spot-checked on three cells)
    int64_t nn[3] = {1, 1, 1}, np[3] = {1, 1, 1}, ncells = 1;
    for (int k = 0; k < c->dim; ++k) { if (d->box.n[k] < 1) throw Error("box.n must be positive"); nn[k] = (int64_t)c->k_u * d->box.n[k] + 1; np[k] = (int64_t)d->box.n[k] + 1; ncells *= d->box.n[k]; }
    if (nn[0] * nn[1] * nn[2] * c->dim != c->n_u || np[0] * np[1] * np[2] != c->n_p || ncells != c->n_cells) throw Error("box does not match n_dofs_u / n_dofs_p / n_cells");
    const int n1 = c->k_u + 1;
    const int64_t ncx = d->box.n[0], ncy = d->box.n[1];
    for (int64_t cell : {(int64_t)0, c->n_cells / 2, c->n_cells - 1}) {
      const int64_t ci = cell % ncx, cj = (cell / ncx) % ncy, ck = cell / (ncx * ncy);
      for (int sidx = 0; sidx < c->ns_u; ++sidx) {
        const int a = sidx % n1, b = (sidx / n1) % n1, cc = sidx / (n1 * n1);
        const int64_t node = ((ck * c->k_u + cc) * nn[1] + (cj * c->k_u + b)) * nn[0] + (ci * c->k_u + a);
        for (int k = 0; k < c->dim; ++k) if (d->cell_dofs_u[(cell * c->ns_u + sidx) * c->dim + k] != node * c->dim + k) throw Error("cell_dofs_u is not the lexicographic box numbering");
      }
      for (int v = 0; v < c->nv; ++v) {
        const int a = v & 1, b = (v >> 1) & 1, cc = v >> 2;
        if (d->cell_dofs_p[cell * c->nv + v] != ((ck + cc) * np[1] + (cj + b)) * np[0] + (ci + a)) throw Error("cell_dofs_p is not the lexicographic box numbering");
      }
    }
  }
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount; }
  if (const char *v = std::getenv("PORO_MF_VARIANT")) c->mf_variant = (std::string(v) == "gather" || std::string(v) == "0") ? 0 : 1;
  c->box.enabled = d->box.enabled;
  for (int k = 0; k < 3; ++k) { c->box.n[k] = d->box.enabled && k < c->dim ? d->box.n[k] : 1; c->box.h[k] = d->box.h[k]; c->box.nn[k] = c->k_u * c->box.n[k] + 1; }

  // tables -> one device buffer
  std::vector<double> T; std::vector<size_t> off;
  auto push = [&](const double *p, size_t n) { off.push_back(T.size()); T.insert(T.end(), p, p + n); };
  const int dim = c->dim, nf = 2 * dim;
  push(f.w_qu, f.nq_u); push(f.w_qp, f.nq_p); push(f.w_qf, f.nq_f); push(f.u_qu, (size_t)f.nq_u * f.ns_u); push(f.du_qu, (size_t)f.nq_u * f.ns_u * dim);
  push(f.du_qp, (size_t)f.nq_p * f.ns_u * dim); push(f.q1_qu, (size_t)f.nq_u * f.ns_p); push(f.dq1_qu, (size_t)f.nq_u * f.ns_p * dim);
  push(f.q1_qp, (size_t)f.nq_p * f.ns_p); push(f.dq1_qp, (size_t)f.nq_p * f.ns_p * dim); push(f.u_qf, (size_t)nf * f.nq_f * f.ns_u); push(f.dq1_qf, (size_t)nf * f.nq_f * f.ns_p * dim);
  c->tables.upload(T);
  const double *tb = c->tables.p;
  c->fe = FeTablesDev{f.nq_u, f.nq_p, f.nq_f, f.ns_u, f.ns_p, tb + off[0], tb + off[1], tb + off[2], tb + off[3], tb + off[4], tb + off[5], tb + off[6], tb + off[7], tb + off[8], tb + off[9], tb + off[10], tb + off[11]};

  c->cell_dofs_u.upload(d->cell_dofs_u, c->n_cells * c->dpc_u); c->cell_dofs_p.upload(d->cell_dofs_p, c->n_cells * c->dpc_p);
  { std::vector<double> X((size_t)c->n_cells * c->nv * dim);
    for (int64_t i = 0; i < c->n_cells * c->nv; ++i) for (int k = 0; k < dim; ++k) X[i * dim + k] = d->vertex_coords[(int64_t)d->cell_vertices[i] * dim + k];
    c->cell_X.upload(X); }
  { std::vector<int32_t> cells; colour_cells(c->n_cells, d->n_vertices, c->nv, d->cell_vertices, cells, c->color_off); c->color_cells.upload(cells); }
  { std::vector<uint8_t> m(c->n_u, 0); std::vector<double> v(c->n_u, 0.0);
    for (int64_t i = 0; i < d->n_dirichlet; ++i) { const int32_t dof = d->dirichlet_dof[i]; if (dof < 0 || dof >= c->n_u) throw Error("dirichlet_dof out of range"); m[dof] = 1; v[dof] = d->dirichlet_value[i]; }
    c->dir_mask.upload(m); c->dir_val.upload(v);
    // hanging-node constraints (locally refined meshes): operator-level condensation, see include/poroel_hip.h poro_constraints
    if (d->cons_u.n || d->cons_p.n) {
      if (d->box.enabled) throw Error("constraint lists belong to general (non-box) meshes: assembled-CSR operator or the general matrix-free one");
      if (c->comm.part.n_ranks > 1) throw Error("constraint lists are implemented for one rank");
    }
    upload_constraints(c->cons_u, d->cons_u, c->n_u, &m, "cons_u");
    { // extension: prescribed pressures (drained boundaries); the rows leave the pressure Newton system exactly like hanging rows do
      std::vector<uint8_t> pm(c->n_p, 0); std::vector<double> pv(c->n_p, 0.0);
      if (d->n_dirichlet_p < 0 || (d->n_dirichlet_p && (!d->dirichlet_dof_p || !d->dirichlet_value_p))) throw Error("bad prescribed-pressure list");
      for (int64_t i = 0; i < d->n_dirichlet_p; ++i) { const int32_t dof = d->dirichlet_dof_p[i]; if (dof < 0 || dof >= c->n_p) throw Error("dirichlet_dof_p out of range"); pm[dof] = 1; pv[dof] = d->dirichlet_value_p[i]; }
      c->n_pdir = d->n_dirichlet_p;
      if (c->n_pdir) { if (c->comm.part.n_ranks > 1) throw Error("prescribed pressures are implemented for one rank"); c->pdir_mask.upload(pm); c->pdir_val.upload(pv); }
      upload_constraints(c->cons_p, d->cons_p, c->n_p, &pm, "cons_p");
      if (c->cons_p.n && c->n_pdir) throw Error("prescribed pressures together with hanging pressure nodes are not supported"); }
    { std::vector<uint8_t> nm((size_t)(c->n_u / c->dim), 0); for (int64_t i = 0; i < d->n_dirichlet; ++i) nm[d->dirichlet_dof[i] / c->dim] |= (uint8_t)(1u << (d->dirichlet_dof[i] % c->dim)); c->node_mask.upload(nm); c->h_node_mask = std::move(nm); }
    if (d->n_dirichlet) c->dir_dofs.upload(d->dirichlet_dof, d->n_dirichlet);
    if (d->box.enabled) {   // are all constrained dofs on the box boundary?  (lets the matrix-free kernels skip mask loads in the interior)
      int64_t nn[3] = {1, 1, 1}; for (int k = 0; k < c->dim; ++k) nn[k] = (int64_t)c->k_u * d->box.n[k] + 1;
      for (int64_t i = 0; i < d->n_dirichlet; ++i) {
        const int64_t node = d->dirichlet_dof[i] / c->dim; const int64_t ci = node % nn[0], cj = (node / nn[0]) % nn[1], ck = node / (nn[0] * nn[1]);
        if (!(ci == 0 || ci == nn[0] - 1 || cj == 0 || cj == nn[1] - 1 || (c->dim == 3 && (ck == 0 || ck == nn[2] - 1)))) { c->mask_anywhere = 1; break; }
      }
    }
    c->h_dir_dof.assign(d->dirichlet_dof, d->dirichlet_dof + d->n_dirichlet); c->h_dir_val.assign(d->dirichlet_value, d->dirichlet_value + d->n_dirichlet); }
  c->n_bfaces = d->n_bfaces; c->n_neumann = d->n_neumann;
  if (d->n_bfaces) { c->bface_cell.upload(d->bface_cell, d->n_bfaces); c->bface_local.upload(d->bface_local, d->n_bfaces); c->bface_id.upload(d->bface_id, d->n_bfaces); }
  if (d->n_neumann) { c->neu_label.upload(d->neumann_label, d->n_neumann); c->neu_comp.upload(d->neumann_component, d->n_neumann); c->neu_val.upload(d->neumann_value, d->n_neumann); }

  // sparsity patterns (PoroElasticPressureSolver.h:80-94, PoroElasticDisplacementSolver.h:140-149)
  { std::vector<int64_t> rp, diag; std::vector<int32_t> col; build_pattern(c->n_p, c->n_cells, c->dpc_p, d->cell_dofs_p, rp, col, diag); upload_csr(c->Ap, c->n_p, rp, col, diag); }
  c->Mp.alloc(c->Ap.nnz); c->Kp.alloc(c->Ap.nnz); c->Jp.alloc(c->Ap.nnz);
  if (c->operator_mode == PORO_OP_CSR) {
    std::vector<int64_t> rp, diag; std::vector<int32_t> col; build_pattern(c->n_u, c->n_cells, c->dpc_u, d->cell_dofs_u, rp, col, diag); upload_csr(c->Au, c->n_u, rp, col, diag);
    c->Au_val.alloc(c->Au.nnz);
  } else if (d->box.enabled) c->Ke.alloc((size_t)c->dpc_u * c->dpc_u);

  hipStream_t s = c->stream;
  const int n_sym = dim * (dim + 1) / 2;
  for (int id : {PORO_VEC_U, PORO_VEC_RHS_U, PORO_VEC_DIAG_U}) { c->vec[id].alloc(c->n_u); c->vec[id].zero(s); }
  for (int id : {PORO_VEC_P, PORO_VEC_P_OLD, PORO_VEC_DP, PORO_VEC_RESIDUAL_P, PORO_VEC_EPSV, PORO_VEC_EPSV0, PORO_VEC_SOURCE_P}) { c->vec[id].alloc(c->n_p); c->vec[id].zero(s); }
  for (int e = 0; e < n_sym; ++e) { c->vec[PORO_VEC_STRAIN0 + e].alloc(c->n_p); c->vec[PORO_VEC_STRAIN0 + e].zero(s); c->vec[PORO_VEC_PROJ_RHS0 + e].alloc(c->n_p); c->vec[PORO_VEC_PROJ_RHS0 + e].zero(s);
                                     c->vec[PORO_VEC_STRESS0 + e].alloc(c->n_p); c->vec[PORO_VEC_STRESS0 + e].zero(s); }
  for (DevBuf<double> *b : {&c->lift_u, &c->neumann_u, &c->diag_u_local, &c->wg_u, &c->wd_u, &c->wh_u}) { b->alloc(c->n_u); b->zero(s); }
  for (DevBuf<double> *b : {&c->diag_J, &c->diag_M, &c->src_local, &c->wg_p, &c->wd_p, &c->wh_p, &c->tmp_p}) { b->alloc(c->n_p); b->zero(s); }
  c->partials.alloc((size_t)4 * kMaxPartials); c->partials.zero(s); c->scal.alloc(1); c->scal.zero(s); c->red.alloc(kScalarSlots); c->red.zero(s);

  // MatrixCreator::create_mass_matrix / create_laplace_matrix (:96-101) + the time-independent well integral (:142-147)
  c->Mp.zero(s); c->Kp.zero(s); c->Jp.zero(s);
  const AsmArgs a = asm_args(c);
  for (size_t k = 0; k + 1 < c->color_off.size(); ++k)
    asm_p_matrices(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], c->Ap.rp.p, c->Ap.col.p, c->Mp.p, c->Kp.p, c->src_local.p);
  PORO_HIP(hipStreamSynchronize(s));

  // uniform box: the coupling / projection right-hand sides have a structured form (kernels_box.hip); check it once against the
  // per-cell kernels on synthetic vectors before it replaces them
  const char *ba = std::getenv("PORO_BOX_ASM");
  if (c->box.enabled && !(ba && std::string(ba) == "0")) {
    c->box_cpl = box_coupling(c->dim, c->k_u, c->box);
    if (!std::getenv("PORO_DIAG_SKIP_SELFCHECK")) {
      std::vector<double> hp(c->n_p), hu(c->n_u);
      for (int64_t i = 0; i < c->n_p; ++i) hp[i] = 1e7 * (1 + 0.3 * std::sin(0.37 * (double)i));
      for (int64_t i = 0; i < c->n_u; ++i) hu[i] = 1e-5 * std::sin(0.11 * (double)i);
      DevBuf<double> tp, tu, zero_u, r1, r2; tp.upload(hp); tu.upload(hu); zero_u.alloc(c->n_u); zero_u.zero(s); r1.alloc(c->n_u); r2.alloc(c->n_u);
      auto compare = [&](const char *what, double *x1, const double *x2, int64_t n) {
        la_axpy(s, x1, -1.0, x2, n);
        la_norm_partials(s, x1, n, c->partials.p, c->partials.p + kMaxPartials); la_norm_partials(s, x2, n, c->partials.p + 2 * kMaxPartials, c->partials.p + 3 * kMaxPartials);
        la_reduce_finish(s, c->partials.p, 4, c->red.p, 2 | 8);
        double h[4]; PORO_HIP(hipMemcpyAsync(h, c->red.p, sizeof(h), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s));
        if (!(h[1] <= 1e-11 * h[3])) throw Error(std::string("structured ") + what + " disagrees with the per-cell kernel: max diff " + std::to_string(h[1]) + " vs max " + std::to_string(h[3]));
      };
      r1.zero(s);
      for (size_t k = 0; k + 1 < c->color_off.size(); ++k) asm_u_rhs(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], tp.p, r1.p);
      la_rhs_u_finish(s, r1.p, zero_u.p, zero_u.p, c->dir_mask.p, c->n_u);
      box_rhs_u(s, c->dim, c->box_cpl, c->mat.biot_alpha, tp.p, zero_u.p, zero_u.p, c->dir_mask.p, r2.p);
      compare("coupling right-hand side", r1.p, r2.p, c->n_u);
      const int ncomp = c->dim * c->dim; int32_t comps[9]; double *o1[9], *o2[9];
      DevBuf<double> q1, q2; q1.alloc((size_t)ncomp * c->n_p); q2.alloc((size_t)ncomp * c->n_p); q1.zero(s);
      for (int e = 0; e < ncomp; ++e) { comps[e] = e; o1[e] = q1.p + (size_t)e * c->n_p; o2[e] = q2.p + (size_t)e * c->n_p; }
      for (int e0 = 0; e0 < ncomp; e0 += 6) {
        const int ne = std::min(6, ncomp - e0);
        for (size_t k = 0; k + 1 < c->color_off.size(); ++k) asm_proj_rhs(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], tu.p, ne, comps + e0, o1 + e0);
        box_proj_rhs(s, c->dim, c->box_cpl, tu.p, ne, comps + e0, o2 + e0);
      }
      compare("projection right-hand side", q1.p, q2.p, (int64_t)ncomp * c->n_p);
    }
    c->box_asm = 1;
  }
}

void sync_source_vector(poro_ctx *c) {   // PORO_VEC_SOURCE_P = the assembled (rank-summed) well integral
  la_copy(c->stream, vec(c, PORO_VEC_SOURCE_P), c->src_local.p, c->n_p);
  exchange_add(c, vec(c, PORO_VEC_SOURCE_P), c->n_p, c->comm.part.plane_p);
}

// lambda_max(D^-1 A_u) from the Lanczos tridiagonal of 25 Jacobi-preconditioned CG steps on a synthetic right-hand side (the constrained rows are
// inert): the largest Ritz value approaches lambda_max from below within a fraction of a percent, far faster than a power iteration
double estimate_lmax_u(poro_ctx *c, const std::function<bool(const double *, double *, double *)> &apply, const DiagVec &dj) {
  hipStream_t s = c->stream; const int64_t n = c->n_u, n_own = owned(c, n, c->comm.part.plane_u);
  std::vector<double> hv(n); for (int64_t i = 0; i < n; ++i) hv[i] = std::sin(0.731 * (double)i) + 0.3 * std::cos(0.013 * (double)i * (double)(i % 7));
  DevBuf<double> r, z, p, ap; r.upload(hv); z.alloc(n); p.alloc(n); ap.alloc(n);
  exchange_add(c, r.p, n, c->comm.part.plane_u);                      // partitioned runs: the start vector was filled by LOCAL index - make the copies of the shared dofs agree (any consistent vector will do)
  la_mask_zero(s, r.p, dj.inert, n);
  la_cheb_first(s, z.p, r.p, dj, 1.0, n);                            // z = D^-1 r
  la_copy(s, p.p, z.p, n);
  double rz = dot_host(c, r.p, z.p, n_own);
  const int K = 25; std::vector<double> al, be;
  for (int k = 0; k < K && rz > 0; ++k) {
    apply(p.p, ap.p, nullptr);
    la_mask_zero(s, ap.p, dj.inert, n);
    const double pap = dot_host(c, p.p, ap.p, n_own);
    if (!(pap > 0)) break;
    const double alpha = rz / pap;
    la_axpy(s, r.p, -alpha, ap.p, n);
    la_cheb_first(s, z.p, r.p, dj, 1.0, n);
    const double rz_new = dot_host(c, r.p, z.p, n_own), beta = rz_new / rz;
    al.push_back(alpha); be.push_back(beta);
    la_xpby(s, p.p, beta, 1.0, z.p, n);                               // p = beta p + z
    rz = rz_new;
  }
  const int m = (int)al.size();
  if (m == 0) return 4.0;
  std::vector<double> T((size_t)m * m, 0.0);
  for (int k = 0; k < m; ++k) {
    T[(size_t)k * m + k] = 1.0 / al[k] + (k > 0 ? be[k - 1] / al[k - 1] : 0.0);
    if (k + 1 < m) T[(size_t)k * m + k + 1] = T[(size_t)(k + 1) * m + k] = std::sqrt(be[k]) / al[k];
  }
  return 1.05 * sym_lambda_max(m, T);
}

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
  if (c->stream) (void)hipStreamDestroy(c->stream);
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
    for (auto &kv : c->vec) { DevBuf<double> &d = c->vec_saved[kv.first]; if (d.n != kv.second.n) d.alloc(kv.second.n); la_copy(c->stream, d.p, kv.second.p, (int64_t)kv.second.n); }
    PORO_HIP(hipStreamSynchronize(c->stream)); return 0;
  });
}
int poro_state_restore(poro_ctx *c) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (c->vec_saved.empty()) throw Error("state_restore without state_save");
    for (auto &kv : c->vec_saved) la_copy(c->stream, c->vec.at(kv.first).p, kv.second.p, (int64_t)kv.second.n);
    // the solves that follow repeat earlier ones: forget the iteration-count history, so that a measurement of a repeated step cannot profit from a perfect prediction
    for (int *h : {c->pcg_hint_u, c->pcg_hint_fdm_u, c->pcg_hint_cheb_u}) h[0] = h[1] = 0;
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
    exchange_add(c, vec(c, PORO_VEC_RHS_U), c->n_u, c->comm.part.plane_u);
    if (c->cons_u.n) {
      // condensed right-hand side C^T (b - A x_inh), x_inh = the constraints' inhomogeneities (distribute_local_to_global, :280-286)
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
    // stream-ordered: the right-hand side is consumed by kernels of the same stream (a caller that wants the host to wait calls poro_ctx_synchronize)
    return 0;
  });
}

int poro_supports_preconditioner(poro_ctx *c, int32_t which_system, int32_t prec) {
  if (!c) return 0;
  if (prec == PORO_PREC_NONE || prec == PORO_PREC_JACOBI) return 1;
  if (which_system == 0 && c->cons_u.n) return prec == PORO_PREC_CHEBYSHEV;      // condensed operators exist at operator level only: Jacobi and the polynomial built on it
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
    if (c->cons_u.n && opts->preconditioner != PORO_PREC_JACOBI && opts->preconditioner != PORO_PREC_NONE && opts->preconditioner != PORO_PREC_CHEBYSHEV)
      throw Error("meshes with constraint lists: PORO_PREC_JACOBI / CHEBYSHEV / NONE only (the operator is condensed on the fly)");
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
      apply_A_u(c, x, y, mode, nullptr, false, nullptr);
      la_cons_reduce(c->stream, c->cons_u, y);
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
        if (oct) {
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
    exchange_add(c, R, c->n_p, c->comm.part.plane_p);
    la_cons_reduce(s, c->cons_p, R);                                              // constraints.condense(residual) (:153)
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
    if ((c->cons_p.n || c->n_pdir) && opts->preconditioner != PORO_PREC_JACOBI && opts->preconditioner != PORO_PREC_NONE) throw Error("meshes with hanging-node constraints or prescribed pressures: PORO_PREC_JACOBI / NONE only");
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
      exchange_add(c, y, c->n_p, c->comm.part.plane_p);
      la_cons_reduce(c->stream, c->cons_p, y); return false;
    };
    if (opts->preconditioner == PORO_PREC_FDM) {
      build_fdm_p(c);
      const double kk[3] = {jk, jk, jk};
      if (!c->wz_p.p) c->wz_p.alloc(c->n_p);
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *) { fdm_precondition_p(c, ja, kk, g, z); return false; };
      DiagVec dz; dz.full = c->dinv_J.p; dz.z = c->wz_p.p;
      return pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_DP), vec(c, PORO_VEC_RESIDUAL_P), dz, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, &P);
    }
    DiagVec dv; dv.full = c->dinv_J.p; dv.inert = (c->cons_p.n || c->n_pdir) ? c->cons_p.inert.p : nullptr;
    const int rc = pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_DP), vec(c, PORO_VEC_RESIDUAL_P), dv, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
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
    for (int k = 0; k < n_comp; ++k) { exchange_add(c, rhs[k], c->n_p, c->comm.part.plane_p); la_cons_reduce(c->stream, c->cons_p, rhs[k]); }   // StrainProjector.h:191-194
    return 0;
  });
}

int poro_proj_solve(poro_ctx *c, int32_t entry, const poro_solver_opts *opts, poro_solve_info *info) {
  return guarded([&] {
    PORO_HIP(hipSetDevice(c->device));
    if (!c->projection_matrix_ready) throw Error("proj_solve before proj_assemble_matrix");
    if (entry < 0 || entry >= c->dim * (c->dim + 1) / 2) throw Error("rhs_entry out of range");
    if (c->cons_p.n && opts->preconditioner != PORO_PREC_JACOBI && opts->preconditioner != PORO_PREC_NONE) throw Error("meshes with hanging-node constraints: PORO_PREC_JACOBI / NONE only");
    if (opts->preconditioner == PORO_PREC_ILU0) {
      return pcg_ilu0(c, c->Ap, c->Mp.p, c->ilu_M, c->ilu_M_valid, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
    }
    if (opts->preconditioner == PORO_PREC_SSOR) return pcg_ssor(c, c->Ap, c->Mp.p, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
    const bool stencil = c->operator_mode == PORO_OP_MATRIX_FREE && c->box.enabled;
    auto apply = [&](const double *x, double *y, double *) {
      la_cons_expand(c->stream, c->cons_p, const_cast<double *>(x), false);      // condensed projection matrix (StrainProjector.h:104-105)
      if (stencil) { Timed tm(c, "apply_p_stencil"); p_stencil_apply(c->stream, c->dim, c->box, 1.0, 0.0, x, y); }
      else { Timed tm(c, "apply_p_csr"); la_csr_spmv(c->stream, c->Ap, c->Mp.p, x, y); }
      exchange_add(c, y, c->n_p, c->comm.part.plane_p);
      la_cons_reduce(c->stream, c->cons_p, y); return false;
    };
    if (opts->preconditioner == PORO_PREC_FDM) {
      build_fdm_p(c);
      const double kk[3] = {0, 0, 0};
      if (!c->wz_p.p) c->wz_p.alloc(c->n_p);
      const std::function<bool(const double *, double *, double *)> P = [&](const double *g, double *z, double *) { fdm_precondition_p(c, 1.0, kk, g, z); return false; };
      DiagVec dz; dz.full = c->dinv_M.p; dz.z = c->wz_p.p;
      return pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), dz, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info, &P);
    }
    DiagVec dv; dv.full = c->dinv_M.p; dv.inert = c->cons_p.n ? c->cons_p.inert.p : nullptr;
    const int rc = pcg(c, apply, c->n_p, c->comm.part.plane_p, vec(c, PORO_VEC_STRAIN0 + entry), vec(c, PORO_VEC_PROJ_RHS0 + entry), dv, c->wg_p.p, c->wd_p.p, c->wh_p.p, opts, info);
    la_cons_expand(c->stream, c->cons_p, vec(c, PORO_VEC_STRAIN0 + entry), true);   // constraints.distribute (StrainProjector.h:216)
    return rc;
  });
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
      auto once = [&]() { if (O.built) fdmo_apply(s, O, O.g.p, O.z.p, O.t.p); else fdm_precondition_u(c, g.p, z.p); };
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
