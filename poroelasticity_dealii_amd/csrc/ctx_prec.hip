// Fast-diagonalisation preconditioners on the host side: set-up and application for the Q1 systems and for the displacement blocks, incl. the slab-distributed forms (all-to-all of column groups).
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
// ---- fast diagonalisation of the Q1 box operators (kernels_fdm.hip) -------------------------------------------------------------
bool fdm_p_supported(poro_ctx *c) {
  if (!c->lines.on) return false;
  if (!c->box.enabled && c->comm.multi()) return false;     // tensor-product grids: one rank
  if (c->comm.multi() && !(c->comm.nccl_comm || (c->comm.ar && c->comm.sr))) return false;
  return true;
}
static void upload_dir(FdmDir &D, const std::vector<double> &hcell, bool uniform, FdmOct *fused = nullptr, int dir = 0) {
  const int n_cells = (int)hcell.size();
  std::vector<double> S, lam;
  if (uniform) q1_eig(n_cells, hcell[0], S, lam); else fdmu_eig_1d(1, hcell, false, false, S, lam);   // (closed form on a uniform line)
  if (fused) fdmo_scalar_upload_dir(*fused, dir, S, lam, n_cells + 1);
  const int n = n_cells + 1;
  std::vector<double> St((size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) St[(size_t)j * n + i] = S[(size_t)i * n + j];
  D.n = n; D.S.upload(S); D.St.upload(St); D.lam.upload(lam);
}
void build_fdm_p(poro_ctx *c) {
  if (c->fdm_p.built) return;
  if (!fdm_p_supported(c)) throw Error("PORO_PREC_FDM needs a uniform box (poro_desc.box.enabled) and, when partitioned, an initialised communicator");
  c->fdm_p.dim = c->dim;
  // one rank, 3D, lines of at most 80 nodes: the three-launch form through the block-FDM transform kernel (kernels_fdmo.hip) instead of six single-direction launches
  int np3[3] = {c->lines.n[0] + 1, c->lines.n[1] + 1, c->dim == 3 ? c->lines.n[2] + 1 : 1};
  const bool fused = !c->comm.multi() && fdmo_scalar_usable(c->dim, np3) && !std::getenv("PORO_FDM_P_UNFUSED");
  if (fused) fdmo_scalar_init(c->fdm_p_fused, np3, c->stream);
  bool fused_slab = false;   // decided below, once the global line length is known
  for (int d = 0; d < c->dim; ++d) upload_dir(c->fdm_p.dir[d], c->lines.hcell[d], c->lines.uniform, fused ? &c->fdm_p_fused : nullptr, d);   // local slab; the last direction is replaced below when partitioned
  if (!c->comm.multi()) c->fdm_p_fused.built = fused;
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
    // slab form of the same three-launch kernels (kernels_fdmo.hip): 3D, local planes and the global line of at most 80 vertices
    { int g3[3] = {np3[0], np3[1], acc + 1};
      fused_slab = c->dim == 3 && fdmo_scalar_usable(3, np3) && fdmo_scalar_usable(3, g3) && !std::getenv("PORO_FDM_P_UNFUSED");
      if (fused_slab) {
        fdmo_scalar_init_slab(c->fdm_p_fused, np3, r, F.layers, c->stream);
        for (int d = 0; d < 2; ++d) { std::vector<double> S, lam; q1_eig(c->box.n[d], c->box.h[d], S, lam); fdmo_scalar_upload_dir(c->fdm_p_fused, d, S, lam, c->box.n[d] + 1); }
        std::vector<double> S, lam; q1_eig(acc, c->box.h[last], S, lam); fdmo_scalar_upload_dir(c->fdm_p_fused, 2, S, lam, acc + 1);
        c->fdm_p_fused.built = true;
      } }
    upload_dir(F.last, std::vector<double>((size_t)acc, c->box.h[last]), true);
    const size_t blk = (size_t)std::max(F.max_own, F.max_nl) * F.C;
    F.sendbuf.alloc(blk * N); F.recvbuf.alloc(blk * N); F.tz1.alloc((size_t)F.ng * F.C); F.tz2.alloc((size_t)F.ng * F.C);
    F.sendbuf.zero(c->stream); F.recvbuf.zero(c->stream); F.tz1.zero(c->stream); F.tz2.zero(c->stream);
    // the four window copies of an application, one entry per peer: local grid -> send blocks (own planes), gathered blocks -> whole lines, whole lines -> send blocks
    // (every rank's planes incl. the shared ones), scattered blocks -> local grid
    std::vector<FdmWindow> W((size_t)4 * N);
    auto ncols_of = [&](int q) { return std::max<int64_t>(0, std::min<int64_t>(F.C, F.ncol_total - (int64_t)q * F.C)); };
    const int own_r = F.layers[r] + (r == N - 1 ? 1 : 0), nl_r = c->box.n[last] + 1;
    for (int q = 0; q < N; ++q) {
      W[q] = FdmWindow{own_r, ncols_of(q), (int64_t)q * F.C, 0};
      W[N + q] = FdmWindow{F.layers[q] + (q == N - 1 ? 1 : 0), F.C, 0, F.off[q]};
      W[2 * N + q] = FdmWindow{F.layers[q] + 1, F.C, 0, F.off[q]};
      W[3 * N + q] = FdmWindow{nl_r, ncols_of(q), (int64_t)q * F.C, 0};
    }
    F.windows.upload(W);
    F.built = true;
  }
  c->fdm_p.built = true;
}
// every rank sends block q of `send` (blk doubles) to rank q and receives block q of `recv` from it
void alltoall_blocks(poro_ctx *c, double *send, double *recv, int64_t blk, bool self_in_place) {
  Comm &cm = c->comm; const int N = cm.part.n_ranks, r = cm.part.rank;
  Timed tm(c, "alltoall");
  if (!self_in_place) PORO_HIP(hipMemcpyAsync(recv + (size_t)r * blk, send + (size_t)r * blk, blk * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
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
  if (!c->comm.multi()) {
    if (c->fdm_p_fused.built && k[0] == k[1] && k[1] == k[2]) fdmo_scalar_apply(s, c->fdm_p_fused, a, k[0], g, z);
    else fdm_apply(s, c->fdm_p, a, k, g, z, c->fdm_t1.p, c->fdm_t2.p);
    return;
  }
  if (c->fdm_p_fused.built && c->fdm_p_fused.slab.on && k[0] == k[1] && k[1] == k[2]) {
    // the slab form of the fused kernels: x / y sweeps on the local planes straight into the exchange buffer, whole z lines per column share, scatter, y / x sweeps
    FdmOct &O = c->fdm_p_fused; auto &S = O.slab; double *send = S.buf.p, *recv = S.buf.p + S.recv_off;
    fdmo_scalar_slab_pass(s, O, 1, a, k[0], g, S.buf.p);
    alltoall_blocks(c, send, recv, (int64_t)S.max_own * S.scols, true);
    fdmo_scalar_slab_pass(s, O, 2, a, k[0], S.buf.p, S.tz.p);
    fdmo_slab_scatter_pack(s, O, nullptr);
    alltoall_blocks(c, send, recv, (int64_t)S.max_nl * S.scols, true);
    fdmo_scalar_slab_pass(s, O, 3, a, k[0], S.buf.p, z);
    return;
  }
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
  const int64_t blk1 = (int64_t)F.max_own * F.C;
  fdm_window_batch(s, const_cast<double *>(cur), F.sendbuf.p, F.recvbuf.p, r, true, F.windows.p, N, F.max_own, F.C, SIp, blk1);
  alltoall_blocks(c, F.sendbuf.p, F.recvbuf.p, blk1, true);
  fdm_window_batch(s, F.tz1.p, F.recvbuf.p, nullptr, r, false, F.windows.p + N, N, F.max_own, F.C, F.C, blk1);       // (the owned planes of all ranks cover every global plane: tz1 is overwritten in full)
  FdmScale sc{}; sc.a = a; sc.ncol = F.C; sc.col0 = (int64_t)r * F.C; sc.col_total = F.ncol_total;
  for (int d = 0; d < 3; ++d) { sc.lam[d] = d < dim ? (d == last ? F.last.lam.p : L.dir[d].lam.p) : nullptr; sc.k[d] = d < dim ? k[d] : 0.0; sc.n[d] = d < dim ? (d == last ? F.ng : L.dir[d].n) : 1; }
  fdm_transform(s, F.last.St.p, F.ng, F.C, 1, F.tz1.p, F.tz2.p, &sc);
  fdm_transform(s, F.last.S.p, F.ng, F.C, 1, F.tz2.p, F.tz1.p, nullptr);
  // scatter back: every rank gets all of its planes (shared ones included) of every column group
  const int64_t blk2 = (int64_t)F.max_nl * F.C;
  fdm_window_batch(s, F.tz1.p, F.sendbuf.p, F.recvbuf.p, r, true, F.windows.p + 2 * N, N, F.max_nl, F.C, F.C, blk2);
  alltoall_blocks(c, F.sendbuf.p, F.recvbuf.p, blk2, true);
  double *back = dim == 3 ? t2 : t1;
  fdm_window_batch(s, back, F.recvbuf.p, nullptr, r, false, F.windows.p + 3 * N, N, F.max_nl, F.C, SIp, blk2);
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
  const int dim = c->dim, last = dim - 1; const int64_t nn[3] = {c->lines.nn[0], c->lines.nn[1], dim == 3 ? c->lines.nn[2] : 1};
  std::string why;
  FdmU &F = c->fdm_u;
  if (!c->lines.on || !c->interleaved_u || (multi && !c->box.enabled)) why = "needs a uniform box (or, on one rank, a tensor-product grid) with node-interleaved displacement dofs";
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
  // lines of more than 320 points only have the even / odd (blocked) transform kernels: every component needs the same condition at both ends of such a direction.
  // Checked here, after the ranks have agreed on the face flags, so that poro_supports_preconditioner() is authoritative and no rank throws alone at solve time
  if (why.empty()) for (int d = 0; d < dim; ++d) {
    const int64_t line = (multi && d == last) ? 0 : nn[d];        // (the partitioned direction's GLOBAL line length is only known in build_fdm_u; its limit of 4096 is checked there on every rank alike)
    if (line > 320 && !(dim == 2 && !multi)) for (int comp = 0; comp < dim; ++comp) if (F.fix[comp][d][0] != F.fix[comp][d][1]) why = "a grid line of more than 320 points needs the same Dirichlet condition at both of its ends (even / odd transforms)";   // (2D on one rank: the planar form takes lines of any length with any end conditions)
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
  for (int d = 0; d < 3; ++d) F.nn[d] = d < dim ? c->lines.nn[d] : 1;
  const double l2g = c->mat.lame_lambda + 2 * c->mat.shear_G, G = c->mat.shear_G;
  for (int comp = 0; comp < dim; ++comp) for (int d = 0; d < dim; ++d) F.coef[comp][d] = d == comp ? l2g : G;
  int n_cells_last = c->lines.n[last];
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
  // octant form (kernels_fdmo.hip): one rank, 3D, every direction mirror-symmetric for every component, half lines of at most 128 entries
  // slab partitions: the quadrant form (x, y split locally; the z butterfly next to the all-to-all) under the same conditions on the GLOBAL line
  bool oct_ok = !F.single && !std::getenv("PORO_FDMU_NO_OCT");
  bool planar = false, planar_split = true;     // 2D, one rank: the quadrant form with a single plane, transforms as batched GEMMs (lines of any length; without the parity split where the end conditions differ)
  { int nn3[3] = {F.nn[0], F.nn[1], F.nn[2]}, sym3[3] = {F.nn[0], F.nn[1], multi ? F.ng : F.nn[2]};
    planar = dim == 2 && !multi && fdmo_planar_usable(dim, nn3);
    oct_ok = oct_ok && (planar || fdmo_usable(dim, sym3));
    bool symmetric = true;
    for (int d = 0; d < dim; ++d) for (int comp = 0; comp < dim; ++comp) symmetric = symmetric && F.fix[comp][d][0] == F.fix[comp][d][1];
    planar_split = symmetric;
    if (!planar) oct_ok = oct_ok && symmetric;      // (the planar form also runs without the parity split; the 3D forms need it)
    if (oct_ok && planar) fdmo_init_planar(c->fdm_oct, nn3, F.coef, c->stream, planar_split);
    else if (oct_ok && !multi) fdmo_init(c->fdm_oct, nn3, F.coef, c->stream);
    if (oct_ok && multi) { std::vector<int> node_layers(F.n_ranks); for (int q = 0; q < F.n_ranks; ++q) node_layers[q] = ku * F.layers[q];
                           fdmo_init_slab(c->fdm_oct, nn3, F.coef, F.rank, node_layers, c->comm.part.has_upper != 0, c->stream); } }
  // eigenpairs per (direction, end conditions); components with the same end conditions share the host work.  A direction takes the even / odd
  // form (half the MFMA work) when every component has the same condition at both ends there - all components of a pass share one kernel
  for (int d = 0; d < dim; ++d) {
    std::vector<double> S[4], lam[4]; bool have[4] = {false, false, false, false};
    const bool global_dir = multi && d == last;
    const int ncell = global_dir ? n_cells_last : c->lines.n[d], nnode = ku * ncell + 1;
    const std::vector<double> hcell = global_dir ? std::vector<double>((size_t)ncell, c->box.h[d]) : c->lines.hcell[d];
    bool allow_split = true;
    for (int comp = 0; comp < dim; ++comp) allow_split = allow_split && F.fix[comp][d][0] == F.fix[comp][d][1];
    for (int attempt = 0; attempt < 2; ++attempt) {
      bool all_split = true;
      for (int comp = 0; comp < dim; ++comp) {
        const int key = F.fix[comp][d][0] * 2 + F.fix[comp][d][1];
        if (!have[key]) { fdmu_eig_1d(ku, hcell, F.fix[comp][d][0], F.fix[comp][d][1], S[key], lam[key]); have[key] = true; }
        FdmuDir &D = global_dir ? F.last_global[comp] : F.dir[comp][d];
        if (!(planar && oct_ok && nnode > 320 && !allow_split)) fdmu_upload_dir(D, S[key], lam[key], nnode, global_dir ? false : F.single, allow_split);   // (the nodal kernels have no full-length form beyond 320 points; the planar form does not need them)
        all_split = all_split && D.split;
        if (oct_ok && attempt == 0) oct_ok = planar ? fdmo_upload_dir_planar(c->fdm_oct, comp, d, S[key], lam[key], nnode) : fdmo_upload_dir(c->fdm_oct, comp, d, S[key], lam[key], nnode);
      }
      if (!allow_split || all_split) break;
      allow_split = false;                       // the numerical symmetry check failed for some component: the whole direction in the full form
    }
  }
  c->fdmu_t1.alloc(c->n_u); c->fdmu_t2.alloc(c->n_u); c->fdmu_t1.zero(c->stream); c->fdmu_t2.zero(c->stream);   // (only finite values ever live in the scratch arrays)
  if (!c->wz_u.p) { c->wz_u.alloc(c->n_u); c->wz_u.zero(c->stream); }
  if (oct_ok && !planar) fdmo_finalize(c->fdm_oct);
  c->fdm_oct.built = oct_ok;
  F.built = true;
}
// slab form of the octant kernels: g, z in quadrant layout.  x / y sweeps on the local planes, whole z lines per column share between two all-to-alls
void fdm_precondition_u_slab(poro_ctx *c, const double *g, double *z, const PcgScalars *gate) {
  Timed tm(c, "precondition_u_fdm");
  hipStream_t s = c->stream; FdmOct &O = c->fdm_oct; auto &S = O.slab;
  double *send = S.buf.p, *recv = S.buf.p + S.recv_off;
  fdmo_slab_pass(s, O, 1, g, S.buf.p, gate);                                          // x, y forward on the local planes; the owned planes go straight into the exchange buffer
  alltoall_blocks(c, send, recv, (int64_t)S.max_own * S.scols, true);
  if (S.zboth) { Timed tp(c, "fdm_u_slab_z_stage"); fdmo_slab_pass(s, O, 2, S.buf.p, S.buf.p, gate); }   // whole z lines of this rank's column share, both parity parts per workgroup: gathered planes in, scattered planes out
  else {
    fdmo_slab_pass(s, O, 2, S.buf.p, S.tz.p, gate);                                   // (tile counts without that variant: one parity part per workgroup, then v_k = a + b, v_k' = a - b in a kernel of its own)
    Timed tp(c, "fdm_u_slab_z_stage"); fdmo_slab_scatter_pack(s, O, gate);
  }
  alltoall_blocks(c, send, recv, (int64_t)S.max_nl * S.scols, true);
  fdmo_slab_pass(s, O, 3, S.buf.p, z, gate);                                          // y, x backward, reading the received planes in place
}
// ---- additive two-level preconditioner on refinements of a uniform box (poro_desc.coarse) ----------------------------------------------------
static void upload_interp(poro_ctx::Interp &T, int64_t n_fine, int64_t n_coarse, const int64_t *ptr, const int32_t *node, const double *weight, const char *what) {
  T.n_fine = n_fine; T.n_coarse = n_coarse;
  const int64_t nnz = ptr[n_fine];
  const auto lanes_for = [](int64_t nnz, int64_t rows) { return nnz >= 6 * rows ? 8 : nnz >= 3 * rows ? 4 : 1; };   // mean row length -> lanes per row
  T.lanes = lanes_for(nnz, n_fine); T.lanes_t = lanes_for(nnz, n_coarse);
  std::vector<int64_t> tp((size_t)n_coarse + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) { const int32_t j = node[k]; if (j < 0 || j >= n_coarse) throw Error(std::string("poro_desc.coarse: ") + what + " node out of range"); tp[j + 1]++; }
  { int64_t longest = 0; for (int64_t j = 0; j < n_coarse; ++j) longest = std::max(longest, tp[j + 1]);      // restriction: a few very long rows (refined block) beside single-entry ones
    if (longest >= 48) T.lanes_t = std::max(T.lanes_t, 16); }                     // (measured on the refined 32^3 box and the Gmsh grid: 16 lanes 3 % ahead of 8, 32 lanes behind both)
  for (int64_t j = 0; j < n_coarse; ++j) tp[j + 1] += tp[j];
  std::vector<int32_t> tc((size_t)nnz); std::vector<double> tw((size_t)nnz); std::vector<int64_t> pos(tp.begin(), tp.end() - 1);
  for (int64_t i = 0; i < n_fine; ++i) {
    if (ptr[i + 1] < ptr[i]) throw Error("poro_desc.coarse: ptr not monotone");
    for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) { const int64_t at = pos[node[k]]++; tc[at] = (int32_t)i; tw[at] = weight[k]; }
  }
  T.p_ptr.upload(ptr, (size_t)n_fine + 1); T.p_col.upload(node, (size_t)nnz); T.p_w.upload(weight, (size_t)nnz);
  T.pt_ptr.upload(tp); T.pt_col.upload(tc); T.pt_w.upload(tw);
}
void setup_two_level(poro_ctx *c, const poro_desc *d) {
  auto &T = c->two_level; const int dim = c->dim;
  upload_interp(T, c->n_u / dim, T.box->n_u / dim, d->coarse.ptr, d->coarse.node, d->coarse.weight, "displacement");
  if (d->coarse.ptr_p) {
    if (!d->coarse.node_p || !d->coarse.weight_p) throw Error("poro_desc.coarse: node_p / weight_p missing");
    upload_interp(T.pressure, c->n_p, T.box->n_p, d->coarse.ptr_p, d->coarse.node_p, d->coarse.weight_p, "pressure");
  }
}
bool two_level_supported(poro_ctx *c) {
  if (!c->two_level.box) return false;
  analyse_fdm_u(c->two_level.box);
  return c->two_level.box->fdm_u_state == 1;
}
bool two_level_supported_p(poro_ctx *c) { return c->two_level.box && c->two_level.pressure.n_fine == c->n_p && fdm_p_supported(c->two_level.box); }
// the scalar analogue for the pressure Jacobian a M + kappa K and the projection mass matrix (a = 1, kappa = 0): Jacobi on this mesh + the box's exact fast diagonalisation
void two_level_precondition_p(poro_ctx *c, double a, double kappa, const double *dinv, const double *g, double *z, double omega) {
  Timed tm(c, "precondition_p_two_level");
  auto &T = c->two_level.pressure; poro_ctx *H = c->two_level.box; hipStream_t s = c->stream;
  build_fdm_p(H);
  if (!H->wz_p.p) H->wz_p.alloc(H->n_p);
  double *rc = H->wg_p.p, *zc = H->wz_p.p;
  la_nodal_interp(s, T.pt_ptr.p, T.pt_col.p, T.pt_w.p, T.n_coarse, 1, g, rc, T.lanes_t);
  const double kk[3] = {kappa, kappa, kappa};
  fdm_precondition_p(H, a, kk, rc, zc);
  la_two_level_combine(s, T.p_ptr.p, T.p_col.p, T.p_w.p, T.n_fine, 1, zc, g, dinv, (c->cons_p.n || c->n_pdir) ? c->cons_p.inert.p : nullptr, omega, z, T.lanes);
}
void two_level_precondition_u(poro_ctx *c, const double *g, double *z, double omega) {
  Timed tm(c, "precondition_u_two_level");
  auto &T = c->two_level; poro_ctx *H = T.box; hipStream_t s = c->stream; const int dim = c->dim;
  build_fdm_u(H);
  double *rc = H->wg_u.p, *zc = H->wz_u.p;                                  // the box context's work vectors (it never solves anything itself)
  la_nodal_interp(s, T.pt_ptr.p, T.pt_col.p, T.pt_w.p, T.n_coarse, dim, g, rc, T.lanes_t);              // r_H = P^T g
  FdmOct &O = H->fdm_oct;
  if (O.built) { fdmo_from_nodal(s, O, rc, O.g.p); if (O.planar) fdmo_apply_planar(s, O, O.g.p, O.z.p); else fdmo_apply(s, O, O.g.p, O.z.p, O.t.p); fdmo_to_nodal(s, O, O.z.p, zc); }
  else fdm_precondition_u(H, rc, zc);                                         // z_H = blockdiag(A_H)^-1 r_H (zero on the box's Dirichlet faces)
  la_two_level_combine(s, T.p_ptr.p, T.p_col.p, T.p_w.p, T.n_fine, dim, zc, g, c->dinv_u.p, c->cons_u.inert.p, omega, z, T.lanes);
}

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

}  // namespace ctx_detail
}  // namespace poro
