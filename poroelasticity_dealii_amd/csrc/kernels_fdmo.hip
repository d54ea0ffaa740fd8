// Octant form of the block fast-diagonalisation preconditioner of the displacement system (K-prec; the preconditioner slot of
// PoroElasticDisplacementSolver<dim>::solve, PoroElasticDisplacementSolver.h:302-305) on 3D uniform boxes, one rank (gfx950, wave64, fp64 MFMA).
//
// kernels_fdmu.hip exploits that on a mirror-symmetric line every generalised eigenvector of the 1D FE_Q(k) pencil is even or odd: with
// e_k = v_k + v_k', o_k = v_k - v_k' (k < h = (n + 1) / 2, k' = n - 1 - k; the centre node of an odd line is its own mirror: e = v, o = 0) a 1D
// transform splits into two half-size products.  There the butterflies run inside every transform pass, and the passes still read / write the
// node-interleaved CG vectors (24-byte strides: 2.4 x the useful HBM traffic in the x passes, profiles/r03_fdmu_baseline_counters.txt).  Here the
// butterflies of all three directions are hoisted OUT of the preconditioner: they commute with the 1D transforms of the other directions, so
//     z = H' (sum over the 8 parity octants of independent half-size 3D fast diagonalisations) H g,
// with H the 8-point butterfly.  The CG residual g and z = P^-1 g therefore LIVE in octant form Q[c][o][kz][ky][kx] (o = 4 pz + 2 py + px), and H / H'
// ride in the CG update kernels, which touch every entry anyway (g += alpha A d reads the eight mirror images of A d; d = -z + beta d scatters to them;
// g.z = sum Q_g Q_z and |g|^2 = weighted sum Q_g^2 hold in octant form directly).  What is left of the preconditioner is 24 independent
// (component, octant) blocks of h^3 entries, each a plain 3D tensor transform in contiguous planar storage:
//     pass 1  (x, y fused per z-plane)    X[ky][kx] -> Fy X Fx^T
//     pass 2  (z; per 16 NT columns)      X[kz][col] -> Bz diag(1 / (kx lam_x + ky lam_y + kz lam_z)) Fz X        (in place)
//     pass 3  (y, x fused per z-plane)    X[my][mx] -> By X Bx^T
// = 3 sweeps over the vector instead of 5, every access contiguous.  All three passes are ONE kernel: a block of <= 80 x 80 doubles is staged
// in LDS, two chained GEMMs on v_mfma_f64_16x16x4_f64 with the data passing through LDS between them (in the bank-conflict-free layout the second
// GEMM's operand reads need), the half-size transform matrices as the other operand straight from L2 in MFMA fragment order (each wave only needs
// the 16 rows of its own output tile).  "contract columns": Y = X T^T, wave w owns output column tile w, data = A operand; "contract rows":
// Y = T X, wave s owns output row tile s, data = B operand; either way the accumulator of tile (T, W) holds element (16 T + 4 q + kq, 16 W + j) in
// register q of lane j + 16 kq, so one store routine serves both.
#include "common.hpp"
#include "device_reduce.hpp"
#include <hip/hip_ext.h>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <set>
#include <type_traits>

namespace poro {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

struct OctDims { int nx, ny, nz, hx, hy, hz, hxp; int64_t co; int no, own_z, nc; };   // nodes and half lengths per direction; hxp = row pitch (hx rounded up to even: 16-byte aligned rows, the pad entry stays zero); co = hxp hy hz entries per (component, octant)
// no = 8: octant form.  no = 4 (slab partitions): z is not split - every z index is "its own mirror image" (hz = nz, parity blocks 0..3 only); own_z = planes that count in dot products

// ---- the 8-point butterfly --------------------------------------------------------------------------------------------------------------
// index bit 0 / 1 / 2 = x / y / z.  Forward: in = values at the lower node (bit clear) and its mirror image (bit set), out = even (bit clear) and odd
// (bit set) parts; a direction whose lower index is the centre of an odd line has no mirror: even = the value, odd = 0.
__device__ __forceinline__ void bfly_fwd(double (&v)[8], const bool (&centre)[3]) {
#pragma unroll
  for (int bit = 0; bit < 3; ++bit) {
    const int m = 1 << bit;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (!(i & m)) {
      const double lo = v[i], hi = v[i | m];
      v[i] = centre[bit] ? lo : lo + hi; v[i | m] = centre[bit] ? 0.0 : lo - hi;
    }
  }
}
// Backward: in = the even-mode / odd-mode partial sums (a, b), out = node values v_k = a + b, v_k' = a - b (centre: v = a, the mirror slot is unused)
__device__ __forceinline__ void bfly_bwd(double (&v)[8], const bool (&centre)[3]) {
#pragma unroll
  for (int bit = 0; bit < 3; ++bit) {
    const int m = 1 << bit;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (!(i & m)) {
      const double a = v[i], b = v[i | m];
      v[i] = centre[bit] ? a : a + b; v[i | m] = centre[bit] ? a : a - b;
    }
  }
}
struct OctPos {
  int64_t node[8]; bool centre[3]; bool live[8]; double weight; bool valid, own;
  // lower-octant position idx = (kz hy + ky) hxp + kx -> the eight mirror nodes; live[m]: node m is distinct from the ones with fewer mirrored directions; valid: not a row pad
  __device__ __forceinline__ OctPos(const OctDims &D, int64_t idx) {
    const int kx = (int)(idx % D.hxp), ky = (int)((idx / D.hxp) % D.hy), kz = (int)(idx / ((int64_t)D.hxp * D.hy));
    valid = kx < D.hx;
    const int mx = D.no == 1 ? kx : D.nx - 1 - kx, my = D.no == 1 ? ky : D.ny - 1 - ky, mz = D.no == 8 ? D.nz - 1 - kz : kz;   // (no = 1: no direction is split - the plain nodal values in planar storage)
    centre[0] = mx == kx; centre[1] = my == ky; centre[2] = mz == kz; own = kz < D.own_z;
    weight = (centre[0] ? 1.0 : 0.5) * (centre[1] ? 1.0 : 0.5) * (centre[2] ? 1.0 : 0.5);   // |v|^2 over a mirror orbit = weight * sum of the squared parity parts
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int x = (m & 1) ? mx : kx, y = (m & 2) ? my : ky, z = (m & 4) ? mz : kz;
      node[m] = ((int64_t)z * D.ny + y) * D.nx + x;
      live[m] = !((m & 1) && centre[0]) && !((m & 2) && centre[1]) && !((m & 4) && centre[2]);
    }
  }
};

// Thread mapping of the vector kernels: thread t <-> (lower-octant position t / 3, component t % 3), so that the lanes of a wave walk through the node-interleaved
// vectors contiguously (dof = 3 node + c: 512 contiguous bytes per wave access, ascending for the lower nodes, descending for the mirror images) and through three
// contiguous streams of every octant array.  (First version: one thread per position, all three components - 8-byte accesses at 24-byte stride: 170 us for the
// direction update instead of 45.)
#define PORO_OCT_LOOP(D) for (int64_t t_ = (int64_t)blockIdx.x * kBlock + threadIdx.x; t_ < NC * (D).co; t_ += (int64_t)gridDim.x * kBlock)   /* NC = components per node: 3, or 2 for the planar form */
#define PORO_OCT_DECODE(D) const int64_t idx = t_ / NC; const int c = (int)(t_ - NC * idx); const OctPos P(D, idx); if (!P.valid) continue;

// q = H v: node-interleaved vector (dof = 3 node + c) -> octant form; masked dofs count as zero
template <int NC> __global__ void __launch_bounds__(kBlock) k_fdmo_from_nodal(OctDims D, const double *__restrict__ v, const uint8_t *__restrict__ inert, double *__restrict__ q) {
  PORO_OCT_LOOP(D) {
    PORO_OCT_DECODE(D)
    double w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { const int64_t dof = P.node[m] * NC + c; w[m] = (inert && inert[dof]) ? 0.0 : v[dof]; }
    bfly_fwd(w, P.centre);
#pragma unroll
    for (int o = 0; o < 8; ++o) if (o < D.no) q[(int64_t)(c * D.no + o) * D.co + idx] = w[o];
  }
}
template <int NC> __global__ void __launch_bounds__(kBlock) k_fdmo_to_nodal(OctDims D, const double *__restrict__ r, double *__restrict__ v) {
  PORO_OCT_LOOP(D) {
    PORO_OCT_DECODE(D)
    double w[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) w[o] = o < D.no ? r[(int64_t)(c * D.no + o) * D.co + idx] : 0.0;
    bfly_bwd(w, P.centre);
#pragma unroll
    for (int m = 0; m < 8; ++m) if (P.live[m]) v[P.node[m] * NC + c] = w[m];
  }
}

// ---- CG vector kernels with g, z in octant form (protocol of k_pcg_* in kernels_la.hip) ---------------------------------------------------
// g = H (A x - b), zero on the inert (Dirichlet) dofs
template <int NC> __global__ void __launch_bounds__(kBlock) k_fdmo_init_residual(OctDims D, double *__restrict__ g, const double *__restrict__ Ax, const double *__restrict__ b, const uint8_t *__restrict__ inert) {
  PORO_OCT_LOOP(D) {
    PORO_OCT_DECODE(D)
    double w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { const int64_t dof = P.node[m] * NC + c; w[m] = (inert && inert[dof]) ? 0.0 : Ax[dof] - b[dof]; }
    bfly_fwd(w, P.centre);
#pragma unroll
    for (int o = 0; o < 8; ++o) if (o < D.no) g[(int64_t)(c * D.no + o) * D.co + idx] = w[o];
  }
}
// d = -z (nodal); block partials of g.g and g.z
template <int NC> __global__ void __launch_bounds__(kBlock) k_fdmo_first_direction(OctDims D, double *__restrict__ d, const double *__restrict__ g, const double *__restrict__ z, double *partials) {
  __shared__ double sh[5];
  double gg = 0, gz = 0;
  PORO_OCT_LOOP(D) {
    PORO_OCT_DECODE(D)
    double w[8]; double s2 = 0, sz = 0;
#pragma unroll
    for (int o = 0; o < 8; ++o) { w[o] = 0.0; if (o < D.no) { const int64_t at = (int64_t)(c * D.no + o) * D.co + idx; const double gv = g[at]; w[o] = z[at]; s2 = fma(gv, gv, s2); sz = fma(gv, w[o], sz); } }
    if (P.own) { gg = fma(P.weight, s2, gg); gz += sz; }
    bfly_bwd(w, P.centre);
#pragma unroll
    for (int m = 0; m < 8; ++m) if (P.live[m]) d[P.node[m] * NC + c] = -w[m];
  }
  gg = block_sum(gg, sh); gz = block_sum(gz, sh);
  store_partial(partials, gg); store_partial(partials + kMaxPartials, gz);
}
// g += alpha H (A d) and the partials of g.g (g.z follows in a plain dot of the two octant arrays once z = P^-1 g exists)
template <int NC> __global__ void __launch_bounds__(kBlock) k_fdmo_update_g(OctDims D, PcgScalars *sc, int parity, double *__restrict__ g, const double *__restrict__ h, const uint8_t *__restrict__ inert, const double *partials_dh, double *partials_out, const double *red) {
  __shared__ double sh[5];
  if (sc->done) return;
  if (sc->finishing) { if (blockIdx.x == 0 && threadIdx.x == 0) sc->done = 1; return; }   // (see k_pcg_update_g_fused)
  const double dh = red ? red[0] : sum_partials(partials_dh, sh);
  const double alpha = sc->gh2[parity] / dh;
  double gg = 0;
  PORO_OCT_LOOP(D) {
    PORO_OCT_DECODE(D)
    // inert (Dirichlet) dofs: the residual stays exactly zero whatever the operator left in h.  The octant form exists only where every Dirichlet condition covers a
    // pair of opposite faces (build_fdm_u), so the eight mirror images of a dof are inert together: one mask byte per thread
    if (inert && inert[P.node[0] * NC + c]) continue;
    double w[8]; double s2 = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = h[P.node[m] * NC + c];
    bfly_fwd(w, P.centre);
#pragma unroll
    for (int o = 0; o < 8; ++o) if (o < D.no) { const int64_t at = (int64_t)(c * D.no + o) * D.co + idx; const double gv = fma(alpha, w[o], g[at]); g[at] = gv; s2 = fma(gv, gv, s2); }
    if (P.own) gg = fma(P.weight, s2, gg);
  }
  gg = block_sum(gg, sh);
  store_partial(partials_out, gg);
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->dh = dh; sc->alpha = alpha; }
}
// the same with TWO neighbouring positions (idx, idx + 1: same row, the pitch is even) per thread: 16-byte accesses on the octant side (twice as long contiguous pieces per
// component and instruction), the nodal side reads two dofs 8 NC bytes apart
template <int NC> __global__ void __launch_bounds__(kBlock) k_fdmo_update_g2(OctDims D, PcgScalars *sc, int parity, double *__restrict__ g, const double *__restrict__ h, const uint8_t *__restrict__ inert, const double *partials_dh, double *partials_out, const double *red) {
  __shared__ double sh[5];
  if (sc->done) return;
  if (sc->finishing) { if (blockIdx.x == 0 && threadIdx.x == 0) sc->done = 1; return; }
  const double dh = red ? red[0] : sum_partials(partials_dh, sh);
  const double alpha = sc->gh2[parity] / dh;
  double gg = 0;
  for (int64_t t_ = (int64_t)blockIdx.x * kBlock + threadIdx.x; t_ < NC * (D.co / 2); t_ += (int64_t)gridDim.x * kBlock) {
    const int64_t pr = t_ / NC; const int c = (int)(t_ - NC * pr); const int64_t idx = 2 * pr;
    const OctPos P0(D, idx), P1(D, idx + 1);
    double w0[8], w1[8];
    const bool on0 = P0.valid && !(inert && inert[P0.node[0] * NC + c]), on1 = P1.valid && !(inert && inert[P1.node[0] * NC + c]);
#pragma unroll
    for (int m = 0; m < 8; ++m) { w0[m] = on0 ? h[P0.node[m] * NC + c] : 0.0; w1[m] = on1 ? h[P1.node[m] * NC + c] : 0.0; }
    bfly_fwd(w0, P0.centre); bfly_fwd(w1, P1.centre);
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int o = 0; o < 8; ++o) if (o < D.no) {
      double2 *gp = reinterpret_cast<double2 *>(g + (int64_t)(c * D.no + o) * D.co + idx);
      double2 gv = *gp;
      gv.x = fma(alpha, w0[o], gv.x); gv.y = fma(alpha, w1[o], gv.y);       // (inert dofs and row pads: w = 0, g stays what it is - zero)
      *gp = gv; s0 = fma(gv.x, gv.x, s0); s1 = fma(gv.y, gv.y, s1);
    }
    if (P0.own) gg = fma(P0.weight, s0, fma(P1.weight, s1, gg));
  }
  gg = block_sum(gg, sh);
  store_partial(partials_out, gg);
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->dh = dh; sc->alpha = alpha; }
}
// x += alpha d, then d = beta d - H' z unless the solve just finished (k_pcg_update_d_fused with the explicit z in octant form)
template <int NC> __global__ void __launch_bounds__(kBlock) k_fdmo_update_d(OctDims D, PcgScalars *sc, int parity, int it, double *__restrict__ x, double *__restrict__ d, const double *__restrict__ z, int64_t n_u, const double *partials_in, const double *red) {
  __shared__ double sh[5];
  if (sc->done) return;
  const double gg = red ? red[0] : sum_partials(partials_in, sh), gz = red ? red[1] : sum_partials(partials_in + kMaxPartials, sh);
  const double res = sqrt(gg), gh_old = sc->gh2[parity], alpha = sc->alpha;
  const bool conv = res <= sc->tol, fail = !conv && it >= sc->max_iter;
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->gg = gg; sc->gz = gz; sc->res = res; sc->it = it; }
  if (conv || fail) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_u; i += (int64_t)gridDim.x * kBlock) x[i] = fma(alpha, d[i], x[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->converged = conv ? 1 : 0; sc->finishing = 1; }
    return;
  }
  const double beta = gz / gh_old;
  PORO_OCT_LOOP(D) {          // (two positions per thread, as in k_fdmo_update_g2, were measured here too: 72 instead of 59 us - the paired read-modify-writes of the nodal vectors cost more than the 16-byte octant reads save)
    PORO_OCT_DECODE(D)
    double w[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) w[o] = o < D.no ? z[(int64_t)(c * D.no + o) * D.co + idx] : 0.0;
    bfly_bwd(w, P.centre);
#pragma unroll
    for (int m = 0; m < 8; ++m) if (P.live[m]) {
      const int64_t dof = P.node[m] * NC + c;
      const double dv = d[dof], xv = x[dof];
      x[dof] = fma(alpha, dv, xv); d[dof] = fma(beta, dv, -w[m]);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->gh2[parity ^ 1] = gz; }
}

// block partials of a . b over the planes this rank owns (slab form: the upper shared plane is the neighbour's)
__global__ void __launch_bounds__(kBlock) k_fdmo_dot_owned(OctDims D, const double *__restrict__ a, const double *__restrict__ b, double *partials, const PcgScalars *gate) {
  __shared__ double sh[5];
  if (gate && (gate->done | gate->finishing)) return;
  // every block of a, b is [plane][position]: the owned planes are its leading part (row pads are zeros in both)
  const int64_t per = (int64_t)D.own_z * D.hxp * D.hy;          // (hxp even: pairs)
  const int nblk = D.nc * D.no, share = gridDim.x / nblk, blk = blockIdx.x / max(share, 1), sub = blockIdx.x - blk * share;   // workgroups per block of the arrays; the remainder idles
  double acc = 0;
  if (share > 0 && blk < nblk) {
    const double2 *__restrict__ a2 = reinterpret_cast<const double2 *>(a + blk * D.co), *__restrict__ b2 = reinterpret_cast<const double2 *>(b + blk * D.co);
    for (int64_t e = (int64_t)sub * kBlock + threadIdx.x; e < per / 2; e += (int64_t)share * kBlock) { const double2 u = a2[e], v = b2[e]; acc = fma(u.x, v.x, fma(u.y, v.y, acc)); }
  }
  acc = block_sum(acc, sh);
  store_partial(partials, acc);
}

// ---- slab form: the z transform needs whole global lines.  Columns = (component, quadrant, plane position), grouped in chunks of cw; rank q transforms chunks
//      [q cps, (q + 1) cps).  Two all-to-alls of [rank][plane][share column] buffers; the z butterfly rides in the unpack / pack next to the transposed array ----
struct SlabGeo { int cw, nchunk, cps, chunk_total, rank, my_chunks, hzg, ng, np; int64_t scols, pl, co; };
// transposed array -> send buffer: every rank's planes (shared ones to both owners), v_k = a + b, v_k' = a - b
__global__ void __launch_bounds__(256) k_fdmo_slab_scatter_pack(SlabGeo S, int rows, const int64_t *__restrict__ row_out, const int32_t *__restrict__ row_kz,
                                                                  const double *__restrict__ tz, double *__restrict__ buf, const PcgScalars *gate) {
  if (gate && (gate->done | gate->finishing)) return;
  const int per_row = S.my_chunks * S.cw, total = rows * per_row;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int row = e / per_row, col = e - row * per_row, cl = col / S.cw, j = col - cl * S.cw;
    const int kz = row_kz[row];
    if (S.np == 1) { buf[row_out[row] + col] = tz[((int64_t)cl * S.hzg + kz) * S.cw + j]; continue; }     // (scalar systems: no parity split)
    const int mz = S.ng - 1 - kz, kh = min(kz, mz);
    const double a = tz[((int64_t)(2 * cl) * S.hzg + kh) * S.cw + j], b = tz[((int64_t)(2 * cl + 1) * S.hzg + kh) * S.cw + j];
    buf[row_out[row] + col] = kz == mz ? a : (kz < mz ? a + b : a - b);
  }
}

// ---- the transform pass --------------------------------------------------------------------------------------------------------------------
struct OctPass {
  int mode;                 // 0: contract columns, then rows (pass 1); 1: rows, eigenvalue scaling, rows (pass 2); 2: rows, then columns (pass 3)
  int R, C;                 // valid rows / columns of a block (mode 1: C = columns of a full chunk, the last chunk has fewer)
  int nt_r, nt_c;           // 16-wide tiles covering them
  int kk1, kk2;             // k-steps (of 4) of the two GEMMs
  int nblk;                 // blocks per (component, octant)
  int64_t co_stride, blk_stride, row_stride;
  int bit1, bit2;           // octant bit that selects the parity of the matrices of GEMM 1 / 2
  int hx, pl;               // mode 1: columns of a plane = hx hy; a column's plane offset -> (my, mx)
  const double *T1[3][2], *T2[3][2];                     // [component][parity], MFMA fragment order [tile][4 NT][64]
  const double *lam_z[3][2]; double cz[3]; const double *bxy;   // mode 1: eigenvalues of the line direction; bxy[(4 c + (o & 3)) pl + column] = the other two directions' share
  const double *in_blk[3]; double *out_blk[3]; int use_in_off, use_out_off;   // batched scalar systems (no_shift = 0, up to 3 blocks = right-hand sides in separate vectors): every block's own vector instead of `in` / `out` + block * co_stride
  int bxy_cmul;                 // bxy table: blocks per component (4: displacement system; 0: the scalar systems share one table)
  int no_shift;                 // log2 of the blocks per component (3: octants, 2: quadrants of the slab form, 0: scalar system)
  int slab_z /* parity parts of the transposed blocks: 2, or 1 for the scalar systems; 0: not the slab form */, chunk0, chunk_total, nchunk;   // slab form, pass 2: workgroup = (local chunk, z parity) of the transposed array; its global chunk number gives (component, quadrant, chunk of the plane)
  // slab form: the copies around the all-to-alls ride in the passes.  The exchange buffer is [send | recv], each [rank q][plane][share column]; column X = (block) nchunk cw +
  // plane position belongs to rank q = X / scols, so its address is X + q (pitch - 1) scols + plane scols: pass 1 (slab_io = 1) stores there (planes >= store_planes are the
  // neighbour's and are not sent; the own share goes straight to the receive half), pass 3 (slab_io = 2) loads the scattered planes from there; pass 2 (row_in): offset of every
  // global plane in the gathered buffer - the block is loaded as even / odd combination of a plane and its mirror image
  int slab_io, store_planes, ng, scols, col_unit, rank; float inv_scols; int64_t dest_stride, recv_off; const int64_t *row_in;
  int vec2;                     // rows are 16-byte aligned (even pitch, even chunk offsets): 16-byte block loads; 0: 8-byte loads (the scalar Q1 systems keep their nodal layout)
  const PcgScalars *gate;       // inside a PCG iteration: the launch is a no-op once the solve has finished (the host enqueues iterations ahead of the device-side stopping test)
  unsigned long long *stamps;   // diagnostic (PORO_FDMO_STAMPS): per block 8 words: 100 MHz time at start / block in LDS / GEMM 1 done / intermediate in LDS / GEMM 2 done / stored, HW_ID, XCC_ID
};

constexpr int kFdmoMaxTiles = 8;                                  // 16-wide tiles per half line the transform kernel is instantiated for (half lines of <= 128 entries)
template <int NT> constexpr int pass_waves() { return NT == 5 ? 4 : NT; }   // waves of a workgroup: one per tile row / column, except NT = 5 (four waves share 25 tiles)
template <int NT> struct PassGeom {
  static constexpr int PADN = 16 * NT, KKP = 4 * NT;
  static constexpr int LDA = PADN + 2;                       // data as the A operand: lane (i, kq) reads [16 t + i][4 kk + kq]; LD = 2 * odd (mod 32) keeps a 32-lane group on 32 bank pairs
  static constexpr int LDB = PADN + ((NT & 1) ? 0 : 16);     // data as the B operand: lane (j, kq) reads [4 kk + kq][16 w + j]; LD = 16 (mod 32)
  static constexpr int LDMAX = LDA > LDB ? LDA : LDB;
};

// One item (= one (component, octant, block)) per workgroup; the hardware dispatcher balances the items over the CUs.
// Workgroup shape (census of resident workgroups per CU, profiles/r03_lds_residency.txt): a 5-wave workgroup - the natural shape for 5 x 5 tiles - is given the
// registers of an 8-wave one (two slots on every SIMD), which left 1.2-2 workgroups resident and the matrix pipe idle 55 % of the time (profiles/r03_fdmo_stamps_v1.txt).
// 4-wave workgroups keep three resident at up to 128 registers, one wave of each on every SIMD.  So for NT = 5 FOUR waves share the tiles:
//   passes 1 / 3 (5 x 5 tiles): wave w owns tile row (or column) w in full plus tile (4, w) [(w, 4)]; tile (4, 4) goes to the wave whose number equals the item number
//     mod 4, so that over the items every SIMD gets the same share (6.25 tiles per wave on average; the two GEMM bodies - with and without the seventh tile - are
//     separate branch-free instruction streams);
//   pass 2 (lines are independent): chunks of 64 columns = 5 x 4 tiles, wave w owns (w, 0..3) and (4, w): five tiles each, nothing left over.
// VAR = 0: the octant form on one rank (24 blocks, aligned rows, no exchange buffer, no per-block offsets) with those switches folded at compile time;
// VAR = 1: the quadrant form of the displacement system on slabs (pass 1 stores into / pass 3 loads from the exchange buffer, pass 2 forms the parity parts on load), likewise;
// VAR = 2: everything else by run-time switches (scalar systems, batched right-hand sides)
template <int NT, int MODE, int VAR>
__global__ void __launch_bounds__(64 * pass_waves<NT>())
k_fdmo_pass(OctPass P, const double *in, double *out) {
  // (the kernel argument stays in the kernarg segment - a modified copy would live in scratch because of its dynamically indexed members)
  constexpr bool GENERAL = VAR == 2, SLAB = VAR == 1;
  const int f_vec2 = GENERAL ? P.vec2 : 1, f_slab_z = GENERAL ? P.slab_z : (SLAB && MODE == 1 ? 2 : 0), f_slab_io = GENERAL ? P.slab_io : (SLAB && MODE == 0 ? 1 : SLAB && MODE == 2 ? 2 : 0),
            f_use_in = GENERAL ? P.use_in_off : 0, f_use_out = GENERAL ? P.use_out_off : 0, f_no_shift = GENERAL ? P.no_shift : (SLAB ? 2 : 3), f_bxy_cmul = GENERAL ? P.bxy_cmul : 4;
  const int64_t *const f_row_in = (GENERAL || (SLAB && MODE == 1)) ? P.row_in : nullptr;
  typedef PassGeom<NT> Gm;
  constexpr int NW = pass_waves<NT>();                       // waves
  constexpr bool EXTRA = NT > NW;                            // NT == 5: the fifth tile row / column is shared out
  constexpr int XT = NT - 1;                                 // index of that tile row / column
  constexpr int NL = (MODE == 1 && EXTRA) ? NW : NT;         // tiles a wave loops over beside its own (pass 2: column tiles of a chunk)
  constexpr bool CORNER = EXTRA && MODE != 1;                // tile (XT, XT) exists
  constexpr int NACC = NL + (EXTRA ? 1 : 0) + (CORNER ? 1 : 0);
  constexpr bool kColsFirst = MODE == 0, kColsSecond = MODE == 2;
  constexpr int LD1 = kColsFirst ? Gm::LDA : Gm::LDB, LD2 = kColsSecond ? Gm::LDA : Gm::LDB;
  constexpr int NS = EXTRA ? 2 : 1;                          // fragment streams per wave: its own tile row of T (and the shared one)
  constexpr int PADC = 16 * NL;                              // padded block columns
  extern __shared__ double L[];                              // PADN x LDMAX doubles (dynamic: more than 64 KB from NT = 6 on)
  if (P.gate && (P.gate->done | P.gate->finishing)) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  int b, c, o, co = 0; int64_t base;
  if (MODE == 1 && f_slab_z) {
    const int G = P.chunk0 + (int)(blockIdx.x >> (f_slab_z - 1)), cq = G / P.nchunk;
    if (G >= P.chunk_total) return;                                  // (the last rank's share may be short; workgroup-uniform, before any barrier)
    b = G - cq * P.nchunk; c = cq >> 2; o = (cq & 3) | ((blockIdx.x & (f_slab_z - 1)) << 2); base = (int64_t)blockIdx.x * P.blk_stride;
  } else {
    co = blockIdx.x / P.nblk;
    b = blockIdx.x % P.nblk; c = co >> f_no_shift; o = co & ((1 << f_no_shift) - 1); base = (int64_t)co * P.co_stride + (int64_t)b * P.blk_stride;
  }
  const int64_t base_in = f_use_in ? (int64_t)b * P.blk_stride : base, base_out = f_use_out ? (int64_t)b * P.blk_stride : base;
  if (f_use_in) in = P.in_blk[co];
  if (f_use_out) out = P.out_blk[co];
  const bool heavy = CORNER && w == (int)(blockIdx.x & (NW - 1));   // this wave also computes tile (XT, XT)
  const int R = P.R, C = MODE == 1 ? min(P.C, P.pl - b * P.C) : P.C;
  const double *__restrict__ T1 = P.T1[c][(o >> P.bit1) & 1] + lane, *__restrict__ T2 = P.T2[c][(o >> P.bit2) & 1] + lane;
  // slab form: address of plane position `col` of this block in the exchange buffer
  // (a block's columns lie in at most two shares when a share is at least a block long - up to 12 ranks: the two bases are wave-uniform scalars)
  int slab_q0 = 0, slab_sw = 0; int64_t slab_base[2] = {0, 0};
  if (MODE != 1 && f_slab_io) {
    const int X0 = co * P.col_unit; slab_q0 = X0 / P.scols; slab_sw = (slab_q0 + 1) * P.scols - X0;
    for (int k = 0; k < 2; ++k) slab_base[k] = (int64_t)X0 + (int64_t)(slab_q0 + k) * P.dest_stride + (int64_t)b * P.scols + ((f_slab_io == 2 || slab_q0 + k == P.rank) ? P.recv_off : 0);
  }
  auto slab_at = [&](int col) -> int64_t {
    if (P.scols >= P.col_unit) return (int64_t)col + (col >= slab_sw ? slab_base[1] : slab_base[0]);
    const int X = co * P.col_unit + col;
    int q = (int)((float)X * P.inv_scols); q -= q * P.scols > X; q += (q + 1) * P.scols <= X;      // X / scols (X < 2^24)
    return (int64_t)X + (int64_t)q * P.dest_stride + (int64_t)b * P.scols + ((f_slab_io == 2 || q == P.rank) ? P.recv_off : 0);
  };
  auto stamp = [&](int k) { if (P.stamps && tid == 0) P.stamps[(int64_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime(); };
  stamp(0);
  // ---- block -> LDS, zero padded to PADN x PADC (the padding meets zero columns of T, but must be finite).  Rows are 16-byte aligned (even pitch, even chunk
  //      offsets, even C): 16-byte loads and LDS stores, all loads in flight before the first LDS store ----
  {
    constexpr int HP = PADC / 2, TOT = Gm::PADN * HP, PER = (TOT + 64 * NW - 1) / (64 * NW);
    double2 stage[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 64 * NW, r = e / HP, c2 = 2 * (e - r * HP);
      if (MODE == 1 && f_row_in) {             // gathered planes -> parity part of the global line: e_k = v_k + v_k', o_k = v_k - v_k' (centre plane: e = v, o = 0)
        const bool ok = e < TOT && r < R && c2 < C, split = f_slab_z == 2; const int rr = ok ? r : 0, mr = split ? P.ng - 1 - rr : rr; const int64_t cb = (int64_t)(blockIdx.x >> (f_slab_z - 1)) * P.C + c2;
        const double2 lo = ok ? *reinterpret_cast<const double2 *>(in + f_row_in[rr] + cb) : double2{0.0, 0.0}, hi = (ok && mr != rr) ? *reinterpret_cast<const double2 *>(in + f_row_in[mr] + cb) : double2{0.0, 0.0};
        const bool odd = split && (blockIdx.x & 1);
        stage[u].x = odd ? (mr != rr ? lo.x - hi.x : 0.0) : lo.x + hi.x; stage[u].y = odd ? (mr != rr ? lo.y - hi.y : 0.0) : lo.y + hi.y;
      } else if (MODE == 2 && f_slab_io == 2) {
        if (f_vec2) stage[u] = (e < TOT && r < R && c2 < C) ? *reinterpret_cast<const double2 *>(in + slab_at(r * (int)P.row_stride + c2)) : double2{0.0, 0.0};
        else { stage[u].x = (e < TOT && r < R && c2 < C) ? in[slab_at(r * (int)P.row_stride + c2)] : 0.0; stage[u].y = (e < TOT && r < R && c2 + 1 < C) ? in[slab_at(r * (int)P.row_stride + c2 + 1)] : 0.0; }
      }
      else if (f_vec2) stage[u] = (e < TOT && r < R && c2 < C) ? *reinterpret_cast<const double2 *>(in + base_in + (int64_t)r * P.row_stride + c2) : double2{0.0, 0.0};
      else { const double *src = in + base_in + (int64_t)r * P.row_stride + c2; stage[u].x = (e < TOT && r < R && c2 < C) ? src[0] : 0.0; stage[u].y = (e < TOT && r < R && c2 + 1 < C) ? src[1] : 0.0; }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int e = tid + u * 64 * NW, r = e / HP, c2 = 2 * (e - r * HP); if (e < TOT) *reinterpret_cast<double2 *>(&L[r * LD1 + c2]) = stage[u]; }
  }
  // transform-matrix fragments in two rotating register buffers of 4 k-steps: the 2 NT chunks of the two GEMMs form one sequence, and a buffer is refilled with the
  // chunk after next as soon as the MFMAs that used it have been issued
  constexpr int CHK = EXTRA ? 2 : 4, NCH = Gm::KKP / CHK;     // k-steps per chunk (two fragment streams at NT = 5: smaller chunks keep the kernel within the registers of three resident workgroups), chunks per GEMM
  double tf[2][NS][CHK];
  auto load_chunk = [&](int buf, int step) {                 // step < NCH: chunk `step` of T1, else chunk step - NCH of T2
    const double *__restrict__ T = step < NCH ? T1 : T2; const int ch = step < NCH ? step : step - NCH;
#pragma unroll
    for (int k = 0; k < CHK; ++k) {
      tf[buf][0][k] = T[((int64_t)w * Gm::KKP + CHK * ch + k) * 64];
      if constexpr (EXTRA) tf[buf][1][k] = T[((int64_t)XT * Gm::KKP + CHK * ch + k) * 64];
    }
  };
  v4d acc[NACC];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = v4d{0, 0, 0, 0};
  };
  // one GEMM = NCH chunks of CHK k-steps.  Every tile of the padded problem is computed (the padding is exact zeros: no predicates inside), only whole trailing k-steps are skipped.
  //   contract columns: acc[t] = tile (t, w) = sum_k data(rows of tile t, k) T(rows of tile w, k); extra: acc[NL] = tile (w, XT), acc[NL + 1] = tile (XT, XT)
  //   contract rows:    acc[u] = tile (w, u) = sum_k T(rows of tile w, k) data(k, columns of tile u); extra: acc[NL] = tile (XT, w), acc[NL + 1] = tile (XT, XT)
  auto gemm = [&](auto contract_cols, auto ld_c, auto first_step_c, auto corner_c, int kk_n) {
    constexpr bool kCols = decltype(contract_cols)::value, kCorner = decltype(corner_c)::value; constexpr int LD = decltype(ld_c)::value, S0 = decltype(first_step_c)::value;
    const double *La = kCols ? L + j * LD + kq : L + kq * LD + j;
    const double *Lw = kCols ? La + 16 * w * LD : La + 16 * w;                 // this wave's own tile row / column of the data (runtime w)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int buf = (S0 + ch) & 1;
#pragma unroll
      for (int k = 0; k < CHK; ++k) {
        const int kk = CHK * ch + k;
        if (kk < Gm::KKP - 3 || kk < kk_n) {                 // (kk_n >= KKP - 3 always: wave-uniform branches on the last three k-steps only)
          double d[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) if (t < NL || (kCorner && t == XT)) d[t] = kCols ? La[16 * t * LD + 4 * kk] : La[4 * kk * LD + 16 * t];
          double dw = 0; if constexpr (EXTRA) dw = kCols ? Lw[4 * kk] : Lw[4 * kk * LD];
#pragma unroll
          for (int t = 0; t < NL; ++t) acc[t] = kCols ? __builtin_amdgcn_mfma_f64_16x16x4f64(d[t], tf[buf][0][k], acc[t], 0, 0, 0) : __builtin_amdgcn_mfma_f64_16x16x4f64(tf[buf][0][k], d[t], acc[t], 0, 0, 0);
          if constexpr (EXTRA) acc[NL] = kCols ? __builtin_amdgcn_mfma_f64_16x16x4f64(dw, tf[buf][1][k], acc[NL], 0, 0, 0) : __builtin_amdgcn_mfma_f64_16x16x4f64(tf[buf][1][k], dw, acc[NL], 0, 0, 0);
          if constexpr (kCorner) acc[NL + 1] = kCols ? __builtin_amdgcn_mfma_f64_16x16x4f64(d[XT], tf[buf][1][k], acc[NL + 1], 0, 0, 0) : __builtin_amdgcn_mfma_f64_16x16x4f64(tf[buf][1][k], d[XT], acc[NL + 1], 0, 0, 0);
        }
      }
      if (S0 + ch + 2 < 2 * NCH) load_chunk(buf, S0 + ch + 2);
    }
  };
  // accumulator a holds tile (tr, tc): element (16 tr + 4 q + kq, 16 tc + j)
  auto tile_of = [&](int a, int &tr, int &tc, bool cols) {
    if (a < NL) { tr = cols ? a : w; tc = cols ? w : a; }
    else if (a == NL) { tr = cols ? w : XT; tc = cols ? XT : w; }
    else { tr = XT; tc = XT; }
  };
  typedef std::integral_constant<bool, kColsFirst> CF; typedef std::integral_constant<int, LD1> L1c;
  typedef std::integral_constant<bool, kColsSecond> CS; typedef std::integral_constant<int, LD2> L2c;
  typedef std::integral_constant<int, 0> S0c; typedef std::integral_constant<int, NCH> S1c;
  __builtin_amdgcn_s_setprio(3);   // the few instructions of the load phase go ahead of the co-resident workgroups' matrix streams
  load_chunk(0, 0); load_chunk(1, 1);
  zero_acc();
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();
  stamp(1);
  if (heavy) gemm(CF{}, L1c{}, S0c{}, std::integral_constant<bool, CORNER>{}, P.kk1); else gemm(CF{}, L1c{}, S0c{}, std::false_type{}, P.kk1);
  __syncthreads();                                   // everybody has finished reading the input block
  stamp(2);
  // ---- first result -> LDS in the layout of the second GEMM (MODE 1: divided by the eigenvalue sums on the way) ----
  if constexpr (MODE == 1) {
    // rows of this wave's tiles: tile row w (and the shared row XT); columns: tile column of the accumulator.  One entry at a time (sched_barrier), otherwise the
    // reciprocal sequences of all of them pile up in registers
    const double *lamz = P.lam_z[c][(o >> 2) & 1], *bx = P.bxy + (int64_t)(f_bxy_cmul * c + (o & 3)) * P.pl;
    const double czc = P.cz[c];
    double lzw[4], lzx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { lzw[q] = czc * lamz[16 * w + 4 * q + kq]; lzx[q] = EXTRA ? czc * lamz[16 * XT + 4 * q + kq] : 0.0; }
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      int tr, tc; tile_of(a, tr, tc, false);
      const double bxy = bx[min(b * P.C + 16 * tc + j, P.pl - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // reciprocal by v_rcp_f64 + one Newton step; modes that do not exist carry lam = inf and give exactly 0
        const double den = (a < NL ? lzw[q] : lzx[q]) + bxy;
        double r = __builtin_amdgcn_rcp(den);
        r = den < 1e300 ? fma(r, fma(-den, r, 1.0), r) : 0.0;
        L[(16 * tr + 4 * q + kq) * LD2 + 16 * tc + j] = acc[a][q] * r;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      int tr, tc; tile_of(a, tr, tc, kColsFirst);
      if (a == NL + 1 && !heavy) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) L[(16 * tr + 4 * q + kq) * LD2 + 16 * tc + j] = acc[a][q];
    }
  }
  zero_acc();
  __syncthreads();
  stamp(3);
  if (heavy) gemm(CS{}, L2c{}, S1c{}, std::integral_constant<bool, CORNER>{}, P.kk2); else gemm(CS{}, L2c{}, S1c{}, std::false_type{}, P.kk2);
  if (P.stamps) { __syncthreads(); stamp(4); }
  // ---- store ----
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    int tr, tc; tile_of(a, tr, tc, kColsSecond);
    if (a == NL + 1 && !heavy) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 16 * tr + 4 * q + kq, cc = 16 * tc + j;
      if (MODE == 0 && f_slab_io == 1) { if (r < R && cc < C && b < P.store_planes) out[slab_at(r * (int)P.row_stride + cc)] = acc[a][q]; }
      else if (r < R && cc < C) out[base_out + (int64_t)r * P.row_stride + cc] = acc[a][q];
    }
  }
  if (P.stamps) {
    __builtin_amdgcn_s_waitcnt(0); __syncthreads(); stamp(5);
    if (tid == 0) { unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); P.stamps[(int64_t)blockIdx.x * 8 + 6] = hw; P.stamps[(int64_t)blockIdx.x * 8 + 7] = xcc; }
  }
}

// ---- planar (2D) form: the blocks are whole (component, quadrant) planes of up to ~350 x 350 entries - too large for one workgroup's LDS, so each of the four
//      1D transforms is a batched tiled GEMM of its own: C_b (M x N) = A_b (M x K) B_b (K x N), 64 x 64 tiles, 16-deep LDS stages (double buffered), 4 waves x (2 x 2) MFMA tiles.
//      An operand whose unit stride runs along K is staged "m-major" ([64][16 + 2]), one whose unit stride runs along M / N "k-major" ([16][64 + 16]): both give
//      conflict-free fragment reads (ds_read_b64: 32 lanes per cycle on 32 bank pairs) and contiguous global loads ----
struct Gemm2D {
  int M, N, K, nb;
  int64_t rsA, csA, rsB, csB, rsC;       // element strides (C: unit column stride)
  const double *A[8], *B[8]; double *C[8];   // per batch entry
  int scale; const double *lamM[8], *lamN[8]; double cM[8], cN[8];   // epilogue: C[m][n] /= cM lamM[m] + cN lamN[n] (inf -> 0)
  const PcgScalars *gate;
};
template <bool KCONTIG> struct GemmLds { static constexpr int LD = KCONTIG ? 18 : 80, SIZE = KCONTIG ? 64 * 18 : 16 * 80; };
template <bool AK, bool BK>
__global__ void __launch_bounds__(256) k_fdmo_gemm2d(Gemm2D G) {
  typedef GemmLds<AK> LA; typedef GemmLds<BK> LB;
  __shared__ double As[2][LA::SIZE], Bs[2][LB::SIZE];
  if (G.gate && (G.gate->done | G.gate->finishing)) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1, j = lane & 15, kq = lane >> 4;
  const int tiles_n = (G.N + 63) / 64, tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n, b = blockIdx.y;
  const int m0 = 64 * tm, n0 = 64 * tn;
  const double *__restrict__ Ab = G.A[b], *__restrict__ Bb = G.B[b];
  // global -> registers: 4 consecutive elements along the operand's unit-stride direction per thread
  double ra[4], rb[4];
  auto gload = [&](int k0) {
    if (AK) { const int m = m0 + (tid >> 2), k = k0 + 4 * (tid & 3);
#pragma unroll
      for (int u = 0; u < 4; ++u) ra[u] = (m < G.M && k + u < G.K) ? Ab[(int64_t)m * G.rsA + (int64_t)(k + u) * G.csA] : 0.0; }
    else { const int k = k0 + (tid >> 4), m = m0 + 4 * (tid & 15);
#pragma unroll
      for (int u = 0; u < 4; ++u) ra[u] = (k < G.K && m + u < G.M) ? Ab[(int64_t)(m + u) * G.rsA + (int64_t)k * G.csA] : 0.0; }
    if (BK) { const int n = n0 + (tid >> 2), k = k0 + 4 * (tid & 3);
#pragma unroll
      for (int u = 0; u < 4; ++u) rb[u] = (n < G.N && k + u < G.K) ? Bb[(int64_t)(k + u) * G.rsB + (int64_t)n * G.csB] : 0.0; }
    else { const int k = k0 + (tid >> 4), n = n0 + 4 * (tid & 15);
#pragma unroll
      for (int u = 0; u < 4; ++u) rb[u] = (k < G.K && n + u < G.N) ? Bb[(int64_t)k * G.rsB + (int64_t)(n + u) * G.csB] : 0.0; }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (AK) As[buf][(tid >> 2) * LA::LD + 4 * (tid & 3) + u] = ra[u]; else As[buf][(tid >> 4) * LA::LD + 4 * (tid & 15) + u] = ra[u];
      if (BK) Bs[buf][(tid >> 2) * LB::LD + 4 * (tid & 3) + u] = rb[u]; else Bs[buf][(tid >> 4) * LB::LD + 4 * (tid & 15) + u] = rb[u];
    }
  };
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) acc[a][c] = v4d{0, 0, 0, 0};
  const int nk = (G.K + 15) / 16;
  gload(0); lstore(0);
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) gload(16 * (s + 1));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) { const int m = 32 * wm + 16 * a + j, k = 4 * kk + kq; fa[a] = AK ? As[buf][m * LA::LD + k] : As[buf][k * LA::LD + m]; }
#pragma unroll
      for (int c = 0; c < 2; ++c) { const int n = 32 * wn + 16 * c + j, k = 4 * kk + kq; fb[c] = BK ? Bs[buf][n * LB::LD + k] : Bs[buf][k * LB::LD + n]; }
#pragma unroll
      for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[c], acc[a][c], 0, 0, 0);
    }
    if (s + 1 < nk) lstore(buf ^ 1);       // (the other buffer: everybody finished reading it before the barrier that ended the previous stage)
    __syncthreads();
  }
  double *__restrict__ Cb = G.C[b];
#pragma unroll
  for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c) {
    const int col = n0 + 32 * wn + 16 * c + j;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = m0 + 32 * wm + 16 * a + 4 * q + kq;
      if (row < G.M && col < G.N) {
        double v = acc[a][c][q];
        if (G.scale) { const double den = G.cM[b] * G.lamM[b][row] + G.cN[b] * G.lamN[b][col]; v = den < 1e300 ? v / den : 0.0; }
        Cb[(int64_t)row * G.rsC + col] = v;
      }
    }
  }
}

// ---- slab form, z stage with BOTH parity parts of a line in one workgroup (NT = 2, 4, 5) --------------------------------------------------------------------------
// A workgroup takes 8 NL real columns of the rank's share: it loads every gathered plane ONCE, forms the even part e_k = v_k + v_k' into the left half of the LDS
// block and the odd part o_k = v_k - v_k' into the right half, runs the half-size forward / scale / backward chain on both halves at once (the column tiles of the
// left half use the even-mode matrices, those of the right half the odd-mode ones: two fragment streams per wave, + one for the shared tile row at NT = 5), and
// stores v_k = a + b, v_k' = a - b straight into the scattering all-to-all's buffer - every plane to the one or two ranks that hold it (dst table).  Replaces the
// two-workgroups-per-chunk pass (each parity read every plane pair again) and the separate scatter kernel.
template <int NT>
__global__ void __launch_bounds__(64 * (NT < 4 ? NT : 4))
k_fdmo_zpass_both(OctPass P, const int64_t *__restrict__ dst /* [ng][2]: offsets of a plane's rows in the exchange buffer, -1 = none */, const double *in, double *out) {
  typedef PassGeom<NT> Gm;
  constexpr int NW = NT < 4 ? NT : 4; constexpr bool EXTRA = NT > NW; constexpr int XT = NT - 1;
  constexpr int NL = NW, HT = NL / 2, HC = 16 * HT;            // column tiles of the LDS block (left half: even part, right half: odd part), real columns of a chunk
  static_assert(NL % 2 == 0, "both-parity z pass: an even number of column tiles");
  constexpr int LD = Gm::LDB, NACC = NL + (EXTRA ? 1 : 0), NS = EXTRA ? 3 : 2;
  __shared__ double L[Gm::PADN * Gm::LDB];
  if (P.gate && (P.gate->done | P.gate->finishing)) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
  const int G = P.chunk0 + (int)blockIdx.x, cq = G / P.nchunk;
  if (G >= P.chunk_total) return;
  const int b = G - cq * P.nchunk, c = cq >> 2, qd = cq & 3;
  const int R = P.R, C = min(HC, P.pl - b * HC);
  const int pw = w >= HT ? 1 : 0;                              // parity of the column tile this wave's share of the extra tile row belongs to
  const int64_t cb = (int64_t)blockIdx.x * HC;                 // first column of the chunk inside the rank's share
  // ---- gathered planes -> parity parts in LDS ----
  {
    constexpr int HP = HC / 2, TOT = Gm::PADN * HP, PER = (TOT + 64 * NW - 1) / (64 * NW);
    double2 lo[PER], hi[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 64 * NW, r = e / HP, c2 = 2 * (e - r * HP);
      const bool ok = e < TOT && r < R && c2 < C; const int rr = ok ? r : 0, mr = P.ng - 1 - rr;
      lo[u] = ok ? *reinterpret_cast<const double2 *>(in + P.row_in[rr] + cb + c2) : double2{0.0, 0.0};
      hi[u] = (ok && mr != rr) ? *reinterpret_cast<const double2 *>(in + P.row_in[mr] + cb + c2) : double2{0.0, 0.0};
      if (ok && mr == rr) hi[u] = double2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 64 * NW, r = e / HP, c2 = 2 * (e - r * HP);
      if (e < TOT) {
        const bool centre = r < R && (P.ng - 1 - r) == r;
        *reinterpret_cast<double2 *>(&L[r * LD + c2]) = double2{lo[u].x + hi[u].x, lo[u].y + hi[u].y};
        *reinterpret_cast<double2 *>(&L[r * LD + HC + c2]) = centre ? double2{0.0, 0.0} : double2{lo[u].x - hi[u].x, lo[u].y - hi[u].y};
      }
    }
  }
  constexpr int CHK = 2, NCH = Gm::KKP / CHK;
  double tf[2][NS][CHK];
  const double *__restrict__ T1e = P.T1[c][0] + lane, *__restrict__ T1o = P.T1[c][1] + lane, *__restrict__ T2e = P.T2[c][0] + lane, *__restrict__ T2o = P.T2[c][1] + lane;
  auto load_chunk = [&](int buf, int step) {
    const bool first = step < NCH; const int ch = first ? step : step - NCH;
    const double *__restrict__ Te = first ? T1e : T2e, *__restrict__ To = first ? T1o : T2o;
#pragma unroll
    for (int k = 0; k < CHK; ++k) {
      tf[buf][0][k] = Te[((int64_t)w * Gm::KKP + CHK * ch + k) * 64];
      tf[buf][1][k] = To[((int64_t)w * Gm::KKP + CHK * ch + k) * 64];
      if constexpr (EXTRA) tf[buf][2][k] = (pw ? To : Te)[((int64_t)XT * Gm::KKP + CHK * ch + k) * 64];
    }
  };
  v4d acc[NACC];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = v4d{0, 0, 0, 0};
  };
  // acc[u] = tile (w, u) = sum_k T_par(u)(rows of tile w, k) data(k, columns of tile u); extra: acc[NL] = tile (XT, w)
  auto gemm = [&](auto first_step_c, int kk_n) {
    constexpr int S0 = decltype(first_step_c)::value;
    const double *La = L + kq * LD + j, *Lw = La + 16 * w;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int buf = (S0 + ch) & 1;
#pragma unroll
      for (int k = 0; k < CHK; ++k) {
        const int kk = CHK * ch + k;
        if (kk < Gm::KKP - 3 || kk < kk_n) {
          double d[NL];
#pragma unroll
          for (int t = 0; t < NL; ++t) d[t] = La[4 * kk * LD + 16 * t];
          double dw = 0; if constexpr (EXTRA) dw = Lw[4 * kk * LD];
#pragma unroll
          for (int t = 0; t < NL; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[buf][t < HT ? 0 : 1][k], d[t], acc[t], 0, 0, 0);
          if constexpr (EXTRA) acc[NL] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[buf][2][k], dw, acc[NL], 0, 0, 0);
        }
      }
      if (S0 + ch + 2 < 2 * NCH) load_chunk(buf, S0 + ch + 2);
    }
  };
  __builtin_amdgcn_s_setprio(3);
  load_chunk(0, 0); load_chunk(1, 1);
  zero_acc();
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();
  gemm(std::integral_constant<int, 0>{}, P.kk1);
  __syncthreads();
  // ---- divided by the eigenvalue sums -> LDS (same layout) ----
  {
    const double *bx = P.bxy + (int64_t)(4 * c + qd) * P.pl; const double czc = P.cz[c];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int tr = a < NL ? w : XT, tc = a < NL ? a : w, par = tc >= HT ? 1 : 0;
      const double *lamz = P.lam_z[c][par];
      const double bxy = bx[min(b * HC + 16 * (tc - par * HT) + j, P.pl - 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double den = fma(czc, lamz[16 * tr + 4 * q + kq], bxy);
        double r = __builtin_amdgcn_rcp(den);
        r = den < 1e300 ? fma(r, fma(-den, r, 1.0), r) : 0.0;
        L[(16 * tr + 4 * q + kq) * LD + 16 * tc + j] = acc[a][q] * r;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  zero_acc();
  __syncthreads();
  gemm(std::integral_constant<int, NCH>{}, P.kk2);
  // ---- v_k = a + b, v_k' = a - b -> the scattering all-to-all's buffer ----
  auto put = [&](int row, int col, double a, double bb) {
    if (row >= R || col >= C) return;
    const int mr = P.ng - 1 - row;
    const double vlo = mr == row ? a : a + bb, vhi = a - bb;
#pragma unroll
    for (int t = 0; t < 2; ++t) { const int64_t d0 = dst[2 * row + t]; if (d0 >= 0) out[d0 + cb + col] = vlo; }
    if (mr != row) {
#pragma unroll
      for (int t = 0; t < 2; ++t) { const int64_t d1 = dst[2 * mr + t]; if (d1 >= 0) out[d1 + cb + col] = vhi; }
    }
  };
#pragma unroll
  for (int u = 0; u < HT; ++u) {
#pragma unroll
    for (int q = 0; q < 4; ++q) put(16 * w + 4 * q + kq, 16 * u + j, acc[u][q], acc[u + HT][q]);
  }
  if constexpr (EXTRA) {
    // the shared tile row: waves 0 .. HT-1 hold the even-mode sums of column tile w, waves HT .. hold the odd-mode sums of column tile w - HT: swap through LDS
    __syncthreads();                                 // (everybody has finished reading the block)
    if (pw) {
#pragma unroll
      for (int q = 0; q < 4; ++q) L[((w - HT) * 4 + q) * 64 + lane] = acc[NL][q];
    }
    __syncthreads();
    if (!pw) {
#pragma unroll
      for (int q = 0; q < 4; ++q) put(16 * XT + 4 * q + kq, 16 * w + j, acc[NL][q], L[(w * 4 + q) * 64 + lane]);
    }
  }
}
template <int NT> void launch_zpass_both(hipStream_t s, const OctPass &P, const int64_t *dst, int n_items, const double *in, double *out, hipEvent_t e0, hipEvent_t e1) {
  hipExtLaunchKernelGGL((k_fdmo_zpass_both<NT>), dim3((unsigned)n_items), dim3(64 * (NT < 4 ? NT : 4)), 0, s, e0, e1, 0, P, dst, in, out);
}

// one instantiation: dynamic LDS (more than 64 KB from NT = 6 on: opted in once per device), events attached to the dispatch itself (the kernel's own duration, as rocprofv3 reports it)
template <int NT, int MODE, int VAR> void launch_one(hipStream_t s, int n_items, const OctPass &P, const double *in, double *out, hipEvent_t e0, hipEvent_t e1) {
  constexpr unsigned lds = (unsigned)(PassGeom<NT>::PADN * PassGeom<NT>::LDMAX * sizeof(double));
  auto kernel = k_fdmo_pass<NT, MODE, VAR>;
  if (lds > 64 * 1024) {
    static std::mutex mu; static std::set<int> done; int dev = 0; PORO_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count(dev)) { PORO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done.insert(dev); }
  }
  hipExtLaunchKernelGGL(kernel, dim3((unsigned)n_items), dim3(64 * pass_waves<NT>()), lds, s, e0, e1, 0, P, in, out);
}
template <int NT> void launch_pass(hipStream_t s, const OctPass &P, int n_items, const double *in, double *out, hipEvent_t e0, hipEvent_t e1) {
  const bool plain = P.vec2 == 1 && !P.use_in_off && !P.use_out_off && P.bxy_cmul == 4;
  const bool oct = plain && P.slab_z == 0 && P.slab_io == 0 && !P.row_in && P.no_shift == 3;
  const bool slab_u = plain && P.no_shift == 2 && ((P.mode == 0 && P.slab_io == 1 && !P.slab_z) || (P.mode == 1 && P.slab_z == 2 && P.row_in && !P.slab_io) || (P.mode == 2 && P.slab_io == 2 && !P.slab_z));
  if (oct) {
    if (P.mode == 0) launch_one<NT, 0, 0>(s, n_items, P, in, out, e0, e1); else if (P.mode == 1) launch_one<NT, 1, 0>(s, n_items, P, in, out, e0, e1); else launch_one<NT, 2, 0>(s, n_items, P, in, out, e0, e1);
  } else if (slab_u) {
    if (P.mode == 0) launch_one<NT, 0, 1>(s, n_items, P, in, out, e0, e1); else if (P.mode == 1) launch_one<NT, 1, 1>(s, n_items, P, in, out, e0, e1); else launch_one<NT, 2, 1>(s, n_items, P, in, out, e0, e1);
  } else {
    if (P.mode == 0) launch_one<NT, 0, 2>(s, n_items, P, in, out, e0, e1); else if (P.mode == 1) launch_one<NT, 1, 2>(s, n_items, P, in, out, e0, e1); else launch_one<NT, 2, 2>(s, n_items, P, in, out, e0, e1);
  }
}
void launch_pass_nt(hipStream_t s, int nt, const OctPass &P, int n_blocks, const double *in, double *out, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  switch (nt) {
    case 1: launch_pass<1>(s, P, n_blocks, in, out, e0, e1); break;
    case 2: launch_pass<2>(s, P, n_blocks, in, out, e0, e1); break;
    case 3: launch_pass<3>(s, P, n_blocks, in, out, e0, e1); break;
    case 4: launch_pass<4>(s, P, n_blocks, in, out, e0, e1); break;
    case 5: launch_pass<5>(s, P, n_blocks, in, out, e0, e1); break;
    case 6: launch_pass<6>(s, P, n_blocks, in, out, e0, e1); break;
    case 7: launch_pass<7>(s, P, n_blocks, in, out, e0, e1); break;
    case 8: launch_pass<8>(s, P, n_blocks, in, out, e0, e1); break;
    default: throw Error("fdmo: half lines of more than 128 entries");
  }
}
// columns of a pass-2 chunk: one 16-wide tile per wave (NT = 5: four waves)
inline int pass2_chunk(int nt) { return nt == 5 ? 64 : 16 * nt; }
inline int oct_grid(int64_t co, int nc = 3) { return (int)std::min<int64_t>((nc * co + kBlock - 1) / kBlock, kMaxPartials); }   // threads = positions x components, as many per thread as the partial slots demand
OctDims dims_of(const FdmOct &O) { return OctDims{O.n[0], O.n[1], O.n[2], O.h[0], O.h[1], O.h[2], O.hxp, O.co_stride, O.no, O.own_z, O.nc}; }
SlabGeo geo_of(const FdmOct &O) { const auto &S = O.slab; return SlabGeo{S.cw, S.nchunk, S.cps, S.nb * S.nchunk, S.rank, S.my_chunks, S.hzg, S.ng, S.np, S.scols, (int64_t)O.hxp * O.h[1], O.co_stride}; }

}  // namespace

bool fdmo_usable(int dim, const int nn[3]) {
  if (dim != 3) return false;
  for (int d = 0; d < 3; ++d) if ((nn[d] + 1) / 2 > 16 * kFdmoMaxTiles || nn[d] < 2) return false;
  return true;
}

void fdmo_init(FdmOct &O, const int nn[3], const double coef[3][3], hipStream_t s) {
  int hmax = 1;
  for (int d = 0; d < 3; ++d) { O.n[d] = nn[d]; O.h[d] = (nn[d] + 1) / 2; hmax = std::max(hmax, O.h[d]); for (int c = 0; c < 3; ++c) O.coef[c][d] = coef[c][d]; }
  O.nt = (hmax + 15) / 16;
  O.hxp = (O.h[0] + 1) & ~1;
  O.co_stride = (int64_t)O.hxp * O.h[1] * O.h[2]; O.n_oct = 24 * O.co_stride; O.no = 8; O.own_z = O.n[2];
  O.g.alloc(O.n_oct); O.z.alloc(O.n_oct); O.t.alloc(O.n_oct);
  O.g.zero(s); O.z.zero(s); O.t.zero(s);
}
// chunks, shares, plane tables and buffers of the slab form, after n / h / hxp / nt / co_stride are set: nb blocks per plane position set, np parity parts of a z line
static void slab_layout(FdmOct &O, int nb, int np, int rank, const std::vector<int> &node_layers, hipStream_t s) {
  auto &S = O.slab; const int N = (int)node_layers.size();
  S.nb = nb; S.np = np; S.hzg = np == 2 ? (S.ng + 1) / 2 : S.ng;
  const int64_t pl = (int64_t)O.hxp * O.h[1];
  // displacement system with 2, 4 or 5 tiles per half line: both parity parts of a z line in one workgroup (k_fdmo_zpass_both) - a chunk is then HALF a block wide
  S.zboth = np == 2 && (O.nt == 2 || O.nt == 4 || O.nt == 5) && !std::getenv("PORO_FDMO_SLAB_TWO_PARITY_WORKGROUPS");
  S.cw = S.zboth ? pass2_chunk(O.nt) / 2 : pass2_chunk(O.nt); S.nchunk = (int)((pl + S.cw - 1) / S.cw);
  const int chunk_total = nb * S.nchunk;
  S.cps = (chunk_total + N - 1) / N; S.chunk0 = rank * S.cps; S.my_chunks = std::max(0, std::min(S.cps, chunk_total - S.chunk0)); S.scols = (int64_t)S.cps * S.cw;
  std::vector<int> off(N), own(N), nl(N); int acc = 0; S.max_own = S.max_nl = S.rows_back = 0;
  for (int q = 0; q < N; ++q) { off[q] = acc; acc += node_layers[q]; own[q] = node_layers[q] + (q == N - 1 ? 1 : 0); nl[q] = node_layers[q] + 1; S.max_own = std::max(S.max_own, own[q]); S.max_nl = std::max(S.max_nl, nl[q]); S.rows_back += nl[q]; }
  S.own = own[rank]; S.nl = nl[rank];
  if ((int64_t)std::max(S.max_own, S.max_nl) * chunk_total * S.cw >= (int64_t)1 << 31 || (int64_t)S.rows_back * S.scols >= (int64_t)1 << 31 || (int64_t)chunk_total * S.cw >= (int64_t)1 << 24) throw Error("fdmo: a slab of more than 2^31 transform entries");
  // one buffer [send | recv]; what a rank keeps for itself is written straight into the receive half
  const int64_t blk = (int64_t)std::max(S.max_own, S.max_nl) * S.scols; S.recv_off = blk * N;
  std::vector<int64_t> rin(S.ng), rout(S.rows_back); std::vector<int32_t> rkz(S.rows_back);
  for (int q = 0; q < N; ++q) for (int k = 0; k < own[q]; ++k) rin[off[q] + k] = S.recv_off + ((int64_t)q * S.max_own + k) * S.scols;
  { int r = 0; for (int q = 0; q < N; ++q) for (int k = 0; k < nl[q]; ++k, ++r) { rout[r] = ((int64_t)q * S.max_nl + k) * S.scols + (q == rank ? S.recv_off : 0); rkz[r] = off[q] + k; } }
  S.row_in.upload(rin); S.row_out.upload(rout); S.row_kz.upload(rkz);
  { std::vector<int64_t> d2((size_t)2 * S.ng, -1);           // per global plane: where its rows go in the scattering all-to-all's buffer (one rank, or two for a shared plane)
    for (int r = 0; r < S.rows_back; ++r) { const int kz = rkz[r]; d2[2 * kz + (d2[2 * kz] >= 0 ? 1 : 0)] = rout[r]; }
    S.dst2.upload(d2); }
  S.buf.alloc((size_t)2 * blk * N); S.buf.zero(s);
  if (!S.zboth) { S.tz.alloc((size_t)np * S.cps * S.hzg * S.cw); S.tz.zero(s); }      // (the both-parity z pass goes from the gathered planes straight to the scattered ones)
}
void fdmo_init_slab(FdmOct &O, const int nn[3], const double coef[3][3], int rank, const std::vector<int> &node_layers, bool has_upper, hipStream_t s) {
  auto &S = O.slab; const int N = (int)node_layers.size();
  S.on = true; S.n_ranks = N; S.rank = rank;
  S.ng = 1; for (int q = 0; q < N; ++q) S.ng += node_layers[q];
  if (nn[2] != node_layers[rank] + 1) throw Error("fdmo_init_slab: local planes do not match the layer table");
  int hmax = (S.ng + 1) / 2;
  for (int d = 0; d < 3; ++d) { O.n[d] = nn[d]; O.h[d] = d < 2 ? (nn[d] + 1) / 2 : nn[d]; if (d < 2) hmax = std::max(hmax, O.h[d]); for (int c = 0; c < 3; ++c) O.coef[c][d] = coef[c][d]; }
  O.nt = (hmax + 15) / 16;
  O.hxp = (O.h[0] + 1) & ~1; O.no = 4; O.own_z = has_upper ? nn[2] - 1 : nn[2];
  O.co_stride = (int64_t)O.hxp * O.h[1] * O.h[2]; O.n_oct = 12 * O.co_stride;
  O.g.alloc(O.n_oct); O.z.alloc(O.n_oct); O.t.alloc(O.n_oct);
  O.g.zero(s); O.z.zero(s); O.t.zero(s);
  slab_layout(O, 12, 2, rank, node_layers, s);
}
// scalar Q1 system on a slab: nodal layout [local plane][y][x], whole-length lines everywhere (no parity split)
void fdmo_scalar_init_slab(FdmOct &O, const int nn[3], int rank, const std::vector<int> &node_layers, hipStream_t s) {
  auto &S = O.slab; const int N = (int)node_layers.size();
  S.on = true; S.n_ranks = N; S.rank = rank;
  S.ng = 1; for (int q = 0; q < N; ++q) S.ng += node_layers[q];
  if (nn[2] != node_layers[rank] + 1) throw Error("fdmo_scalar_init_slab: local planes do not match the layer table");
  int hmax = S.ng;
  for (int d = 0; d < 3; ++d) { O.n[d] = nn[d]; O.h[d] = nn[d]; if (d < 2) hmax = std::max(hmax, nn[d]); }
  O.nt = (hmax + 15) / 16; O.hxp = nn[0]; O.no = 1;
  O.co_stride = (int64_t)nn[0] * nn[1] * nn[2]; O.n_oct = O.co_stride;
  slab_layout(O, 1, 1, rank, node_layers, s);
}

bool fdmo_upload_dir(FdmOct &O, int comp, int dir, const std::vector<double> &S, const std::vector<double> &lam, int nn) {
  const int h = (nn + 1) / 2, nt = O.nt, kkp = 4 * nt, padn = 16 * nt;
  if (nn != (O.slab.on && dir == 2 ? O.slab.ng : O.n[dir])) throw Error("fdmo_upload_dir: line length mismatch");
  std::vector<int> grp[2];
  for (int m = 0; m < nn; ++m) {
    if (!(lam[m] < 1e300)) continue;              // removed modes
    double ds = 0, da = 0, nrm = 0;
    for (int k = 0; k < nn; ++k) { const double a = S[(size_t)k * nn + m], b = S[(size_t)(nn - 1 - k) * nn + m]; ds += (a - b) * (a - b); da += (a + b) * (a + b); nrm += a * a; }
    if (ds <= 1e-20 * nrm) grp[0].push_back(m); else if (da <= 1e-20 * nrm) grp[1].push_back(m); else return false;
  }
  for (int p = 0; p < 2; ++p) {
    if ((int)grp[p].size() > h) return false;
    // forward F[m][k] = S[k][mode m] (rows = modes of the parity group, columns = lower-half nodes), backward B[k][m] = S[k][mode m]; both as [tile][4 NT][64]:
    // lane l of fragment (tile, kk) holds element (16 tile + (l & 15), 4 kk + (l >> 4))
    std::vector<double> F((size_t)nt * kkp * 64, 0.0), B((size_t)nt * kkp * 64, 0.0), lp(padn + 16, std::numeric_limits<double>::infinity());
    const int ng = (int)grp[p].size();
    for (int t = 0; t < nt; ++t) for (int kk = 0; kk < kkp; ++kk) for (int l = 0; l < 64; ++l) {
      const int r = 16 * t + (l & 15), cc = 4 * kk + (l >> 4);
      const size_t at = ((size_t)t * kkp + kk) * 64 + l;
      if (r < ng && cc < h) F[at] = S[(size_t)cc * nn + grp[p][r]];
      if (r < h && cc < ng) B[at] = S[(size_t)r * nn + grp[p][cc]];
    }
    for (int m = 0; m < ng; ++m) lp[m] = lam[grp[p][m]];
    O.h_lam[comp][dir][p] = lp;
    O.fwd[comp][dir][p].upload(F); O.bwd[comp][dir][p].upload(B); O.lam[comp][dir][p].upload(lp);
  }
  return true;
}

void fdmo_finalize(FdmOct &O) {
  const int hx = O.h[0], hy = O.h[1], hxp = O.hxp; const size_t pl = (size_t)hxp * hy;
  std::vector<double> B(12 * pl, 1.0);             // (row pads: any finite non-zero value, their data are zeros)
  for (int c = 0; c < 3; ++c) for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px) for (int my = 0; my < hy; ++my) for (int mx = 0; mx < hx; ++mx)
    B[(size_t)(4 * c + 2 * py + px) * pl + (size_t)my * hxp + mx] = O.coef[c][0] * O.h_lam[c][0][px][mx] + O.coef[c][1] * O.h_lam[c][1][py][my];
  O.bxy.upload(B);
}

void fdmo_apply(hipStream_t s, const FdmOct &O, const double *g_oct, double *z_oct, double *scratch, const PcgScalars *gate, hipEvent_t *ev) {
  static int stamp_calls = 0; const char *stamp_path = std::getenv("PORO_FDMO_STAMPS");
  const bool stamping = stamp_path && ++stamp_calls == 3;          // diagnostic: the third application of the process writes its per-block time stamps
  DevBuf<unsigned long long> stamps; std::vector<std::pair<int, int64_t>> stamp_off;
  const int nt = O.nt, hx = O.h[0], hy = O.h[1], hz = O.h[2], hxp = O.hxp;
  auto tiles = [](int n) { return (n + 15) / 16; };
  auto ksteps = [](int n) { return (n + 3) / 4; };
  OctPass P{};
  P.co_stride = O.co_stride; P.hx = hxp; P.pl = hxp * hy;
  P.bxy = O.bxy.p; P.gate = gate; P.vec2 = 1; P.no_shift = 3; P.bxy_cmul = 4;
  for (int c = 0; c < 3; ++c) { P.cz[c] = O.coef[c][2]; for (int p = 0; p < 2; ++p) P.lam_z[c][p] = O.lam[c][2][p].p; }
  // pass 1: per z-plane, X[ky][kx] -> Fy (X Fx^T)
  P.mode = 0; P.R = hy; P.C = hxp; P.nt_r = tiles(hy); P.nt_c = tiles(hx); P.kk1 = ksteps(hx); P.kk2 = ksteps(hy); P.nblk = hz; P.blk_stride = (int64_t)hxp * hy; P.row_stride = hxp; P.bit1 = 0; P.bit2 = 1;
  for (int c = 0; c < 3; ++c) for (int p = 0; p < 2; ++p) { P.T1[c][p] = O.fwd[c][0][p].p; P.T2[c][p] = O.fwd[c][1][p].p; }
  if (stamping) { stamps.alloc((size_t)8 * 24 * (O.h[2] + (hxp * hy + pass2_chunk(nt) - 1) / pass2_chunk(nt) + O.h[2])); stamps.zero(s); }
  P.stamps = stamping ? stamps.p : nullptr; stamp_off.push_back({24 * P.nblk, 0});
  launch_pass_nt(s, nt, P, 24 * P.nblk, g_oct, scratch, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr);
  // pass 2: per chunk of 16 NT columns of a (component, octant) block, X[kz][col] -> Bz scale (Fz X), in place
  const int cw = pass2_chunk(nt);                 // chunk width of pass 2: one column tile per wave (NT = 5: 5 x 4 tiles for the four waves of the workgroup)
  P.mode = 1; P.R = hz; P.C = cw; P.nt_r = tiles(hz); P.nt_c = cw / 16; P.kk1 = ksteps(hz); P.kk2 = ksteps(hz); P.nblk = (hxp * hy + cw - 1) / cw; P.blk_stride = cw; P.row_stride = (int64_t)hxp * hy; P.bit1 = 2; P.bit2 = 2;
  for (int c = 0; c < 3; ++c) for (int p = 0; p < 2; ++p) { P.T1[c][p] = O.fwd[c][2][p].p; P.T2[c][p] = O.bwd[c][2][p].p; }
  if (stamping) P.stamps = stamps.p + 8 * (int64_t)(24 * hz); stamp_off.push_back({24 * P.nblk, 8 * (int64_t)(24 * hz)});
  launch_pass_nt(s, nt, P, 24 * P.nblk, scratch, scratch, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr);
  // pass 3: per z-plane, X[my][mx] -> (By X) Bx^T
  P.mode = 2; P.R = hy; P.C = hxp; P.nt_r = tiles(hy); P.nt_c = tiles(hx); P.kk1 = ksteps(hy); P.kk2 = ksteps(hx); P.nblk = hz; P.blk_stride = (int64_t)hxp * hy; P.row_stride = hxp; P.bit1 = 1; P.bit2 = 0;
  for (int c = 0; c < 3; ++c) for (int p = 0; p < 2; ++p) { P.T1[c][p] = O.bwd[c][1][p].p; P.T2[c][p] = O.bwd[c][0][p].p; }
  if (stamping) P.stamps = stamps.p + 8 * (int64_t)(24 * hz + stamp_off[1].first); stamp_off.push_back({24 * P.nblk, 8 * (int64_t)(24 * hz + stamp_off[1].first)});
  launch_pass_nt(s, nt, P, 24 * P.nblk, scratch, z_oct, ev ? ev[4] : nullptr, ev ? ev[5] : nullptr);
  if (stamping) {
    PORO_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(stamps.n); PORO_HIP(hipMemcpy(h.data(), stamps.p, stamps.n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE *f = std::fopen(stamp_path, "w")) {
      for (int pass = 0; pass < 3; ++pass) for (int b = 0; b < stamp_off[pass].first; ++b) {
        const unsigned long long *r = h.data() + stamp_off[pass].second + 8 * (int64_t)b;
        std::fprintf(f, "%d %d %llu %llu %llu %llu %llu %llu %llu %llu\n", pass, b, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
      }
      std::fclose(f);
    }
  }
}

// ---- slab form: the three sweeps as separate entry points (two all-to-alls sit between them, ctx_prec.hip) ----
void fdmo_slab_pass(hipStream_t s, const FdmOct &O, int pass, const double *in, double *out, const PcgScalars *gate, hipEvent_t e0, hipEvent_t e1) {
  const auto &S = O.slab; const int nt = O.nt, hx = O.h[0], hy = O.h[1], nzl = O.h[2], hxp = O.hxp;
  auto tiles = [](int n) { return (n + 15) / 16; };
  auto ksteps = [](int n) { return (n + 3) / 4; };
  OctPass P{};
  P.co_stride = O.co_stride; P.hx = hxp; P.pl = hxp * hy; P.bxy = O.bxy.p; P.gate = gate; P.vec2 = 1; P.no_shift = 2; P.bxy_cmul = 4;
  for (int c = 0; c < 3; ++c) { P.cz[c] = O.coef[c][2]; for (int p = 0; p < 2; ++p) P.lam_z[c][p] = O.lam[c][2][p].p; }
  if (pass == 2) {
    P.mode = 1; P.R = S.hzg; P.C = S.cw; P.nt_r = tiles(S.hzg); P.nt_c = S.cw / 16; P.kk1 = P.kk2 = ksteps(S.hzg); P.nblk = 1; P.blk_stride = (int64_t)S.hzg * S.cw; P.row_stride = S.cw; P.bit1 = P.bit2 = 2;
    P.slab_z = 2; P.chunk0 = S.chunk0; P.chunk_total = 12 * S.nchunk; P.nchunk = S.nchunk; P.row_in = S.row_in.p; P.ng = S.ng;
    for (int c = 0; c < 3; ++c) for (int p = 0; p < 2; ++p) { P.T1[c][p] = O.fwd[c][2][p].p; P.T2[c][p] = O.bwd[c][2][p].p; }
    if (S.zboth) {       // both parity parts per workgroup: `in` and `out` are the exchange buffer (gathered planes in, scattered planes out)
      if (S.my_chunks > 0) switch (nt) {
        case 2: launch_zpass_both<2>(s, P, S.dst2.p, S.my_chunks, in, out, e0, e1); break;
        case 4: launch_zpass_both<4>(s, P, S.dst2.p, S.my_chunks, in, out, e0, e1); break;
        case 5: launch_zpass_both<5>(s, P, S.dst2.p, S.my_chunks, in, out, e0, e1); break;
        default: throw Error("fdmo: both-parity z pass needs 2, 4 or 5 tiles per half line");
      }
      return;
    }
    if (S.my_chunks > 0) launch_pass_nt(s, nt, P, 2 * S.my_chunks, in, out, e0, e1);
    return;
  }
  const bool first = pass == 1;
  P.mode = first ? 0 : 2; P.R = hy; P.C = hxp; P.nt_r = tiles(hy); P.nt_c = tiles(hx); P.kk1 = ksteps(first ? hx : hy); P.kk2 = ksteps(first ? hy : hx); P.nblk = nzl; P.blk_stride = (int64_t)hxp * hy; P.row_stride = hxp;
  P.bit1 = first ? 0 : 1; P.bit2 = first ? 1 : 0;
  for (int c = 0; c < 3; ++c) for (int p = 0; p < 2; ++p) { P.T1[c][p] = first ? O.fwd[c][0][p].p : O.bwd[c][1][p].p; P.T2[c][p] = first ? O.fwd[c][1][p].p : O.bwd[c][0][p].p; }
  P.slab_io = first ? 1 : 2; P.store_planes = S.own; P.scols = (int)S.scols; P.inv_scols = 1.0f / (float)S.scols; P.col_unit = S.nchunk * S.cw; P.rank = S.rank;
  P.dest_stride = (int64_t)((first ? S.max_own : S.max_nl) - 1) * S.scols; P.recv_off = S.recv_off;
  launch_pass_nt(s, nt, P, 12 * P.nblk, in, out, e0, e1);
}
static int copy_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }
void fdmo_slab_scatter_pack(hipStream_t s, const FdmOct &O, const PcgScalars *gate) {
  const auto &S = O.slab; const SlabGeo G = geo_of(O);
  if (S.my_chunks > 0) hipLaunchKernelGGL(k_fdmo_slab_scatter_pack, copy_grid((int64_t)S.rows_back * S.my_chunks * S.cw), 256, 0, s, G, S.rows_back, S.row_out.p, S.row_kz.p, S.tz.p, S.buf.p, gate);
}
void fdmo_dot_owned(hipStream_t s, const FdmOct &O, const double *a, const double *b, double *partials, const PcgScalars *gate) {
  hipLaunchKernelGGL(k_fdmo_dot_owned, std::max(oct_grid(O.co_stride, O.nc), O.nc * O.no), kBlock, 0, s, dims_of(O), a, b, partials, gate);   // (at least one workgroup per block of the arrays)
}

// ---- the same three sweeps for a SCALAR Q1 system of the box (pressure Jacobian a M + kappa K, projection mass matrix): one "component", no parity octants, the
// vectors keep their nodal layout [z][y][x] (full-length lines, odd pitch: 8-byte block loads).  Replaces six single-direction launches of kernels_fdm.hip.
bool fdmo_scalar_usable(int dim, const int nn[3]) { if (dim != 3) return false; for (int d = 0; d < 3; ++d) if (nn[d] > 16 * kFdmoMaxTiles || nn[d] < 2) return false; return true; }
void fdmo_scalar_init(FdmOct &O, const int nn[3], hipStream_t s) {
  int hmax = 1;
  for (int d = 0; d < 3; ++d) { O.n[d] = nn[d]; O.h[d] = nn[d]; hmax = std::max(hmax, nn[d]); }
  O.nt = (hmax + 15) / 16; O.hxp = nn[0];
  O.co_stride = (int64_t)nn[0] * nn[1] * nn[2]; O.n_oct = O.co_stride;
  O.t.alloc(3 * O.n_oct); O.t.zero(s);      // (scratch for up to three right-hand sides at once)
}
void fdmo_scalar_upload_dir(FdmOct &O, int dir, const std::vector<double> &S, const std::vector<double> &lam, int n) {   // S: n x n row-major, columns = M-orthonormal eigenvectors
  if (n != (O.slab.on && dir == 2 ? O.slab.ng : O.n[dir])) throw Error("fdmo_scalar_upload_dir: line length mismatch");
  const int nt = O.nt, kkp = 4 * nt, padn = 16 * nt;
  std::vector<double> F((size_t)nt * kkp * 64, 0.0), B((size_t)nt * kkp * 64, 0.0), lp(padn + 16, std::numeric_limits<double>::infinity());
  for (int t = 0; t < nt; ++t) for (int kk = 0; kk < kkp; ++kk) for (int l = 0; l < 64; ++l) {
    const int r = 16 * t + (l & 15), cc = 4 * kk + (l >> 4);
    if (r < n && cc < n) { F[((size_t)t * kkp + kk) * 64 + l] = S[(size_t)cc * n + r]; B[((size_t)t * kkp + kk) * 64 + l] = S[(size_t)r * n + cc]; }
  }
  for (int m = 0; m < n; ++m) lp[m] = lam[m];
  O.h_lam[0][dir][0] = lp; O.fwd[0][dir][0].upload(F); O.bwd[0][dir][0].upload(B); O.lam[0][dir][0].upload(lp);
}
static const double *scalar_table(hipStream_t s, FdmOct &O, double a, double kappa);
// z = (a M + kappa K)^-1 g; the x / y share a + kappa (lam_x + lam_y) of the eigenvalue sums is tabulated per plane position, one table per (a, kappa)
void fdmo_scalar_apply(hipStream_t s, FdmOct &O, double a, double kappa, const double *g, double *z, const PcgScalars *gate) {
  const double *gs[1] = {g}; double *zs[1] = {z};
  fdmo_scalar_apply_many(s, O, a, kappa, 1, gs, zs, gate);
}
// the same for up to three right-hand sides in separate vectors: one set of three launches with 3 x the workgroups (the Q1 systems of config 4 fill less than a third of the chip)
void fdmo_scalar_apply_many(hipStream_t s, FdmOct &O, double a, double kappa, int nb, const double *const *g, double *const *z, const PcgScalars *gate) {
  if (nb < 1 || nb > 3) throw Error("fdmo_scalar_apply_many: 1..3 right-hand sides");
  const int nt = O.nt, hx = O.h[0], hy = O.h[1], hz = O.h[2];
  const double *table = scalar_table(s, O, a, kappa);
  auto tiles = [](int n) { return (n + 15) / 16; };
  auto ksteps = [](int n) { return (n + 3) / 4; };
  OctPass P{};
  P.co_stride = O.co_stride; P.hx = hx; P.pl = hx * hy; P.bxy = table; P.gate = gate; P.vec2 = 0; P.no_shift = 0; P.bxy_cmul = 0;
  for (int c = 0; c < nb; ++c) { P.cz[c] = kappa; P.lam_z[c][0] = O.lam[0][2][0].p; P.in_blk[c] = g[c]; P.out_blk[c] = z[c]; }
  P.mode = 0; P.R = hy; P.C = hx; P.nt_r = tiles(hy); P.nt_c = tiles(hx); P.kk1 = ksteps(hx); P.kk2 = ksteps(hy); P.nblk = hz; P.blk_stride = (int64_t)hx * hy; P.row_stride = hx; P.bit1 = 0; P.bit2 = 1;
  for (int c = 0; c < nb; ++c) { P.T1[c][0] = O.fwd[0][0][0].p; P.T2[c][0] = O.fwd[0][1][0].p; }
  P.use_in_off = 1; P.use_out_off = 0;
  launch_pass_nt(s, nt, P, nb * P.nblk, g[0], O.t.p);
  const int cw = pass2_chunk(nt);
  P.mode = 1; P.R = hz; P.C = cw; P.nt_r = tiles(hz); P.nt_c = cw / 16; P.kk1 = ksteps(hz); P.kk2 = ksteps(hz); P.nblk = (hx * hy + cw - 1) / cw; P.blk_stride = cw; P.row_stride = (int64_t)hx * hy; P.bit1 = 2; P.bit2 = 2;
  for (int c = 0; c < nb; ++c) { P.T1[c][0] = O.fwd[0][2][0].p; P.T2[c][0] = O.bwd[0][2][0].p; }
  P.use_in_off = 0;
  launch_pass_nt(s, nt, P, nb * P.nblk, O.t.p, O.t.p);
  P.mode = 2; P.R = hy; P.C = hx; P.nt_r = tiles(hy); P.nt_c = tiles(hx); P.kk1 = ksteps(hy); P.kk2 = ksteps(hx); P.nblk = hz; P.blk_stride = (int64_t)hx * hy; P.row_stride = hx; P.bit1 = 1; P.bit2 = 0;
  for (int c = 0; c < nb; ++c) { P.T1[c][0] = O.bwd[0][1][0].p; P.T2[c][0] = O.bwd[0][0][0].p; }
  P.use_out_off = 1;
  launch_pass_nt(s, nt, P, nb * P.nblk, O.t.p, z[0]);
}

static const double *scalar_table(hipStream_t s, FdmOct &O, double a, double kappa) {
  for (auto &T : O.scalar_tables) if (T.a == a && T.kappa == kappa) return T.t.p;
  const int hx = O.h[0], hy = O.h[1];
  if (O.scalar_tables.size() >= 8) { PORO_HIP(hipStreamSynchronize(s)); O.scalar_tables.pop_front(); }   // (a changing coefficient, e.g. a varying time step: drop the oldest)
  std::vector<double> Bt((size_t)hx * hy);
  for (int my = 0; my < hy; ++my) for (int mx = 0; mx < hx; ++mx) Bt[(size_t)my * hx + mx] = a + kappa * (O.h_lam[0][0][0][mx] + O.h_lam[0][1][0][my]);
  O.scalar_tables.emplace_back(); auto &T = O.scalar_tables.back(); T.a = a; T.kappa = kappa; T.t.upload(Bt); return T.t.p;
}
void fdmo_scalar_slab_pass(hipStream_t s, FdmOct &O, int pass, double a, double kappa, const double *in, double *out) {
  const auto &S = O.slab; const int nt = O.nt, hx = O.h[0], hy = O.h[1], nzl = O.h[2];
  auto tiles = [](int n) { return (n + 15) / 16; };
  auto ksteps = [](int n) { return (n + 3) / 4; };
  OctPass P{};
  P.co_stride = O.co_stride; P.hx = hx; P.pl = hx * hy; P.bxy = scalar_table(s, O, a, kappa); P.vec2 = 0; P.no_shift = 0; P.bxy_cmul = 0;
  P.cz[0] = kappa; P.lam_z[0][0] = O.lam[0][2][0].p;
  if (pass == 2) {
    P.mode = 1; P.R = S.ng; P.C = S.cw; P.nt_r = tiles(S.ng); P.nt_c = S.cw / 16; P.kk1 = P.kk2 = ksteps(S.ng); P.nblk = 1; P.blk_stride = (int64_t)S.hzg * S.cw; P.row_stride = S.cw; P.bit1 = P.bit2 = 2;
    P.slab_z = 1; P.chunk0 = S.chunk0; P.chunk_total = S.nchunk; P.nchunk = S.nchunk; P.row_in = S.row_in.p; P.ng = S.ng; P.vec2 = 1;
    P.T1[0][0] = O.fwd[0][2][0].p; P.T2[0][0] = O.bwd[0][2][0].p;
    if (S.my_chunks > 0) launch_pass_nt(s, nt, P, S.my_chunks, in, out);
    return;
  }
  const bool first = pass == 1;
  P.mode = first ? 0 : 2; P.R = hy; P.C = hx; P.nt_r = tiles(hy); P.nt_c = tiles(hx); P.kk1 = ksteps(first ? hx : hy); P.kk2 = ksteps(first ? hy : hx); P.nblk = nzl; P.blk_stride = (int64_t)hx * hy; P.row_stride = hx;
  P.bit1 = first ? 0 : 1; P.bit2 = first ? 1 : 0;
  P.T1[0][0] = first ? O.fwd[0][0][0].p : O.bwd[0][1][0].p; P.T2[0][0] = first ? O.fwd[0][1][0].p : O.bwd[0][0][0].p;
  P.slab_io = first ? 1 : 2; P.store_planes = S.own; P.scols = (int)S.scols; P.inv_scols = 1.0f / (float)S.scols; P.col_unit = S.nchunk * S.cw; P.rank = S.rank;
  P.dest_stride = (int64_t)((first ? S.max_own : S.max_nl) - 1) * S.scols; P.recv_off = S.recv_off;
  launch_pass_nt(s, nt, P, P.nblk, in, out);
}

#define PORO_OCT_LAUNCH(kernel, O, ...) do { if ((O).nc == 2) hipLaunchKernelGGL(kernel<2>, oct_grid((O).co_stride, 2), kBlock, 0, s, dims_of(O), __VA_ARGS__); \
                                              else hipLaunchKernelGGL(kernel<3>, oct_grid((O).co_stride, 3), kBlock, 0, s, dims_of(O), __VA_ARGS__); } while (0)
// ---- planar form, host side: quadrant layout Q[c][2 py + px][ky][kx] (one plane), transforms as row-major h x h matrices F[mode][node] per (component, direction, parity) ----
bool fdmo_planar_usable(int dim, const int nn[3]) { return dim == 2 && nn[0] >= 2 && nn[1] >= 2 && nn[0] <= 4096 && nn[1] <= 4096; }
void fdmo_init_planar(FdmOct &O, const int nn[3], const double coef[3][3], hipStream_t s, bool split) {
  O.nc = 2; O.no = split ? 4 : 1; O.planar = true;
  for (int d = 0; d < 3; ++d) { O.n[d] = d < 2 ? nn[d] : 1; O.h[d] = d < 2 ? (split ? (nn[d] + 1) / 2 : nn[d]) : 1; for (int c = 0; c < 3; ++c) O.coef[c][d] = coef[c][d]; }
  O.nt = 0; O.hxp = (O.h[0] + 1) & ~1; O.own_z = 1;
  O.co_stride = (int64_t)O.hxp * O.h[1]; O.n_oct = 2 * O.no * O.co_stride;
  O.g.alloc(O.n_oct); O.z.alloc(O.n_oct); O.t.alloc(2 * O.n_oct);
  O.g.zero(s); O.z.zero(s); O.t.zero(s);
}
bool fdmo_upload_dir_planar(FdmOct &O, int comp, int dir, const std::vector<double> &S, const std::vector<double> &lam, int nn) {
  if (nn != O.n[dir]) throw Error("fdmo_upload_dir_planar: line length mismatch");
  if (O.no == 1) {          // no parity split (different conditions at the two ends of a line): the full transform F[mode][node], modes that do not exist carry lam = inf
    std::vector<double> F((size_t)nn * nn, 0.0), lp((size_t)nn + 16, std::numeric_limits<double>::infinity());
    for (int m = 0; m < nn; ++m) { if (!(lam[m] < 1e300)) continue; for (int k = 0; k < nn; ++k) F[(size_t)m * nn + k] = S[(size_t)k * nn + m]; lp[m] = lam[m]; }
    O.h_lam[comp][dir][0] = lp; O.fwd[comp][dir][0].upload(F); O.lam[comp][dir][0].upload(lp);
    return true;
  }
  const int h = (nn + 1) / 2;
  std::vector<int> grp[2];
  for (int m = 0; m < nn; ++m) {
    if (!(lam[m] < 1e300)) continue;
    double ds = 0, da = 0, nrm = 0;
    for (int k = 0; k < nn; ++k) { const double a = S[(size_t)k * nn + m], b = S[(size_t)(nn - 1 - k) * nn + m]; ds += (a - b) * (a - b); da += (a + b) * (a + b); nrm += a * a; }
    if (ds <= 1e-20 * nrm) grp[0].push_back(m); else if (da <= 1e-20 * nrm) grp[1].push_back(m); else return false;
  }
  for (int p = 0; p < 2; ++p) {
    if ((int)grp[p].size() > h) return false;
    std::vector<double> F((size_t)h * h, 0.0), lp((size_t)h + 16, std::numeric_limits<double>::infinity());
    const int ng = (int)grp[p].size();
    for (int m = 0; m < ng; ++m) { for (int k = 0; k < h; ++k) F[(size_t)m * h + k] = S[(size_t)k * nn + grp[p][m]]; lp[m] = lam[grp[p][m]]; }
    O.h_lam[comp][dir][p] = lp; O.fwd[comp][dir][p].upload(F); O.lam[comp][dir][p].upload(lp);
  }
  return true;
}
// z = blockdiag(A_cc)^-1 g in quadrant form: T = X Fx^T, U = (Fy T) / (cx lam_x + cy lam_y), V = Fy^T U, Z = V Fx  - four batched GEMMs over the 8 (component, quadrant) planes
void fdmo_apply_planar(hipStream_t s, const FdmOct &O, const double *g, double *z, const PcgScalars *gate) {
  const int hx = O.h[0], hy = O.h[1], hxp = O.hxp; const int64_t co = O.co_stride;
  double *t1 = O.t.p, *t2 = O.t.p + O.n_oct;
  const int no = O.no, nb = 2 * no;                 // blocks: (component, quadrant) - or the two components alone without the parity split
  Gemm2D G{}; G.nb = nb; G.gate = gate; G.rsC = hxp;
  auto launch = [&](bool ak, bool bk) {
    const dim3 grid((unsigned)(((G.M + 63) / 64) * ((G.N + 63) / 64)), (unsigned)nb);
    if (ak && bk) hipLaunchKernelGGL((k_fdmo_gemm2d<true, true>), grid, 256, 0, s, G);
    else if (ak) hipLaunchKernelGGL((k_fdmo_gemm2d<true, false>), grid, 256, 0, s, G);
    else hipLaunchKernelGGL((k_fdmo_gemm2d<false, false>), grid, 256, 0, s, G);
  };
  // 1: T[ky][mx] = sum_kx X[ky][kx] Fx[mx][kx]
  G.M = hy; G.N = hx; G.K = hx; G.rsA = hxp; G.csA = 1; G.rsB = 1; G.csB = hx; G.scale = 0;
  for (int b = 0; b < nb; ++b) { const int c = b / no, q = b % no; G.A[b] = g + (int64_t)b * co; G.B[b] = O.fwd[c][0][q & 1].p; G.C[b] = t1 + (int64_t)b * co; }
  launch(true, true);
  // 2: U[my][mx] = sum_ky Fy[my][ky] T[ky][mx], divided by the eigenvalue sums
  G.M = hy; G.N = hx; G.K = hy; G.rsA = hy; G.csA = 1; G.rsB = hxp; G.csB = 1; G.scale = 1;
  for (int b = 0; b < nb; ++b) { const int c = b / no, q = b % no; G.A[b] = O.fwd[c][1][q >> 1].p; G.B[b] = t1 + (int64_t)b * co; G.C[b] = t2 + (int64_t)b * co;
                                G.lamM[b] = O.lam[c][1][q >> 1].p; G.lamN[b] = O.lam[c][0][q & 1].p; G.cM[b] = O.coef[c][1]; G.cN[b] = O.coef[c][0]; }
  launch(true, false);
  // 3: V[ky][mx] = sum_my Fy[my][ky] U[my][mx]
  G.rsA = 1; G.csA = hy; G.scale = 0;
  for (int b = 0; b < nb; ++b) { G.B[b] = t2 + (int64_t)b * co; G.C[b] = t1 + (int64_t)b * co; }
  launch(false, false);
  // 4: Z[ky][kx] = sum_mx V[ky][mx] Fx[mx][kx]
  G.M = hy; G.N = hx; G.K = hx; G.rsA = hxp; G.csA = 1; G.rsB = hx; G.csB = 1;
  for (int b = 0; b < nb; ++b) { const int c = b / no, q = b % no; G.A[b] = t1 + (int64_t)b * co; G.B[b] = O.fwd[c][0][q & 1].p; G.C[b] = z + (int64_t)b * co; }
  launch(true, false);
}

void fdmo_from_nodal(hipStream_t s, const FdmOct &O, const double *v, double *q) { PORO_OCT_LAUNCH(k_fdmo_from_nodal, O, v, (const uint8_t *)nullptr, q); }
void fdmo_to_nodal(hipStream_t s, const FdmOct &O, const double *r, double *v) { PORO_OCT_LAUNCH(k_fdmo_to_nodal, O, r, v); }
void fdmo_init_residual(hipStream_t s, const FdmOct &O, double *g, const double *Ax, const double *b, const uint8_t *inert) { PORO_OCT_LAUNCH(k_fdmo_init_residual, O, g, Ax, b, inert); }
void fdmo_first_direction(hipStream_t s, const FdmOct &O, double *d, const double *g, const double *z, double *partials) { PORO_OCT_LAUNCH(k_fdmo_first_direction, O, d, g, z, partials); }
void fdmo_update_g(hipStream_t s, const FdmOct &O, PcgScalars *sc, int parity, double *g, const double *h, const uint8_t *inert, const double *partials_dh, double *partials_out, const double *red) {
  static const bool single = std::getenv("PORO_FDMO_UPDATE_G_SINGLE") != nullptr;
  if (single) PORO_OCT_LAUNCH(k_fdmo_update_g, O, sc, parity, g, h, inert, partials_dh, partials_out, red);
  else PORO_OCT_LAUNCH(k_fdmo_update_g2, O, sc, parity, g, h, inert, partials_dh, partials_out, red);
}
void fdmo_update_d(hipStream_t s, const FdmOct &O, PcgScalars *sc, int parity, int it, double *x, double *d, const double *z, const double *partials_in, const double *red) {
  PORO_OCT_LAUNCH(k_fdmo_update_d, O, sc, parity, it, x, d, z, (int64_t)O.nc * O.n[0] * O.n[1] * O.n[2], partials_in, red);
}

}  // namespace poro
