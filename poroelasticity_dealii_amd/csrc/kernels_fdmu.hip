// Block fast-diagonalisation preconditioner of the DISPLACEMENT system (K-prec, SURVEY 8f-1 "a stronger preconditioner") on uniform
// boxes (gfx950, wave64, fp64 / fp32 MFMA).
//
// On a box the diagonal blocks of A_u (PoroElasticDisplacementSolver.h:230-242, one block per displacement component c) are
//   A_cc = sum_d coef(c,d) (K_d in direction d) (x) (M_e in the other directions),   coef = lambda + 2G (d == c) | G (d != c),
// with the 1D FE_Q(k_u) mass / stiffness matrices M_d, K_d; a Dirichlet condition on component c over a whole face removes the end
// node of that direction from the 1D matrices of component c, so the structure survives the constraints.  With the generalised 1D
// eigen-decompositions K S = M S Lambda, S^T M S = I (host, once per mesh; one set per component because the end conditions differ)
//   A_cc^-1 = (Sx (x) Sy (x) Sz) diag(sum_d coef(c,d) lam_d)^-1 (Sx (x) Sy (x) Sz)^T   EXACTLY.
// z = blockdiag(A_cc)^-1 g is spectrally equivalent to A_u^-1 (Korn): CG needs ~20 iterations independent of h instead of O(1/h)
// with Jacobi (245 at BASELINE config 4).  The off-diagonal blocks A_ab (products of first-derivative matrices) are not separable and
// stay in the Krylov iteration.
//
// One application = 2 dim dense (n_d x n_d) transforms along the grid lines of a [component][z][y][x] array: GEMM-shaped work
// (2 x 3 x 6 x 145^4 = 1.6e10 flop at config 4), so it runs on the matrix cores.  One kernel serves every pass:
//   load a panel of 32 grid lines (all points of each line) into LDS -> D = T1 * panel on v_mfma_f64_16x16x4_f64 (T as the A operand,
//   streamed from L2 in MFMA fragment order, 512 contiguous bytes per wave and k-step; panel columns as the B operand from LDS)
//   -> [last direction only: scale by the inverse eigenvalue sums, write back to LDS, second GEMM with T2 = S] -> LDS -> coalesced store.
// The first pass reads the node-interleaved residual (dof = node * dim + c), the last pass writes the interleaved z; in between the
// data are component-planar.  Passes: x, y, z (forward + scaling + backward fused), y, x  = 5 sweeps over the vector in 3D.
// Rows / columns of constrained end nodes are zero in S, so z = 0 on Dirichlet dofs (they are inert inside PCG anyway).
#include "common.hpp"
#include <cmath>
#include <limits>
#include <type_traits>

namespace poro {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <class T> struct Mfma;
template <> struct Mfma<double> { typedef v4d acc_t; static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); } };
template <> struct Mfma<float> { typedef v4f acc_t; static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); } };

constexpr int kNT = 32;        // grid lines per panel (two 16-column MFMA tiles)
constexpr int kLdK = kNT + 16; // LDS row stride of a [k][line] panel: the four k-rows of a B fragment land in disjoint banks
constexpr int kThreads = 256;  // 4 waves: wave w owns column tile w & 1 and the output row tiles of half w >> 1

struct FdmuPass {
  int nK;               // points per line (= output rows)
  int MT, KK;           // 16-row output tiles, 4-deep k-steps (both ceil)
  int64_t SI;           // element stride along the line: 1 (x), nx (y), nx ny (z)
  int64_t n_lines;      // lines per component
  int64_t comp_stride;  // planar arrays: elements per component
  int ncomp;
  int in_interleaved, out_interleaved;   // array layout: element (line point e, component c) at e * ncomp + c instead of c * comp_stride + e
  int x_layout;         // 1: LDS panel stored [line][k] (contiguous lines: the x direction), 0: [k][line]
  int ld_line;          // x_layout: LDS stride between lines
  // fused scaling (last direction): D(m, line) /= kd * lam_d[m] + k0 * lam0[i] + k1 * lam1[j], (i, j) = grid position of the line
  int fused; int n0; int64_t col0, col_total; double kd[3], k0[3], k1[3];
  int reg_form;         // 1: register-panel kernel (T1 / T2 point at the chunked fragment order of k_fdmu_reg)
  int split;            // 1: even / odd form (k_fdmu_split): every component's 1D eigenvectors are symmetric or antisymmetric about the line's centre
  int split_dir;        // split, not fused: 0 forward (nodes -> modes), 1 backward
  int n_even[3];        // split: even modes per component (mode order along the line: even modes first, then the odd ones)
  int blk;              // split form for lines of more than 160 points (k_fdmu_blk): the output rows are produced in blocks of 80 per parity, the transform matrix is
  int blk_kk[3], blk_nch[3], blk_mb[3];   //   packed [row block][chunk]...; per component: k-steps, chunks per row block and row blocks of THIS pass (forward / backward differ)
  int scale_on_load;    // blk backward pass of the last direction: the coefficients are divided by the eigenvalue sums as they are loaded
  const void *T1[3], *T2[3];             // per component: transform matrices in MFMA fragment order [MT][KK][64]
  const double *lam_d[3], *lam0[3], *lam1[3];
};

// element index of point k of line n inside one component's grid
__device__ __forceinline__ int64_t line_base(const FdmuPass &P, int64_t n) { const int64_t o = n / P.SI; return o * P.SI * P.nK + (n - o * P.SI); }

template <class TC, class TIn, class TOut, int MTH>
__global__ void __launch_bounds__(kThreads)
k_fdmu_pass(FdmuPass P, const TIn *__restrict__ in, TOut *__restrict__ out) {
  extern __shared__ double lds_raw[];
  TC *L = reinterpret_cast<TC *>(lds_raw);
  typedef typename Mfma<TC>::acc_t acc_t;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.y;
  const int64_t n0 = (int64_t)blockIdx.x * kNT;
  const int rows = P.KK * 4 > P.MT * 16 ? P.KK * 4 : P.MT * 16;      // panel rows: k (input) or m (output), zero padded
  const int s_n = P.x_layout ? P.ld_line : 1, s_k = P.x_layout ? 1 : kLdK;

  // ---- load the panel: element (k, line) -> L[line * s_n + k * s_k]; padded rows and lines are zero ----
  if (P.x_layout) {
    // lines are contiguous in memory (planar) or ncomp-strided (interleaved): run along k
    const int total = kNT * rows;
    for (int idx = tid; idx < total; idx += kThreads) {
      const int ln = idx / rows, k = idx - ln * rows;
      const int64_t n = n0 + ln;
      TC v = 0;
      if (k < P.nK && n < P.n_lines) {
        const int64_t e = n * P.nK + k;
        v = (TC)(P.in_interleaved ? in[e * P.ncomp + c] : in[(int64_t)c * P.comp_stride + e]);
      }
      L[ln * s_n + k] = v;
    }
  } else {
    const int ln = tid & (kNT - 1);
    const int64_t n = n0 + ln;
    const bool lv = n < P.n_lines;
    const int64_t base = (int64_t)c * P.comp_stride + (lv ? line_base(P, n) : 0);
    for (int k = tid / kNT; k < rows; k += kThreads / kNT) L[k * kLdK + ln] = (lv && k < P.nK) ? (TC)in[base + (int64_t)k * P.SI] : (TC)0;
  }
  __syncthreads();

  const int nt = w & 1, mh = w >> 1;
  const int i16 = lane & 15, kq = lane >> 4;
  const TC *Bp = L + (nt * 16 + i16) * s_n + kq * s_k;       // B fragment of k-step kk: Bp[4 kk s_k]
  acc_t acc[MTH];

  auto gemm = [&](const TC *__restrict__ Tf) {
#pragma unroll
    for (int t = 0; t < MTH; ++t) acc[t] = acc_t{0, 0, 0, 0};
    const TC *Ap = Tf + ((int64_t)(mh * MTH) * P.KK) * 64 + lane;   // fragment (m-tile, kk) at ((mt * KK) + kk) * 64 + lane
    int kk = 0;
    for (; kk + 4 <= P.KK; kk += 4) {          // 4 k-steps per trip, every operand load issued before the first MFMA
      TC a[4][MTH], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        b[u] = Bp[(kk + u) * 4 * s_k];
#pragma unroll
        for (int t = 0; t < MTH; ++t) a[u][t] = (mh * MTH + t < P.MT) ? Ap[((int64_t)t * P.KK + kk + u) * 64] : (TC)0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < MTH; ++t) acc[t] = Mfma<TC>::run(a[u][t], b[u], acc[t]);
    }
    for (; kk < P.KK; ++kk) {
      const TC b = Bp[kk * 4 * s_k];
#pragma unroll
      for (int t = 0; t < MTH; ++t) { const TC a = (mh * MTH + t < P.MT) ? Ap[((int64_t)t * P.KK + kk) * 64] : (TC)0; acc[t] = Mfma<TC>::run(a, b, acc[t]); }
    }
  };
  // D fragment -> LDS: register q of a lane holds row (lane >> 4) + 4 q of the tile, column lane & 15
  auto acc_to_lds = [&](bool scale) {
    double base = 0; bool lvalid = true;
    if (scale) {
      const int64_t col = P.col0 + n0 + nt * 16 + i16;           // global line index = (j, i) grid position
      lvalid = n0 + nt * 16 + i16 < P.n_lines && col < P.col_total;
      if (lvalid) {
        if (P.lam1[c]) { const int64_t j = col / P.n0; base = P.k0[c] * P.lam0[c][col - j * P.n0] + P.k1[c] * P.lam1[c][j]; }
        else if (P.lam0[c]) base = P.k0[c] * P.lam0[c][col];
      }
    }
#pragma unroll
    for (int t = 0; t < MTH; ++t) {
      const int mt = mh * MTH + t;
      if (mt >= P.MT) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = mt * 16 + kq + 4 * q;
        double v = (double)acc[t][q];
        if (scale) v = (lvalid && m < P.nK) ? v / (base + P.kd[c] * P.lam_d[c][m]) : 0.0;   // padded / constrained modes carry lam = inf
        L[(nt * 16 + i16) * s_n + m * s_k] = (TC)v;
      }
    }
  };

  gemm(reinterpret_cast<const TC *>(P.T1[c]));
  __syncthreads();                       // everybody has finished reading the input panel
  acc_to_lds(P.fused != 0);
  if (P.fused) {
    // rows >= 16 MT of the panel still hold input data; the second GEMM's k runs to 4 KK <= rows: clear what the first result did not cover
    for (int idx = tid; idx < (rows - P.MT * 16) * kNT; idx += kThreads) { const int k = P.MT * 16 + idx / kNT, ln = idx % kNT; L[ln * s_n + k * s_k] = 0; }
    __syncthreads();
    gemm(reinterpret_cast<const TC *>(P.T2[c]));
    __syncthreads();
    acc_to_lds(false);
  }
  __syncthreads();

  // ---- store the panel (rows m < nK) ----
  if (P.x_layout) {
    const int total = kNT * P.nK;
    for (int idx = tid; idx < total; idx += kThreads) {
      const int ln = idx / P.nK, m = idx - ln * P.nK;
      const int64_t n = n0 + ln;
      if (n >= P.n_lines) continue;
      const int64_t e = n * P.nK + m;
      const TC v = L[ln * s_n + m];
      if (P.out_interleaved) out[e * P.ncomp + c] = (TOut)v; else out[(int64_t)c * P.comp_stride + e] = (TOut)v;
    }
  } else {
    const int ln = tid & (kNT - 1);
    const int64_t n = n0 + ln;
    if (n < P.n_lines) {
      const int64_t base = (int64_t)c * P.comp_stride + line_base(P, n);
      for (int m = tid / kNT; m < P.nK; m += kThreads / kNT) out[base + (int64_t)m * P.SI] = (TOut)L[m * kLdK + ln];
    }
  }
}

// ---- register-panel form (lines of <= 160 points, fp64): no LDS traffic for the data at all ------------------------------------------
// A wave owns 16 grid lines.  Their points go straight from global memory into MFMA B-fragment registers (lane (j, kq) holds points
// k = 4 kk + kq of line j for every k-step kk: <= 40 doubles), the transform matrix streams through LDS in chunks of 4 k-steps shared by
// the 4 waves of the workgroup (double buffered, one barrier per chunk, 16-byte LDS reads feeding two MFMAs each), and the D fragments are
// stored from the accumulators.  The D layout of v_mfma_f64_16x16x4_f64 (register q of a lane = row kq + 4 q of the tile, column j) IS the
// B layout of the next GEMM (k-step 4 mt + q of the same lane), so the fused last-direction pass chains its two GEMMs without moving data.
template <int NCH> struct RegGeom { static constexpr int MTP = (NCH + 1) / 2, MT = 2 * MTP, KKP = 4 * NCH, CHUNK = 4 * MTP * 128; };

// MODE 0: data as the B operand (y / z passes: 16 consecutive lines per tile are contiguous in memory, loads and stores run in 128-byte pieces);
// MODE 1: the same with the scaling and the second GEMM fused (last direction); MODE 2: data as the A operand (x passes: the SAME registers,
// only the operand order of the MFMA changes, and D comes out transposed - lane = point of the line - so the stores are contiguous along x).
// The data fragments of chunk ch + kAhead are requested while chunk ch multiplies (the loads stay behind the chunk's barrier).
template <int NCH, int MODE>
__global__ void __launch_bounds__(kThreads, MODE == 0 ? 1 : 2)   // (measured per mode: the register cap of two waves per SIMD pays for the fused and the x passes only)
k_fdmu_reg(FdmuPass P, const double *__restrict__ in, double *__restrict__ out) {
  typedef RegGeom<NCH> Gm;
  constexpr int kAhead = NCH < 3 ? NCH : 3;
  __shared__ double LT[2][Gm::CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.y, j = lane & 15, kq = lane >> 4;
  const int64_t n_tile = (int64_t)blockIdx.x * 64 + w * 16;
  const int64_t n = n_tile + j;
  const bool valid = n < P.n_lines;
  // element (k, line n) of the input / output arrays at base + k * stride.  Loads are never predicated: lines behind the end re-read the
  // last line and points behind the line's end re-read its last point (finite values that only meet zero columns of T or are never stored)
  const int64_t nc = valid ? n : P.n_lines - 1;
  const int64_t lb = P.SI == 1 ? nc * P.nK : line_base(P, nc);
  const int64_t in_stride = P.in_interleaved ? P.SI * P.ncomp : P.SI, out_stride = P.out_interleaved ? P.SI * P.ncomp : P.SI;
  const double *in_lane = in + (P.in_interleaved ? lb * P.ncomp + c : (int64_t)c * P.comp_stride + lb);
  const int k_last = P.nK - 1;

  double b[Gm::KKP];
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int kk = 4 * ch + u; const int k = min(4 * kk + kq, k_last); b[kk] = in_lane[(int64_t)k * in_stride]; }
  };
#pragma unroll
  for (int ch = 0; ch < kAhead; ++ch) load_chunk(ch);

  v4d acc[Gm::MT];
  // D = T * B (or B^T * T^T for MODE 2) with T in chunked fragment order [chunk][k-step in chunk][tile pair][lane][2]
  auto gemm = [&](const double *__restrict__ Tg, double (&bb)[Gm::KKP], const bool stream_b, const bool swapped) {
#pragma unroll
    for (int t = 0; t < Gm::MT; ++t) acc[t] = v4d{0, 0, 0, 0};
    constexpr int PER = Gm::CHUNK / 2 / kThreads;   // double2 per thread and chunk = MTP
    double2 stage[PER];
    const double2 *src = reinterpret_cast<const double2 *>(Tg);
#pragma unroll
    for (int i = 0; i < PER; ++i) reinterpret_cast<double2 *>(LT[0])[tid + i * kThreads] = src[tid + i * kThreads];
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (ch + 1 < NCH) {
#pragma unroll
        for (int i = 0; i < PER; ++i) stage[i] = src[(int64_t)(ch + 1) * (Gm::CHUNK / 2) + tid + i * kThreads];
      }
      if (stream_b && ch + kAhead < NCH) load_chunk(ch + kAhead);
      const double2 *La = reinterpret_cast<const double2 *>(LT[ch & 1]) + lane;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = 4 * ch + u;
        if (ch + 1 < NCH || u == 0 || kk < P.KK) {         // wave-uniform, last chunk only: k-steps behind the line length are skipped
#pragma unroll
          for (int p = 0; p < Gm::MTP; ++p) {
            const double2 a = La[(u * Gm::MTP + p) * 64];
            if (swapped) {
              acc[2 * p] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[kk], a.x, acc[2 * p], 0, 0, 0);
              acc[2 * p + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[kk], a.y, acc[2 * p + 1], 0, 0, 0);
            } else {
              acc[2 * p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bb[kk], acc[2 * p], 0, 0, 0);
              acc[2 * p + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bb[kk], acc[2 * p + 1], 0, 0, 0);
            }
          }
        }
      }
      if (ch + 1 < NCH) {
#pragma unroll
        for (int i = 0; i < PER; ++i) reinterpret_cast<double2 *>(LT[(ch + 1) & 1])[tid + i * kThreads] = stage[i];
      }
      __syncthreads();
    }
  };

  gemm(reinterpret_cast<const double *>(P.T1[c]), b, true, MODE == 2);
  if constexpr (MODE == 1) {
    double base = 0;
    const int64_t col = min(P.col0 + nc, P.col_total - 1);     // (lines behind the end: any valid column, the result is never stored)
    if (P.lam1[c]) { const int64_t jj = col / P.n0; base = P.k0[c] * P.lam0[c][col - jj * P.n0] + P.k1[c] * P.lam1[c][jj]; }
    else if (P.lam0[c]) base = P.k0[c] * P.lam0[c][col];
    double b2[Gm::KKP];
    const double *lam_lane = P.lam_d[c]; const double kdc = P.kd[c];
#pragma unroll
    for (int mt = 0; mt < NCH; ++mt) {        // D register (mt, q) of this lane = B operand of k-step 4 mt + q.  (Tiles come in pairs: with an odd number of chunks the last tile, mt = NCH, holds rows
                                              //  behind the line's end only - exact zeros - and b2 has no k-steps for it: looping to MT wrote past b2, which broke lines of 33-48, 65-80, ... points in this
                                              //  non-symmetric form until round 3)
      double den[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) den[q] = fma(kdc, lam_lane[min(16 * mt + kq + 4 * q, k_last)], base);   // removed modes carry lam = inf -> factor 0
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // reciprocal by v_rcp_f64 + one Newton step (full division sequences would cost a quarter of the GEMM); den = inf gives exactly 0
        double r = __builtin_amdgcn_rcp(den[q]);
        r = den[q] < 1e300 ? fma(r, fma(-den[q], r, 1.0), r) : 0.0;
        b2[4 * mt + q] = acc[mt][q] * r;       // rows behind the line's end are exact zeros of T
      }
      __builtin_amdgcn_sched_barrier(0);       // keep the eigenvalue loads of the next tile from piling up in registers
    }
    gemm(reinterpret_cast<const double *>(P.T2[c]), b2, false, false);
  }
  if constexpr (MODE == 2) {
    // transposed D: lane j = point 16 t + j of the line, register q = line kq + 4 q of the tile (x direction: SI == 1)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t nl = n_tile + kq + 4 * q;
      if (nl >= P.n_lines) continue;
      const int64_t ob = P.out_interleaved ? nl * P.nK * P.ncomp + c : (int64_t)c * P.comp_stride + nl * P.nK;
#pragma unroll
      for (int t = 0; t < Gm::MT; ++t) { const int m = 16 * t + j; if (m < P.nK) out[ob + (int64_t)m * out_stride] = acc[t][q]; }
    }
  } else if (valid) {
    const int64_t out_base = P.out_interleaved ? lb * P.ncomp + c : (int64_t)c * P.comp_stride + lb;
#pragma unroll
    for (int t = 0; t < Gm::MT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int m = 16 * t + kq + 4 * q; if (m < P.nK) out[out_base + (int64_t)m * out_stride] = acc[t][q]; }
  }
}

// ---- even / odd form ---------------------------------------------------------------------------------------------------------------
// On a uniform mesh with the same end condition at both ends the 1D matrices are persymmetric, so every eigenvector is symmetric or
// antisymmetric about the centre of the line.  With e_k = x_k + x_{n-1-k}, o_k = x_k - x_{n-1-k} (k in the lower half, the centre node kept as
// it is) the forward transform splits into two half-size products (even modes from e, odd modes from o) and the backward transform into two
// half-size products followed by x_k = a_k + b_k, x_{n-1-k} = a_k - b_k: HALF the MFMA work of the full transform.  A lane loads point
// 4 kk + kq of its line AND its mirror image, so the butterflies stay inside the lane; the two products of a pass run as the two MFMAs fed by
// one 16-byte LDS read (tile pair p = even tile p, odd tile p) and share the chunked streaming of k_fdmu_reg.  Modes are stored along the line
// with the even ones first; the eigenvalue arrays are permuted to match.  MODE 0: forward, 1: backward, 2: forward + scaling + backward.
template <int NCH> struct SplitGeom { static constexpr int P = NCH, KKP = 4 * NCH, CHUNK = 4 * NCH * 128; };

template <int NCH, int MODE>
__global__ void __launch_bounds__(kThreads, 2)
k_fdmu_split(FdmuPass P, const double *__restrict__ in, double *__restrict__ out) {
  typedef SplitGeom<NCH> Gm;
  constexpr int kAhead = NCH < 2 ? NCH : 2;
  __shared__ double LT[2][Gm::CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.y, j = lane & 15, kq = lane >> 4;
  const int64_t n = (int64_t)blockIdx.x * 64 + w * 16 + j;
  const bool valid = n < P.n_lines;
  const int64_t nc = valid ? n : P.n_lines - 1;
  const int64_t lb = P.SI == 1 ? nc * P.nK : line_base(P, nc);
  const int64_t in_stride = P.in_interleaved ? P.SI * P.ncomp : P.SI, out_stride = P.out_interleaved ? P.SI * P.ncomp : P.SI;
  const double *in_lane = in + (P.in_interleaved ? lb * P.ncomp + c : (int64_t)c * P.comp_stride + lb);
  double *out_lane = out + (P.out_interleaved ? lb * P.ncomp + c : (int64_t)c * P.comp_stride + lb);
  const int nn = P.nK, h = (nn + 1) / 2, ne = P.n_even[c], k_last = nn - 1;
  const int KK = (h + 3) / 4;

  // first operand pair: forward = (e, o) from a point and its mirror image; backward = (even, odd) coefficient groups
  double b0[Gm::KKP], b1[Gm::KKP];
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = 4 * ch + u, k = 4 * kk + kq;
      if (MODE == 1) {
        b0[kk] = in_lane[(int64_t)min(k, k_last) * in_stride];
        b1[kk] = in_lane[(int64_t)min(ne + k, k_last) * in_stride];
      } else {
        const int kc = min(k, h - 1), km = k_last - kc;
        const double lo = in_lane[(int64_t)kc * in_stride], hi = in_lane[(int64_t)km * in_stride];
        b0[kk] = km == kc ? lo : lo + hi;       // the centre node of an odd line is its own mirror image
        b1[kk] = lo - hi;
      }
    }
  };
#pragma unroll
  for (int ch = 0; ch < kAhead; ++ch) load_chunk(ch);

  v4d acc[2 * Gm::P];
  auto gemm = [&](const double *__restrict__ Tg, double (&be)[Gm::KKP], double (&bo)[Gm::KKP], const bool stream_b) {
#pragma unroll
    for (int t = 0; t < 2 * Gm::P; ++t) acc[t] = v4d{0, 0, 0, 0};
    constexpr int PER = Gm::CHUNK / 2 / kThreads;
    double2 stage[PER > 0 ? PER : 1];
    const double2 *src = reinterpret_cast<const double2 *>(Tg);
    for (int i = tid; i < Gm::CHUNK / 2; i += kThreads) reinterpret_cast<double2 *>(LT[0])[i] = src[i];
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (ch + 1 < NCH) {
#pragma unroll
        for (int i = 0; i < PER; ++i) stage[i] = src[(int64_t)(ch + 1) * (Gm::CHUNK / 2) + tid + i * kThreads];
      }
      if (stream_b && ch + kAhead < NCH) load_chunk(ch + kAhead);
      const double2 *La = reinterpret_cast<const double2 *>(LT[ch & 1]) + lane;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = 4 * ch + u;
        if (ch + 1 < NCH || u == 0 || kk < KK) {
#pragma unroll
          for (int p = 0; p < Gm::P; ++p) {
            const double2 a = La[(u * Gm::P + p) * 64];
            acc[2 * p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, be[kk], acc[2 * p], 0, 0, 0);
            acc[2 * p + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bo[kk], acc[2 * p + 1], 0, 0, 0);
          }
        }
      }
      if (ch + 1 < NCH) {
        if constexpr (PER > 0) {
#pragma unroll
          for (int i = 0; i < PER; ++i) reinterpret_cast<double2 *>(LT[(ch + 1) & 1])[tid + i * kThreads] = stage[i];
        } else {
          for (int i = tid; i < Gm::CHUNK / 2; i += kThreads) reinterpret_cast<double2 *>(LT[(ch + 1) & 1])[i] = src[(int64_t)(ch + 1) * (Gm::CHUNK / 2) + i];
        }
      }
      __syncthreads();
    }
  };

  gemm(reinterpret_cast<const double *>(P.T1[c]), b0, b1, true);
  if constexpr (MODE == 2) {
    double base = 0;
    const int64_t col = min(P.col0 + nc, P.col_total - 1);
    if (P.lam1[c]) { const int64_t jj = col / P.n0; base = P.k0[c] * P.lam0[c][col - jj * P.n0] + P.k1[c] * P.lam1[c][jj]; }
    else if (P.lam0[c]) base = P.k0[c] * P.lam0[c][col];
    double c0[Gm::KKP], c1[Gm::KKP];
    const double *lam_lane = P.lam_d[c]; const double kdc = P.kd[c];
#pragma unroll
    for (int p = 0; p < Gm::P; ++p) {       // D register (tile p, q) of this lane = B operand of k-step 4 p + q of the same parity group
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = 16 * p + kq + 4 * q;
        const double de = fma(kdc, lam_lane[min(m, k_last)], base), dof_ = fma(kdc, lam_lane[min(ne + m, k_last)], base);
        double re = __builtin_amdgcn_rcp(de), ro = __builtin_amdgcn_rcp(dof_);
        re = de < 1e300 ? fma(re, fma(-de, re, 1.0), re) : 0.0; ro = dof_ < 1e300 ? fma(ro, fma(-dof_, ro, 1.0), ro) : 0.0;
        c0[4 * p + q] = m < ne ? acc[2 * p][q] * re : 0.0;
        c1[4 * p + q] = ne + m < nn ? acc[2 * p + 1][q] * ro : 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    gemm(reinterpret_cast<const double *>(P.T2[c]), c0, c1, false);
  }
  if (!valid) return;
  if constexpr (MODE == 0) {                 // modes: even ones at m, odd ones behind them; slots behind the last mode get the zeros of the padded rows
#pragma unroll
    for (int p = 0; p < Gm::P; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = 16 * p + kq + 4 * q;
        if (m < ne) out_lane[(int64_t)m * out_stride] = acc[2 * p][q];
        if (ne + m < nn) out_lane[(int64_t)(ne + m) * out_stride] = acc[2 * p + 1][q];
      }
  } else {                                   // nodes: x_k = a + b, x_{n-1-k} = a - b
#pragma unroll
    for (int p = 0; p < Gm::P; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 16 * p + kq + 4 * q;
        if (k >= h) continue;
        const double a = acc[2 * p][q], b = acc[2 * p + 1][q];
        const int km = k_last - k;
        out_lane[(int64_t)k * out_stride] = km == k ? a : a + b;
        if (km != k) out_lane[(int64_t)km * out_stride] = a - b;
      }
  }
}

// ---- even / odd form for long lines (> 160 points: 2D config 2 has 673, a 128^3 Q2 box 257) ------------------------------------------------
// Same arithmetic as k_fdmu_split, but the 2 x 5 accumulator tiles cover only 80 output rows per parity: the rows are produced block by block, each block
// streaming ALL k-steps of its slice of the transform matrix through LDS while the data fragments are re-read (L2) and, for the forward direction,
// butterflied again.  Loops are runtime loops (the line length is arbitrary), the data fragments of the next chunk are requested one chunk ahead.
// MODE 0 forward, 1 backward (optionally dividing the coefficients by the eigenvalue sums on load: the last direction, whose forward / backward
// pair cannot be chained through registers at this length).
template <int MODE>
__global__ void __launch_bounds__(kThreads, 2)
k_fdmu_blk(FdmuPass P, const double *__restrict__ in, double *__restrict__ out) {
  constexpr int PP = 5, CHUNK = 4 * PP * 128, PER = CHUNK / 2 / kThreads;
  __shared__ double LT[2][CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.y, j = lane & 15, kq = lane >> 4;
  const int64_t n = (int64_t)blockIdx.x * 64 + w * 16 + j;
  const bool valid = n < P.n_lines;
  const int64_t nc = valid ? n : P.n_lines - 1;
  const int64_t lb = P.SI == 1 ? nc * P.nK : line_base(P, nc);
  const int64_t in_stride = P.in_interleaved ? P.SI * P.ncomp : P.SI, out_stride = P.out_interleaved ? P.SI * P.ncomp : P.SI;
  const double *in_lane = in + (P.in_interleaved ? lb * P.ncomp + c : (int64_t)c * P.comp_stride + lb);
  double *out_lane = out + (P.out_interleaved ? lb * P.ncomp + c : (int64_t)c * P.comp_stride + lb);
  const int nn = P.nK, h = (nn + 1) / 2, ne = P.n_even[c], k_last = nn - 1;
  const int NCHK = P.blk_nch[c], KK = P.blk_kk[c];
  double base = 0; const double *lam_lane = nullptr; double kdc = 0;
  if (MODE == 1 && P.scale_on_load) {
    const int64_t col = min(P.col0 + nc, P.col_total - 1);
    if (P.lam1[c]) { const int64_t jj = col / P.n0; base = P.k0[c] * P.lam0[c][col - jj * P.n0] + P.k1[c] * P.lam1[c][jj]; }
    else if (P.lam0[c]) base = P.k0[c] * P.lam0[c][col];
    lam_lane = P.lam_d[c]; kdc = P.kd[c];
  }
  auto load_chunk = [&](int ch, double (&b0)[4], double (&b1)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 4 * (4 * ch + u) + kq;
      if (MODE == 1) {
        const int ie = min(k, k_last), io = min(ne + k, k_last);
        double ve = in_lane[(int64_t)ie * in_stride], vo = in_lane[(int64_t)io * in_stride];
        if (lam_lane) {
          const double de = fma(kdc, lam_lane[ie], base), dd = fma(kdc, lam_lane[io], base);
          double re = __builtin_amdgcn_rcp(de), ro = __builtin_amdgcn_rcp(dd);
          re = de < 1e300 ? fma(re, fma(-de, re, 1.0), re) : 0.0; ro = dd < 1e300 ? fma(ro, fma(-dd, ro, 1.0), ro) : 0.0;
          ve *= re; vo *= ro;
        }
        b0[u] = ve; b1[u] = vo;
      } else {
        const int kc = min(k, h - 1), km = k_last - kc;
        const double lo = in_lane[(int64_t)kc * in_stride], hi = in_lane[(int64_t)km * in_stride];
        b0[u] = km == kc ? lo : lo + hi; b1[u] = lo - hi;
      }
    }
  };
  // one row block per workgroup (grid z): a 2D mesh has few lines (673 per component at config 2), the row blocks supply the missing parallelism
  for (int mb = blockIdx.z; mb < P.blk_mb[c]; mb += gridDim.z) {
    v4d acc[2 * PP];
#pragma unroll
    for (int t = 0; t < 2 * PP; ++t) acc[t] = v4d{0, 0, 0, 0};
    const double2 *src = reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(P.T1[c]) + (int64_t)mb * NCHK * CHUNK);
    double2 stage[PER];
    double b0[4], b1[4], n0[4], n1[4];
    __syncthreads();                                        // the previous row block has finished with both LDS buffers
#pragma unroll
    for (int i = 0; i < PER; ++i) reinterpret_cast<double2 *>(LT[0])[tid + i * kThreads] = src[tid + i * kThreads];
    load_chunk(0, b0, b1);
    __syncthreads();
    for (int ch = 0; ch < NCHK; ++ch) {
      const bool more = ch + 1 < NCHK;
      if (more) {
#pragma unroll
        for (int i = 0; i < PER; ++i) stage[i] = src[(int64_t)(ch + 1) * (CHUNK / 2) + tid + i * kThreads];
        load_chunk(ch + 1, n0, n1);
      }
      const double2 *La = reinterpret_cast<const double2 *>(LT[ch & 1]) + lane;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (4 * ch + u < KK) {
#pragma unroll
          for (int p = 0; p < PP; ++p) {
            const double2 a = La[(u * PP + p) * 64];
            acc[2 * p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b0[u], acc[2 * p], 0, 0, 0);
            acc[2 * p + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b1[u], acc[2 * p + 1], 0, 0, 0);
          }
        }
      }
      if (more) {
#pragma unroll
        for (int i = 0; i < PER; ++i) reinterpret_cast<double2 *>(LT[(ch + 1) & 1])[tid + i * kThreads] = stage[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) { b0[u] = n0[u]; b1[u] = n1[u]; }
      }
      __syncthreads();
    }
    if (valid) {
#pragma unroll
      for (int p = 0; p < PP; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * (PP * mb + p) + kq + 4 * q;
          if (MODE == 0) {
            if (r < ne) out_lane[(int64_t)r * out_stride] = acc[2 * p][q];
            if (ne + r < nn) out_lane[(int64_t)(ne + r) * out_stride] = acc[2 * p + 1][q];
          } else if (r < h) {
            const double a = acc[2 * p][q], b = acc[2 * p + 1][q];
            const int km = k_last - r;
            out_lane[(int64_t)r * out_stride] = km == r ? a : a + b;
            if (km != r) out_lane[(int64_t)km * out_stride] = a - b;
          }
        }
    }
  }
}

inline int split_nch(int nK) { const int need = ((nK + 1) / 2 + 15) / 16; for (int v : {1, 2, 3, 4, 5}) if (need <= v) return v; return 0; }

void launch_split(hipStream_t s, const FdmuPass &P, const double *in, double *out) {
  const dim3 grid((unsigned)((P.n_lines + 63) / 64), (unsigned)P.ncomp);
  if (P.blk) {
    if (P.fused) throw Error("launch_split: the blocked form runs the last direction as two passes");
    int mbz = 1; for (int c = 0; c < P.ncomp; ++c) mbz = std::max(mbz, P.blk_mb[c]);
    const dim3 gridz(grid.x, grid.y, (unsigned)mbz);
    if (P.split_dir == 0) hipLaunchKernelGGL((k_fdmu_blk<0>), gridz, dim3(kThreads), 0, s, P, in, out);
    else hipLaunchKernelGGL((k_fdmu_blk<1>), gridz, dim3(kThreads), 0, s, P, in, out);
    return;
  }
  const int mode = P.fused ? 2 : P.split_dir;
  switch (split_nch(P.nK) * 4 + mode) {
#define PORO_SPLIT_CASE(N) case 4 * N: hipLaunchKernelGGL((k_fdmu_split<N, 0>), grid, dim3(kThreads), 0, s, P, in, out); break; \
                           case 4 * N + 1: hipLaunchKernelGGL((k_fdmu_split<N, 1>), grid, dim3(kThreads), 0, s, P, in, out); break; \
                           case 4 * N + 2: hipLaunchKernelGGL((k_fdmu_split<N, 2>), grid, dim3(kThreads), 0, s, P, in, out); break;
    PORO_SPLIT_CASE(1) PORO_SPLIT_CASE(2) PORO_SPLIT_CASE(3) PORO_SPLIT_CASE(4) PORO_SPLIT_CASE(5)
#undef PORO_SPLIT_CASE
    default: throw Error("launch_split: line too long");
  }
}

inline int reg_nch(int nK) { const int need = (nK + 15) / 16; for (int v : {1, 2, 3, 5, 7, 10}) if (need <= v) return v; return 0; }

void launch_reg(hipStream_t s, const FdmuPass &P, const double *in, double *out) {
  const dim3 grid((unsigned)((P.n_lines + 63) / 64), (unsigned)P.ncomp);
  const int mode = P.fused ? 1 : (P.SI == 1 && !std::getenv("PORO_FDMU_NO_SWAP")) ? 2 : 0;
  switch (reg_nch(P.nK) * 4 + mode) {
#define PORO_REG_CASE(N) case 4 * N: hipLaunchKernelGGL((k_fdmu_reg<N, 0>), grid, dim3(kThreads), 0, s, P, in, out); break; \
                         case 4 * N + 1: hipLaunchKernelGGL((k_fdmu_reg<N, 1>), grid, dim3(kThreads), 0, s, P, in, out); break; \
                         case 4 * N + 2: hipLaunchKernelGGL((k_fdmu_reg<N, 2>), grid, dim3(kThreads), 0, s, P, in, out); break;
    PORO_REG_CASE(1) PORO_REG_CASE(2) PORO_REG_CASE(3) PORO_REG_CASE(5) PORO_REG_CASE(7) PORO_REG_CASE(10)
#undef PORO_REG_CASE
    default: throw Error("launch_reg: line too long");
  }
}

template <class TC, class TIn, class TOut>
void launch_pass(hipStream_t s, const FdmuPass &P, const TIn *in, TOut *out) {
  if constexpr (std::is_same<TC, double>::value && std::is_same<TIn, double>::value && std::is_same<TOut, double>::value) {
    if (P.split) { launch_split(s, P, in, out); return; }
    if (P.reg_form) { launch_reg(s, P, in, out); return; }
  }
  const int rows = std::max(P.KK * 4, P.MT * 16);
  const size_t lds = sizeof(TC) * (size_t)(P.x_layout ? kNT * P.ld_line : rows * kLdK);
  const dim3 grid((unsigned)((P.n_lines + kNT - 1) / kNT), (unsigned)P.ncomp);
  const int mth = (P.MT + 1) / 2;
#define PORO_FDMU_CASE(M)                                                                                                          \
  { static bool set_##M[64] = {false}; int dev = 0; (void)hipGetDevice(&dev);                                                    \
    if (dev < 64 && !set_##M[dev]) { PORO_HIP(hipFuncSetAttribute((const void *)k_fdmu_pass<TC, TIn, TOut, M>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set_##M[dev] = true; } \
    hipLaunchKernelGGL((k_fdmu_pass<TC, TIn, TOut, M>), grid, dim3(kThreads), lds, s, P, in, out); }
  if (mth <= 1) PORO_FDMU_CASE(1)
  else if (mth <= 2) PORO_FDMU_CASE(2)
  else if (mth <= 3) PORO_FDMU_CASE(3)
  else if (mth <= 5) PORO_FDMU_CASE(5)
  else if (mth <= 7) PORO_FDMU_CASE(7)
  else if (mth <= 10) PORO_FDMU_CASE(10)
  else throw Error("fast diagonalisation of the displacement system: more than 320 nodes per grid line");
#undef PORO_FDMU_CASE
}

// ---- host: 1D matrices, generalised eigen-decomposition ------------------------------------------------------------------------
// symmetric eigenproblem by cyclic Jacobi rotations (n <= 320: a few 1e8 flop, once per mesh); V's columns are the eigenvectors
void jacobi_eig(int n, std::vector<double> &A, std::vector<double> &V, std::vector<double> &w) {
  V.assign((size_t)n * n, 0.0); for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) { diag += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
    if (off <= 1e-30 * diag || off == 0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; ++k) {   // columns p, q
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = cs * akp - sn * akq; A[(size_t)k * n + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < n; ++k) {   // rows p, q
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = cs * apk - sn * aqk; A[(size_t)q * n + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = cs * vkp - sn * vkq; V[(size_t)k * n + q] = sn * vkp + cs * vkq;
        }
      }
  }
  w.resize(n); for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

// symmetric eigenproblem by Householder tridiagonalisation + implicit QL with accumulated transformations (the classical tred2 / tql2 pair):
// O(n^3) with a small constant, for the long lines (n = 671 in BASELINE config 2) where the Jacobi sweeps above would take minutes.
// A is overwritten; V's columns are the eigenvectors.
void householder_ql_eig(int n, std::vector<double> &A, std::vector<double> &V, std::vector<double> &w) {
  std::vector<double> d(n, 0.0), e(n, 0.0);
  auto a = [&](int i, int j) -> double & { return A[(size_t)i * n + j]; };
  for (int i = n - 1; i > 0; --i) {
    const int l = i - 1; double h = 0, scale = 0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(a(i, k));
      if (scale == 0.0) e[i] = a(i, l);
      else {
        for (int k = 0; k <= l; ++k) { a(i, k) /= scale; h += a(i, k) * a(i, k); }
        double f = a(i, l), g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * g; h -= f * g; a(i, l) = f - g; f = 0.0;
        for (int j = 0; j <= l; ++j) {
          a(j, i) = a(i, j) / h;
          g = 0.0;
          for (int k = 0; k <= j; ++k) g += a(j, k) * a(i, k);
          for (int k = j + 1; k <= l; ++k) g += a(k, j) * a(i, k);
          e[j] = g / h; f += e[j] * a(i, j);
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) {
          f = a(i, j); e[j] = g = e[j] - hh * f;
          for (int k = 0; k <= j; ++k) a(j, k) -= f * e[k] + g * a(i, k);
        }
      }
    } else e[i] = a(i, l);
    d[i] = h;
  }
  d[0] = 0.0; e[0] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0)
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        for (int k = 0; k <= l; ++k) g += a(i, k) * a(k, j);
        for (int k = 0; k <= l; ++k) a(k, j) -= g * a(k, i);
      }
    d[i] = a(i, i); a(i, i) = 1.0;
    for (int j = 0; j <= l; ++j) a(j, i) = a(i, j) = 0.0;
  }
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) { const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]); if (std::fabs(e[m]) <= 1e-16 * dd) break; }
      if (m != l) {
        if (iter++ == 200) throw Error("householder_ql_eig: no convergence");
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]), r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double sn = 1.0, cs = 1.0, p = 0.0; int i;
        for (i = m - 1; i >= l; --i) {
          double f = sn * e[i]; const double b = cs * e[i];
          e[i + 1] = (r = std::hypot(f, g));
          if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
          sn = f / r; cs = g / r; g = d[i + 1] - p;
          r = (d[i] - g) * sn + 2.0 * cs * b;
          d[i + 1] = g + (p = sn * r); g = cs * r - b;
          for (int k = 0; k < n; ++k) { f = a(k, i + 1); a(k, i + 1) = sn * a(k, i) + cs * f; a(k, i) = cs * a(k, i) - sn * f; }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[m] = 0.0;
      }
    } while (m != l);
  }
  V = A; w = d;
}

// FE_Q(k) mass / stiffness matrices of n_cells cells of length h (dense, nn = k n_cells + 1); element matrices as in kernels_kron.hip
void fe1d(int k, const std::vector<double> &hc, std::vector<double> &M, std::vector<double> &K) {
  const int n_cells = (int)hc.size(), nn = k * n_cells + 1; M.assign((size_t)nn * nn, 0.0); K.assign((size_t)nn * nn, 0.0);
  static const double M2[3][3] = {{4, 2, -1}, {2, 16, 2}, {-1, 2, 4}}, K2[3][3] = {{7, -8, 1}, {-8, 16, -8}, {1, -8, 7}};
  static const double M1[2][2] = {{2, 1}, {1, 2}}, K1[2][2] = {{1, -1}, {-1, 1}};
  for (int c = 0; c < n_cells; ++c)
    for (int a = 0; a <= k; ++a) for (int b = 0; b <= k; ++b) {
      const size_t at = (size_t)(k * c + a) * nn + (k * c + b);
      const double h = hc[c];
      if (k == 2) { M[at] += h / 30.0 * M2[a][b]; K[at] += K2[a][b] / (3.0 * h); } else { M[at] += h / 6.0 * M1[a][b]; K[at] += K1[a][b] / h; }
    }
}

}  // namespace

// generalised eigenpairs K s = lam M s of the 1D FE_Q(k) matrices with the end nodes lo / hi removed when fix_lo / fix_hi:
// S (nn x nn row-major, S^T M S = I on the free block, zero rows for removed nodes, zero columns behind the n_free modes), lam (inf behind n_free)
void fdmu_eig_1d(int k, int n_cells, double h, bool fix_lo, bool fix_hi, std::vector<double> &S, std::vector<double> &lam) {
  fdmu_eig_1d(k, std::vector<double>((size_t)n_cells, h), fix_lo, fix_hi, S, lam);
}
void fdmu_eig_1d(int k, const std::vector<double> &hc, bool fix_lo, bool fix_hi, std::vector<double> &S, std::vector<double> &lam) {
  const int n_cells = (int)hc.size();
  std::vector<double> M, K; fe1d(k, hc, M, K);
  const int nn = k * n_cells + 1, f0 = fix_lo ? 1 : 0, nf = nn - f0 - (fix_hi ? 1 : 0);
  S.assign((size_t)nn * nn, 0.0); lam.assign(nn, std::numeric_limits<double>::infinity());
  if (nf <= 0) return;
  // Cholesky M_ff = L L^T, C = L^-1 K_ff L^-T, C = Q W Q^T, S_ff = L^-T Q
  std::vector<double> Lc((size_t)nf * nf, 0.0), C((size_t)nf * nf);
  for (int i = 0; i < nf; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = M[(size_t)(i + f0) * nn + (j + f0)];
      for (int p = 0; p < j; ++p) s -= Lc[(size_t)i * nf + p] * Lc[(size_t)j * nf + p];
      if (i == j) { if (!(s > 0)) throw Error("fdmu_eig_1d: mass matrix not positive definite"); Lc[(size_t)i * nf + i] = std::sqrt(s); }
      else Lc[(size_t)i * nf + j] = s / Lc[(size_t)j * nf + j];
    }
  // X = L^-1 K_ff (forward substitution on columns), C = X L^-T = (L^-1 X^T)^T
  std::vector<double> X((size_t)nf * nf);
  for (int col = 0; col < nf; ++col)
    for (int i = 0; i < nf; ++i) {
      double s = K[(size_t)(i + f0) * nn + (col + f0)];
      for (int p = 0; p < i; ++p) s -= Lc[(size_t)i * nf + p] * X[(size_t)p * nf + col];
      X[(size_t)i * nf + col] = s / Lc[(size_t)i * nf + i];
    }
  for (int row = 0; row < nf; ++row)         // solve L y = X[row, :]^T  ->  C[:, row] = y
    for (int i = 0; i < nf; ++i) {
      double s = X[(size_t)row * nf + i];
      for (int p = 0; p < i; ++p) s -= Lc[(size_t)i * nf + p] * C[(size_t)p * nf + row];
      C[(size_t)i * nf + row] = s / Lc[(size_t)i * nf + i];
    }
  for (int i = 0; i < nf; ++i) for (int j = i + 1; j < nf; ++j) { const double a = 0.5 * (C[(size_t)i * nf + j] + C[(size_t)j * nf + i]); C[(size_t)i * nf + j] = C[(size_t)j * nf + i] = a; }
  std::vector<double> Q, wv;
  if (nf > 96) householder_ql_eig(nf, C, Q, wv); else jacobi_eig(nf, C, Q, wv);
  for (int j = 0; j < nf; ++j) {             // back substitution L^T s = q_j
    std::vector<double> sv(nf);
    for (int i = nf - 1; i >= 0; --i) {
      double s = Q[(size_t)i * nf + j];
      for (int p = i + 1; p < nf; ++p) s -= Lc[(size_t)p * nf + i] * sv[p];
      sv[i] = s / Lc[(size_t)i * nf + i];
    }
    for (int i = 0; i < nf; ++i) S[(size_t)(i + f0) * nn + j] = sv[i];
    lam[j] = std::max(wv[j], 0.0);
  }
  // the same condition at both ends: M and K are persymmetric, every eigenvector is symmetric or antisymmetric about the centre up to the rounding of
  // the eigen-solver (1e-9 for the clustered top of a 671-point spectrum).  Make that exact - the even / odd transform kernels rely on it - and restore
  // the M-normalisation.
  if (fix_lo == fix_hi)
    for (int j = 0; j < nf; ++j) {
      double ds = 0, da = 0;
      for (int k = 0; k < nn; ++k) { const double a = S[(size_t)k * nn + j], b = S[(size_t)(nn - 1 - k) * nn + j]; ds += (a - b) * (a - b); da += (a + b) * (a + b); }
      const double sgn = ds <= da ? 1.0 : -1.0;
      if (std::min(ds, da) > 1e-8 * std::max(ds, da)) continue;         // (not the expected structure: left alone, the full-length kernels take over)
      for (int k = 0; k < nn / 2; ++k) {
        const double a = S[(size_t)k * nn + j], b = S[(size_t)(nn - 1 - k) * nn + j], v = 0.5 * (a + sgn * b);
        S[(size_t)k * nn + j] = v; S[(size_t)(nn - 1 - k) * nn + j] = sgn * v;
      }
      if ((nn & 1) && sgn < 0) S[(size_t)(nn / 2) * nn + j] = 0.0;
      double nrm = 0;
      for (int i = 0; i < nn; ++i) { double t = 0; for (int p = std::max(0, i - 2 * k); p <= std::min(nn - 1, i + 2 * k); ++p) t += M[(size_t)i * nn + p] * S[(size_t)p * nn + j]; nrm += S[(size_t)i * nn + j] * t; }
      const double sc = 1.0 / std::sqrt(nrm);
      for (int i = 0; i < nn; ++i) S[(size_t)i * nn + j] *= sc;
    }
}

double sym_lambda_max(int n, const std::vector<double> &A) {
  std::vector<double> B = A, V, w; jacobi_eig(n, B, V, w);
  double m = 0; for (double v : w) m = std::max(m, v);
  return m;
}
// largest eigenvalue of D^-1/2 A D^-1/2 for a small dense symmetric matrix (row-major n x n), D = diag(A): the rigorous element-level bound
// lambda_max(D^-1 A_global) <= max_e lambda_max(diag(A_e)^-1 A_e) of the Chebyshev preconditioner
double jacobi_scaled_lambda_max(int n, const std::vector<double> &A) {
  std::vector<double> B((size_t)n * n), V, w;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) B[(size_t)i * n + j] = 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]) / std::sqrt(A[(size_t)i * n + i] * A[(size_t)j * n + j]);
  jacobi_eig(n, B, V, w);
  double m = 0; for (double v : w) m = std::max(m, v);
  return m;
}

namespace {
// MFMA A-fragment order of the (nn x nn) matrix Tm (row-major; transposed access when `transpose`): [MT][KK][64], lane -> row 16 mt + (lane & 15), column 4 kk + (lane >> 4)
template <class TC> void upload_fragments(DevBuf<double> &dst, const std::vector<double> &Tm, int nn, bool transpose) {
  const int MT = (nn + 15) / 16, KK = (nn + 3) / 4;
  std::vector<TC> f((size_t)MT * KK * 64, (TC)0);
  for (int mt = 0; mt < MT; ++mt) for (int kk = 0; kk < KK; ++kk) for (int l = 0; l < 64; ++l) {
    const int r = 16 * mt + (l & 15), cc = 4 * kk + (l >> 4);
    if (r < nn && cc < nn) f[((size_t)mt * KK + kk) * 64 + l] = (TC)(transpose ? Tm[(size_t)cc * nn + r] : Tm[(size_t)r * nn + cc]);
  }
  const size_t bytes = f.size() * sizeof(TC);
  dst.alloc((bytes + 7) / 8);
  PORO_HIP(hipMemcpy(dst.p, f.data(), bytes, hipMemcpyHostToDevice));
}
}  // namespace

// chunked order of k_fdmu_reg: [chunk][k-step u][tile pair p][lane][2]; lane -> row 16 (2p + e) + (lane & 15), column 4 (4 chunk + u) + (lane >> 4)
static void upload_chunked(DevBuf<double> &dst, const std::vector<double> &Tm, int nn, bool transpose) {
  const int nch = reg_nch(nn), mtp = (nch + 1) / 2;
  std::vector<double> f((size_t)nch * 4 * mtp * 128, 0.0);
  for (int ch = 0; ch < nch; ++ch) for (int u = 0; u < 4; ++u) for (int p = 0; p < mtp; ++p) for (int l = 0; l < 64; ++l) for (int e = 0; e < 2; ++e) {
    const int r = 16 * (2 * p + e) + (l & 15), cc = 4 * (4 * ch + u) + (l >> 4);
    if (r < nn && cc < nn) f[(((size_t)(ch * 4 + u) * mtp + p) * 64 + l) * 2 + e] = transpose ? Tm[(size_t)cc * nn + r] : Tm[(size_t)r * nn + cc];
  }
  dst.upload(f);
}

// even / odd form: classify the eigenvectors by their symmetry about the centre, pack the half-size matrices in the chunked pair order of k_fdmu_split
// ([chunk][k-step u][tile p][lane][2]: element 0 = even product, 1 = odd product; lane -> row 16 p + (lane & 15), column 4 (4 chunk + u) + (lane >> 4))
static bool upload_split(FdmuDir &D, const std::vector<double> &S, const std::vector<double> &lam, int nn) {
  const int nch = split_nch(nn);
  const int h = (nn + 1) / 2;
  std::vector<int> even, odd;
  for (int m = 0; m < nn; ++m) {
    if (!(lam[m] < 1e300)) continue;              // removed modes
    double ds = 0, da = 0, nrm = 0;
    for (int k = 0; k < nn; ++k) { const double a = S[(size_t)k * nn + m], b = S[(size_t)(nn - 1 - k) * nn + m]; ds += (a - b) * (a - b); da += (a + b) * (a + b); nrm += a * a; }
    if (ds <= 1e-20 * nrm) even.push_back(m); else if (da <= 1e-20 * nrm) odd.push_back(m); else return false;
  }
  const int ne = (int)even.size(), no = (int)odd.size();
  std::vector<double> lp(nn, std::numeric_limits<double>::infinity());
  for (int i = 0; i < ne; ++i) lp[i] = lam[even[i]];
  for (int i = 0; i < no; ++i) lp[ne + i] = lam[odd[i]];
  if (!nch) {
    // long lines: blocked packing of k_fdmu_blk, [row block][chunk][k-step u][tile p][lane][2] with 80 rows per parity and block
    constexpr int PP = 5;
    auto pack_blk = [&](DevBuf<double> &dst, bool forward, int &kk_out, int &nch_out, int &mb_out) {
      const int rows = forward ? std::max(ne, no) : h, cols = forward ? h : std::max(ne, no);
      const int kk = (cols + 3) / 4, nchk = (kk + 3) / 4, mb = (rows + 16 * PP - 1) / (16 * PP);
      std::vector<double> f((size_t)mb * nchk * 4 * PP * 128, 0.0);
      for (int b = 0; b < mb; ++b) for (int ch = 0; ch < nchk; ++ch) for (int u = 0; u < 4; ++u) for (int p = 0; p < PP; ++p) for (int l = 0; l < 64; ++l) for (int e = 0; e < 2; ++e) {
        const int r = 16 * (PP * b + p) + (l & 15), cc = 4 * (4 * ch + u) + (l >> 4);
        const std::vector<int> &grp = e ? odd : even;
        double v = 0;
        if (forward) { if (r < (int)grp.size() && cc < h) v = S[(size_t)cc * nn + grp[r]]; }
        else { if (r < h && cc < (int)grp.size()) v = S[(size_t)r * nn + grp[cc]]; }
        f[((((size_t)(b * nchk + ch) * 4 + u) * PP + p) * 64 + l) * 2 + e] = v;
      }
      dst.upload(f); kk_out = kk; nch_out = nchk; mb_out = mb;
    };
    pack_blk(D.fwd, true, D.blk_kk[0], D.blk_nch[0], D.blk_mb[0]); pack_blk(D.bwd, false, D.blk_kk[1], D.blk_nch[1], D.blk_mb[1]);
    D.lam.upload(lp); D.n_even = ne; D.split = true; D.blk = true;
    return true;
  }
  if (ne > 16 * nch || no > 16 * nch || h > 16 * nch) return false;
  auto pack = [&](DevBuf<double> &dst, bool forward) {
    std::vector<double> f((size_t)nch * 4 * nch * 128, 0.0);
    for (int ch = 0; ch < nch; ++ch) for (int u = 0; u < 4; ++u) for (int p = 0; p < nch; ++p) for (int l = 0; l < 64; ++l) for (int e = 0; e < 2; ++e) {
      const int r = 16 * p + (l & 15), cc = 4 * (4 * ch + u) + (l >> 4);
      const std::vector<int> &grp = e ? odd : even;
      double v = 0;
      if (forward) { if (r < (int)grp.size() && cc < h) v = S[(size_t)cc * nn + grp[r]]; }      // rows = modes of the parity group, columns = lower-half nodes
      else { if (r < h && cc < (int)grp.size()) v = S[(size_t)r * nn + grp[cc]]; }              // rows = lower-half nodes, columns = modes
      f[(((size_t)(ch * 4 + u) * nch + p) * 64 + l) * 2 + e] = v;
    }
    dst.upload(f);
  };
  pack(D.fwd, true); pack(D.bwd, false);
  D.lam.upload(lp); D.n_even = ne; D.split = true; D.blk = false;
  return true;
}

void fdmu_upload_dir(FdmuDir &D, const std::vector<double> &S, const std::vector<double> &lam, int nn, bool single, bool allow_split) {
  D.n = nn; D.split = false; D.blk = false; D.n_even = 0; D.reg_form = !single && reg_nch(nn) > 0 && !std::getenv("PORO_FDMU_LDS_FORM");
  if (!single && allow_split && !std::getenv("PORO_FDMU_LDS_FORM") && !std::getenv("PORO_FDMU_NO_SPLIT") && upload_split(D, S, lam, nn)) return;
  if (nn > 320) throw Error("fast diagonalisation of the displacement system: a line of more than 320 points needs the even / odd form (the same Dirichlet condition at both ends of every direction)");
  if (D.reg_form) { upload_chunked(D.fwd, S, nn, true); upload_chunked(D.bwd, S, nn, false); D.lam.upload(lam); return; }
  if (single) { upload_fragments<float>(D.fwd, S, nn, true); upload_fragments<float>(D.bwd, S, nn, false); }
  else { upload_fragments<double>(D.fwd, S, nn, true); upload_fragments<double>(D.bwd, S, nn, false); }
  D.lam.upload(lam);
}

// z = blockdiag(A_cc)^-1 g.  g, z: node-interleaved vectors of the local grid nn[0] x nn[1] (x nn[2]); t1, t2: planar scratch of the same size.
// z_lines != null (partitioned run): the last-direction pass works on whole global lines of this rank's column group, which the caller
// gathers / scatters around it (see ctx.hip); then only the passes of the leading directions run here (stage 0: forward, 1: backward).
// fills the pass descriptor of direction d of a grid nn[0] x nn[1] (x nn[2]) from the per-(component, direction) data `dirs[c]`
static void fill_dir(FdmuPass &P, const FdmuDir *const dirs[3], int dim, bool fwd) {
  P.reg_form = dirs[0]->reg_form ? 1 : 0;
  P.split = 1; P.split_dir = fwd ? 0 : 1; P.blk = 1;
  for (int c = 0; c < dim; ++c) { P.split = P.split && dirs[c]->split; P.blk = P.blk && dirs[c]->blk; P.n_even[c] = dirs[c]->n_even; }
  if (!P.split) P.blk = 0;
  if (P.blk) { const int w = fwd ? 0 : 1; for (int c = 0; c < dim; ++c) { P.blk_kk[c] = dirs[c]->blk_kk[w]; P.blk_nch[c] = dirs[c]->blk_nch[w]; P.blk_mb[c] = dirs[c]->blk_mb[w]; } }
}
template <class TC> static void fdmu_apply_t(hipStream_t s, const FdmU &F, const double *g, double *z, void *t1v, void *t2v, int stage) {
  TC *t1 = reinterpret_cast<TC *>(t1v), *t2 = reinterpret_cast<TC *>(t2v);
  const int dim = F.dim; const int64_t nx = F.nn[0], ny = F.nn[1], nz = dim == 3 ? F.nn[2] : 1, nnode = nx * ny * nz;
  auto pass = [&](int d, bool fwd, bool fused) {
    FdmuPass P{};
    P.nK = F.nn[d]; P.MT = (P.nK + 15) / 16; P.KK = (P.nK + 3) / 4;
    P.SI = d == 0 ? 1 : d == 1 ? nx : nx * ny; P.n_lines = nnode / P.nK; P.comp_stride = nnode; P.ncomp = dim;
    P.x_layout = d == 0 ? 1 : 0; P.ld_line = std::max(P.KK * 4, P.MT * 16) + 2;
    P.fused = fused ? 1 : 0; P.n0 = (int)nx; P.col0 = 0; P.col_total = P.n_lines;
    const FdmuDir *dirs[3] = {&F.dir[0][d], &F.dir[dim > 1 ? 1 : 0][d], &F.dir[dim > 2 ? 2 : 0][d]};
    fill_dir(P, dirs, dim, fwd);
    for (int c = 0; c < dim; ++c) {
      const FdmuDir &D = F.dir[c][d];
      P.T1[c] = (fwd || fused) ? (const void *)D.fwd.p : (const void *)D.bwd.p; P.T2[c] = D.bwd.p;
      P.lam_d[c] = D.lam.p; P.kd[c] = F.coef[c][d];
      P.lam0[c] = F.dir[c][0].lam.p; P.k0[c] = F.coef[c][0];
      P.lam1[c] = dim == 3 ? F.dir[c][1].lam.p : nullptr; P.k1[c] = dim == 3 ? F.coef[c][1] : 0.0;
    }
    return P;
  };
  const int last = dim - 1;
  if (stage == 0 || stage == 2) {
    FdmuPass P = pass(0, true, false); P.in_interleaved = 1;
    launch_pass<TC, double, TC>(s, P, g, t1);                                    // x forward: g (interleaved) -> t1 (planar)
    if (dim == 3) { P = pass(1, true, false); launch_pass<TC, TC, TC>(s, P, t1, t2); }   // y forward: t1 -> t2
  }
  TC *cur = dim == 3 ? t2 : t1, *other = dim == 3 ? t1 : t2;                    // where the data are after the leading directions
  if (stage == 2) {                                                               // single rank: last direction forward + scale + backward
    FdmuPass P = pass(last, true, true);
    if (P.blk) {                                                                  // long lines: two passes, the scaling rides on the backward pass's loads
      P.fused = 0; launch_pass<TC, TC, TC>(s, P, cur, other);
      FdmuPass Q = pass(last, false, false); Q.scale_on_load = 1; launch_pass<TC, TC, TC>(s, Q, other, cur);
    } else { launch_pass<TC, TC, TC>(s, P, cur, other); std::swap(cur, other); }
  }
  if (stage == 1 || stage == 2) {
    // stage 1 (partitioned): the caller scattered the result of the last direction back into `cur`
    if (dim == 3) {
      FdmuPass P = pass(1, false, false);
      launch_pass<TC, TC, TC>(s, P, cur, other); std::swap(cur, other);
    }
    FdmuPass P = pass(0, false, false); P.out_interleaved = 1;
    launch_pass<TC, TC, double>(s, P, cur, z);
  }
}
// copy a (component, planes, columns) window between a planar grid array [c][grid_planes][grid_stride] and a dense, zero-padded block
// [c][n_planes_pad][C] (to_block) or back: the pieces of the all-to-all that hands whole lines of the partitioned direction to a rank
__global__ void k_fdmu_window(double *dst, const double *src, int to_block, int ncomp, int n_planes, int n_planes_pad, int64_t C, int64_t ncols_valid,
                              int64_t grid_stride, int64_t grid_planes, int64_t grid_col0, int64_t grid_plane0) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)ncomp * n_planes_pad * C) return;
  const int64_t cc = idx % C, rest = idx / C, k = rest % n_planes_pad, comp = rest / n_planes_pad;
  const bool valid = k < n_planes && cc < ncols_valid;
  const int64_t g = (comp * grid_planes + grid_plane0 + k) * grid_stride + grid_col0 + cc;
  if (to_block) dst[idx] = valid ? src[g] : 0.0;
  else if (valid) dst[g] = src[idx];
}
void fdmu_window(hipStream_t s, double *dst, const double *src, bool to_block, int ncomp, int n_planes, int n_planes_pad, int64_t C, int64_t ncols_valid,
                 int64_t grid_stride, int64_t grid_planes, int64_t grid_col0, int64_t grid_plane0) {
  const int64_t n = (int64_t)ncomp * n_planes_pad * C;
  if (n) hipLaunchKernelGGL(k_fdmu_window, (unsigned)((n + 255) / 256), 256, 0, s, dst, src, to_block ? 1 : 0, ncomp, n_planes, n_planes_pad, C, ncols_valid, grid_stride, grid_planes, grid_col0, grid_plane0);
}

void fdmu_apply(hipStream_t s, const FdmU &F, const double *g, double *z, void *t1, void *t2, int stage) {
  if (F.single) fdmu_apply_t<float>(s, F, g, z, t1, t2, stage); else fdmu_apply_t<double>(s, F, g, z, t1, t2, stage);
}

// the fused last-direction pass on a column-distributed array [component][global line point][C local columns] (partitioned runs)
template <class TC> static void fdmu_lines_t(hipStream_t s, const FdmU &F, const FdmuDir *last_dir /*[dim]*/, int64_t C, int64_t col0, int64_t ncol_valid, void *in_v, void *out_v) {
  const int dim = F.dim, last = dim - 1;
  auto pass = [&](bool fwd, bool fused) {
    FdmuPass P{};
    P.nK = last_dir[0].n; P.MT = (P.nK + 15) / 16; P.KK = (P.nK + 3) / 4;
    P.SI = C; P.n_lines = C; P.comp_stride = (int64_t)P.nK * C; P.ncomp = dim; P.x_layout = 0; P.ld_line = 0;
    P.fused = fused ? 1 : 0; P.n0 = F.nn[0]; P.col0 = col0; P.col_total = col0 + ncol_valid;   // padding columns hold zeros and stay zero
    const FdmuDir *dirs[3] = {&last_dir[0], &last_dir[dim > 1 ? 1 : 0], &last_dir[dim > 2 ? 2 : 0]};
    fill_dir(P, dirs, dim, fwd);
    for (int c = 0; c < dim; ++c) {
      P.T1[c] = (fwd || fused) ? last_dir[c].fwd.p : last_dir[c].bwd.p; P.T2[c] = last_dir[c].bwd.p; P.lam_d[c] = last_dir[c].lam.p; P.kd[c] = F.coef[c][last];
      P.lam0[c] = F.dir[c][0].lam.p; P.k0[c] = F.coef[c][0];
      P.lam1[c] = dim == 3 ? F.dir[c][1].lam.p : nullptr; P.k1[c] = dim == 3 ? F.coef[c][1] : 0.0;
    }
    return P;
  };
  TC *in = reinterpret_cast<TC *>(in_v), *out = reinterpret_cast<TC *>(out_v);
  FdmuPass P = pass(true, true);
  if (P.blk) {
    P.fused = 0; launch_pass<TC, TC, TC>(s, P, in, out);
    FdmuPass Q = pass(false, false); Q.scale_on_load = 1; launch_pass<TC, TC, TC>(s, Q, out, in);
    PORO_HIP(hipMemcpyAsync(out, in, sizeof(TC) * (size_t)dim * P.nK * C, hipMemcpyDeviceToDevice, s));
  } else launch_pass<TC, TC, TC>(s, P, in, out);
}
void fdmu_lines(hipStream_t s, const FdmU &F, const FdmuDir *last_dir, int64_t C, int64_t col0, int64_t ncol_valid, void *in, void *out) {
  if (F.single) fdmu_lines_t<float>(s, F, last_dir, C, col0, ncol_valid, in, out); else fdmu_lines_t<double>(s, F, last_dir, C, col0, ncol_valid, in, out);
}

}  // namespace poro
