// K-apply-u, sum-factorised form for 3D Q2 on a uniform box (gfx950, wave64).
//
// On a tensor-product mesh with constant coefficients the elasticity operator of
// PoroElasticDisplacementSolver::assemble_system (:230-242) is EXACTLY a sum of Kronecker products of the 1D banded
// matrices M (mass), K (stiffness) and C (C[m][n] = int phi_m' phi_n) of the three directions:
//   A_aa = (lambda+2G) K_a (x) M (x) M + G sum_{d != a} K_d (x) M (x) M,   A_ab = lambda C_a (x) C_b^T (x) M + G C_a^T (x) C_b (x) M
// (validated against the assembled matrix to 1e-15 in tools/kron_proto.py).  With C = O + D (off-diagonal part + boundary
// diagonal) and C^T = -O + D, one application is 9 z-sweeps + 15 y-sweeps + 9 x-sweeps of <= 5-point 1D stencils:
// ~160 FMA per node instead of ~800 for the element-matrix gather, which turns the kernel from FP64-issue-bound into HBM-bound.
//
// Mapping: a workgroup owns a 60 x 12 node column of the mesh (64 x 16 with halo) and MARCHES along z over a chunk of planes.
//   z: each thread keeps a 5-plane window of its nodes in registers (sliding, planes processed in vertex/mid pairs);
//   y: the nine z-stage fields of a plane go through LDS; waves hold rows of ONE parity so band coefficients are SGPRs;
//   x: a thread owns an (even, odd) node pair; contributions to the neighbour pairs travel by wave shuffles (scatter form).
// Out-of-domain nodes are loaded as zeros, so only the centre coefficients know about the domain boundary.
// No atomics; results are bitwise reproducible.  Dirichlet rows/columns handled as in k_mf_apply.
#include "common.hpp"

namespace poro {
namespace {

constexpr int TXL = 32;          // x-pairs per row of the tile (64 nodes, 60 valid)
constexpr int TYR = 16;          // rows of the tile (12 valid)
constexpr int NFLD = 12;         // 9 z-stage fields + 3 boundary-plane fields
constexpr int VX = 60, VY = 12;  // valid outputs per tile

struct D2 { double e, o; };
__device__ inline D2 operator*(double s, D2 v) { return {s * v.e, s * v.o}; }
__device__ inline D2 operator+(D2 a, D2 b) { return {a.e + b.e, a.o + b.o}; }
__device__ inline void fma2(D2 &acc, double s, D2 v) { acc.e = fma(s, v.e, acc.e); acc.o = fma(s, v.o, acc.o); }

struct Kron1D { double M[3][3], K[3][3], C[3][3]; };
struct KronArgs {
  int nn[3]; int ntx, nty, nzc, chunk;   // chunk = planes per z-chunk (even)
  Kron1D e[3];
  double lam, G;
  const uint8_t *mask; const double *diag_local; int constrained, mask_anywhere;
};

__device__ inline int64_t xcd_remap(int64_t bid, int64_t n) {
  const int64_t q = n / 8, r = n % 8, xcd = bid % 8, idx = bid / 8;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

__global__ void __launch_bounds__(512, 2)
k_kron3_q2(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double lds_raw[];
  D2 *L = reinterpret_cast<D2 *>(lds_raw);                 // [NFLD][TYR][TXL]
  const int tid = threadIdx.x;
  const int NX = a.nn[0], NY = a.nn[1], NZ = a.nn[2];
  const int64_t nblocks = (int64_t)a.ntx * a.nty * a.nzc;
  const int64_t tile = xcd_remap(blockIdx.x, nblocks);
  const int zc = (int)(tile % a.nzc), tyi = (int)((tile / a.nzc) % a.nty), txi = (int)(tile / ((int64_t)a.nzc * a.nty));
  const int X0 = VX * txi - 2, Y0 = VY * tyi - 2;
  const int k0 = zc * a.chunk, k1 = min(NZ, k0 + a.chunk);

  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, rw = lane >> 5, px = lane & 31;
  // rows of one parity per wave; waves 0,1 hold the halo rows (0,14) / (1,15) and only feed the y-stage of the others
  const int r = (w < 2) ? (w + 14 * rw) : (4 * ((w - 2) >> 1) + 2 + (w & 1) + 2 * rw);
  const bool odd_row = (w & 1) != 0, halo_wave = w < 2;
  const int j = Y0 + r, ie = X0 + 2 * px, io = ie + 1;
  const bool vj = j >= 0 && j < NY, ve = vj && ie >= 0 && ie < NX, vo = vj && io >= 0 && io < NX;
  const bool out_e = ve && px >= 1 && px <= 30 && r >= 2 && r <= 13, out_o = vo && px >= 1 && px <= 30 && r >= 2 && r <= 13;
  const bool bnd_xy_e = ie == 0 || ie == NX - 1 || j == 0 || j == NY - 1, bnd_xy_o = j == 0 || j == NY - 1;   // odd nodes are never on an x face

  const Kron1D &EX = a.e[0], &EY = a.e[1], &EZ = a.e[2];
  // centre coefficients (the only place the domain boundary enters): vertex rows sum the parts of the cells that exist
  const double cMx = (ie > 0 ? EX.M[2][2] : 0.0) + (ie < NX - 1 ? EX.M[0][0] : 0.0);
  const double cKx = (ie > 0 ? EX.K[2][2] : 0.0) + (ie < NX - 1 ? EX.K[0][0] : 0.0);
  const double cDx = (ie > 0 ? EX.C[2][2] : 0.0) + (ie < NX - 1 ? EX.C[0][0] : 0.0);
  const double cMy = odd_row ? EY.M[1][1] : (j > 0 ? EY.M[2][2] : 0.0) + (j < NY - 1 ? EY.M[0][0] : 0.0);
  const double cKy = odd_row ? EY.K[1][1] : (j > 0 ? EY.K[2][2] : 0.0) + (j < NY - 1 ? EY.K[0][0] : 0.0);
  const double cDy = odd_row ? EY.C[1][1] : (j > 0 ? EY.C[2][2] : 0.0) + (j < NY - 1 ? EY.C[0][0] : 0.0);

  const double lam = a.lam, G = a.G, l2g = lam + 2 * G, c1 = -(lam + G), c2 = lam - G, c3 = G - lam, c4 = lam + G;

  auto load_plane = [&](int p, double (&v)[6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = 0.0;
    if (p < 0 || p >= NZ || !vj) return;
    const int64_t base = ((int64_t)p * NY + j) * NX;
    const bool zb = p == 0 || p == NZ - 1;
    if (ve) {
      const int64_t d0 = (base + ie) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = x[d0 + c];
      if (a.constrained && (a.mask_anywhere || zb || bnd_xy_e)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (a.mask[d0 + c]) v[c] = 0.0;
      }
    }
    if (vo) {
      const int64_t d0 = (base + io) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[3 + c] = x[d0 + c];
      if (a.constrained && (a.mask_anywhere || zb || bnd_xy_o)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (a.mask[d0 + c]) v[3 + c] = 0.0;
      }
    }
  };

  double W0[6], W1[6], W2[6], W3[6], W4[6], P0[6], P1[6];
  load_plane(k0 - 2, W0); load_plane(k0 - 1, W1); load_plane(k0, W2); load_plane(k0 + 1, W3); load_plane(k0 + 2, W4);

  // one plane: z-stage in registers -> LDS -> y-stage -> x-stage -> store
  auto plane = [&](const int kk, const bool oddz) {
    D2 f[9];       // mz_x mz_y mz_z kz_x kz_y kz_z oz_x oz_y oz_z
    D2 wz[3];
    const bool has_w = !oddz && (kk == 0 || kk == NZ - 1);
    if (!oddz) {
      const double cM = (kk > 0 ? EZ.M[2][2] : 0.0) + (kk < NZ - 1 ? EZ.M[0][0] : 0.0);
      const double cK = (kk > 0 ? EZ.K[2][2] : 0.0) + (kk < NZ - 1 ? EZ.K[0][0] : 0.0);
      const double cD = (kk > 0 ? EZ.C[2][2] : 0.0) + (kk < NZ - 1 ? EZ.C[0][0] : 0.0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        f[c].e = EZ.M[2][0] * W0[c] + EZ.M[2][1] * W1[c] + cM * W2[c] + EZ.M[0][1] * W3[c] + EZ.M[0][2] * W4[c];
        f[c].o = EZ.M[2][0] * W0[3 + c] + EZ.M[2][1] * W1[3 + c] + cM * W2[3 + c] + EZ.M[0][1] * W3[3 + c] + EZ.M[0][2] * W4[3 + c];
        f[3 + c].e = EZ.K[2][0] * W0[c] + EZ.K[2][1] * W1[c] + cK * W2[c] + EZ.K[0][1] * W3[c] + EZ.K[0][2] * W4[c];
        f[3 + c].o = EZ.K[2][0] * W0[3 + c] + EZ.K[2][1] * W1[3 + c] + cK * W2[3 + c] + EZ.K[0][1] * W3[3 + c] + EZ.K[0][2] * W4[3 + c];
        f[6 + c].e = EZ.C[2][0] * W0[c] + EZ.C[2][1] * W1[c] + EZ.C[0][1] * W3[c] + EZ.C[0][2] * W4[c];
        f[6 + c].o = EZ.C[2][0] * W0[3 + c] + EZ.C[2][1] * W1[3 + c] + EZ.C[0][1] * W3[3 + c] + EZ.C[0][2] * W4[3 + c];
        wz[c].e = cD * W2[c]; wz[c].o = cD * W2[3 + c];
      }
    } else {   // mid plane: couples to kk-1, kk, kk+1 = W2, W3, W4
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        f[c].e = EZ.M[1][0] * W2[c] + EZ.M[1][1] * W3[c] + EZ.M[1][2] * W4[c];
        f[c].o = EZ.M[1][0] * W2[3 + c] + EZ.M[1][1] * W3[3 + c] + EZ.M[1][2] * W4[3 + c];
        f[3 + c].e = EZ.K[1][0] * W2[c] + EZ.K[1][1] * W3[c] + EZ.K[1][2] * W4[c];
        f[3 + c].o = EZ.K[1][0] * W2[3 + c] + EZ.K[1][1] * W3[3 + c] + EZ.K[1][2] * W4[3 + c];
        f[6 + c].e = EZ.C[1][0] * W2[c] + EZ.C[1][2] * W4[c];
        f[6 + c].o = EZ.C[1][0] * W2[3 + c] + EZ.C[1][2] * W4[3 + c];
        wz[c].e = EZ.C[1][1] * W3[c]; wz[c].o = EZ.C[1][1] * W3[3 + c];   // = 0 for Lagrange Q2
      }
    }
    __syncthreads();                                   // every wave is done reading the previous plane's fields
#pragma unroll
    for (int q = 0; q < 9; ++q) L[(q * TYR + r) * TXL + px] = f[q];
    if (has_w) {
#pragma unroll
      for (int c = 0; c < 3; ++c) L[((9 + c) * TYR + r) * TXL + px] = wz[c];
    }
    __syncthreads();
    if (halo_wave) return;

    // ---- y-stage: banded sweeps over the rows of the tile ----
    // band coefficients of this wave's row parity (wave-uniform -> SGPR)
    const double yMm2 = EY.M[2][0], yMm1 = odd_row ? EY.M[1][0] : EY.M[2][1], yMp1 = odd_row ? EY.M[1][2] : EY.M[0][1], yMp2 = EY.M[0][2];
    const double yKm2 = EY.K[2][0], yKm1 = odd_row ? EY.K[1][0] : EY.K[2][1], yKp1 = odd_row ? EY.K[1][2] : EY.K[0][1], yKp2 = EY.K[0][2];
    const double yOm2 = EY.C[2][0], yOm1 = odd_row ? EY.C[1][0] : EY.C[2][1], yOp1 = odd_row ? EY.C[1][2] : EY.C[0][1], yOp2 = EY.C[0][2];
    D2 nm2, nm1, np1, np2;
    auto nb = [&](int q) {
      nm1 = L[(q * TYR + r - 1) * TXL + px]; np1 = L[(q * TYR + r + 1) * TXL + px];
      if (!odd_row) { nm2 = L[(q * TYR + r - 2) * TXL + px]; np2 = L[(q * TYR + r + 2) * TXL + px]; }
    };
    auto sweepM = [&](D2 own) { D2 s = cMy * own; fma2(s, yMm1, nm1); fma2(s, yMp1, np1); if (!odd_row) { fma2(s, yMm2, nm2); fma2(s, yMp2, np2); } return s; };
    auto sweepK = [&](D2 own) { D2 s = cKy * own; fma2(s, yKm1, nm1); fma2(s, yKp1, np1); if (!odd_row) { fma2(s, yKm2, nm2); fma2(s, yKp2, np2); } return s; };
    auto sweepO = [&]() { D2 s = yOm1 * nm1; fma2(s, yOp1, np1); if (!odd_row) { fma2(s, yOm2, nm2); fma2(s, yOp2, np2); } return s; };

    D2 XK[3], XM[3], XO[3], XD[3];
    { nb(0); const D2 My = sweepM(f[0]), Ky = sweepK(f[0]), Oy = sweepO(), Dy = cDy * f[0];   // mz_x
      XK[0] = l2g * My; XM[0] = G * Ky; XO[1] = c1 * Oy + c3 * Dy; XD[1] = c2 * Oy + c4 * Dy; }
    { nb(1); const D2 My = sweepM(f[1]), Ky = sweepK(f[1]), Oy = sweepO(), Dy = cDy * f[1];   // mz_y
      XK[1] = G * My; XM[1] = l2g * Ky; XO[0] = c1 * Oy + c2 * Dy; XD[0] = c3 * Oy + c4 * Dy; }
    { nb(2); const D2 My = sweepM(f[2]), Ky = sweepK(f[2]);                                    // mz_z
      XK[2] = G * My; XM[2] = G * Ky; }
    { nb(3); fma2(XM[0], G, sweepM(f[3])); }                                                   // kz_x
    { nb(4); fma2(XM[1], G, sweepM(f[4])); }                                                   // kz_y
    { nb(5); fma2(XM[2], l2g, sweepM(f[5])); }                                                 // kz_z
    { nb(6); const D2 My = sweepM(f[6]); XO[2] = c1 * My; XD[2] = c2 * My; }                   // oz_x
    { nb(7); const D2 Oy = sweepO(), Dy = cDy * f[7]; fma2(XM[2], c1, Oy); fma2(XM[2], c2, Dy); }   // oz_y
    { nb(8); const D2 My = sweepM(f[8]), Oy = sweepO(), Dy = cDy * f[8];                       // oz_z
      fma2(XO[0], c1, My); fma2(XD[0], c3, My); fma2(XM[1], c1, Oy); fma2(XM[1], c3, Dy); }
    if (has_w) {   // first / last plane: the boundary diagonal of C_z
      { nb(9); const D2 My = sweepM(wz[0]); fma2(XO[2], c3, My); fma2(XD[2], c4, My); }                                      // wz_x
      { nb(10); const D2 Oy = sweepO(), Dy = cDy * wz[1]; fma2(XM[2], c3, Oy); fma2(XM[2], c4, Dy); }                         // wz_y
      { nb(11); const D2 My = sweepM(wz[2]), Oy = sweepO(), Dy = cDy * wz[2];                                                 // wz_z
        fma2(XO[0], c2, My); fma2(XD[0], c4, My); fma2(XM[1], c2, Oy); fma2(XM[1], c4, Dy); }
    }

    // ---- x-stage, scatter form: own-pair terms + three messages per component through wave shuffles ----
    double ye[3], yo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const D2 FK = XK[c], FM = XM[c], FO = XO[c], FD = XD[c];
      double se = cKx * FK.e + EX.K[0][1] * FK.o + cMx * FM.e + EX.M[0][1] * FM.o + EX.C[0][1] * FO.o + cDx * FD.e;
      double so = EX.K[1][0] * FK.e + EX.K[1][1] * FK.o + EX.M[1][0] * FM.e + EX.M[1][1] * FM.o + EX.C[1][0] * FO.e + EX.C[1][1] * FD.o;
      const double toR = EX.K[2][0] * FK.e + EX.K[2][1] * FK.o + EX.M[2][0] * FM.e + EX.M[2][1] * FM.o + EX.C[2][0] * FO.e + EX.C[2][1] * FO.o;   // row i+2 (vertex), its left cell
      const double toLe = EX.K[0][2] * FK.e + EX.M[0][2] * FM.e + EX.C[0][2] * FO.e;   // row i-2 (vertex), its right cell
      const double toLo = EX.K[1][2] * FK.e + EX.M[1][2] * FM.e + EX.C[1][2] * FO.e;   // row i-1 (mid)
      se += __shfl_up(toR, 1, 32) + __shfl_down(toLe, 1, 32);
      so += __shfl_down(toLo, 1, 32);
      ye[c] = se; yo[c] = so;
    }
    const int64_t base = ((int64_t)kk * NY + j) * NX;
    const bool zb = kk == 0 || kk == NZ - 1;
    if (out_e) {
      const int64_t d0 = (base + ie) * 3;
      if (a.constrained && (a.mask_anywhere || zb || bnd_xy_e)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (a.mask[d0 + c]) ye[c] = a.diag_local[d0 + c] * x[d0 + c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) y[d0 + c] = ye[c];
    }
    if (out_o) {
      const int64_t d0 = (base + io) * 3;
      if (a.constrained && (a.mask_anywhere || zb || bnd_xy_o)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) if (a.mask[d0 + c]) yo[c] = a.diag_local[d0 + c] * x[d0 + c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) y[d0 + c] = yo[c];
    }
  };

  for (int k = k0; k < k1; k += 2) {
    load_plane(k + 3, P0); load_plane(k + 4, P1);       // prefetch the next pair while this one computes
    plane(k, false);
    if (k + 1 < k1) plane(k + 1, true);
#pragma unroll
    for (int c = 0; c < 6; ++c) { W0[c] = W2[c]; W1[c] = W3[c]; W2[c] = W4[c]; W3[c] = P0[c]; W4[c] = P1[c]; }
  }
}

// 1D element matrices of FE_Q(2) on a cell of length h: M = h int phi_a phi_b, K = (1/h) int phi_a' phi_b', C = int phi_a' phi_b
Kron1D make_1d(double h) {
  // 4-point Gauss-Legendre on [0,1] (exact to degree 7)
  const double gx[4] = {0.0694318442029737124, 0.3300094782075718676, 0.6699905217924281324, 0.9305681557970262876};
  const double gw[4] = {0.1739274225687269287, 0.3260725774312730713, 0.3260725774312730713, 0.1739274225687269287};
  Kron1D e{};
  for (int q = 0; q < 4; ++q) {
    const double t = gx[q];
    const double v[3] = {2 * (t - 0.5) * (t - 1), 4 * t * (1 - t), 2 * t * (t - 0.5)}, d[3] = {4 * t - 3, 4 - 8 * t, 4 * t - 1};
    for (int i = 0; i < 3; ++i) for (int jj = 0; jj < 3; ++jj) { e.M[i][jj] += h * gw[q] * v[i] * v[jj]; e.K[i][jj] += gw[q] * d[i] * d[jj] / h; e.C[i][jj] += gw[q] * d[i] * v[jj]; }
  }
  return e;
}

}  // namespace

bool kron_supported(int dim, int k_u) { return dim == 3 && k_u == 2; }

void kron_apply(hipStream_t s, const MfArgs &m, const double *x, double *y, bool constrained, int n_cus) {
  KronArgs a{};
  for (int d = 0; d < 3; ++d) { a.nn[d] = 2 * m.box.n[d] + 1; a.e[d] = make_1d(m.box.h[d]); }
  a.ntx = (a.nn[0] + VX - 1) / VX; a.nty = (a.nn[1] + VY - 1) / VY;
  // z-chunks: enough workgroups to fill the chip (one 512-thread workgroup per CU), each chunk an even number of planes
  const int cols = a.ntx * a.nty;
  int nzc = (n_cus + cols - 1) / cols; if (nzc < 1) nzc = 1;
  int chunk = (a.nn[2] + nzc - 1) / nzc; chunk += chunk & 1; if (chunk < 8) chunk = 8;
  a.chunk = chunk; a.nzc = (a.nn[2] + chunk - 1) / chunk;
  a.lam = m.lam; a.G = m.G; a.mask = m.mask; a.diag_local = m.diag_local; a.constrained = constrained ? 1 : 0; a.mask_anywhere = m.mask_anywhere;
  const size_t lds = (size_t)NFLD * TYR * TXL * sizeof(D2);
  static bool attr_set = false;
  if (!attr_set) { PORO_HIP(hipFuncSetAttribute((const void *)k_kron3_q2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set = true; }
  hipLaunchKernelGGL(k_kron3_q2, (unsigned)(a.ntx * a.nty * a.nzc), 512, lds, s, a, x, y);
}

}  // namespace poro
