// K-apply-u for GENERAL meshes (Gmsh quads, graded / locally refined boxes): the matrix-free cell loop of
// PoroElasticDisplacementSolver::assemble_system (:206-246) applied to a vector instead of scattered into a matrix.
//   y_i = sum_K sum_q (C : eps(u_h)(x_q)) : eps(phi_i)(x_q) JxW_q,   eps(u_h) from the cell's dof values (get_function_gradients form)
// One wavefront per cell.  The cell's dof indices, dof values, vertex coordinates, J^-1 / JxW and the stress at every quadrature point are
// staged in LDS; phase 1 (one lane per quadrature point) evaluates the displacement gradient through the reference shape gradients and
// MappingQ1's J^-1, forms sigma = lambda tr(eps) I + 2 G eps (the isotropic Gassmann tensor, ConstitutiveModel.h:45-57) times JxW;
// phase 2 (one lane per cell dof) tests it with grad phi_i.  2 n_q n_s dim^2 flops per cell and phase instead of the dpc^2 of an element
// matrix, no element matrix in memory.  Cells are processed colour by colour (no two cells of a colour share a dof), so the scatter is
// a plain read-modify-write: no atomics, bitwise reproducible.  Dirichlet columns are masked on load; the Dirichlet ROWS are left to the
// caller (inert inside PCG; poro_apply_operator finishes them from the constraint list).
#include "common.hpp"
#include <cmath>
#include <cstdlib>

namespace poro {
namespace {

template <int DIM> __device__ inline double jac_inv(const double *X, const double *dN, double *Ji) {
  constexpr int NV = 1 << DIM;
  double J[DIM][DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a)
#pragma unroll
    for (int b = 0; b < DIM; ++b) { double s = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) s += X[v * DIM + a] * dN[v * DIM + b];
      J[a][b] = s; }
  if constexpr (DIM == 2) {
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0], id = 1.0 / det;
    Ji[0] = J[1][1] * id; Ji[1] = -J[0][1] * id; Ji[2] = -J[1][0] * id; Ji[3] = J[0][0] * id;
    return det;
  } else {
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
    Ji[0] = c00 * id; Ji[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; Ji[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    Ji[3] = c01 * id; Ji[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; Ji[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    Ji[6] = c02 * id; Ji[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; Ji[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    return det;
  }
}

constexpr int kMaxNq = 27, kMaxDpc = 81;

// mode 0: y += A x (cells of one colour); mode 1: y += diag(A)
template <int DIM> __global__ void __launch_bounds__(64)
k_mfg(AsmArgs a, const int32_t *__restrict__ cells, const double *__restrict__ x, double *__restrict__ y, int constrained, int mode) {
  constexpr int NV = 1 << DIM;
  __shared__ double sX[NV * DIM], sU[kMaxDpc], sJi[kMaxNq * DIM * DIM], sJxW[kMaxNq], sS[kMaxNq * DIM * DIM];
  __shared__ int32_t sDof[kMaxDpc];
  __shared__ uint8_t sDir[kMaxDpc];
  const int tid = threadIdx.x;
  const int64_t cell = cells[blockIdx.x];
  const int nq = a.fe.nq_u, ns = a.ns_u, dpc = a.dpc_u;
  const double lam = a.mat.lame_lambda, G = a.mat.shear_G;
  for (int i = tid; i < NV * DIM; i += 64) sX[i] = a.cell_X[cell * NV * DIM + i];
  for (int i = tid; i < dpc; i += 64) {
    const int32_t dof = a.cell_dofs_u[cell * dpc + i];
    const uint8_t m = constrained ? a.dir_mask[dof] : (uint8_t)0;
    sDof[i] = dof; sDir[i] = m; sU[i] = (mode == 0 && !m) ? x[dof] : 0.0;
  }
  __syncthreads();
  for (int q = tid; q < nq; q += 64) {
    double *Ji = sJi + q * DIM * DIM;
    const double det = jac_inv<DIM>(sX, a.fe.dq1_qu + (size_t)q * NV * DIM, Ji);
    const double jxw = det * a.fe.w_qu[q];
    sJxW[q] = jxw;
    if (mode == 0) {
      double R[DIM][DIM];                       // reference-space gradient of u_h: R[c][b] = sum_s u[s, c] d phi_s / d xi_b
#pragma unroll
      for (int c = 0; c < DIM; ++c)
#pragma unroll
        for (int b = 0; b < DIM; ++b) R[c][b] = 0;
      for (int s = 0; s < ns; ++s) {
        const double *gr = a.fe.du_qu + (size_t)(q * ns + s) * DIM;
#pragma unroll
        for (int c = 0; c < DIM; ++c) { const double us = sU[s * DIM + c];
#pragma unroll
          for (int b = 0; b < DIM; ++b) R[c][b] = fma(us, gr[b], R[c][b]); }
      }
      double g[DIM][DIM], tr = 0;              // g[c][d] = d u_c / d x_d = sum_b R[c][b] Ji[b][d]
#pragma unroll
      for (int c = 0; c < DIM; ++c)
#pragma unroll
        for (int d = 0; d < DIM; ++d) { double t = 0;
#pragma unroll
          for (int b = 0; b < DIM; ++b) t = fma(R[c][b], Ji[b * DIM + d], t);
          g[c][d] = t; }
#pragma unroll
      for (int c = 0; c < DIM; ++c) tr += g[c][c];
#pragma unroll
      for (int c = 0; c < DIM; ++c)
#pragma unroll
        for (int d = 0; d < DIM; ++d) sS[(q * DIM + c) * DIM + d] = jxw * (G * (g[c][d] + g[d][c]) + (c == d ? lam * tr : 0.0));
    }
  }
  __syncthreads();
  for (int i = tid; i < dpc; i += 64) {
    if (sDir[i]) continue;
    const int s = i / DIM, c = i % DIM;
    double acc = 0;
    for (int q = 0; q < nq; ++q) {
      const double *gr = a.fe.du_qu + (size_t)(q * ns + s) * DIM, *Ji = sJi + q * DIM * DIM;
      double gx[DIM];                           // grad phi_s in real space
#pragma unroll
      for (int d = 0; d < DIM; ++d) { double t = 0;
#pragma unroll
        for (int b = 0; b < DIM; ++b) t = fma(Ji[b * DIM + d], gr[b], t);
        gx[d] = t; }
      if (mode == 0) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) acc = fma(sS[(q * DIM + c) * DIM + d], gx[d], acc);
      } else {                                  // (C : eps(phi_i)) : eps(phi_i) = lambda g_c^2 + G (|g|^2 + g_c^2)
        double n2 = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) n2 = fma(gx[d], gx[d], n2);
        acc = fma(sJxW[q], lam * gx[c] * gx[c] + G * (n2 + gx[c] * gx[c]), acc);
      }
    }
    y[sDof[i]] += acc;
  }
}

// ---- sum-factorised form for hexahedra (3D, Q1 / Q2): the kernel the large general meshes run --------------------------------------------------------
// The element above spends 2 n_q n_s dim^2 flops per phase and keeps 27 of 64 lanes busy on a Q2 hexahedron.  FE_Q(k) is a tensor product, so the gradient of u_h
// at the (k+1)^3 Gauss points follows from three sweeps with the (k+1) x (k+1) 1D value / derivative tables (and the test with grad phi_i from the transposed
// sweeps): ~300 FMAs per point instead of ~1200, and one lane per point with SEVERAL cells per workgroup: 8 Q2 cells (32 Q1 cells) share a 256-thread workgroup,
// a cell's points live in one wave.  Per cell: dof indices and values, vertex coordinates staged once; J^-1 from MappingQ1 at the lane's own point.
struct Sf1D { double N[3][3], D[3][3], w[3], xi[3]; };   // [quadrature point][node]: values / derivatives of the 1D Lagrange basis (equidistant nodes on [0, 1]) at the Gauss points

// every exchange of the sum-factorised kernel stays inside one cell's lanes, i.e. inside one wavefront (32 or 8 lanes per cell): LDS operations of a wave complete in
// issue order, so a compiler-level fence replaces the workgroup barrier and the four waves of a workgroup drift apart freely (645 -> 626 us at 72^3 cells)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
template <int N1, bool AFFINE>
__global__ void __launch_bounds__(256)
k_mfg3_sf(AsmArgs a, Sf1D T, const int32_t *__restrict__ cells, int n_cells, const double *__restrict__ x, double *__restrict__ y, int constrained) {
  constexpr int NP = N1 * N1 * N1, LPC = N1 == 3 ? 32 : 8, CPW = 256 / LPC;
  __shared__ double sU[CPW][3][NP], sA[CPW][6][NP], sB[CPW][9][NP], sX[CPW][24];
  const int tid = threadIdx.x, cs = tid / LPC, p = tid - cs * LPC;
  const int64_t slot = (int64_t)blockIdx.x * CPW + cs;
  const bool live = p < NP && slot < n_cells;
  const int64_t cell = slot < n_cells ? cells[slot] : cells[0];
  const int i = p % N1, j = (p / N1) % N1, k = p / (N1 * N1);
  const double lam = a.mat.lame_lambda, G = a.mat.shear_G;
  int32_t dof[3] = {0, 0, 0}; bool dir[3] = {true, true, true};
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dof[c] = a.cell_dofs_u[cell * (3 * NP) + p * 3 + c];
      dir[c] = constrained && a.dir_mask[dof[c]];
      sU[cs][c][p] = dir[c] ? 0.0 : x[dof[c]];
    }
  }
  if constexpr (AFFINE) { if (slot < n_cells) for (int e = p; e < 10; e += LPC) sX[cs][e] = a.cell_geo[cell * 10 + e]; }
  else if (slot < n_cells) for (int e = p; e < 24; e += LPC) sX[cs][e] = a.cell_X[cell * 24 + e];
  wave_sync();
  double R[3][3];                                         // reference-space gradient of u_h at the lane's quadrature point
  {
    // sweep along xi: thread (qi = i, j, k)
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { double vN = 0, vD = 0;
#pragma unroll
        for (int m = 0; m < N1; ++m) { const double u = sU[cs][c][m + N1 * (j + N1 * k)]; vN = fma(T.N[i][m], u, vN); vD = fma(T.D[i][m], u, vD); }
        sA[cs][2 * c][p] = vN; sA[cs][2 * c + 1][p] = vD; }
    }
    wave_sync();
    // sweep along eta: thread (qi, qj = j, k)
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { double nn = 0, dn = 0, nd = 0;
#pragma unroll
        for (int m = 0; m < N1; ++m) { const double vN = sA[cs][2 * c][i + N1 * (m + N1 * k)], vD = sA[cs][2 * c + 1][i + N1 * (m + N1 * k)];
          nn = fma(T.N[j][m], vN, nn); dn = fma(T.N[j][m], vD, dn); nd = fma(T.D[j][m], vN, nd); }
        sB[cs][3 * c][p] = nn; sB[cs][3 * c + 1][p] = dn; sB[cs][3 * c + 2][p] = nd; }
    }
    wave_sync();
    // sweep along zeta: thread (qi, qj, qk = k)
#pragma unroll
    for (int c = 0; c < 3; ++c) { double gx = 0, gy = 0, gz = 0;
      if (live) {
#pragma unroll
        for (int m = 0; m < N1; ++m) { const int at = i + N1 * (j + N1 * m);
          gx = fma(T.N[k][m], sB[cs][3 * c + 1][at], gx); gy = fma(T.N[k][m], sB[cs][3 * c + 2][at], gy); gz = fma(T.D[k][m], sB[cs][3 * c][at], gz); }
      }
      R[c][0] = gx; R[c][1] = gy; R[c][2] = gz; }
  }
  // MappingQ1 at the point: J[a][b] = sum_v X_v[a] dN_v / dxi_b, N_v trilinear (vertices lexicographic)
  double S[3][3];                                         // reference-space flux: S[c][b] = sum_d sigma[c][d] Jinv[b][d] * JxW
  {
    double Ji[3][3], det;                                 // Ji[b][d] = d xi_b / d x_d
    if constexpr (AFFINE) {                               // one Jacobian per cell, inverted at set-up
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int d = 0; d < 3; ++d) Ji[b][d] = live ? sX[cs][3 * b + d] : 0.0;
      det = sX[cs][9];
    } else {
    const double lx[2] = {1.0 - T.xi[i], T.xi[i]}, ly[2] = {1.0 - T.xi[j], T.xi[j]}, lz[2] = {1.0 - T.xi[k], T.xi[k]}, dl[2] = {-1.0, 1.0};
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int vi = v & 1, vj = (v >> 1) & 1, vk = v >> 2;
      const double d0 = dl[vi] * ly[vj] * lz[vk], d1 = lx[vi] * dl[vj] * lz[vk], d2 = lx[vi] * ly[vj] * dl[vk];
#pragma unroll
      for (int r = 0; r < 3; ++r) { const double X = sX[cs][v * 3 + r]; J[r][0] = fma(X, d0, J[r][0]); J[r][1] = fma(X, d1, J[r][1]); J[r][2] = fma(X, d2, J[r][2]); }
    }
    const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02; const double id = live ? 1.0 / det : 0.0;
    Ji[0][0] = c00 * id; Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    Ji[1][0] = c01 * id; Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    Ji[2][0] = c02 * id; Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    }
    const double jxw = det * T.w[i] * T.w[j] * T.w[k];
    double g[3][3], tr = 0;                               // g[c][d] = d u_c / d x_d
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) { double t = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b) t = fma(R[c][b], Ji[b][d], t);
        g[c][d] = t; }
    tr = g[0][0] + g[1][1] + g[2][2];
    double sg[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) sg[c][d] = jxw * (G * (g[c][d] + g[d][c]) + (c == d ? lam * tr : 0.0));   // isotropic Gassmann tensor (ConstitutiveModel.h:45-57) times JxW
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int b = 0; b < 3; ++b) { double t = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) t = fma(sg[c][d], Ji[b][d], t);
        S[c][b] = t; }
  }
  // transposed sweeps: y_c(a, b, cc) = sum_q [D(qi,a) N(qj,b) N(qk,cc) S_c0 + N(qi,a) D(qj,b) N(qk,cc) S_c1 + N(qi,a) N(qj,b) D(qk,cc) S_c2]
  wave_sync();
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int b = 0; b < 3; ++b) sB[cs][3 * c + b][p] = S[c][b];
  }
  wave_sync();
  double E[3][3];
  if (live) {                                             // contract qk -> node cc = k: thread (qi, qj, cc)
#pragma unroll
    for (int c = 0; c < 3; ++c) { double e0 = 0, e1 = 0, e2 = 0;
#pragma unroll
      for (int m = 0; m < N1; ++m) { const int at = i + N1 * (j + N1 * m);
        e0 = fma(T.N[m][k], sB[cs][3 * c][at], e0); e1 = fma(T.N[m][k], sB[cs][3 * c + 1][at], e1); e2 = fma(T.D[m][k], sB[cs][3 * c + 2][at], e2); }
      E[c][0] = e0; E[c][1] = e1; E[c][2] = e2; }
  }
  wave_sync();
  if (live) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int b = 0; b < 3; ++b) sB[cs][3 * c + b][p] = E[c][b];
  }
  wave_sync();
  if (live) {                                             // contract qj -> node b = j: thread (qi, b, cc)
#pragma unroll
    for (int c = 0; c < 3; ++c) { double f0 = 0, f1 = 0;
#pragma unroll
      for (int m = 0; m < N1; ++m) { const int at = i + N1 * (m + N1 * k);
        f0 = fma(T.N[m][j], sB[cs][3 * c][at], f0); f1 = fma(T.D[m][j], sB[cs][3 * c + 1][at], f1); f1 = fma(T.N[m][j], sB[cs][3 * c + 2][at], f1); }
      sA[cs][2 * c][p] = f0; sA[cs][2 * c + 1][p] = f1; }
  }
  wave_sync();
  if (live) {                                             // contract qi -> node a = i: thread (a, b, cc) = node p; coloured scatter (no two cells of a colour share a dof)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = 0;
#pragma unroll
      for (int m = 0; m < N1; ++m) { const int at = m + N1 * (j + N1 * k); v = fma(T.D[m][i], sA[cs][2 * c][at], v); v = fma(T.N[m][i], sA[cs][2 * c + 1][at], v); }
      if (!dir[c]) y[dof[c]] += v;
    }
  }
}


// ---- the same for quadrilaterals (2D, Q1 / Q2): 16 lanes per Q2 cell (9 points), 4 per Q1 cell; 16 / 64 cells per workgroup -----------------------------------------
template <int N1>
__global__ void __launch_bounds__(256)
k_mfg2_sf(AsmArgs a, Sf1D T, const int32_t *__restrict__ cells, int n_cells, const double *__restrict__ x, double *__restrict__ y, int constrained) {
  constexpr int NP = N1 * N1, LPC = N1 == 3 ? 16 : 4, CPW = 256 / LPC;
  __shared__ double sU[CPW][2][NP], sA[CPW][4][NP], sX[CPW][8];
  const int tid = threadIdx.x, cs = tid / LPC, p = tid - cs * LPC;
  const int64_t slot = (int64_t)blockIdx.x * CPW + cs;
  const bool live = p < NP && slot < n_cells;
  const int64_t cell = slot < n_cells ? cells[slot] : cells[0];
  const int i = p % N1, j = (p / N1) % N1;
  const double lam = a.mat.lame_lambda, G = a.mat.shear_G;
  int32_t dof[2] = {0, 0}; bool dir[2] = {true, true};
  if (live) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      dof[c] = a.cell_dofs_u[cell * (2 * NP) + p * 2 + c];
      dir[c] = constrained && a.dir_mask[dof[c]];
      sU[cs][c][p] = dir[c] ? 0.0 : x[dof[c]];
    }
  }
  if (slot < n_cells) for (int e = p; e < 8; e += LPC) sX[cs][e] = a.cell_X[cell * 8 + e];
  wave_sync();
  if (live) {                                             // sweep along xi: thread (qi = i, j)
#pragma unroll
    for (int c = 0; c < 2; ++c) { double vN = 0, vD = 0;
#pragma unroll
      for (int m = 0; m < N1; ++m) { const double u = sU[cs][c][m + N1 * j]; vN = fma(T.N[i][m], u, vN); vD = fma(T.D[i][m], u, vD); }
      sA[cs][2 * c][p] = vN; sA[cs][2 * c + 1][p] = vD; }
  }
  wave_sync();
  double R[2][2];                                         // reference-space gradient at the lane's quadrature point (qi = i, qj = j)
#pragma unroll
  for (int c = 0; c < 2; ++c) { double gx = 0, gy = 0;
    if (live) {
#pragma unroll
      for (int m = 0; m < N1; ++m) { gx = fma(T.N[j][m], sA[cs][2 * c + 1][i + N1 * m], gx); gy = fma(T.D[j][m], sA[cs][2 * c][i + N1 * m], gy); }
    }
    R[c][0] = gx; R[c][1] = gy; }
  double S[2][2];                                         // reference-space flux: S[c][b] = sum_d sigma[c][d] Jinv[b][d] * JxW
  {
    const double lx[2] = {1.0 - T.xi[i], T.xi[i]}, ly[2] = {1.0 - T.xi[j], T.xi[j]}, dl[2] = {-1.0, 1.0};
    double J[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int vi = v & 1, vj = v >> 1;
      const double d0 = dl[vi] * ly[vj], d1 = lx[vi] * dl[vj];
#pragma unroll
      for (int r = 0; r < 2; ++r) { const double X = sX[cs][v * 2 + r]; J[r][0] = fma(X, d0, J[r][0]); J[r][1] = fma(X, d1, J[r][1]); }
    }
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0], id = live ? 1.0 / det : 0.0;
    const double Ji[2][2] = {{J[1][1] * id, -J[0][1] * id}, {-J[1][0] * id, J[0][0] * id}};   // Ji[b][d] = d xi_b / d x_d
    const double jxw = det * T.w[i] * T.w[j];
    double g[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int d = 0; d < 2; ++d) g[c][d] = fma(R[c][0], Ji[0][d], R[c][1] * Ji[1][d]);
    const double tr = g[0][0] + g[1][1];
    double sg[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int d = 0; d < 2; ++d) sg[c][d] = jxw * (G * (g[c][d] + g[d][c]) + (c == d ? lam * tr : 0.0));
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int b = 0; b < 2; ++b) S[c][b] = fma(sg[c][0], Ji[b][0], sg[c][1] * Ji[b][1]);
  }
  wave_sync();
  if (live) {
#pragma unroll
    for (int c = 0; c < 2; ++c) { sA[cs][2 * c][p] = S[c][0]; sA[cs][2 * c + 1][p] = S[c][1]; }
  }
  wave_sync();
  double F[2][2];
  if (live) {                                             // contract qj -> node b = j: thread (qi, b)
#pragma unroll
    for (int c = 0; c < 2; ++c) { double f0 = 0, f1 = 0;
#pragma unroll
      for (int m = 0; m < N1; ++m) { f0 = fma(T.N[m][j], sA[cs][2 * c][i + N1 * m], f0); f1 = fma(T.D[m][j], sA[cs][2 * c + 1][i + N1 * m], f1); }
      F[c][0] = f0; F[c][1] = f1; }
  }
  wave_sync();
  if (live) {
#pragma unroll
    for (int c = 0; c < 2; ++c) { sA[cs][2 * c][p] = F[c][0]; sA[cs][2 * c + 1][p] = F[c][1]; }
  }
  wave_sync();
  if (live) {                                             // contract qi -> node a = i; coloured scatter
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double v = 0;
#pragma unroll
      for (int m = 0; m < N1; ++m) { v = fma(T.D[m][i], sA[cs][2 * c][m + N1 * j], v); v = fma(T.N[m][i], sA[cs][2 * c + 1][m + N1 * j], v); }
      if (!dir[c]) y[dof[c]] += v;
    }
  }
}

Sf1D sf_tables(int k) {
  Sf1D T{};
  const int n1 = k + 1;
  if (k == 1) { const double g = 0.5 / std::sqrt(3.0); T.xi[0] = 0.5 - g; T.xi[1] = 0.5 + g; T.w[0] = T.w[1] = 0.5; }
  else { const double g = 0.5 * std::sqrt(0.6); T.xi[0] = 0.5 - g; T.xi[1] = 0.5; T.xi[2] = 0.5 + g; T.w[0] = T.w[2] = 5.0 / 18.0; T.w[1] = 8.0 / 18.0; }
  for (int q = 0; q < n1; ++q) {
    const double t = T.xi[q];
    if (k == 1) { T.N[q][0] = 1 - t; T.N[q][1] = t; T.D[q][0] = -1; T.D[q][1] = 1; }
    else { T.N[q][0] = 2 * (t - 0.5) * (t - 1); T.N[q][1] = 4 * t * (1 - t); T.N[q][2] = 2 * t * (t - 0.5); T.D[q][0] = 4 * t - 3; T.D[q][1] = 4 - 8 * t; T.D[q][2] = 4 * t - 1; }
  }
  return T;
}

}  // namespace

// y = A_u x (mode 0) or y = diag(A_u) (mode 1) over the colour classes; y is zeroed here
void mfg_apply(hipStream_t s, const AsmArgs &a, const int32_t *color_cells, const std::vector<int64_t> &color_off, int64_t n_u, const double *x, double *y, bool constrained, int mode) {
  if (a.fe.nq_u > kMaxNq || a.dpc_u > kMaxDpc) throw Error("mfg_apply: element too large");
  PORO_HIP(hipMemsetAsync(y, 0, n_u * sizeof(double), s));
  static const bool no_sf = std::getenv("PORO_MFG_NO_SUMFAC") != nullptr;
  const bool sf = (a.dim == 3 || a.dim == 2) && mode == 0 && !no_sf && (a.k_u == 1 || a.k_u == 2);
  const Sf1D T = sf ? sf_tables(a.k_u) : Sf1D{};
  for (size_t k = 0; k + 1 < color_off.size(); ++k) {
    const int64_t nc = color_off[k + 1] - color_off[k];
    if (!nc) continue;
    if (sf && a.dim == 2) {
      const int cpw = a.k_u == 2 ? 16 : 64;
      if (a.k_u == 2) hipLaunchKernelGGL(k_mfg2_sf<3>, (unsigned)((nc + cpw - 1) / cpw), 256, 0, s, a, T, color_cells + color_off[k], (int)nc, x, y, constrained ? 1 : 0);
      else hipLaunchKernelGGL(k_mfg2_sf<2>, (unsigned)((nc + cpw - 1) / cpw), 256, 0, s, a, T, color_cells + color_off[k], (int)nc, x, y, constrained ? 1 : 0);
      continue;
    }
    if (sf) {
      const int cpw = a.k_u == 2 ? 8 : 32; const unsigned grid = (unsigned)((nc + cpw - 1) / cpw);
      if (a.k_u == 2 && a.cell_geo) hipLaunchKernelGGL((k_mfg3_sf<3, true>), grid, 256, 0, s, a, T, color_cells + color_off[k], (int)nc, x, y, constrained ? 1 : 0);
      else if (a.k_u == 2) hipLaunchKernelGGL((k_mfg3_sf<3, false>), grid, 256, 0, s, a, T, color_cells + color_off[k], (int)nc, x, y, constrained ? 1 : 0);
      else if (a.cell_geo) hipLaunchKernelGGL((k_mfg3_sf<2, true>), grid, 256, 0, s, a, T, color_cells + color_off[k], (int)nc, x, y, constrained ? 1 : 0);
      else hipLaunchKernelGGL((k_mfg3_sf<2, false>), grid, 256, 0, s, a, T, color_cells + color_off[k], (int)nc, x, y, constrained ? 1 : 0);
      continue;
    }
    if (a.dim == 2) hipLaunchKernelGGL(k_mfg<2>, (unsigned)nc, 64, 0, s, a, color_cells + color_off[k], x, y, constrained ? 1 : 0, mode);
    else hipLaunchKernelGGL(k_mfg<3>, (unsigned)nc, 64, 0, s, a, color_cells + color_off[k], x, y, constrained ? 1 : 0, mode);
  }
}

}  // namespace poro
