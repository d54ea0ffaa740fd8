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

}  // namespace

// y = A_u x (mode 0) or y = diag(A_u) (mode 1) over the colour classes; y is zeroed here
void mfg_apply(hipStream_t s, const AsmArgs &a, const int32_t *color_cells, const std::vector<int64_t> &color_off, int64_t n_u, const double *x, double *y, bool constrained, int mode) {
  if (a.fe.nq_u > kMaxNq || a.dpc_u > kMaxDpc) throw Error("mfg_apply: element too large");
  PORO_HIP(hipMemsetAsync(y, 0, n_u * sizeof(double), s));
  for (size_t k = 0; k + 1 < color_off.size(); ++k) {
    const int64_t nc = color_off[k + 1] - color_off[k];
    if (!nc) continue;
    if (a.dim == 2) hipLaunchKernelGGL(k_mfg<2>, (unsigned)nc, 64, 0, s, a, color_cells + color_off[k], x, y, constrained ? 1 : 0, mode);
    else hipLaunchKernelGGL(k_mfg<3>, (unsigned)nc, 64, 0, s, a, color_cells + color_off[k], x, y, constrained ? 1 : 0, mode);
  }
}

}  // namespace poro
