// Fast-diagonalisation preconditioner (K-prec, SURVEY 8f-1 "a stronger preconditioner") for tensor-product operators on uniform boxes.
//
// On a box every operator of the path that CG has to invert is (block-wise) a short sum of Kronecker products of 1D matrices:
//   pressure Jacobian      J = a Mx(x)My(x)Mz + kappa (Kx(x)My(x)Mz + Mx(x)Ky(x)Mz + Mx(x)My(x)Kz)     (PoroElasticPressureSolver.h:158-169)
//   projection mass matrix M = Mx(x)My(x)Mz                                                            (StrainProjector.h:101-106)
// With the generalised 1D eigen-decompositions K_d S_d = M_d S_d Lambda_d, S_d^T M_d S_d = I (computed once per mesh on the host),
//   (a M + sum_d k_d K_d-terms)^-1 = (Sx(x)Sy(x)Sz) diag(a + k_x lam_x[i] + k_y lam_y[j] + k_z lam_z[k])^-1 (Sx(x)Sy(x)Sz)^T :
// an EXACT inverse applied as 2*dim batched dense (n_d x n_d) transforms along the grid lines.  That is GEMM-shaped fp64 work, so it
// runs on the matrix cores (v_mfma_f64_16x16x4_f64): one wavefront owns a 16 x 64 output tile, operands come straight from L2 (the
// whole pressure vector is 3 MB, S_d is 42 KB at BASELINE config 4), no LDS staging is needed at these sizes.
// Used as the preconditioner inside SolverCG's recurrence, so the stopping rule and the converged solution are those of the
// reference's cg.solve; CG then needs 1-2 iterations instead of ~150 with Jacobi.
#include "common.hpp"

namespace poro {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

// out(l', r) = sum_l T[l'][l] in(l, r) along a strided direction: element (l, r) lives at (r % SI) + l * SI + (r / SI) * SI * n_l.
// MFMA operand layout (f64 16x16x4): A[i][k] and B[k][j] one value per lane with i|j = lane & 15, k = lane >> 4;
// D: register q of a lane holds row (lane >> 4) + 4 q, column lane & 15.  Grid: x = 64-line groups, y = 16-row groups of l'.
__global__ void __launch_bounds__(64)
k_fdm_dir(const double *__restrict__ T, int n_l, int64_t SI, int64_t nr, const double *__restrict__ in, double *__restrict__ out, FdmScale sc, int has_scale) {
  const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
  const int row0 = blockIdx.y * 16;
  const int64_t col0 = (int64_t)blockIdx.x * 64;
  int64_t boff[4]; bool bval[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t r = col0 + 16 * c + i;
    bval[c] = r < nr;
    const int64_t outer = r / SI, inner = r - outer * SI;
    boff[c] = inner + outer * SI * n_l;
  }
  const bool aval = row0 + i < n_l;
  const double *Arow = T + (int64_t)(aval ? row0 + i : 0) * n_l;
  v4d acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = v4d{0, 0, 0, 0};
  for (int l0 = 0; l0 < n_l; l0 += 4) {
    const int l = l0 + kq; const bool lv = l < n_l;
    const double a = (aval && lv) ? Arow[l] : 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double b = (bval[c] && lv) ? in[boff[c] + (int64_t)l * SI] : 0.0;
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (!bval[c]) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = row0 + kq + 4 * q;
      if (row >= n_l) continue;
      const int64_t addr = boff[c] + (int64_t)row * SI;
      double v = acc[c][q];
      if (has_scale) {
        const int64_t n01 = (int64_t)sc.n[0] * sc.n[1];
        const int kk = (int)(addr / n01); const int64_t rem = addr - (int64_t)kk * n01; const int jj = (int)(rem / sc.n[0]), ii = (int)(rem - (int64_t)jj * sc.n[0]);
        double D = sc.a + sc.k[0] * sc.lam[0][ii] + sc.k[1] * sc.lam[1][jj];
        if (sc.lam[2]) D += sc.k[2] * sc.lam[2][kk];
        v /= D;
      }
      out[addr] = v;
    }
  }
}

// the same along the contiguous (x) direction: element (l, r) at l + r * n_l.  Here the lines are the MFMA rows:
// out[r][l'] = sum_l in[r][l] T[l'][l].  Grid: x = 16-line groups, y = 64-column groups of l'.
__global__ void __launch_bounds__(64)
k_fdm_x(const double *__restrict__ T, int n_l, int64_t nr, const double *__restrict__ in, double *__restrict__ out) {
  const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int c0 = blockIdx.y * 64;
  const bool aval = r0 + i < nr;
  const double *Arow = in + (aval ? r0 + i : 0) * n_l;
  const double *Brow[4]; bool bval[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { const int lp = c0 + 16 * c + i; bval[c] = lp < n_l; Brow[c] = T + (int64_t)(bval[c] ? lp : 0) * n_l; }
  v4d acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = v4d{0, 0, 0, 0};
  for (int l0 = 0; l0 < n_l; l0 += 4) {
    const int l = l0 + kq; const bool lv = l < n_l;
    const double a = (aval && lv) ? Arow[l] : 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double b = (bval[c] && lv) ? Brow[c][l] : 0.0;
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int lp = c0 + 16 * c + i;
    if (lp >= n_l) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = r0 + kq + 4 * q;
      if (r < nr) out[r * n_l + lp] = acc[c][q];
    }
  }
}

}  // namespace

void fdm_transform(hipStream_t s, const double *T, int n_l, int64_t SI, int64_t n_outer, const double *in, double *out, const FdmScale *scale) {
  const int64_t nr = SI * n_outer;
  if (!nr || !n_l) return;
  if (SI == 1 && !scale) {
    hipLaunchKernelGGL(k_fdm_x, dim3((unsigned)((nr + 15) / 16), (unsigned)((n_l + 63) / 64)), 64, 0, s, T, n_l, nr, in, out);
  } else {
    FdmScale sc{}; if (scale) sc = *scale;
    hipLaunchKernelGGL(k_fdm_dir, dim3((unsigned)((nr + 63) / 64), (unsigned)((n_l + 15) / 16)), 64, 0, s, T, n_l, SI, nr, in, out, sc, scale ? 1 : 0);
  }
}

// z = (S0 (x) S1 (x) S2) D^-1 (S0 (x) S1 (x) S2)^T g on an n0 x n1 (x n2) grid, x fastest.  t1, t2: scratch vectors of the grid size.
void fdm_apply(hipStream_t s, const FdmScalar &F, double a, const double k[3], const double *g, double *z, double *t1, double *t2) {
  const int dim = F.dim; const int n0 = F.dir[0].n, n1 = F.dir[1].n, n2 = dim == 3 ? F.dir[2].n : 1;
  FdmScale sc{}; sc.a = a;
  for (int d = 0; d < 3; ++d) { sc.lam[d] = d < dim ? F.dir[d].lam.p : nullptr; sc.k[d] = d < dim ? k[d] : 0.0; sc.n[d] = d < dim ? F.dir[d].n : 1; }
  if (dim == 2) {
    fdm_transform(s, F.dir[0].St.p, n0, 1, n1, g, t1, nullptr);
    fdm_transform(s, F.dir[1].St.p, n1, n0, 1, t1, t2, &sc);
    fdm_transform(s, F.dir[1].S.p, n1, n0, 1, t2, t1, nullptr);
    fdm_transform(s, F.dir[0].S.p, n0, 1, n1, t1, z, nullptr);
  } else {
    fdm_transform(s, F.dir[0].St.p, n0, 1, (int64_t)n1 * n2, g, t1, nullptr);
    fdm_transform(s, F.dir[1].St.p, n1, n0, n2, t1, t2, nullptr);
    fdm_transform(s, F.dir[2].St.p, n2, (int64_t)n0 * n1, 1, t2, t1, &sc);
    fdm_transform(s, F.dir[2].S.p, n2, (int64_t)n0 * n1, 1, t1, t2, nullptr);
    fdm_transform(s, F.dir[1].S.p, n1, n0, n2, t2, t1, nullptr);
    fdm_transform(s, F.dir[0].S.p, n0, 1, (int64_t)n1 * n2, t1, z, nullptr);
  }
}

// ---- host side: generalised symmetric-definite eigenproblem K s = lam M s (dense, n <= ~600), cyclic Jacobi on L^-1 K L^-T ---------
void gen_sym_eig(int n, const std::vector<double> &K, const std::vector<double> &M, std::vector<double> &S, std::vector<double> &lam) {
  std::vector<double> L((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {                                  // Cholesky M = L L^T
    double d = M[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0)) throw Error("fast diagonalisation: 1D mass matrix is not positive definite");
    L[(size_t)j * n + j] = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) {
      double v = M[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = v / L[(size_t)j * n + j];
    }
  }
  std::vector<double> C(K);                                      // C = L^-1 K L^-T: forward substitution on the columns, then on the rows
  for (int col = 0; col < n; ++col)
    for (int i = 0; i < n; ++i) { double v = C[(size_t)i * n + col]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * C[(size_t)k * n + col]; C[(size_t)i * n + col] = v / L[(size_t)i * n + i]; }
  for (int row = 0; row < n; ++row)
    for (int i = 0; i < n; ++i) { double v = C[(size_t)row * n + i]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * C[(size_t)row * n + k]; C[(size_t)row * n + i] = v / L[(size_t)i * n + i]; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) { const double v = 0.5 * (C[(size_t)i * n + j] + C[(size_t)j * n + i]); C[(size_t)i * n + j] = C[(size_t)j * n + i] = v; }
  std::vector<double> V((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  double scale = 0; for (double v : C) scale = std::max(scale, std::fabs(v));
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += C[(size_t)p * n + q] * C[(size_t)p * n + q];
    if (std::sqrt(off) <= 1e-15 * scale * n) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = C[(size_t)p * n + q];
        if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 1e-17 * scale) continue;
        const double theta = (C[(size_t)q * n + q] - C[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; ++k) { const double ckp = C[(size_t)k * n + p], ckq = C[(size_t)k * n + q]; C[(size_t)k * n + p] = cs * ckp - sn * ckq; C[(size_t)k * n + q] = sn * ckp + cs * ckq; }
        for (int k = 0; k < n; ++k) { const double cpk = C[(size_t)p * n + k], cqk = C[(size_t)q * n + k]; C[(size_t)p * n + k] = cs * cpk - sn * cqk; C[(size_t)q * n + k] = sn * cpk + cs * cqk; }
        for (int k = 0; k < n; ++k) { const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q]; V[(size_t)k * n + p] = cs * vkp - sn * vkq; V[(size_t)k * n + q] = sn * vkp + cs * vkq; }
      }
  }
  lam.resize(n); for (int i = 0; i < n; ++i) lam[i] = C[(size_t)i * n + i];
  S.assign((size_t)n * n, 0.0);                                  // S = L^-T V: back substitution per column
  for (int col = 0; col < n; ++col)
    for (int i = n - 1; i >= 0; --i) { double v = V[(size_t)i * n + col]; for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * S[(size_t)k * n + col]; S[(size_t)i * n + col] = v / L[(size_t)i * n + i]; }
}

}  // namespace poro
