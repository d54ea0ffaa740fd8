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
  // 4 k-steps per trip with all 20 operand loads issued before the first MFMA: the kernel is latency-bound (operands come from L2)
  for (int l0 = 0; l0 < n_l; l0 += 16) {
    double a[4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int l = l0 + 4 * u + kq; const bool lv = l < n_l;
      a[u] = (aval && lv) ? Arow[l] : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) b[u][c] = (bval[c] && lv) ? in[boff[c] + (int64_t)l * SI] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][c], acc[c], 0, 0, 0);
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
        int ii, jj, kk; bool ok = true;
        if (sc.ncol > 0) {             // column-distributed layout of the partitioned solve: [line of the last direction][ncol local columns]
          const int line = (int)(addr / sc.ncol); const int64_t col = sc.col0 + (addr - (int64_t)line * sc.ncol);
          ok = col < sc.col_total;     // padding columns hold zeros
          if (sc.lam[2]) { jj = (int)(col / sc.n[0]); ii = (int)(col - (int64_t)jj * sc.n[0]); kk = line; } else { ii = (int)col; jj = line; kk = 0; }
        } else {
          const int64_t n01 = (int64_t)sc.n[0] * sc.n[1];
          kk = (int)(addr / n01); const int64_t rem = addr - (int64_t)kk * n01; jj = (int)(rem / sc.n[0]); ii = (int)(rem - (int64_t)jj * sc.n[0]);
        }
        if (ok) {
          double D = sc.a + sc.k[0] * sc.lam[0][ii] + sc.k[1] * sc.lam[1][jj];
          if (sc.lam[2]) D += sc.k[2] * sc.lam[2][kk];
          v /= D;
        }
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
  for (int l0 = 0; l0 < n_l; l0 += 16) {
    double a[4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int l = l0 + 4 * u + kq; const bool lv = l < n_l;
      a[u] = (aval && lv) ? Arow[l] : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) b[u][c] = (bval[c] && lv) ? Brow[c][l] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][c], acc[c], 0, 0, 0);
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

// copy a (planes x columns) window between a strided grid and a dense, zero-padded [n_planes_pad][C] block (to_block) or back
__global__ void k_fdm_window(double *dst, const double *src, int to_block, int n_planes, int n_planes_pad, int64_t C, int64_t ncols_valid,
                             int64_t grid_stride, int64_t grid_col0, int64_t grid_plane0) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_planes_pad * C) return;
  const int64_t k = idx / C, cc = idx - k * C;
  const bool valid = k < n_planes && cc < ncols_valid;
  const int64_t g = (grid_plane0 + k) * grid_stride + grid_col0 + cc;
  if (to_block) dst[idx] = valid ? src[g] : 0.0;
  else if (valid) dst[g] = src[idx];
}

__global__ void __launch_bounds__(256) k_fdm_window_batch(double *grid, double *dense, double *dense_self, int self, int to_block, const FdmWindow *__restrict__ win, int n_planes_pad, int64_t C, int64_t grid_stride, int64_t blk) {
  const int q = blockIdx.y; const FdmWindow W = win[q];
  double *blkp = ((to_block && q == self) ? dense_self : dense) + (int64_t)q * blk;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < (int64_t)n_planes_pad * C; idx += (int64_t)gridDim.x * 256) {
    const int64_t k = idx / C, cc = idx - k * C;
    const bool valid = k < W.n_planes && cc < W.ncols_valid;
    const int64_t g = (W.grid_plane0 + k) * grid_stride + W.grid_col0 + cc;
    if (to_block) blkp[idx] = valid ? grid[g] : 0.0;
    else if (valid) grid[g] = blkp[idx];
  }
}

}  // namespace

void fdm_window_batch(hipStream_t s, double *grid, double *dense, double *dense_self, int self, bool to_block, const FdmWindow *win, int n_peers, int n_planes_pad, int64_t C, int64_t grid_stride, int64_t blk) {
  const int64_t n = (int64_t)n_planes_pad * C;
  if (n && n_peers) hipLaunchKernelGGL(k_fdm_window_batch, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024), (unsigned)n_peers), 256, 0, s, grid, dense, dense_self, self, to_block ? 1 : 0, win, n_planes_pad, C, grid_stride, blk);
}
void fdm_window(hipStream_t s, double *dst, const double *src, bool to_block, int n_planes, int n_planes_pad, int64_t C, int64_t ncols_valid, int64_t grid_stride,
                int64_t grid_col0, int64_t grid_plane0) {
  const int64_t n = (int64_t)n_planes_pad * C;
  if (n) hipLaunchKernelGGL(k_fdm_window, (unsigned)((n + 255) / 256), 256, 0, s, dst, src, to_block ? 1 : 0, n_planes, n_planes_pad, C, ncols_valid, grid_stride, grid_col0, grid_plane0);
}

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
  FdmScale sc{}; sc.a = a; sc.ncol = 0;
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

// ---- host side: K1 s = lam M1 s for the 1D Q1 matrices on n cells of size h (natural boundary conditions) ---------------------------
// M1 = h/6 tridiag(1 4 1) (corners 2), K1 = 1/h tridiag(-1 2 -1) (corners 1): the eigenvectors are the cosines s_j(i) = cos(j pi i / n),
// lam_j = 6/h^2 (1 - cos t)/(2 + cos t), t = j pi / n (insert into an interior and a boundary row); columns scaled to s^T M1 s = 1.
void q1_eig(int n_cells, double h, std::vector<double> &S, std::vector<double> &lam) {
  const int n = n_cells + 1; const double pi = 3.14159265358979323846;
  S.assign((size_t)n * n, 0.0); lam.assign(n, 0.0);
  std::vector<double> v(n);
  for (int j = 0; j < n; ++j) {
    const double t = j * pi / n_cells, ct = std::cos(t);
    lam[j] = 6.0 / (h * h) * (1.0 - ct) / (2.0 + ct);
    for (int i = 0; i < n; ++i) v[i] = std::cos(t * i);
    double m = 0;
    for (int i = 0; i < n; ++i) {
      double Mv = (i > 0 && i < n - 1 ? 4.0 : 2.0) * v[i];
      if (i > 0) Mv += v[i - 1];
      if (i < n - 1) Mv += v[i + 1];
      m += v[i] * Mv * h / 6.0;
    }
    const double sc = 1.0 / std::sqrt(m);
    for (int i = 0; i < n; ++i) S[(size_t)i * n + j] = v[i] * sc;
  }
}

}  // namespace poro
