// K-apply-u: matrix-free y = A_u x for the elasticity operator on a uniform box mesh (gfx950, wave64).
// Replaces SparseMatrix::vmult of the displacement system inside SolverCG (PoroElasticDisplacementSolver.h:305)
// without ever forming the 189-entries-per-row CSR matrix: all cells of the box share one element matrix
// Ke (assembled once by K-asm-u from the reference's bilinear form, :230-242), so one application needs only
// x (read once per tile + halo), y (written once) and the 52 KB Ke, which stays in the scalar / L2 caches.
//
// Formulation (owner computes, no atomics, bitwise reproducible): a workgroup owns a tile of nodes; x of the
// tile plus its halo is staged in LDS (Dirichlet columns zeroed on the way in).  Nodes are sorted by their
// position in the cell (vertex / edge / face / centre classes for Q2): one wavefront handles 64 nodes of ONE
// class, so the local row index inside every adjacent cell is wave-uniform and the Ke rows are fetched through
// the scalar data path (s_load) and broadcast as SGPR operands of v_fma_f64, while the 64 lanes read their x
// values from LDS.  Each lane gathers y_i = sum_{cells K adjacent to i} sum_j Ke[row_K(i)][j] x_j.
// Constrained rows return diag_i x_i (ConstraintMatrix elimination).
#include "common.hpp"
#include "p_stencil.hpp"

namespace poro {
namespace {

template <int DIM, int K> struct Cfg {
  static constexpr int N1 = K + 1;
  static constexpr int NS = DIM == 2 ? N1 * N1 : N1 * N1 * N1;
  static constexpr int DPC = NS * DIM;
  static constexpr int LAT0 = DIM == 2 ? 8 : 4, LAT1 = DIM == 2 ? 8 : 4, LAT2 = DIM == 2 ? 1 : 4;
  static constexpr int NW = (K == 2) ? (1 << DIM) : 4;
  static constexpr int T0 = 2 * LAT0, T1 = 2 * LAT1, T2 = DIM == 2 ? 1 : (K == 2 ? 2 * LAT2 : LAT2);
  static constexpr int HLO = K, HHI = 1;
  static constexpr int E0 = T0 + HLO + HHI, E1 = T1 + HLO + HHI, E2 = DIM == 2 ? 1 : T2 + HLO + HHI;
  static constexpr int LDS_NODES = E0 * E1 * E2;
};

struct MfGeom { int nn[3]; int nc[3]; int nt[3]; int64_t n_tiles; };

// blocks b and b+8 share an XCD (MI355X dispatches round-robin over the 8 XCDs): give every XCD one contiguous
// range of tiles so halo re-reads hit that XCD's L2 (speed only, never correctness)
__device__ inline int64_t xcd_tile(int64_t bid, int64_t n) {
  const int64_t q = n / 8, r = n % 8, xcd = bid % 8, idx = bid / 8;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

template <int DIM, int K, bool DIAG_ONLY>
__global__ void __launch_bounds__((Cfg<DIM, K>::NW * 64))
k_mf_apply(MfGeom g, const double *__restrict__ Ke, const double *__restrict__ x, double *__restrict__ y, const uint8_t *__restrict__ mask,
           const double *__restrict__ diag_local, int constrained, double *__restrict__ dot_partials) {
  using C = Cfg<DIM, K>;
  __shared__ double sx[DIAG_ONLY ? 1 : C::LDS_NODES * DIM];   // [comp][node] (SoA: conflict-light ds_read_b64)
  __shared__ double sdot[C::NW];
  const int tid = threadIdx.x;
  double dot_acc = 0.0;                                        // x.y over this workgroup's tiles (fused d.Ad of PCG)
  // one tile per workgroup, except when the fused dot product limits the grid to the number of partial slots
  for (int64_t tile_i = blockIdx.x; tile_i < g.n_tiles; tile_i += gridDim.x) {
  const int64_t tile = gridDim.x == g.n_tiles ? xcd_tile(tile_i, g.n_tiles) : tile_i;
  int t[3];
  t[0] = (int)(tile % g.nt[0]); t[1] = (int)((tile / g.nt[0]) % g.nt[1]); t[2] = (int)(tile / ((int64_t)g.nt[0] * g.nt[1]));
  const int org[3] = {t[0] * C::T0, t[1] * C::T1, DIM == 3 ? t[2] * C::T2 : 0};

  if constexpr (!DIAG_ONLY) {
    for (int idx = tid; idx < C::LDS_NODES * DIM; idx += C::NW * 64) {
      const int c = idx % DIM, nd = idx / DIM;
      const int l0 = nd % C::E0, l1 = (nd / C::E0) % C::E1, l2 = nd / (C::E0 * C::E1);
      const int g0 = org[0] - C::HLO + l0, g1 = org[1] - C::HLO + l1, g2 = DIM == 3 ? org[2] - C::HLO + l2 : 0;
      double v = 0.0;
      if (g0 >= 0 && g0 < g.nn[0] && g1 >= 0 && g1 < g.nn[1] && g2 >= 0 && g2 < g.nn[2]) {
        const int64_t dof = (((int64_t)g2 * g.nn[1] + g1) * g.nn[0] + g0) * DIM + c;
        v = x[dof];
        if (constrained && mask[dof]) v = 0.0;
      }
      sx[c * C::LDS_NODES + nd] = v;
    }
    __syncthreads();
  }

  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lat[3] = {lane % C::LAT0, (lane / C::LAT0) % C::LAT1, lane / (C::LAT0 * C::LAT1)};
  int woff[3], vertex[3];   // wave-uniform: offset of this wave's node lattice, and whether its nodes are vertex-type per direction
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (K == 2) { woff[d] = (w >> d) & 1; vertex[d] = !woff[d]; }
    else { woff[d] = (d < 2) ? ((w >> d) & 1) * (d == 0 ? C::LAT0 : C::LAT1) : 0; vertex[d] = 1; }
  }
  int node[3]; bool in_range = true;
#pragma unroll
  for (int d = 0; d < DIM; ++d) { node[d] = org[d] + K * lat[d] + woff[d]; in_range = in_range && node[d] < g.nn[d]; }
  if (DIM == 2) node[2] = 0;

  double acc[DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a) acc[a] = 0.0;

  const int nopt0 = vertex[0] ? 2 : 1, nopt1 = vertex[1] ? 2 : 1, nopt2 = DIM == 3 ? (vertex[2] ? 2 : 1) : 1;
  for (int o2 = 0; o2 < nopt2; ++o2)
    for (int o1 = 0; o1 < nopt1; ++o1)
      for (int o0 = 0; o0 < nopt0; ++o0) {
        const int o[3] = {o0, o1, o2};
        int cell[3], loc[3]; bool exists = in_range;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          if (vertex[d]) { cell[d] = node[d] / K - 1 + o[d]; loc[d] = o[d] ? 0 : K; }
          else { cell[d] = (node[d] - 1) / 2; loc[d] = 1; }
          exists = exists && cell[d] >= 0 && cell[d] < g.nc[d];
        }
        const int li = loc[0] + C::N1 * (loc[1] + (DIM == 3 ? C::N1 * loc[2] : 0));   // wave-uniform local scalar node
        const double *__restrict__ row = Ke + (size_t)li * DIM * C::DPC;
        if constexpr (DIAG_ONLY) {
          if (exists) {
#pragma unroll
            for (int a = 0; a < DIM; ++a) acc[a] += row[a * C::DPC + li * DIM + a];
          }
        } else {
          if (exists) {
            const int b0 = cell[0] * K - (org[0] - C::HLO), b1 = cell[1] * K - (org[1] - C::HLO), b2 = DIM == 3 ? cell[2] * K - (org[2] - C::HLO) : 0;
            const double *xs = sx + (b2 * C::E1 + b1) * C::E0 + b0;
#pragma unroll
            for (int j2 = 0; j2 < (DIM == 3 ? C::N1 : 1); ++j2)
#pragma unroll
              for (int j1 = 0; j1 < C::N1; ++j1)
#pragma unroll
                for (int j0 = 0; j0 < C::N1; ++j0) {
                  const int j = j0 + C::N1 * (j1 + C::N1 * j2);
                  const int off = (j2 * C::E1 + j1) * C::E0 + j0;
#pragma unroll
                  for (int b = 0; b < DIM; ++b) {
                    const double xv = xs[b * C::LDS_NODES + off];
#pragma unroll
                    for (int a = 0; a < DIM; ++a) acc[a] += row[a * C::DPC + j * DIM + b] * xv;
                  }
                }
          }
        }
      }
  if (in_range) {
    const int64_t nodeid = ((int64_t)node[2] * g.nn[1] + node[1]) * g.nn[0] + node[0];
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
      const int64_t dof = nodeid * DIM + a;
      double v = acc[a];
      if constexpr (!DIAG_ONLY) {
        const double xv = x[dof];
        if (constrained && mask[dof]) v = diag_local[dof] * xv;
        dot_acc = fma(xv, v, dot_acc);
      }
      y[dof] = v;
    }
  }
  if (tile_i + gridDim.x < g.n_tiles) __syncthreads();          // the next tile refills sx
  }
  if constexpr (!DIAG_ONLY) {
    if (dot_partials) {                                          // fixed shuffle tree, waves summed in index order: bitwise reproducible
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_xor(dot_acc, off, 64);
      if ((tid & 63) == 0) sdot[tid >> 6] = dot_acc;
      __syncthreads();
      if (tid == 0) { double tsum = 0; for (int q = 0; q < C::NW; ++q) tsum += sdot[q]; dot_partials[blockIdx.x] = tsum; }
    }
  }
}

template <int DIM, int K> MfGeom make_geom(const BoxDev &box) {
  using C = Cfg<DIM, K>;
  MfGeom g{};
  const int T[3] = {C::T0, C::T1, C::T2};
  g.n_tiles = 1;
  for (int d = 0; d < 3; ++d) {
    g.nc[d] = d < DIM ? box.n[d] : 1; g.nn[d] = d < DIM ? K * box.n[d] + 1 : 1;
    g.nt[d] = d < DIM ? (g.nn[d] + T[d] - 1) / T[d] : 1; g.n_tiles *= g.nt[d];
  }
  return g;
}

template <int DIM, int K> void launch(hipStream_t s, const MfArgs &a, const double *x, double *y, bool constrained, bool diag_only, double *dot_partials) {
  using C = Cfg<DIM, K>;
  const MfGeom g = make_geom<DIM, K>(a.box);
  const unsigned grid = (unsigned)(dot_partials ? std::min<int64_t>(g.n_tiles, kMaxPartials) : g.n_tiles);   // one partial slot per workgroup
  if (diag_only) hipLaunchKernelGGL((k_mf_apply<DIM, K, true>), grid, C::NW * 64, 0, s, g, a.Ke, x, y, a.mask, a.diag_local, 0, (double *)nullptr);
  else hipLaunchKernelGGL((k_mf_apply<DIM, K, false>), grid, C::NW * 64, 0, s, g, a.Ke, x, y, a.mask, a.diag_local, constrained ? 1 : 0, dot_partials);
}

void dispatch(hipStream_t s, const MfArgs &a, const double *x, double *y, bool constrained, bool diag_only, double *dot_partials = nullptr) {
  if (a.dim == 2 && a.k_u == 1) launch<2, 1>(s, a, x, y, constrained, diag_only, dot_partials);
  else if (a.dim == 2 && a.k_u == 2) launch<2, 2>(s, a, x, y, constrained, diag_only, dot_partials);
  else if (a.dim == 3 && a.k_u == 1) launch<3, 1>(s, a, x, y, constrained, diag_only, dot_partials);
  else if (a.dim == 3 && a.k_u == 2) launch<3, 2>(s, a, x, y, constrained, diag_only, dot_partials);
  else throw Error("mf_apply: unsupported dim / degree");
}

}  // namespace

// ---- K-apply-p: y = (a M + kappa K) x for the Q1 pressure space on a uniform box ---------------------------------------------
// Replaces SparseMatrix::vmult of the Jacobian (PoroElasticPressureSolver.h:179) and of the projection mass matrix
// (StrainProjector.h:213): M = Mx (x) My (x) Mz, K = Kx (x) My (x) Mz + Mx (x) Ky (x) Mz + Mx (x) My (x) Kz with the tridiagonal 1D
// Q1 matrices M1 = h/6 (1,4,1), K1 = 1/h (-1,2,-1) (boundary rows: h/6 (2,1), 1/h (1,-1)): a 3^DIM-point constant-coefficient
// stencil.  One thread per node; the 3 MB vector stays in L2, so the kernel costs microseconds where the CSR form streams 125 MB.
template <int DIM> __global__ void __launch_bounds__(256)
k_p_stencil(int n0, int n1, int n2, double h0, double h1, double h2, double a, double kappa, const double *__restrict__ x, double *__restrict__ y) {
  const int64_t node = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nn = (int64_t)n0 * n1 * n2;
  if (node >= nn) return;
  y[node] = p_stencil_row<DIM>(n0, n1, n2, h0, h1, h2, a, kappa, node, x);
}

// K-res-p on a uniform box: R = -((M t + kappa (K p)) + q), same association as k_residual / the reference (PoroElasticPressureSolver.h:113-155)
template <int DIM> __global__ void __launch_bounds__(256)
k_p_residual_stencil(int n0, int n1, int n2, double h0, double h1, double h2, double kappa, const double *__restrict__ t, const double *__restrict__ p,
                     const double *__restrict__ src, double *__restrict__ R) {
  const int64_t node = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= (int64_t)n0 * n1 * n2) return;
  const double s1 = p_stencil_row<DIM>(n0, n1, n2, h0, h1, h2, 1.0, 0.0, node, t), s2 = p_stencil_row<DIM>(n0, n1, n2, h0, h1, h2, 0.0, 1.0, node, p);
  R[node] = -((s1 + s2 * kappa) + src[node]);
}
void p_residual_stencil(hipStream_t s, int dim, const BoxDev &box, double kappa, const double *t, const double *p, const double *src, double *R) {
  const int n0 = box.n[0] + 1, n1 = box.n[1] + 1, n2 = dim == 3 ? box.n[2] + 1 : 1;
  const unsigned grid = (unsigned)(((int64_t)n0 * n1 * n2 + 255) / 256);
  if (dim == 2) hipLaunchKernelGGL(k_p_residual_stencil<2>, grid, 256, 0, s, n0, n1, n2, box.h[0], box.h[1], 1.0, kappa, t, p, src, R);
  else hipLaunchKernelGGL(k_p_residual_stencil<3>, grid, 256, 0, s, n0, n1, n2, box.h[0], box.h[1], box.h[2], kappa, t, p, src, R);
}

void p_stencil_apply(hipStream_t s, int dim, const BoxDev &box, double a, double kappa, const double *x, double *y) {
  const int n0 = box.n[0] + 1, n1 = box.n[1] + 1, n2 = dim == 3 ? box.n[2] + 1 : 1;
  const int64_t nn = (int64_t)n0 * n1 * n2;
  const unsigned grid = (unsigned)((nn + 255) / 256);
  if (dim == 2) hipLaunchKernelGGL(k_p_stencil<2>, grid, 256, 0, s, n0, n1, n2, box.h[0], box.h[1], 1.0, a, kappa, x, y);
  else hipLaunchKernelGGL(k_p_stencil<3>, grid, 256, 0, s, n0, n1, n2, box.h[0], box.h[1], box.h[2], a, kappa, x, y);
}

// dot_partials (optional, kMaxPartials slots zeroed once by the caller): per-workgroup partial sums of x.y over all local rows
void mf_apply(hipStream_t s, const MfArgs &a, const double *x, double *y, bool constrained, double *dot_partials) { dispatch(s, a, x, y, constrained, false, dot_partials); }
void mf_diag(hipStream_t s, const MfArgs &a, double *diag) { dispatch(s, a, nullptr, diag, false, true); }

}  // namespace poro
