// Uniform-box forms of the two cell loops that run every fixed-stress iteration besides the solves:
//   K-rhs-u : b_i = int alpha p_h div(phi_i)                      (PoroElasticDisplacementSolver.h:231-236 inside the loop :203-288)
//   K-proj  : r^c_i = int psi_i eps_c(u_h)                         (StrainProjector.h:150-195)
// On a box with constant Jacobian both are the SAME rectangular Kronecker operator and its transpose:
//   G_c = W^c_x (x) W^c_y (x) W^c_z,   W^c_d = D_d if d == c else N_d,
//   N_d[i][a] = int phi_i psi_a dx_d,  D_d[i][a] = int phi_i' psi_a dx_d      (1D Q_k x Q1, assembled from 2-column local blocks)
//   b_c = alpha G_c p + (lifting + Neumann),      r^(c,c) = G_c^T u_c,   r^(a,b) = (G_b^T u_a + G_a^T u_b) / 2.
// (The reference integrates these products with QGauss(k+1) / QGauss(2); both are exact for them, so the quadrature sums equal the
// 1D integrals up to rounding.)  One thread per OUTPUT node, gather form: no colouring, no atomics, no index arrays - the pressure
// vector (3 MB) and the 5^dim neighbourhood of u come through L1/L2, HBM sees each vector once.
// The generic per-cell kernels (kernels_asm.hip) remain the path for unstructured meshes and check this one at set-up.
#include "common.hpp"

namespace poro {
namespace {

constexpr int kTB = 256;

// weights of direction d at u index i: nonzero p indices a0 .. a0+2 (entries may be zero / out of range -> weight 0)
struct W3 { int a0; double wN[3], wD[3]; };
template <int K> __device__ __forceinline__ W3 u_row_weights(const BoxCoupling &B, int d, int i) {
  constexpr int k = K; const int n = B.n[d], e = i / k, s = i - e * k;
  W3 w; w.wN[0] = w.wN[1] = w.wN[2] = 0; w.wD[0] = w.wD[1] = w.wD[2] = 0;
  if (s != 0) {                       // interior node of cell e
    w.a0 = e;
    w.wN[0] = B.N[1][0] * B.h[d]; w.wN[1] = B.N[1][1] * B.h[d]; w.wD[0] = B.D[1][0]; w.wD[1] = B.D[1][1];   // K = 2: the only interior node is s = 1
  } else {                            // vertex e: last node of cell e-1 and first node of cell e
    w.a0 = e - 1;
    if (e >= 1) { w.wN[0] += B.N[k][0] * B.h[d]; w.wN[1] += B.N[k][1] * B.h[d]; w.wD[0] += B.D[k][0]; w.wD[1] += B.D[k][1]; }
    if (e <= n - 1) { w.wN[1] += B.N[0][0] * B.h[d]; w.wN[2] += B.N[0][1] * B.h[d]; w.wD[1] += B.D[0][0]; w.wD[2] += B.D[0][1]; }
  }
  return w;
}

template <int DIM, int K> __global__ void __launch_bounds__(kTB)
k_box_rhs_u(BoxCoupling B, double alpha, const double *__restrict__ p, const double *__restrict__ lift, const double *__restrict__ neu,
            const uint8_t *__restrict__ mask, double *__restrict__ rhs) {
  const int nu0 = K * B.n[0] + 1, nu1 = K * B.n[1] + 1, nu2 = DIM == 3 ? K * B.n[2] + 1 : 1;
  const int64_t node = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (node >= (int64_t)nu0 * nu1 * nu2) return;
  const int i0 = (int)(node % nu0), i1 = (int)((node / nu0) % nu1), i2 = (int)(node / ((int64_t)nu0 * nu1));
  const int np0 = B.n[0] + 1, np1 = B.n[1] + 1;
  const W3 w0 = u_row_weights<K>(B, 0, i0), w1 = u_row_weights<K>(B, 1, i1);
  W3 w2; w2.a0 = 0; w2.wN[0] = 1; w2.wN[1] = w2.wN[2] = 0; w2.wD[0] = w2.wD[1] = w2.wD[2] = 0;
  if constexpr (DIM == 3) w2 = u_row_weights<K>(B, 2, i2);
  double acc[3] = {0, 0, 0};
#pragma unroll
  for (int c = 0; c < (DIM == 3 ? 3 : 1); ++c) {
    if (w2.wN[c] == 0.0 && w2.wD[c] == 0.0) continue;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (w1.wN[b] == 0.0 && w1.wD[b] == 0.0) continue;
      const double nn = w1.wN[b] * w2.wN[c], dn = w1.wD[b] * w2.wN[c], nd = w1.wN[b] * w2.wD[c];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (w0.wN[a] == 0.0 && w0.wD[a] == 0.0) continue;
        const double pv = p[(int64_t)(w0.a0 + a) + (int64_t)np0 * ((w1.a0 + b) + (int64_t)np1 * (w2.a0 + c))];
        acc[0] = fma(pv, w0.wD[a] * nn, acc[0]);
        acc[1] = fma(pv, w0.wN[a] * dn, acc[1]);
        if constexpr (DIM == 3) acc[2] = fma(pv, w0.wN[a] * nd, acc[2]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < DIM; ++c) {
    const int64_t dof = node * DIM + c;
    rhs[dof] = mask[dof] ? 0.0 : (alpha * acc[c] + neu[dof]) + lift[dof];      // same finish as k_rhs_u_finish
  }
}

// weights of direction d at p index a: u indices i0 .. i0+2k
struct W5 { int i0; double wN[5], wD[5]; };
template <int K> __device__ __forceinline__ W5 p_row_weights(const BoxCoupling &B, int d, int a) {
  constexpr int k = K; const int n = B.n[d];
  W5 w; w.i0 = k * (a - 1);
#pragma unroll
  for (int m = 0; m < 5; ++m) { w.wN[m] = 0; w.wD[m] = 0; }
#pragma unroll
  for (int s = 0; s <= k; ++s) {
    if (a >= 1) { w.wN[s] += B.N[s][1] * B.h[d]; w.wD[s] += B.D[s][1]; }                 // cell a-1, psi_1
    if (a <= n - 1) { w.wN[k + s] += B.N[s][0] * B.h[d]; w.wD[k + s] += B.D[s][0]; }     // cell a, psi_0
  }
  return w;
}

struct BoxProjOut { double *rhs[6]; int t1[6], t2[6]; int n; };

template <int DIM, int K> __global__ void __launch_bounds__(kTB)
k_box_proj_rhs(BoxCoupling B, const double *__restrict__ u, BoxProjOut out) {
  const int np0 = B.n[0] + 1, np1 = B.n[1] + 1, np2 = DIM == 3 ? B.n[2] + 1 : 1;
  const int64_t node = (int64_t)blockIdx.x * kTB + threadIdx.x;
  if (node >= (int64_t)np0 * np1 * np2) return;
  const int a0 = (int)(node % np0), a1 = (int)((node / np0) % np1), a2 = (int)(node / ((int64_t)np0 * np1));
  const int nu0 = K * B.n[0] + 1, nu1 = K * B.n[1] + 1;
  constexpr int span = 2 * K + 1;
  const W5 w0 = p_row_weights<K>(B, 0, a0), w1 = p_row_weights<K>(B, 1, a1);
  W5 w2; w2.i0 = 0; for (int m = 0; m < 5; ++m) { w2.wN[m] = 0; w2.wD[m] = 0; } w2.wN[0] = 1;
  if constexpr (DIM == 3) w2 = p_row_weights<K>(B, 2, a2);
  double G[DIM][DIM];                 // G[comp][dir] = int psi_node d u_comp / d x_dir
#pragma unroll
  for (int c = 0; c < DIM; ++c)
#pragma unroll
    for (int d = 0; d < DIM; ++d) G[c][d] = 0;
#pragma unroll
  for (int kz = 0; kz < (DIM == 3 ? span : 1); ++kz) {
    if (w2.wN[kz] == 0.0 && w2.wD[kz] == 0.0) continue;
#pragma unroll
    for (int jy = 0; jy < span; ++jy) {
      if (w1.wN[jy] == 0.0 && w1.wD[jy] == 0.0) continue;
      const double nn = w1.wN[jy] * w2.wN[kz], dn = w1.wD[jy] * w2.wN[kz], nd = w1.wN[jy] * w2.wD[kz];
      const int64_t rowbase = (int64_t)nu0 * ((w1.i0 + jy) + (int64_t)nu1 * (w2.i0 + kz));
#pragma unroll
      for (int ix = 0; ix < span; ++ix) {
        if (w0.wN[ix] == 0.0 && w0.wD[ix] == 0.0) continue;
        const double *uv = u + (rowbase + w0.i0 + ix) * DIM;
        const double gx = w0.wD[ix] * nn, gy = w0.wN[ix] * dn, gz = w0.wN[ix] * nd;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          const double uc = uv[c];
          G[c][0] = fma(uc, gx, G[c][0]); G[c][1] = fma(uc, gy, G[c][1]);
          if constexpr (DIM == 3) G[c][2] = fma(uc, gz, G[c][2]);
        }
      }
    }
  }
  auto sel = [&](int c, int d) {
    double r = 0;
#pragma unroll
    for (int cc = 0; cc < DIM; ++cc)
#pragma unroll
      for (int dd = 0; dd < DIM; ++dd) if (cc == c && dd == d) r = G[cc][dd];
    return r;
  };
  for (int e = 0; e < out.n; ++e) {
    const int t1 = out.t1[e], t2 = out.t2[e];
    out.rhs[e][node] = t1 == t2 ? sel(t1, t1) : (sel(t1, t2) + sel(t2, t1)) / 2;   // StrainProjector.h:177-181 / ConstitutiveModel.h:27-42
  }
}

// ---- K-asm-u on a uniform box: every cell matrix is the same Ke, so each CSR entry is written ONCE by its owner (no colouring, no
// read-modify-write): A(r, c) = sum over the <= 2^dim cells that contain both nodes of Ke[local(r)][local(c)], Dirichlet rows / columns
// eliminated as in distribute_local_to_global (sum of |K_ii| on the constrained diagonal).  L lanes per CSR row as in k_spmv.
// x / d for 0 <= x < 2^31 through the reciprocal (one correction step): the kernel decomposes a node index per CSR entry
__device__ __forceinline__ int div_by(int x, int d, double inv) { int q = (int)((double)x * inv); const int r = x - q * d; if (r < 0) --q; else if (r >= d) ++q; return q; }
template <int DIM, int K, int L> __global__ void __launch_bounds__(256)
k_box_asm_u(int n0, int n1, int n2, const double *__restrict__ Ke, int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
            const uint8_t *__restrict__ mask, double *__restrict__ val) {
  constexpr int N1 = K + 1, NS = DIM == 2 ? N1 * N1 : N1 * N1 * N1, DPC = NS * DIM;
  const int lane = threadIdx.x % L;
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) / L;
  if (row >= n_rows) return;
  const int nc[3] = {n0, n1, n2};
  const int nx = K * n0 + 1, ny = K * n1 + 1;
  const double inx = 1.0 / nx, iny = 1.0 / ny;
  const int node_r = (int)(row / DIM), cr = (int)(row - (int64_t)node_r * DIM);
  const int qr = div_by(node_r, nx, inx), zr = DIM == 3 ? div_by(qr, ny, iny) : 0;
  const int ir[3] = {node_r - qr * nx, qr - zr * ny, zr};
  const bool mrow = mask[row] != 0;
  for (int64_t e = rp[row] + lane; e < rp[row + 1]; e += L) {
    const int32_t c = col[e];
    const int node_c = c / DIM, cc = c - node_c * DIM;
    const int qc = div_by(node_c, nx, inx), zc = DIM == 3 ? div_by(qc, ny, iny) : 0;
    const int ic[3] = {node_c - qc * nx, qc - zc * ny, zc};
    int c0[3], cnt[3];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {                            // cells of direction d that contain both indices: K c <= i, j <= K c + K
      const int hi = ir[d] > ic[d] ? ir[d] : ic[d], lo = ir[d] < ic[d] ? ir[d] : ic[d];
      int cmin = (hi - K + K - 1) / K; if (hi - K < 0) cmin = 0;
      int cmax = lo / K; if (cmax > nc[d] - 1) cmax = nc[d] - 1;
      c0[d] = cmin; cnt[d] = cmax - cmin + 1;
    }
    double v = 0.0;
    const bool diag = c == (int32_t)row;
    if (!(mrow && !diag) && !(mask[c] && !mrow)) {
      for (int kz = 0; kz < (DIM == 3 ? cnt[2] : 1); ++kz)
        for (int ky = 0; ky < cnt[1]; ++ky)
          for (int kx = 0; kx < cnt[0]; ++kx) {
            const int cx = c0[0] + kx, cy = c0[1] + ky, cz = DIM == 3 ? c0[2] + kz : 0;
            const int li = (DIM == 3 ? (ir[2] - K * cz) * N1 * N1 : 0) + (ir[1] - K * cy) * N1 + (ir[0] - K * cx);
            const int lj = (DIM == 3 ? (ic[2] - K * cz) * N1 * N1 : 0) + (ic[1] - K * cy) * N1 + (ic[0] - K * cx);
            const double k = Ke[(size_t)(li * DIM + cr) * DPC + lj * DIM + cc];
            v += mrow ? fabs(k) : k;
          }
    }
    val[e] = v;
  }
}

}  // namespace

template <int DIM, int K> static void launch_box_asm(hipStream_t s, const BoxDev &box, const double *Ke, const CsrDev &A, const uint8_t *mask, double *val) {
  const int64_t grid = (A.n * 64 + 255) / 256;
  hipLaunchKernelGGL((k_box_asm_u<DIM, K, 64>), (unsigned)grid, 256, 0, s, box.n[0], box.n[1], DIM == 3 ? box.n[2] : 1, Ke, A.n, A.rp.p, A.col.p, mask, val);
}
void box_asm_u_matrix(hipStream_t s, int dim, int k_u, const BoxDev &box, const double *Ke, const CsrDev &A, const uint8_t *mask, double *val) {
  if (dim == 2 && k_u == 1) launch_box_asm<2, 1>(s, box, Ke, A, mask, val);
  else if (dim == 2) launch_box_asm<2, 2>(s, box, Ke, A, mask, val);
  else if (k_u == 1) launch_box_asm<3, 1>(s, box, Ke, A, mask, val);
  else launch_box_asm<3, 2>(s, box, Ke, A, mask, val);
}

// 1D local blocks on the unit interval: N[s][t] = int phi_s psi_t, D[s][t] = int phi_s' psi_t (Lagrange Q_k on equidistant nodes, Q1)
BoxCoupling box_coupling(int dim, int k_u, const BoxDev &box) {
  if (k_u < 1 || k_u > 2) throw Error("box assembly: displacement degree must be 1 or 2");
  BoxCoupling B{}; B.k = k_u;
  for (int d = 0; d < 3; ++d) { B.n[d] = d < dim ? box.n[d] : 1; B.h[d] = d < dim ? box.h[d] : 1.0; }
  static const double gx[3] = {0.5 - 0.3872983346207417, 0.5, 0.5 + 0.3872983346207417}, gw[3] = {5.0 / 18, 8.0 / 18, 5.0 / 18};   // exact to degree 5
  const int ns = k_u + 1;
  for (int s = 0; s < ns; ++s)
    for (int t = 0; t < 2; ++t) {
      double vn = 0, vd = 0;
      for (int q = 0; q < 3; ++q) {
        const double x = gx[q];
        double phi = 1, dphi = 0;                        // Lagrange basis s on nodes j / k
        for (int j = 0; j < ns; ++j) if (j != s) phi *= (x - (double)j / k_u) / ((double)(s - j) / k_u);
        for (int m = 0; m < ns; ++m) if (m != s) { double term = 1.0 / ((double)(s - m) / k_u); for (int j = 0; j < ns; ++j) if (j != s && j != m) term *= (x - (double)j / k_u) / ((double)(s - j) / k_u); dphi += term; }
        const double psi = t ? x : 1 - x;
        vn += gw[q] * phi * psi; vd += gw[q] * dphi * psi;
      }
      B.N[s][t] = vn; B.D[s][t] = vd;
    }
  return B;
}

void box_rhs_u(hipStream_t s, int dim, const BoxCoupling &B, double alpha, const double *p, const double *lift, const double *neu, const uint8_t *mask, double *rhs) {
  int64_t nn = 1; for (int d = 0; d < dim; ++d) nn *= (int64_t)B.k * B.n[d] + 1;
  const unsigned grid = (unsigned)((nn + kTB - 1) / kTB);
  if (dim == 2 && B.k == 1) hipLaunchKernelGGL((k_box_rhs_u<2, 1>), grid, kTB, 0, s, B, alpha, p, lift, neu, mask, rhs);
  else if (dim == 2) hipLaunchKernelGGL((k_box_rhs_u<2, 2>), grid, kTB, 0, s, B, alpha, p, lift, neu, mask, rhs);
  else if (B.k == 1) hipLaunchKernelGGL((k_box_rhs_u<3, 1>), grid, kTB, 0, s, B, alpha, p, lift, neu, mask, rhs);
  else hipLaunchKernelGGL((k_box_rhs_u<3, 2>), grid, kTB, 0, s, B, alpha, p, lift, neu, mask, rhs);
}

void box_proj_rhs(hipStream_t s, int dim, const BoxCoupling &B, const double *u, int n_comp, const int32_t *tensor_components, double *const *rhs) {
  int64_t nn = 1; for (int d = 0; d < dim; ++d) nn *= (int64_t)B.n[d] + 1;
  BoxProjOut out{}; out.n = n_comp;
  for (int e = 0; e < n_comp; ++e) { out.rhs[e] = rhs[e]; out.t1[e] = tensor_components[e] / dim; out.t2[e] = tensor_components[e] % dim; }
  const unsigned grid = (unsigned)((nn + kTB - 1) / kTB);
  if (dim == 2 && B.k == 1) hipLaunchKernelGGL((k_box_proj_rhs<2, 1>), grid, kTB, 0, s, B, u, out);
  else if (dim == 2) hipLaunchKernelGGL((k_box_proj_rhs<2, 2>), grid, kTB, 0, s, B, u, out);
  else if (B.k == 1) hipLaunchKernelGGL((k_box_proj_rhs<3, 1>), grid, kTB, 0, s, B, u, out);
  else hipLaunchKernelGGL((k_box_proj_rhs<3, 2>), grid, kTB, 0, s, B, u, out);
}

}  // namespace poro
