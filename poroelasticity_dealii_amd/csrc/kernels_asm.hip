// Cell-local finite-element assembly kernels (gfx950).  One workgroup per cell; the cell's dof indices,
// vertex coordinates, J^-1 / JxW at every quadrature point and the real-space shape gradients are staged
// in LDS, the bilinear forms are evaluated from LDS, and the results are scattered into CSR / vectors.
// Cells are processed colour by colour (no two cells of a colour share a dof), so the scatter needs no
// atomics and the result is bitwise reproducible; inside a cell each wavefront owns whole rows of the
// cell matrix, so one CSR row segment is updated by consecutive lanes of one wave.
//   K-asm-u   PoroElasticDisplacementSolver::assemble_system matrix part   (:216-246, :279-286)
//   K-rhs-u   ... right-hand-side part (alpha p div phi_i :230-234, Neumann :249-277)
//   K-asm-p   MatrixCreator::create_mass_matrix / create_laplace_matrix    (PoroElasticPressureSolver.h:96-101)
//   K-src-p   VectorTools::create_right_hand_side with SinglePhaseWell     (:142-147, right_hand_side.h:99-116)
//   K-proj    StrainProjector::assemble_projection_rhs                     (StrainProjector.h:109-198)
#include "common.hpp"

namespace poro {
namespace {

constexpr int kMaxNq = 27, kMaxDpc = 81, kMaxNv = 8;


// MappingQ1: J_ab = sum_v X_v[a] dN_v/dxi_b ; returns det J, writes J^-1
template <int DIM> __device__ inline double jacobian_inverse(const double *X /*[nv][DIM]*/, const double *dN /*[nv][DIM]*/, double *Ji /*[DIM*DIM]*/) {
  constexpr int NV = 1 << DIM;
  double J[DIM][DIM];
#pragma unroll
  for (int a = 0; a < DIM; ++a)
#pragma unroll
    for (int b = 0; b < DIM; ++b) {
      double s = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) s += X[v * DIM + a] * dN[v * DIM + b];
      J[a][b] = s;
    }
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

__device__ inline int64_t csr_find(const int32_t *col, int64_t b, int64_t e, int32_t c) {
  while (b < e) { const int64_t m = (b + e) >> 1; if (col[m] < c) b = m + 1; else e = m; }
  return b;
}

// ---- K-asm-u: cell stiffness matrix + Dirichlet elimination + CSR scatter + lifting vector -------------
// mode 0: scatter into CSR (constraints applied) and accumulate lift_i = -sum_{j constrained} K_ij g_j
// mode 1: dump the raw cell matrix of `single_cell` into Ke (reference element matrix of the matrix-free operator)
//
// The reference evaluates (C:eps_i):eps_j for every (i, j, q) (:230-242, ConstitutiveModel.h:9-57).  With i = (s_i, c_i), j = (s_j, c_j)
//   K_ij = lambda H^{c_i c_j}[s_i][s_j] + G H^{c_j c_i}[s_i][s_j] + delta(c_i, c_j) G sum_d H^{dd}[s_i][s_j],
//   H^{ab}[s][t] = sum_q JxW_q d_a phi_s(x_q) d_b phi_t(x_q)          (dim^2 small dense products, K = n_q)
// so the quadrature work is dim^2 GEMMs of n_s x n_q x n_s per cell instead of dpc^2 n_q contractions (6x fewer flops for Q2 hexes),
// and it is GEMM-shaped: it runs on the matrix cores (v_mfma_f64_16x16x4_f64), operands from LDS.  The CSR position of an entry is
// looked up once per scalar node pair when the numbering is node-interleaved (dof = node * dim + component: then the dim rows of a
// node share one column pattern made of dim-wide groups), which cuts the binary searches by dim^2.
typedef double v4d __attribute__((ext_vector_type(4)));
struct AsmSmem {   // carve-up of the dynamic LDS block (doubles first, then 8 / 4 / 1-byte arrays)
  double *G, *H, *Ji, *JxW, *X, *Gval; int64_t *Rp; int32_t *Dof, *Slot; uint8_t *Dir;
  __host__ __device__ static size_t bytes(int dim, int ns, int nq, int nv) {
    const int dpc = ns * dim;
    return sizeof(double) * ((size_t)nq * ns * dim + (size_t)dim * dim * ns * ns + (size_t)nq * dim * dim + nq + (size_t)nv * dim + dpc) + sizeof(int64_t) * dpc +
           sizeof(int32_t) * ((size_t)dpc + (size_t)ns * ns) + dpc + 16;
  }
  __device__ AsmSmem(double *base, int dim, int ns, int nq, int nv) {
    const int dpc = ns * dim;
    G = base; H = G + (size_t)nq * ns * dim; Ji = H + (size_t)dim * dim * ns * ns; JxW = Ji + (size_t)nq * dim * dim; X = JxW + nq; Gval = X + (size_t)nv * dim;
    Rp = reinterpret_cast<int64_t *>(Gval + dpc); Dof = reinterpret_cast<int32_t *>(Rp + dpc); Slot = Dof + dpc; Dir = reinterpret_cast<uint8_t *>(Slot + (size_t)ns * ns);
  }
};
template <int DIM, int NT> __global__ void __launch_bounds__(NT)
k_asm_u_matrix(AsmArgs a, const int32_t *__restrict__ cells, int32_t single_cell, int mode, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
               double *__restrict__ val, double *__restrict__ lift, double *__restrict__ Ke, int interleaved) {
  extern __shared__ double smem_asm[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t cell = mode == 1 ? single_cell : cells[blockIdx.x];
  const int nq = a.fe.nq_u, ns = a.ns_u, dpc = a.dpc_u, nv = a.nv;
  AsmSmem S(smem_asm, DIM, ns, nq, nv);
  for (int i = tid; i < nv * DIM; i += nt) S.X[i] = a.cell_X[cell * nv * DIM + i];
  for (int i = tid; i < dpc; i += nt) {
    const int32_t dof = a.cell_dofs_u[cell * dpc + i];
    S.Dof[i] = dof; const uint8_t m = a.dir_mask[dof]; S.Dir[i] = m; S.Gval[i] = m ? a.dir_val[dof] : 0.0;
    if (mode == 0) S.Rp[i] = rp[dof];
  }
  __syncthreads();
  for (int q = tid; q < nq; q += nt) {
    const double det = jacobian_inverse<DIM>(S.X, a.fe.dq1_qu + (size_t)q * nv * DIM, S.Ji + q * DIM * DIM);
    S.JxW[q] = det * a.fe.w_qu[q];
  }
  // CSR slots of the scalar node pairs (independent of the quadrature work; its global loads overlap with the next phases)
  if (mode == 0 && interleaved)
    for (int e = tid; e < ns * ns; e += nt) {
      const int32_t r0 = S.Dof[(e / ns) * DIM];
      S.Slot[e] = (int32_t)(csr_find(col, rp[r0], rp[r0 + 1], S.Dof[(e % ns) * DIM]) - rp[r0]);
    }
  __syncthreads();
  for (int idx = tid; idx < nq * ns; idx += nt) {          // real-space gradients [q][s][d]
    const int q = idx / ns;
    const double *gr = a.fe.du_qu + (size_t)idx * DIM, *Ji = S.Ji + q * DIM * DIM;
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      double g = 0;
#pragma unroll
      for (int b = 0; b < DIM; ++b) g += Ji[b * DIM + c] * gr[b];
      S.G[idx * DIM + c] = g;
    }
  }
  __syncthreads();
  // H^{ab} = (JxW G_a)^T G_b on 16 x 16 output tiles; MFMA operand layout: A[i][k], B[k][j] with i | j = lane & 15, k = lane >> 4;
  // D register r of a lane = row (lane >> 4) + 4 r, column lane & 15
  {
    const int lane = tid & 63, w = tid >> 6, nw = nt >> 6, T = (ns + 15) / 16, ntile = DIM * DIM * T * T;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = w; t < ntile; t += nw) {
      const int ab = t / (T * T), ti = (t / T) % T, tj = t % T, ca = ab / DIM, cb = ab % DIM;
      const int si = ti * 16 + li, sj = tj * 16 + li;
      v4d acc = {0, 0, 0, 0};
      for (int q0 = 0; q0 < nq; q0 += 4) {
        const int q = q0 + lk; const bool qv = q < nq;
        const double av = (qv && si < ns) ? S.G[(q * ns + si) * DIM + ca] * S.JxW[q] : 0.0;
        const double bv = (qv && sj < ns) ? S.G[(q * ns + sj) * DIM + cb] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + lk + 4 * r, cc = tj * 16 + li;
        if (row < ns && cc < ns) S.H[(size_t)ab * ns * ns + row * ns + cc] = acc[r];
      }
    }
  }
  __syncthreads();
  const double lam = a.mat.lame_lambda, G = a.mat.shear_G;
  auto entry = [&](int i, int j) {
    const int si = i / DIM, ci = i % DIM, sj = j / DIM, cj = j % DIM, o = si * ns + sj, nn = ns * ns;
    double t = lam * S.H[(ci * DIM + cj) * nn + o] + G * S.H[(cj * DIM + ci) * nn + o];
    if (ci == cj) {
      double tr = 0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) tr += S.H[(d * DIM + d) * nn + o];
      t += G * tr;
    }
    return t;
  };
  if (mode == 1) {
    for (int e = tid; e < dpc * dpc; e += nt) Ke[e] = entry(e / dpc, e % dpc);
    return;
  }
  auto pos = [&](int i, int j) -> int64_t {
    if (interleaved) return S.Rp[i] + S.Slot[(i / DIM) * ns + (j / DIM)] + (j % DIM);
    const int32_t r = S.Dof[i]; return csr_find(col, S.Rp[i], rp[r + 1], S.Dof[j]);
  };
  // read-modify-write of the CSR values in batches of 8 independent entries per thread (cells of one colour share no dof and the
  // (i, j) pairs of a cell are distinct, so the positions never collide): the loads of a batch are in flight together
  constexpr int U = 8;
  for (int e0 = tid; e0 < dpc * dpc; e0 += nt * U) {
    int64_t ps[U]; double vs[U], old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * nt; ps[u] = -1; vs[u] = 0;
      if (e >= dpc * dpc) continue;
      const int i = e / dpc, j = e % dpc;
      if (S.Dir[i]) { if (i == j) { ps[u] = pos(i, i); vs[u] = fabs(entry(i, i)); } continue; }
      if (S.Dir[j]) continue;
      ps[u] = pos(i, j); vs[u] = entry(i, j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) old[u] = ps[u] >= 0 ? val[ps[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) if (ps[u] >= 0) val[ps[u]] = old[u] + vs[u];
  }
  // lifting of the inhomogeneous Dirichlet values: rhs_i -= sum_{j constrained} K_ij g_j (distribute_local_to_global, :281-286)
  for (int i = tid; i < dpc; i += nt) {
    if (S.Dir[i]) continue;
    double s = 0; bool any = false;
    for (int j = 0; j < dpc; ++j) if (S.Dir[j] && S.Gval[j] != 0.0) { s += entry(i, j) * S.Gval[j]; any = true; }
    if (any) lift[S.Dof[i]] -= s;
  }
}

// ---- K-rhs-u: b_i += int alpha p_h div(phi_i)  (one wave per cell) -------------------------------------------
template <int DIM> __global__ void __launch_bounds__(64)
k_asm_u_rhs(AsmArgs a, const int32_t *__restrict__ cells, const double *__restrict__ p, double *__restrict__ rhs) {
  __shared__ double sJi[kMaxNq * DIM * DIM];
  __shared__ double sC[kMaxNq];
  __shared__ double sX[kMaxNv * DIM];
  __shared__ double sP[kMaxNv];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t cell = cells[blockIdx.x];
  const int nq = a.fe.nq_u, ns = a.ns_u, dpc = a.dpc_u, nv = a.nv;
  for (int i = tid; i < nv * DIM; i += nt) sX[i] = a.cell_X[cell * nv * DIM + i];
  for (int i = tid; i < nv; i += nt) sP[i] = p[a.cell_dofs_p[cell * nv + i]];
  __syncthreads();
  for (int q = tid; q < nq; q += nt) {
    const double det = jacobian_inverse<DIM>(sX, a.fe.dq1_qu + (size_t)q * nv * DIM, sJi + q * DIM * DIM);
    double ph = 0;                                               // pressure_fe_values.get_function_values (:211-212)
    for (int k = 0; k < nv; ++k) ph += sP[k] * a.fe.q1_qu[q * nv + k];
    sC[q] = a.mat.biot_alpha * ph * (det * a.fe.w_qu[q]);
  }
  __syncthreads();
  for (int i = tid; i < dpc; i += nt) {
    const int s = i / DIM, c = i % DIM;
    double acc = 0;
    for (int q = 0; q < nq; ++q) {
      const double *gr = a.fe.du_qu + (size_t)(q * ns + s) * DIM, *Ji = sJi + q * DIM * DIM;
      double g = 0;                                               // trace(eps(phi_i)) = d phi_s / d x_c
#pragma unroll
      for (int b = 0; b < DIM; ++b) g += Ji[b * DIM + c] * gr[b];
      acc += sC[q] * g;
    }
    rhs[a.cell_dofs_u[cell * dpc + i]] += acc;
  }
}

// ---- Neumann faces: b_i += phi_i t_l n_c JxW_f (:249-277); few faces, constant in time -> float atomics are fine ----
template <int DIM> __global__ void k_asm_u_neumann(AsmArgs a, int64_t n_bf, const int32_t *bf_cell, const int32_t *bf_local, const int32_t *bf_id, int n_neu,
                                                   const int32_t *label, const int32_t *comp, const double *value, double *rhs) {
  const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= n_bf * n_neu) return;
  const int64_t bf = item / n_neu; const int l = (int)(item % n_neu);
  if (bf_id[bf] != label[l]) return;
  constexpr int NV = 1 << DIM;
  const int64_t cell = bf_cell[bf];
  const int f = bf_local[bf], nd = f / 2, side = f % 2, ci = comp[l];
  double X[NV * DIM];
  for (int i = 0; i < NV * DIM; ++i) X[i] = a.cell_X[cell * NV * DIM + i];
  const int nqf = a.fe.nq_f, ns = a.ns_u, dpc = a.dpc_u;
  for (int q = 0; q < nqf; ++q) {
    const double *dN = a.fe.dq1_qf + (size_t)(f * nqf + q) * NV * DIM;
    double J[DIM][DIM];
    for (int r = 0; r < DIM; ++r) for (int b = 0; b < DIM; ++b) { double s = 0; for (int v = 0; v < NV; ++v) s += X[v * DIM + r] * dN[v * DIM + b]; J[r][b] = s; }
    double cof[DIM];   // det(J) J^-T e_nd
    if constexpr (DIM == 2) { if (nd == 0) { cof[0] = J[1][1]; cof[1] = -J[0][1]; } else { cof[0] = -J[1][0]; cof[1] = J[0][0]; } }
    else { const int a1 = (nd + 1) % 3, a2 = (nd + 2) % 3; for (int r = 0; r < 3; ++r) { const int r1 = (r + 1) % 3, r2 = (r + 2) % 3; cof[r] = J[r1][a1] * J[r2][a2] - J[r1][a2] * J[r2][a1]; } }
    double len = 0; for (int r = 0; r < DIM; ++r) len += cof[r] * cof[r]; len = sqrt(len);
    const double sgn = side ? 1.0 : -1.0;
    const double neumann_value = value[l] * (sgn * cof[ci] / len), jxwf = len * a.fe.w_qf[q];
    for (int s = 0; s < ns; ++s) {
      const double phi = a.fe.u_qf[(size_t)(f * nqf + q) * ns + s];
      if (phi != 0.0) atomicAdd(&rhs[a.cell_dofs_u[cell * dpc + s * DIM + ci]], phi * neumann_value * jxwf);
    }
  }
}

// ---- K-asm-p + K-src-p: Q1 mass / Laplace matrices and the well source integral (one wave per cell) -----------
template <int DIM> __global__ void __launch_bounds__(64)
k_asm_p(AsmArgs a, const int32_t *__restrict__ cells, const int64_t *__restrict__ rp, const int32_t *__restrict__ col, double *__restrict__ M,
        double *__restrict__ K, double *__restrict__ src) {
  constexpr int NV = 1 << DIM;
  __shared__ double sG[NV * NV * DIM];   // [q][v][d]
  __shared__ double sJi[NV * DIM * DIM];
  __shared__ double sJxW[NV], sS[NV];
  __shared__ double sX[NV * DIM];
  __shared__ int32_t sDof[NV];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t cell = cells[blockIdx.x];
  const int nq = a.fe.nq_p;
  for (int i = tid; i < NV * DIM; i += nt) sX[i] = a.cell_X[cell * NV * DIM + i];
  for (int i = tid; i < NV; i += nt) sDof[i] = a.cell_dofs_p[cell * NV + i];
  __syncthreads();
  for (int q = tid; q < nq; q += nt) {
    const double det = jacobian_inverse<DIM>(sX, a.fe.dq1_qp + (size_t)q * NV * DIM, sJi + q * DIM * DIM);
    sJxW[q] = det * a.fe.w_qp[q];
    double xq[DIM];
    for (int c = 0; c < DIM; ++c) { double s = 0; for (int v = 0; v < NV; ++v) s += sX[v * DIM + c] * a.fe.q1_qp[q * NV + v]; xq[c] = s; }
    // SinglePhaseWell::value: z-axis cylinder, pi = 3.1415926 (right_hand_side.h:106-109)
    const double r2 = xq[0] * xq[0] + xq[1] * xq[1];
    sS[q] = (r2 <= a.mat.r_well * a.mat.r_well) ? -a.mat.flow_rate / (3.1415926 * a.mat.r_well * a.mat.r_well) : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < nq * NV; idx += nt) {
    const int q = idx / NV;
    const double *gr = a.fe.dq1_qp + (size_t)idx * DIM, *Ji = sJi + q * DIM * DIM;
    for (int c = 0; c < DIM; ++c) { double g = 0; for (int b = 0; b < DIM; ++b) g += Ji[b * DIM + c] * gr[b]; sG[idx * DIM + c] = g; }
  }
  __syncthreads();
  for (int e = tid; e < NV * NV; e += nt) {
    const int i = e / NV, j = e % NV;
    double m = 0, k = 0;
    for (int q = 0; q < nq; ++q) {
      m += a.fe.q1_qp[q * NV + i] * a.fe.q1_qp[q * NV + j] * sJxW[q];
      double dot = 0;
      for (int c = 0; c < DIM; ++c) dot += sG[(q * NV + i) * DIM + c] * sG[(q * NV + j) * DIM + c];
      k += dot * sJxW[q];
    }
    const int32_t r = sDof[i];
    const int64_t pos = csr_find(col, rp[r], rp[r + 1], sDof[j]);
    M[pos] += m; K[pos] += k;
  }
  for (int i = tid; i < NV; i += nt) {
    double s = 0;
    for (int q = 0; q < nq; ++q) s += a.fe.q1_qp[q * NV + i] * sS[q] * sJxW[q];
    src[sDof[i]] += s;
  }
}

// ---- K-proj: r^c_i += sum_q phi^p_i eps_c(u_h)(x_q) JxW (one wave per cell) -----------------------------------
struct ProjOut { double *rhs[6]; int comp[6]; int n; };
template <int DIM> __global__ void __launch_bounds__(64)
k_proj_rhs(AsmArgs a, const int32_t *__restrict__ cells, const double *__restrict__ u, ProjOut out) {
  constexpr int NV = 1 << DIM;
  __shared__ double sJi[NV * DIM * DIM];
  __shared__ double sJxW[NV];
  __shared__ double sX[NV * DIM];
  __shared__ double sU[kMaxDpc];
  __shared__ double sGrad[NV * DIM * DIM];   // [q][component][direction]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t cell = cells[blockIdx.x];
  const int nq = a.fe.nq_p, ns = a.ns_u, dpc = a.dpc_u;
  for (int i = tid; i < NV * DIM; i += nt) sX[i] = a.cell_X[cell * NV * DIM + i];
  for (int i = tid; i < dpc; i += nt) sU[i] = u[a.cell_dofs_u[cell * dpc + i]];
  __syncthreads();
  for (int q = tid; q < nq; q += nt) {
    const double det = jacobian_inverse<DIM>(sX, a.fe.dq1_qp + (size_t)q * NV * DIM, sJi + q * DIM * DIM);
    sJxW[q] = det * a.fe.w_qp[q];
  }
  __syncthreads();
  // get_function_gradients (StrainProjector.h:164-165): grad[q][comp][dir] = sum_s u[s,comp] dphi_s/dx_dir
  for (int idx = tid; idx < nq * DIM * DIM; idx += nt) {
    const int q = idx / (DIM * DIM), comp = (idx / DIM) % DIM, dir = idx % DIM;
    const double *Ji = sJi + q * DIM * DIM;
    double acc = 0;
    for (int s = 0; s < ns; ++s) {
      const double *gr = a.fe.du_qp + (size_t)(q * ns + s) * DIM;
      double g = 0;
      for (int b = 0; b < DIM; ++b) g += Ji[b * DIM + dir] * gr[b];
      acc += sU[s * DIM + comp] * g;
    }
    sGrad[idx] = acc;
  }
  __syncthreads();
  for (int e = tid; e < out.n * NV; e += nt) {
    const int c = e / NV, i = e % NV;
    const int t1 = out.comp[c] / DIM, t2 = out.comp[c] % DIM;   // StrainProjector.h:177-181
    double acc = 0;
    for (int q = 0; q < nq; ++q) {
      const double *g = sGrad + q * DIM * DIM;
      const double strain = (t1 == t2) ? g[t1 * DIM + t1] : (g[t1 * DIM + t2] + g[t2 * DIM + t1]) / 2;   // ConstitutiveModel.h:27-42
      acc += a.fe.q1_qp[q * NV + i] * strain * sJxW[q];
    }
    out.rhs[c][a.cell_dofs_p[cell * NV + i]] += acc;
  }
}

}  // namespace

#define PORO_DIM_DISPATCH(dim, CALL2, CALL3) do { if ((dim) == 2) { CALL2; } else { CALL3; } } while (0)

// workgroup size by cell-matrix size: the phases parallelise over n_s^2 .. dpc^2 items, and LDS (73 KB for Q2 hexes) allows two
// workgroups per CU, so big cells take 512 threads (4 waves / SIMD resident) and small ones fewer (measured, tools/asm_bench.py)
template <int DIM, int NT> static void launch_asm_u_nt(hipStream_t s, const AsmArgs &a, unsigned grid, const int32_t *cells, int32_t cell, int mode, const int64_t *rp, const int32_t *col,
                                                       double *val, double *lift, double *Ke, int interleaved) {
  const size_t lds = AsmSmem::bytes(DIM, a.ns_u, a.fe.nq_u, a.nv);
  // Q2 hexes need 73 KB of LDS per workgroup: above the 64 KB default of dynamic LDS.  Function attributes are per device: one flag per device index
  static bool attr_set[64] = {false}; int dev = 0; (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_asm_u_matrix<DIM, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (dev >= 0 && dev < 64) attr_set[dev] = true; }
  hipLaunchKernelGGL((k_asm_u_matrix<DIM, NT>), grid, NT, lds, s, a, cells, cell, mode, rp, col, val, lift, Ke, interleaved);
}
template <int DIM> static void launch_asm_u(hipStream_t s, const AsmArgs &a, unsigned grid, const int32_t *cells, int32_t cell, int mode, const int64_t *rp, const int32_t *col, double *val,
                                            double *lift, double *Ke, int interleaved) {
  int nt = a.dpc_u >= 64 ? 512 : a.dpc_u >= 24 ? 128 : 64;
  if (const char *e = std::getenv("PORO_ASM_THREADS")) nt = std::atoi(e);
  switch (nt) {
    case 64: launch_asm_u_nt<DIM, 64>(s, a, grid, cells, cell, mode, rp, col, val, lift, Ke, interleaved); break;
    case 128: launch_asm_u_nt<DIM, 128>(s, a, grid, cells, cell, mode, rp, col, val, lift, Ke, interleaved); break;
    case 512: launch_asm_u_nt<DIM, 512>(s, a, grid, cells, cell, mode, rp, col, val, lift, Ke, interleaved); break;
    case 1024: launch_asm_u_nt<DIM, 1024>(s, a, grid, cells, cell, mode, rp, col, val, lift, Ke, interleaved); break;
    default: launch_asm_u_nt<DIM, 256>(s, a, grid, cells, cell, mode, rp, col, val, lift, Ke, interleaved);
  }
}
void asm_u_matrix(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n, const int64_t *rp, const int32_t *col, double *val, double *lift) {
  if (!n) return;
  if (a.dim == 2) launch_asm_u<2>(s, a, (unsigned)n, cells, 0, 0, rp, col, val, lift, nullptr, a.interleaved_u);
  else launch_asm_u<3>(s, a, (unsigned)n, cells, 0, 0, rp, col, val, lift, nullptr, a.interleaved_u);
}
void asm_u_element_matrix(hipStream_t s, const AsmArgs &a, int32_t cell, double *Ke) {
  if (a.dim == 2) launch_asm_u<2>(s, a, 1, nullptr, cell, 1, nullptr, nullptr, nullptr, nullptr, Ke, 0);
  else launch_asm_u<3>(s, a, 1, nullptr, cell, 1, nullptr, nullptr, nullptr, nullptr, Ke, 0);
}
void asm_u_rhs(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n, const double *p, double *rhs) {
  if (!n) return;
  PORO_DIM_DISPATCH(a.dim, hipLaunchKernelGGL(k_asm_u_rhs<2>, (unsigned)n, 64, 0, s, a, cells, p, rhs), hipLaunchKernelGGL(k_asm_u_rhs<3>, (unsigned)n, 64, 0, s, a, cells, p, rhs));
}
void asm_u_neumann(hipStream_t s, const AsmArgs &a, int64_t n_bf, const int32_t *bc, const int32_t *bl, const int32_t *bi, int n_neu, const int32_t *label,
                   const int32_t *comp, const double *value, double *rhs) {
  const int64_t items = n_bf * n_neu;
  if (!items) return;
  const unsigned grid = (unsigned)((items + 63) / 64);
  PORO_DIM_DISPATCH(a.dim, hipLaunchKernelGGL(k_asm_u_neumann<2>, grid, 64, 0, s, a, n_bf, bc, bl, bi, n_neu, label, comp, value, rhs),
                    hipLaunchKernelGGL(k_asm_u_neumann<3>, grid, 64, 0, s, a, n_bf, bc, bl, bi, n_neu, label, comp, value, rhs));
}
void asm_p_matrices(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n, const int64_t *rp, const int32_t *col, double *M, double *K, double *src) {
  if (!n) return;
  PORO_DIM_DISPATCH(a.dim, hipLaunchKernelGGL(k_asm_p<2>, (unsigned)n, 64, 0, s, a, cells, rp, col, M, K, src), hipLaunchKernelGGL(k_asm_p<3>, (unsigned)n, 64, 0, s, a, cells, rp, col, M, K, src));
}
void asm_proj_rhs(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n, const double *u, int n_comp, const int32_t *comps, double *const *rhs) {
  if (!n || !n_comp) return;
  ProjOut out{}; out.n = n_comp;
  for (int c = 0; c < n_comp; ++c) { out.rhs[c] = rhs[c]; out.comp[c] = comps[c]; }
  PORO_DIM_DISPATCH(a.dim, hipLaunchKernelGGL(k_proj_rhs<2>, (unsigned)n, 64, 0, s, a, cells, u, out), hipLaunchKernelGGL(k_proj_rhs<3>, (unsigned)n, 64, 0, s, a, cells, u, out));
}

}  // namespace poro
