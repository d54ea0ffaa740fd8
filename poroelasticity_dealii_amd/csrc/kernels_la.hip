// Sparse / dense linear-algebra kernels of the Krylov solves (gfx950, wave64):
//   K-spmv  CSR fp64 SpMV (int32 columns, int64 row pointers), sub-wave row groups, shuffle reduction
//   K-vec   fused PCG vector updates with deterministic two-stage reductions (no float atomics)
//   K-res-p fused pressure residual, K-jac-p Jacobian value AXPY
// All of these are HBM-bound streaming kernels: 8-16 B per lane coalesced accesses, grid-stride loops,
// grids capped at kMaxPartials blocks for the reducing kernels so block partials fit one 8 KB slab.
// Replaces deal.II SparseMatrix::vmult, Vector ops and SolverCG's inner loop
// (PoroElasticDisplacementSolver.h:300-305, PoroElasticPressureSolver.h:122-153,162-167,176-179).
#include "common.hpp"
#include "device_reduce.hpp"

namespace poro {
namespace {

__global__ void k_post(Mailbox *mb, unsigned long long seq, const double *src, int n, const PcgScalars *sc) {
  if ((int)threadIdx.x < n) mb->vals[threadIdx.x] = src[threadIdx.x];
  if (sc && threadIdx.x == 32) mb->sc = *sc;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(const_cast<unsigned long long *>(&mb->seq), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_fill(double *x, double v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) x[i] = v;
}
__global__ void k_axpy(double *y, double a, const double *x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) y[i] += a * x[i];
}
__global__ void k_dot(const double *a, const double *b, int64_t n, double *partials, const PcgScalars *gate) {
  __shared__ double sh[4];
  if (gate && (gate->done | gate->finishing)) return;
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) s += a[i] * b[i];
  s = block_sum(s, sh);
  store_partial(partials, s);
}
__global__ void k_norm(const double *a, int64_t n, double *p2, double *pinf) {
  __shared__ double sh[4];
  double s = 0, m = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) { const double v = a[i]; s += v * v; m = fmax(m, fabs(v)); }
  s = block_sum(s, sh); m = block_max(m, sh);
  store_partial(p2, s); store_partial(pinf, m);
}
// one block; set k is a max-reduction when bit k of max_mask is set, else a sum
__global__ void k_reduce_finish(const double *partials, int n_sets, double *red, int max_mask) {
  __shared__ double sh[4];
  for (int k = 0; k < n_sets; ++k) {
    const bool is_max = (max_mask >> k) & 1;
    double v = 0;
    for (int i = threadIdx.x; i < kMaxPartials; i += kBlock) { const double t = partials[k * kMaxPartials + i]; v = is_max ? fmax(v, t) : v + t; }
    v = is_max ? block_max(v, sh) : block_sum(v, sh);
    if (threadIdx.x == 0) red[k] = v;
  }
}

template <int L> __global__ void k_spmv(int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ col, const double *__restrict__ val,
                                        const double *__restrict__ x, double *__restrict__ y) {
  const int lane = threadIdx.x % L;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (row >= n) return;   // whole L-groups leave together (kBlock % L == 0)
  const int64_t b = rp[row], e = rp[row + 1];
  double s = 0;
  // 4 (value, column) pairs per lane are requested before the first gather of x is issued: two dependent memory latencies per trip
  // instead of two per element (a Q2 hex row has 189 entries = one trip of a 64-lane group)
  constexpr int U = 4;
  for (int64_t j0 = b + lane; j0 < e; j0 += (int64_t)L * U) {
    double v[U]; int32_t c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t j = j0 + (int64_t)u * L; const bool ok = j < e; v[u] = ok ? val[j] : 0.0; c[u] = col[ok ? j : b]; }
#pragma unroll
    for (int u = 0; u < U; ++u) s = fma(v[u], x[c[u]], s);
  }
#pragma unroll
  for (int off = L / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, L);
  if (lane == 0) y[row] = s;
}
template <int L> __global__ void k_residual(int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ col, const double *__restrict__ M,
                                            const double *__restrict__ K, double kappa, const double *__restrict__ t, const double *__restrict__ p,
                                            const double *__restrict__ src, double *__restrict__ R) {
  const int lane = threadIdx.x % L;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / L;
  if (row >= n) return;
  const int64_t b = rp[row], e = rp[row + 1];
  double s1 = 0, s2 = 0;
  constexpr int U = 4;                                    // as k_spmv: the loads of a trip are issued before its gathers
  for (int64_t j0 = b + lane; j0 < e; j0 += (int64_t)L * U) {
    double vm[U], vk[U]; int32_t c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t j = j0 + (int64_t)u * L; const bool ok = j < e; vm[u] = ok ? M[j] : 0.0; vk[u] = ok ? K[j] : 0.0; c[u] = col[ok ? j : b]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { s1 = fma(vm[u], t[c[u]], s1); s2 = fma(vk[u], p[c[u]], s2); }
  }
#pragma unroll
  for (int off = L / 2; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, L); s2 += __shfl_xor(s2, off, L); }
  // same association as the reference: residual = M t; tmp = (K p) * kappa; residual += tmp; += source; *= -1
  if (lane == 0) R[row] = -((s1 + s2 * kappa) + src[row]);
}
__global__ void k_pressure_tmp(double *t, const double *ev, const double *ev0, const double *p, const double *po, double c1, double c2, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) t[i] = (ev[i] - ev0[i]) * c1 + (p[i] - po[i]) * c2;
}
__global__ void k_jacobian(double *J, const double *M, const double *K, double a, double kappa, int64_t nnz) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * kBlock) J[i] = M[i] * a + kappa * K[i];
}
__global__ void k_reciprocal(double *y, const double *x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) y[i] = 1.0 / x[i];
}
__global__ void k_pointwise_mul(double *y, const double *x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) y[i] *= x[i];
}
__global__ void k_cons_expand(int64_t n, const int32_t *__restrict__ dof, const int64_t *__restrict__ ptr, const int32_t *__restrict__ master, const double *__restrict__ w,
                              const double *__restrict__ inhom, double *x) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double s = inhom ? inhom[i] : 0.0;
  for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) s += w[k] * x[master[k]];     // masters are unconstrained: no read-after-write inside the launch
  x[dof[i]] = s;
}
__global__ void k_cons_gather(int64_t n_masters, const int32_t *__restrict__ t_master, const int64_t *__restrict__ t_ptr, const int32_t *__restrict__ t_dof,
                              const double *__restrict__ t_w, double *y) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_masters) return;
  double s = 0;
  for (int64_t k = t_ptr[i]; k < t_ptr[i + 1]; ++k) s += t_w[k] * y[t_dof[k]];
  y[t_master[i]] += s;
}
__global__ void k_cons_zero(int64_t n, const int32_t *__restrict__ dof, double *y) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) y[dof[i]] = 0.0;
}
__global__ void k_csr_diag(int64_t n, const int64_t *pos, const double *val, double *diag) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) diag[i] = val[pos[i]];
}
struct StrainPtrs { const double *p[3]; };
__global__ void k_sum_strains(double *ev, StrainPtrs s, int n, int64_t len) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
    double v = 0;
    for (int k = 0; k < n; ++k) v += s.p[k][i];
    ev[i] = v;
  }
}
__global__ void k_set_constrained(double *x, const uint8_t *mask, const double *val, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) if (mask[i]) x[i] = val[i];
}
__global__ void k_rhs_u_finish(double *rhs, const double *lift, const double *neu, const uint8_t *mask, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) rhs[i] = mask[i] ? 0.0 : (rhs[i] + neu[i]) + lift[i];
}

// ---- SparseMatrix::precondition_SSOR in natural row order, level-scheduled: dst = (D + wU)^-1 w(2-w) D (D + wL)^-1 src ------
// one thread per row of the current dependency level; the in-row sum runs in ascending column order like the reference's loop
__global__ void k_ssor_fwd(int64_t n_lvl, const int32_t *__restrict__ rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                           const double *__restrict__ val, const int64_t *__restrict__ dpos, double om, const double *__restrict__ src, double *dst) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_lvl) return;
  const int32_t r = rows[t];
  double s = 0;
  for (int64_t j = rp[r]; j < dpos[r]; ++j) s += val[j] * dst[col[j]];
  dst[r] = (src[r] - s * om) / val[dpos[r]];
}
__global__ void k_ssor_mid(int64_t n, const double *__restrict__ val, const int64_t *__restrict__ dpos, double om, double *dst) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i] *= om * (2. - om) * val[dpos[i]];
}
__global__ void k_ssor_bwd(int64_t n_lvl, const int32_t *__restrict__ rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                           const double *__restrict__ val, const int64_t *__restrict__ dpos, double om, double *dst) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_lvl) return;
  const int32_t r = rows[t];
  double s = 0;
  for (int64_t j = dpos[r] + 1; j < rp[r + 1]; ++j) s += val[j] * dst[col[j]];
  dst[r] = (dst[r] - s * om) / val[dpos[r]];
}
// ILU(0) solves z = U^-1 L^-1 g with the factors stored on A's pattern (unit lower L below the diagonal, U on and above it)
__global__ void k_ilu_fwd(int64_t n_lvl, const int32_t *__restrict__ rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                          const double *__restrict__ lu, const int64_t *__restrict__ dpos, const double *__restrict__ src, double *dst) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_lvl) return;
  const int32_t r = rows[t];
  double s = src[r];
  for (int64_t j = rp[r]; j < dpos[r]; ++j) s -= lu[j] * dst[col[j]];
  dst[r] = s;
}
__global__ void k_ilu_bwd(int64_t n_lvl, const int32_t *__restrict__ rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                          const double *__restrict__ lu, const int64_t *__restrict__ dpos, double *dst) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_lvl) return;
  const int32_t r = rows[t];
  double s = dst[r];
  for (int64_t j = dpos[r] + 1; j < rp[r + 1]; ++j) s -= lu[j] * dst[col[j]];
  dst[r] = s / lu[dpos[r]];
}
// ILU(0) factorisation, level-scheduled like the triangular solves: the rows of one level depend only on rows of earlier levels (launches), so they
// factorise in parallel, ONE WAVE PER ROW.  The row lives in LDS while it is worked on (row-wise IKJ on A's own pattern, the pivots k in ascending order
// exactly as a sequential sweep would take them; the products are rounded separately like the sequential reference, no fused multiply-add): for every
// lower entry a_ik the lanes take the upper entries of row k, find their column in row i by binary search and subtract l_ik u_kj.
constexpr int kIluMaxRow = 512;
__global__ void __launch_bounds__(64)
k_ilu0_level(int64_t n_lvl, const int32_t *__restrict__ rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col, const int64_t *__restrict__ dpos, double *lu, int *flag) {
  __shared__ double sval[kIluMaxRow];
  __shared__ int32_t scol[kIluMaxRow];
  const int lane = threadIdx.x;
  const int32_t i = rows[blockIdx.x];
  const int64_t r0 = rp[i]; const int len = (int)(rp[i + 1] - r0), nlow = (int)(dpos[i] - r0);
  for (int t = lane; t < len; t += 64) { sval[t] = lu[r0 + t]; scol[t] = col[r0 + t]; }
  __syncthreads();
  for (int kk = 0; kk < nlow; ++kk) {
    const int32_t k = scol[kk];
    const double lik = sval[kk] / lu[dpos[k]];
    __syncthreads();                                   // every lane has read a_ik before it is replaced by l_ik
    if (lane == 0) sval[kk] = lik;
    for (int64_t jj = dpos[k] + 1 + lane; jj < rp[k + 1]; jj += 64) {
      const int32_t cj = col[jj];
      int lo = kk + 1, hi = len - 1, pos = -1;           // columns are sorted and cj > k
      while (lo <= hi) { const int mid = (lo + hi) >> 1; const int32_t cm = scol[mid]; if (cm == cj) { pos = mid; break; } if (cm < cj) lo = mid + 1; else hi = mid - 1; }
      if (pos >= 0) sval[pos] = __dsub_rn(sval[pos], __dmul_rn(lik, lu[jj]));
    }
    __syncthreads();
  }
  if (lane == 0 && !(fabs(sval[nlow]) > 0)) atomicMax(flag, i + 1);     // zero pivot: report the (1-based) row
  for (int t = lane; t < len; t += 64) lu[r0 + t] = sval[t];
}
__global__ void k_xpby(double *y, double a, double b, const double *x, int64_t n) {   // y = a y + b x
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) y[i] = a * y[i] + b * x[i];
}

// ---- PCG (deal.II SolverCG structure: g = A x - b, d = -P^-1 g) ----------------------------------
// INVERSE Jacobi diagonal (z = g * dinv: no fp64 division in the streaming kernels) either as a full vector or
// dictionary-compressed: class byte per node (ncomp dofs each) + table[class][comp]
struct DiagRef { const double *full; const uint8_t *cls; const double *tab; int ncomp; const double *z;   // z != null: explicit z = P^-1 g computed by the caller
                 double *z1_out = nullptr; double z1_scale = 0; };
// NC = components per node of the dictionary form (0: full vector); NC is a template parameter so that i / NC is a multiply
template <int NC> __device__ inline double diag_at(const DiagRef &D, int64_t i) {
  if constexpr (NC == 0) return D.full[i];
  else { const uint32_t node = (uint32_t)i / (uint32_t)NC; return D.tab[(uint32_t)D.cls[node] * NC + ((uint32_t)i - node * NC)]; }
}
// inert != null: those dofs (Dirichlet rows) are kept out of the Krylov system - their residual is zero by definition, exactly as in the
// reference, where the constrained rows of A x = b are satisfied identically by the warm start (constraints.distribute)
__global__ void k_pcg_init_residual(double *g, const double *Ax, const double *b, const uint8_t *inert, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) g[i] = (inert && inert[i]) ? 0.0 : Ax[i] - b[i];
}
__global__ void k_mask_zero(double *x, const uint8_t *mask, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) if (mask[i]) x[i] = 0.0;
}
template <int NC> __global__ void k_pcg_first_direction(double *d, const double *g, DiagRef D, int prec, int64_t n, int64_t n_owned, double *partials) {
  __shared__ double sh[4];
  double gg = 0, gz = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double gi = g[i], z = D.z ? D.z[i] : prec ? gi * diag_at<NC>(D, i) : gi;
    d[i] = -z;
    if (i < n_owned) { gg += gi * gi; gz += gi * z; }
  }
  gg = block_sum(gg, sh); gz = block_sum(gz, sh);
  store_partial(partials, gg); store_partial(partials + kMaxPartials, gz);
}
__global__ void k_pcg_dot_dh(const PcgScalars *sc, const double *d, const double *h, int64_t n_owned, double *partials) {
  __shared__ double sh[4];
  if (sc->done) return;
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_owned; i += (int64_t)gridDim.x * kBlock) s += d[i] * h[i];
  s = block_sum(s, sh);
  store_partial(partials, s);
}
__global__ void k_scalars_sum(const PcgScalars *sc, const double *partials, int n_sets, double *red) {
  __shared__ double sh[4];
  if (sc && sc->done) return;
  for (int k = 0; k < n_sets; ++k) {
    double v = 0;
    for (int i = threadIdx.x; i < kMaxPartials; i += kBlock) v += partials[k * kMaxPartials + i];
    v = block_sum(v, sh);
    if (threadIdx.x == 0) red[k] = v;
  }
}
__global__ void k_scalars_start(PcgScalars *sc, const double *red, double abs_tol, double rel_tol, int max_iter, int stop_rule) {
  const double bb = red[0], gg = red[1], gz = red[2];
  sc->tol = fmax(abs_tol, rel_tol * sqrt(stop_rule == PORO_STOP_REDUCTION ? gg : bb));   // relative to ||b|| or to the warm start's residual
  sc->res0 = sc->res = sqrt(gg);
  sc->gg = gg; sc->gz = gz; sc->gh2[0] = gz; sc->gh2[1] = gz; sc->dh = 0; sc->alpha = 0; sc->beta = 0;
  sc->it = 0; sc->max_iter = max_iter;
  sc->converged = sc->res <= sc->tol; sc->done = sc->converged; sc->finishing = 0;
}
// g += alpha h and the partials of g.g, g.z (z = g / diag).  red != null (partitioned run): the all-reduced scalars are read instead
// of the local block partials.  x is NOT touched here: x += alpha d rides in k_pcg_update_d_fused, which reads d anyway.
template <int NC> __global__ void k_pcg_update_g_fused(PcgScalars *sc, int parity, double *g, const double *h, DiagRef D, int prec, int64_t n, int64_t n_owned,
                                     const double *partials_dh, const double *red, double *partials_out) {
  __shared__ double sh[5];
  if (sc->done) return;
  // the previous iteration finished the solve: latch `done` here, one launch later, so that inside the finishing
  // k_pcg_update_d_fused no block could see done = 1 and skip its share of the last x update
  if (sc->finishing) { if (blockIdx.x == 0 && threadIdx.x == 0) sc->done = 1; return; }
  const double dh = red ? red[0] : sum_partials(partials_dh, sh);
  const double alpha = sc->gh2[parity] / dh;
  double gg = 0, gz = 0;
  // a zero reciprocal diagonal marks an inert (Dirichlet) dof: its residual stays exactly zero whatever the operator wrote into h there
  auto one = [&](int64_t i, double &gi) {
    const double Di = diag_at<NC>(D, i);
    if (D.z1_out) D.z1_out[i] = Di == 0.0 ? 0.0 : D.z1_scale * Di * gi;          // first iterate of the polynomial preconditioner rides with the residual update
    if (Di == 0.0) { gi = 0.0; return; }
    if (i < n_owned) { gg += gi * gi; if (!D.z) gz += gi * (prec ? gi * Di : gi); }   // explicit preconditioner: g.z follows in its own dot kernel once z = P^-1 g exists
  };
  // 16-byte accesses (two dofs per lane and step): the arrays are hipMalloc-aligned; an odd tail element goes to one thread
  const int64_t n2 = n >> 1;
  double2 *g2 = reinterpret_cast<double2 *>(g); const double2 *h2 = reinterpret_cast<const double2 *>(h);
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n2; q += (int64_t)gridDim.x * kBlock) {
    double2 gv = g2[q]; const double2 hv = h2[q];
    gv.x = fma(alpha, hv.x, gv.x); gv.y = fma(alpha, hv.y, gv.y);
    one(2 * q, gv.x); one(2 * q + 1, gv.y);
    g2[q] = gv;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { double gi = fma(alpha, h[n - 1], g[n - 1]); one(n - 1, gi); g[n - 1] = gi; }
  gg = block_sum(gg, sh); gz = block_sum(gz, sh);
  store_partial(partials_out, gg); store_partial(partials_out + kMaxPartials, gz);
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->dh = dh; sc->alpha = alpha; }
}
// x += alpha d (always: SolverCG updates x before it checks the residual), then d = beta d - z unless the solve just finished.
// `it` = 1-based index of this iteration, supplied by the host (launch order), so no block depends on a control word that another
// block of the same launch updates; only block 0 writes the control words.
template <int NC> __global__ void k_pcg_update_d_fused(PcgScalars *sc, int parity, int it, double *x, double *d, const double *g, DiagRef D, int prec, int64_t n,
                                     const double *partials_in, const double *red) {
  __shared__ double sh[5];
  if (sc->done) return;
  const double gg = red ? red[0] : sum_partials(partials_in, sh), gz = red ? red[1] : sum_partials(partials_in + kMaxPartials, sh);
  const double res = sqrt(gg), gh_old = sc->gh2[parity], alpha = sc->alpha;
  const bool conv = res <= sc->tol, fail = !conv && it >= sc->max_iter;      // SolverControl::check order: success first, then the cap
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->gg = gg; sc->gz = gz; sc->res = res; sc->it = it; }
  if (conv || fail) {   // every block takes this branch together (identical inputs) and still owes its share of the last x update;
                        // `done` itself is raised by the next launch (k_pcg_update_g_fused), never inside this one
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) x[i] = fma(alpha, d[i], x[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->converged = conv ? 1 : 0; sc->finishing = 1; }
    return;
  }
  const double beta = gz / gh_old;
  const int64_t n2 = n >> 1;
  double2 *d2 = reinterpret_cast<double2 *>(d); const double2 *g2 = reinterpret_cast<const double2 *>(g);
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n2; q += (int64_t)gridDim.x * kBlock) {
    // x is touched once per iteration and nowhere else: non-temporal accesses keep it from evicting d, g and h (3 x 73 MB at config 4),
    // which then stay resident in the 256 MB memory-side cache between the three kernels of an iteration (measured: -19 us / iteration)
    double2 gv{0, 0}; if (!D.z) gv = g2[q];          // (an explicit z makes the residual unnecessary here: one pass less)
    double2 dv = d2[q], xv;
    xv.x = __builtin_nontemporal_load(&x[2 * q]); xv.y = __builtin_nontemporal_load(&x[2 * q + 1]);
    const double z0 = D.z ? D.z[2 * q] : prec ? gv.x * diag_at<NC>(D, 2 * q) : gv.x, z1 = D.z ? D.z[2 * q + 1] : prec ? gv.y * diag_at<NC>(D, 2 * q + 1) : gv.y;
    xv.x = fma(alpha, dv.x, xv.x); xv.y = fma(alpha, dv.y, xv.y);
    dv.x = fma(beta, dv.x, -z0); dv.y = fma(beta, dv.y, -z1);
    __builtin_nontemporal_store(xv.x, &x[2 * q]); __builtin_nontemporal_store(xv.y, &x[2 * q + 1]); d2[q] = dv;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1; const double gi = D.z ? 0.0 : g[i], z = D.z ? D.z[i] : prec ? gi * diag_at<NC>(D, i) : gi, di = d[i];
    x[i] = fma(alpha, di, x[i]); d[i] = fma(beta, di, -z);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->beta = beta; sc->gh2[parity ^ 1] = gz; }
}
template <int NC> __global__ void k_cheb_first(double *z, const double *g, DiagRef D, double scale, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) z[i] = scale * diag_at<NC>(D, i) * g[i];
}
template <int NC> __global__ void k_cheb_step(double *znew, const double *zj, const double *g, const double *Az, DiagRef D, double omega, int64_t n, int64_t n_owned, double *gz_partials) {
  __shared__ double sh[4];
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double gi = g[i];
    const double zn = fma(omega * diag_at<NC>(D, i), gi - Az[i], zj[i]);
    znew[i] = zn;
    if (i < n_owned) acc = fma(gi, zn, acc);
  }
  if (gz_partials) { acc = block_sum(acc, sh); store_partial(gz_partials, acc); }
}
template <int NC> __global__ void k_cheb_fix_planes(double *znew, const double *zj, const double *g, const double *own_lo, const double *nbr_lo, const double *own_hi, const double *nbr_hi,
                                                   DiagRef D, double omega, int64_t n, int64_t plane) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2 * plane) return;
  const bool hi = t >= plane; const int64_t k = hi ? t - plane : t;
  const double *own = hi ? own_hi : own_lo, *nbr = hi ? nbr_hi : nbr_lo;
  if (!own) return;
  const int64_t i = hi ? n - plane + k : k;
  znew[i] = fma(omega * diag_at<NC>(D, i), g[i] - (own[k] + nbr[k]), zj[i]);      // own + neighbour: the same two numbers on both ranks, and the sum commutes
}
template <class F> void dispatch_lanes(int L, F &&f) {
  switch (L) {
    case 2: f(std::integral_constant<int, 2>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    case 8: f(std::integral_constant<int, 8>()); break;
    case 16: f(std::integral_constant<int, 16>()); break;
    case 32: f(std::integral_constant<int, 32>()); break;
    default: f(std::integral_constant<int, 64>()); break;
  }
}

}  // namespace

__global__ void __launch_bounds__(kBlock) k_nodal_interp(const int64_t *__restrict__ ptr, const int32_t *__restrict__ col, const double *__restrict__ w, int64_t n_rows, int ncomp, const double *__restrict__ in, double *__restrict__ out,
                                                       const double *__restrict__ g, const double *__restrict__ dinv, const uint8_t *__restrict__ inert, double omega) {
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n_rows * ncomp; t += (int64_t)gridDim.x * kBlock) {
    const int64_t row = t / ncomp; const int c = (int)(t - row * ncomp);
    double acc = 0;
    for (int64_t k = ptr[row]; k < ptr[row + 1]; ++k) acc = fma(w[k], in[(int64_t)col[k] * ncomp + c], acc);
    if (g) acc = (inert && inert[t]) ? 0.0 : fma(omega * dinv[t], g[t], acc);
    out[t] = acc;
  }
}
// long rows (the restriction P^T of a refined block: up to (4k - 1)^dim entries per coarse node next to single-entry rows; the prolongation: 1 .. (k + 1)^dim): LANES
// consecutive lanes share a row, consecutive entries to consecutive lanes, partial sums folded with shuffles (a fixed order: reproducible)
template <int LANES, int NC>
__global__ void __launch_bounds__(kBlock) k_nodal_interp_wide(const int64_t *__restrict__ ptr, const int32_t *__restrict__ col, const double *__restrict__ w, int64_t n_rows, const double *__restrict__ in, double *__restrict__ out,
                                                            const double *__restrict__ g, const double *__restrict__ dinv, const uint8_t *__restrict__ inert, double omega) {
  // a lane group per ROW: an entry's column and weight are read once for the NC components of the node (adjacent in memory)
  const int lane = threadIdx.x % LANES;
  const int64_t per_pass = (int64_t)gridDim.x * (kBlock / LANES);
  const int64_t last = ((n_rows + per_pass - 1) / per_pass) * per_pass;        // whole groups stay in the loop together (the shuffles need every lane)
  for (int64_t row = (int64_t)blockIdx.x * (kBlock / LANES) + threadIdx.x / LANES; row < last; row += per_pass) {
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0;
    if (row < n_rows) {
      for (int64_t k = ptr[row] + lane; k < ptr[row + 1]; k += LANES) {
        const double wk = w[k]; const double *v = in + (int64_t)col[k] * NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = fma(wk, v[c], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int m = LANES / 2; m >= 1; m >>= 1) acc[c] += __shfl_xor(acc[c], m, LANES);
    if (row < n_rows && lane < NC) {                       // lane c stores component c
      double v = acc[0];
#pragma unroll
      for (int c = 1; c < NC; ++c) if (lane == c) v = acc[c];
      const int64_t t = row * NC + lane;
      if (g) v = (inert && inert[t]) ? 0.0 : fma(omega * dinv[t], g[t], v);
      out[t] = v;
    }
  }
}
template <int LANES> static void nodal_interp_wide(hipStream_t s, const int64_t *ptr, const int32_t *col, const double *w, int64_t n_rows, int ncomp, const double *in, double *out, const double *g, const double *dinv, const uint8_t *inert, double omega) {
  const unsigned grid = grid_for(n_rows * LANES);
  if (ncomp == 1) hipLaunchKernelGGL((k_nodal_interp_wide<LANES, 1>), grid, kBlock, 0, s, ptr, col, w, n_rows, in, out, g, dinv, inert, omega);
  else if (ncomp == 2) hipLaunchKernelGGL((k_nodal_interp_wide<LANES, 2>), grid, kBlock, 0, s, ptr, col, w, n_rows, in, out, g, dinv, inert, omega);
  else hipLaunchKernelGGL((k_nodal_interp_wide<LANES, 3>), grid, kBlock, 0, s, ptr, col, w, n_rows, in, out, g, dinv, inert, omega);
}
static void nodal_interp_launch(hipStream_t s, int lanes, const int64_t *ptr, const int32_t *col, const double *w, int64_t n_rows, int ncomp, const double *in, double *out, const double *g, const double *dinv, const uint8_t *inert, double omega) {
  if (!n_rows) return;
  if (ncomp < 1 || ncomp > 3) throw Error("nodal interpolation: 1..3 components per node");
  if (lanes >= 16) nodal_interp_wide<16>(s, ptr, col, w, n_rows, ncomp, in, out, g, dinv, inert, omega);
  else if (lanes >= 8) nodal_interp_wide<8>(s, ptr, col, w, n_rows, ncomp, in, out, g, dinv, inert, omega);
  else if (lanes >= 4) nodal_interp_wide<4>(s, ptr, col, w, n_rows, ncomp, in, out, g, dinv, inert, omega);
  else hipLaunchKernelGGL(k_nodal_interp, grid_for(n_rows * ncomp), kBlock, 0, s, ptr, col, w, n_rows, ncomp, in, out, g, dinv, inert, omega);
}
void la_nodal_interp(hipStream_t s, const int64_t *ptr, const int32_t *col, const double *w, int64_t n_rows, int ncomp, const double *in, double *out, int lanes) {
  nodal_interp_launch(s, lanes, ptr, col, w, n_rows, ncomp, in, out, nullptr, nullptr, nullptr, 0.0);
}
void la_two_level_combine(hipStream_t s, const int64_t *ptr, const int32_t *col, const double *w, int64_t n_rows, int ncomp, const double *zc, const double *g, const double *dinv, const uint8_t *inert, double omega, double *z, int lanes) {
  nodal_interp_launch(s, lanes, ptr, col, w, n_rows, ncomp, zc, z, g, dinv, inert, omega);
}
struct ManyPairs { const double *y[3], *b[3]; };
__global__ void __launch_bounds__(kBlock) k_residual_norms_many(ManyPairs V, int nb, int64_t n, double *partials) {
  __shared__ double sh[5];
  for (int e = 0; e < nb; ++e) {
    double rr = 0, bb = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) { const double bv = V.b[e][i], r = V.y[e][i] - bv; rr = fma(r, r, rr); bb = fma(bv, bv, bb); }
    rr = block_sum(rr, sh); bb = block_sum(bb, sh);
    store_partial(partials + (size_t)(2 * e) * kMaxPartials, rr); store_partial(partials + (size_t)(2 * e + 1) * kMaxPartials, bb);
  }
}
void la_residual_norms_many(hipStream_t s, int nb, const double *const *y, const double *const *b, int64_t n, double *partials) {
  ManyPairs V{}; for (int e = 0; e < nb; ++e) { V.y[e] = y[e]; V.b[e] = b[e]; }
  hipLaunchKernelGGL(k_residual_norms_many, reduce_grid(n), kBlock, 0, s, V, nb, n, partials);
}
void la_post(hipStream_t s, Mailbox *mb, unsigned long long seq, const double *src, int n, const PcgScalars *sc) { hipLaunchKernelGGL(k_post, 1, 64, 0, s, mb, seq, src, n, sc); }
void la_fill(hipStream_t s, double *x, double v, int64_t n) { if (n) hipLaunchKernelGGL(k_fill, grid_for(n), kBlock, 0, s, x, v, n); }
// device-to-device copy as a kernel: hipMemcpyAsync costs the host ~50 us per call (measured between the back-to-back copies of poro_state_restore), a launch ~5 us
__global__ void k_copy(double *__restrict__ y, const double *__restrict__ x, int64_t n) {
  const int64_t n2 = n >> 1;
  double2 *y2 = reinterpret_cast<double2 *>(y); const double2 *x2 = reinterpret_cast<const double2 *>(x);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) y2[i] = x2[i];
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = x[n - 1];
}
void la_copy(hipStream_t s, double *y, const double *x, int64_t n) {
  if (!n || y == x) return;
  if (((uintptr_t)y | (uintptr_t)x) & 15) { PORO_HIP(hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, s)); return; }   // (sub-vectors at odd offsets)
  hipLaunchKernelGGL(k_copy, grid_for(n, 8), kBlock, 0, s, y, x, n);
}
// up to 24 vectors in one launch (state snapshot / rollback: two dozen mostly small vectors, 7 us each as launches of their own); 16-byte aligned pointers
struct CopyMany { double *dst[24]; const double *src[24]; int64_t n[24]; };
__global__ void __launch_bounds__(kBlock) k_copy_many(CopyMany V) {
  const int v = blockIdx.y; const int64_t n = V.n[v], n2 = n >> 1;
  const double2 *__restrict__ s2 = reinterpret_cast<const double2 *>(V.src[v]); double2 *__restrict__ d2 = reinterpret_cast<double2 *>(V.dst[v]);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) d2[i] = s2[i];
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) V.dst[v][n - 1] = V.src[v][n - 1];
}
void la_copy_many(hipStream_t s, int count, double *const *dst, const double *const *src, const int64_t *n) {
  for (int base = 0; base < count; base += 24) {
    CopyMany V{}; int m = 0; int64_t longest = 0;
    for (int k = base; k < count && m < 24; ++k) {
      if (!n[k] || dst[k] == src[k]) continue;
      if (((uintptr_t)dst[k] | (uintptr_t)src[k]) & 15) { la_copy(s, dst[k], src[k], n[k]); continue; }
      V.dst[m] = dst[k]; V.src[m] = src[k]; V.n[m] = n[k]; longest = std::max(longest, n[k]); ++m;
    }
    if (m) hipLaunchKernelGGL(k_copy_many, dim3((unsigned)std::min<int64_t>((longest / 2 + kBlock - 1) / kBlock + 1, 2048), (unsigned)m), kBlock, 0, s, V);
  }
}
void la_axpy(hipStream_t s, double *y, double a, const double *x, int64_t n) { if (n) hipLaunchKernelGGL(k_axpy, grid_for(n), kBlock, 0, s, y, a, x, n); }
void la_add_range(hipStream_t s, double *y, const double *x, int64_t n) { la_axpy(s, y, 1.0, x, n); }
// both interface planes of a slab in one launch (either pair may be null)
__global__ void k_add_two_ranges(double *y0, const double *x0, double *y1, const double *x1, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) { if (y0) y0[i] += x0[i]; if (y1) y1[i] += x1[i]; }
}
void la_add_two_ranges(hipStream_t s, double *y0, const double *x0, double *y1, const double *x1, int64_t n) {
  if (n && (y0 || y1)) hipLaunchKernelGGL(k_add_two_ranges, grid_for(n), kBlock, 0, s, y0, x0, y1, x1, n);
}
void la_dot_partials(hipStream_t s, const double *a, const double *b, int64_t n, double *partials, const PcgScalars *gate) {
  hipLaunchKernelGGL(k_dot, reduce_grid(n), kBlock, 0, s, a, b, n, partials, gate);
}
void la_norm_partials(hipStream_t s, const double *a, int64_t n, double *p2, double *pinf) {
  hipLaunchKernelGGL(k_norm, reduce_grid(n), kBlock, 0, s, a, n, p2, pinf);
}
void la_reduce_finish(hipStream_t s, const double *partials, int n_sets, double *red, int max_mask) {
  hipLaunchKernelGGL(k_reduce_finish, 1, kBlock, 0, s, partials, n_sets, red, max_mask);
}
void la_csr_spmv(hipStream_t s, const CsrDev &A, const double *val, const double *x, double *y) {
  if (!A.n) return;
  dispatch_lanes(A.lanes_per_row, [&](auto L) {
    constexpr int l = decltype(L)::value;
    const int64_t grid = (A.n * l + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_spmv<l>, (unsigned)grid, kBlock, 0, s, A.n, A.rp.p, A.col.p, val, x, y);
  });
}
void la_csr_residual(hipStream_t s, const CsrDev &A, const double *M, const double *K, double kappa, const double *t, const double *p, const double *src,
                     double *R) {
  if (!A.n) return;
  dispatch_lanes(A.lanes_per_row, [&](auto L) {
    constexpr int l = decltype(L)::value;
    const int64_t grid = (A.n * l + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_residual<l>, (unsigned)grid, kBlock, 0, s, A.n, A.rp.p, A.col.p, M, K, kappa, t, p, src, R);
  });
}
void la_pressure_tmp(hipStream_t s, double *t, const double *ev, const double *ev0, const double *p, const double *po, double c1, double c2, int64_t n) {
  hipLaunchKernelGGL(k_pressure_tmp, grid_for(n), kBlock, 0, s, t, ev, ev0, p, po, c1, c2, n);
}
void la_jacobian(hipStream_t s, double *J, const double *M, const double *K, double a, double kappa, int64_t nnz) {
  hipLaunchKernelGGL(k_jacobian, grid_for(nnz), kBlock, 0, s, J, M, K, a, kappa, nnz);
}
void la_xpby(hipStream_t s, double *y, double a, double b, const double *x, int64_t n) { if (n) hipLaunchKernelGGL(k_xpby, grid_for(n), kBlock, 0, s, y, a, b, x, n); }
void la_ssor_apply(hipStream_t s, const CsrDev &A, const double *val, const SsorLevels &lv, double omega, const double *src, double *dst) {
  for (size_t l = 0; l + 1 < lv.fwd_off.size(); ++l) {
    const int64_t n = lv.fwd_off[l + 1] - lv.fwd_off[l];
    hipLaunchKernelGGL(k_ssor_fwd, (unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, s, n, lv.fwd_rows.p + lv.fwd_off[l], A.rp.p, A.col.p, val, A.diag_pos.p, omega, src, dst);
  }
  hipLaunchKernelGGL(k_ssor_mid, grid_for(A.n), kBlock, 0, s, A.n, val, A.diag_pos.p, omega, dst);
  for (size_t l = 0; l + 1 < lv.bwd_off.size(); ++l) {
    const int64_t n = lv.bwd_off[l + 1] - lv.bwd_off[l];
    hipLaunchKernelGGL(k_ssor_bwd, (unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, s, n, lv.bwd_rows.p + lv.bwd_off[l], A.rp.p, A.col.p, val, A.diag_pos.p, omega, dst);
  }
}
void la_ilu0_factor(hipStream_t s, const CsrDev &A, const SsorLevels &lv, double *lu, int *flag) {
  for (size_t l = 0; l + 1 < lv.fwd_off.size(); ++l) {
    const int64_t n = lv.fwd_off[l + 1] - lv.fwd_off[l];
    if (n) hipLaunchKernelGGL(k_ilu0_level, (unsigned)n, 64, 0, s, n, lv.fwd_rows.p + lv.fwd_off[l], A.rp.p, A.col.p, A.diag_pos.p, lu, flag);
  }
}
void la_ilu_apply(hipStream_t s, const CsrDev &A, const double *lu, const SsorLevels &lv, const double *src, double *dst) {
  for (size_t l = 0; l + 1 < lv.fwd_off.size(); ++l) {
    const int64_t n = lv.fwd_off[l + 1] - lv.fwd_off[l];
    hipLaunchKernelGGL(k_ilu_fwd, (unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, s, n, lv.fwd_rows.p + lv.fwd_off[l], A.rp.p, A.col.p, lu, A.diag_pos.p, src, dst);
  }
  for (size_t l = 0; l + 1 < lv.bwd_off.size(); ++l) {
    const int64_t n = lv.bwd_off[l + 1] - lv.bwd_off[l];
    hipLaunchKernelGGL(k_ilu_bwd, (unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, s, n, lv.bwd_rows.p + lv.bwd_off[l], A.rp.p, A.col.p, lu, A.diag_pos.p, dst);
  }
}
void la_reciprocal(hipStream_t s, double *y, const double *x, int64_t n) { if (n) hipLaunchKernelGGL(k_reciprocal, grid_for(n), kBlock, 0, s, y, x, n); }
void la_pointwise_mul(hipStream_t s, double *y, const double *x, int64_t n) { if (n) hipLaunchKernelGGL(k_pointwise_mul, grid_for(n), kBlock, 0, s, y, x, n); }
void la_cons_expand(hipStream_t s, const ConsDev &C, double *x, bool with_inhom) {
  if (C.n) hipLaunchKernelGGL(k_cons_expand, (unsigned)((C.n + kBlock - 1) / kBlock), kBlock, 0, s, C.n, C.dof.p, C.ptr.p, C.master.p, C.weight.p, with_inhom ? C.inhom.p : (const double *)nullptr, x);
}
void la_cons_reduce(hipStream_t s, const ConsDev &C, double *y) {
  if (!C.n) return;
  if (C.n_masters) hipLaunchKernelGGL(k_cons_gather, (unsigned)((C.n_masters + kBlock - 1) / kBlock), kBlock, 0, s, C.n_masters, C.t_master.p, C.t_ptr.p, C.t_dof.p, C.t_weight.p, y);
  hipLaunchKernelGGL(k_cons_zero, (unsigned)((C.n + kBlock - 1) / kBlock), kBlock, 0, s, C.n, C.dof.p, y);
}
// general partition: pack the shared entries for the neighbours; sum the received partial rows in ascending rank order (src < 0: the own value)
__global__ void k_ifc_pack(int64_t m, const int32_t *__restrict__ dof, const double *__restrict__ v, double *__restrict__ send) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < m) send[j] = v[dof[j]];
}
__global__ void k_ifc_sum(int64_t m, const int32_t *__restrict__ sh_dof, const int64_t *__restrict__ sh_ptr, const int32_t *__restrict__ sh_src, const double *__restrict__ recv, double *v) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  const int32_t dof = sh_dof[i]; const double own = v[dof];
  double acc = 0;
  for (int64_t k = sh_ptr[i]; k < sh_ptr[i + 1]; ++k) { const int32_t src = sh_src[k]; acc += src < 0 ? own : recv[src]; }
  v[dof] = acc;
}
void la_ifc_pack(hipStream_t s, const IfcDev &I, const double *v) {
  if (I.m_send) hipLaunchKernelGGL(k_ifc_pack, (unsigned)((I.m_send + kBlock - 1) / kBlock), kBlock, 0, s, I.m_send, I.dof.p, v, I.send.p);
}
void la_ifc_sum(hipStream_t s, const IfcDev &I, double *v) {
  if (I.m_shared) hipLaunchKernelGGL(k_ifc_sum, (unsigned)((I.m_shared + kBlock - 1) / kBlock), kBlock, 0, s, I.m_shared, I.sh_dof.p, I.sh_ptr.p, I.sh_src.p, I.recv.p, v);
}
void la_csr_diag(hipStream_t s, const CsrDev &A, const double *val, double *diag) {
  hipLaunchKernelGGL(k_csr_diag, grid_for(A.n), kBlock, 0, s, A.n, A.diag_pos.p, val, diag);
}
// sigma_ij = 2 G eps_ij + lambda tr(eps) delta_ij on the packed symmetric entries (2D: xx xy yy; 3D: xx xy xz yy yz zz)
struct SymPtrs { const double *eps[6]; double *sig[6]; };
__global__ void k_effective_stress(SymPtrs P, int dim, double lam, double G, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int ne = dim == 2 ? 3 : 6;
    const double tr = dim == 2 ? P.eps[0][i] + P.eps[2][i] : P.eps[0][i] + P.eps[3][i] + P.eps[5][i];
    for (int e = 0; e < ne; ++e) {
      const bool diag = dim == 2 ? (e == 0 || e == 2) : (e == 0 || e == 3 || e == 5);
      P.sig[e][i] = 2.0 * G * P.eps[e][i] + (diag ? lam * tr : 0.0);
    }
  }
}
void la_effective_stress(hipStream_t s, const double *const *strains, double *const *stresses, int dim, double lam, double G, int64_t len) {
  SymPtrs P{}; const int ne = dim == 2 ? 3 : 6;
  for (int k = 0; k < ne; ++k) { P.eps[k] = strains[k]; P.sig[k] = stresses[k]; }
  hipLaunchKernelGGL(k_effective_stress, grid_for(len), kBlock, 0, s, P, dim, lam, G, len);
}
void la_sum_strains(hipStream_t s, double *ev, const double *const *strains, int n, int64_t len) {
  StrainPtrs sp{}; for (int k = 0; k < n; ++k) sp.p[k] = strains[k];
  hipLaunchKernelGGL(k_sum_strains, grid_for(len), kBlock, 0, s, ev, sp, n, len);
}
void la_set_constrained(hipStream_t s, double *x, const uint8_t *mask, const double *val, int64_t n) {
  hipLaunchKernelGGL(k_set_constrained, grid_for(n), kBlock, 0, s, x, mask, val, n);
}
void la_rhs_u_finish(hipStream_t s, double *rhs, const double *lift, const double *neu, const uint8_t *mask, int64_t n) {
  hipLaunchKernelGGL(k_rhs_u_finish, grid_for(n), kBlock, 0, s, rhs, lift, neu, mask, n);
}
// dictionary form only for 2 / 3 components and 32-bit dof indices
static int diag_nc(const DiagVec &dv, int64_t n) { return (dv.cls && (dv.ncomp == 2 || dv.ncomp == 3) && n < (int64_t)4000000000ll) ? dv.ncomp : 0; }
void la_cheb_first(hipStream_t s, double *z, const double *g, const DiagVec &dv, double scale, int64_t n) {
  const DiagRef D{dv.full, dv.cls, dv.tab, dv.ncomp, nullptr};
  switch (diag_nc(dv, n)) {
    case 2: hipLaunchKernelGGL(k_cheb_first<2>, grid_for(n), kBlock, 0, s, z, g, D, scale, n); break;
    case 3: hipLaunchKernelGGL(k_cheb_first<3>, grid_for(n), kBlock, 0, s, z, g, D, scale, n); break;
    default: hipLaunchKernelGGL(k_cheb_first<0>, grid_for(n), kBlock, 0, s, z, g, D, scale, n);
  }
}
void la_cheb_fix_planes(hipStream_t s, double *znew, const double *zj, const double *g, const double *own_lo, const double *nbr_lo, const double *own_hi, const double *nbr_hi, const DiagVec &dv, double omega, int64_t n, int64_t plane) {
  const DiagRef D{dv.full, dv.cls, dv.tab, dv.ncomp, nullptr};
  const unsigned grid = (unsigned)((2 * plane + kBlock - 1) / kBlock);
  switch (diag_nc(dv, n)) {
    case 2: hipLaunchKernelGGL(k_cheb_fix_planes<2>, grid, kBlock, 0, s, znew, zj, g, own_lo, nbr_lo, own_hi, nbr_hi, D, omega, n, plane); break;
    case 3: hipLaunchKernelGGL(k_cheb_fix_planes<3>, grid, kBlock, 0, s, znew, zj, g, own_lo, nbr_lo, own_hi, nbr_hi, D, omega, n, plane); break;
    default: hipLaunchKernelGGL(k_cheb_fix_planes<0>, grid, kBlock, 0, s, znew, zj, g, own_lo, nbr_lo, own_hi, nbr_hi, D, omega, n, plane);
  }
}
void la_cheb_step(hipStream_t s, double *znew, const double *zj, const double *g, const double *Az, const DiagVec &dv, double omega, int64_t n, int64_t n_owned, double *gz_partials) {
  const DiagRef D{dv.full, dv.cls, dv.tab, dv.ncomp, nullptr};
  switch (diag_nc(dv, n)) {
    case 2: hipLaunchKernelGGL(k_cheb_step<2>, reduce_grid(n), kBlock, 0, s, znew, zj, g, Az, D, omega, n, n_owned, gz_partials); break;
    case 3: hipLaunchKernelGGL(k_cheb_step<3>, reduce_grid(n), kBlock, 0, s, znew, zj, g, Az, D, omega, n, n_owned, gz_partials); break;
    default: hipLaunchKernelGGL(k_cheb_step<0>, reduce_grid(n), kBlock, 0, s, znew, zj, g, Az, D, omega, n, n_owned, gz_partials);
  }
}
void pcg_init_residual(hipStream_t s, double *g, const double *Ax, const double *b, const uint8_t *inert, int64_t n) {
  hipLaunchKernelGGL(k_pcg_init_residual, grid_for(n), kBlock, 0, s, g, Ax, b, inert, n);
}
void la_mask_zero(hipStream_t s, double *x, const uint8_t *mask, int64_t n) { if (n) hipLaunchKernelGGL(k_mask_zero, grid_for(n), kBlock, 0, s, x, mask, n); }
void pcg_first_direction(hipStream_t s, double *d, const double *g, const DiagVec &dv, int prec, int64_t n, int64_t n_owned, double *partials) {
  const DiagRef D{dv.full, dv.cls, dv.tab, dv.ncomp, dv.z};
  switch (diag_nc(dv, n)) {
    case 2: hipLaunchKernelGGL(k_pcg_first_direction<2>, reduce_grid(n), kBlock, 0, s, d, g, D, prec, n, n_owned, partials); break;
    case 3: hipLaunchKernelGGL(k_pcg_first_direction<3>, reduce_grid(n), kBlock, 0, s, d, g, D, prec, n, n_owned, partials); break;
    default: hipLaunchKernelGGL(k_pcg_first_direction<0>, reduce_grid(n), kBlock, 0, s, d, g, D, prec, n, n_owned, partials);
  }
}
void pcg_dot_dh(hipStream_t s, const PcgScalars *sc, const double *d, const double *h, int64_t n_owned, double *partials) {
  hipLaunchKernelGGL(k_pcg_dot_dh, reduce_grid(n_owned), kBlock, 0, s, sc, d, h, n_owned, partials);
}
// ---- single-reduction PCG (Chronopoulos & Gear): partitioned runs pay ONE all-reduce per iteration ------------------------------------
// With z = P^-1 g and w = A z the three dots g.z, w.z, g.g of an iteration are independent of each other, so they travel in one reduction;
// A d follows from the recurrence s = -w + beta s instead of a second operator application.
__global__ void k_cg1_dots(const double *__restrict__ g, const double *__restrict__ z, const double *__restrict__ w, const double *__restrict__ b, int64_t n_owned, double *partials) {
  __shared__ double sh[4];
  double gz = 0, wz = 0, gg = 0, bb = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_owned; i += (int64_t)gridDim.x * kBlock) {
    const double gi = g[i], zi = z[i];
    gz += gi * zi; wz += w[i] * zi; gg += gi * gi;
    if (b) { const double bi = b[i]; bb += bi * bi; }
  }
  gz = block_sum(gz, sh); store_partial(partials, gz);
  wz = block_sum(wz, sh); store_partial(partials + kMaxPartials, wz);
  gg = block_sum(gg, sh); store_partial(partials + 2 * kMaxPartials, gg);
  bb = block_sum(bb, sh); store_partial(partials + 3 * kMaxPartials, bb);
}
// red = {g.z, w.z, g.g, b.b} after the all-reduce.  SolverCG's control flow: stop when ||g|| <= tol (checked before the update that would follow), give up at max_iter
__global__ void k_cg1_scalars(Cg1State *st, const double *red, int first, double abs_tol, double rel_tol, int max_iter, int stop_rule) {
  const double gamma = red[0], delta = red[1], gg = red[2];
  if (first) {
    st->tol = fmax(abs_tol, rel_tol * sqrt(stop_rule == PORO_STOP_REDUCTION ? gg : red[3]));
    st->res0 = sqrt(gg); st->it = 0; st->done = 0; st->converged = 0; st->gamma_old = 0; st->alpha_old = 0;
  }
  if (st->done) return;
  st->res = sqrt(gg);
  if (st->res <= st->tol) { st->converged = 1; st->done = 1; return; }
  if (st->it >= max_iter) { st->done = 1; return; }
  const double beta = first ? 0.0 : gamma / st->gamma_old;
  const double alpha = first ? gamma / delta : gamma / (delta - beta * gamma / st->alpha_old);
  st->alpha = alpha; st->beta = beta; st->gamma_old = gamma; st->alpha_old = alpha; st->it += 1;
}
// d = -z + beta d, s = -w + beta s (= A d), x += alpha d, g += alpha s; inert (Dirichlet) dofs keep d = s = 0 whatever the operator left in w there
template <int NC> __global__ void k_cg1_update(const Cg1State *st, double *d, double *sv, double *x, double *g, const double *z /* may be z1_out */, const double *__restrict__ w, const uint8_t *inert, int64_t n,
                                              DiagRef D, double *z1_out, double z1_scale) {
  if (st->done) return;
  const double alpha = st->alpha, beta = st->beta;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (inert && inert[i]) { if (z1_out) z1_out[i] = 0.0; continue; }
    const double di = fma(beta, d[i], -z[i]), si = fma(beta, sv[i], -w[i]);
    const double gi = fma(alpha, si, g[i]);
    d[i] = di; sv[i] = si; x[i] = fma(alpha, di, x[i]); g[i] = gi;
    if (z1_out) z1_out[i] = z1_scale * diag_at<NC>(D, i) * gi;      // first iterate of the polynomial preconditioner of the NEXT iteration (DiagVec::z1_out), saves a kernel
  }
}
void cg1_dots(hipStream_t s, const double *g, const double *z, const double *w, const double *b, int64_t n_owned, double *partials) {
  hipLaunchKernelGGL(k_cg1_dots, reduce_grid(n_owned), kBlock, 0, s, g, z, w, b, n_owned, partials);
}
void cg1_scalars(hipStream_t s, Cg1State *st, const double *red, int first, double abs_tol, double rel_tol, int max_iter, int stop_rule) {
  hipLaunchKernelGGL(k_cg1_scalars, 1, 1, 0, s, st, red, first, abs_tol, rel_tol, max_iter, stop_rule);
}
void cg1_update(hipStream_t s, const Cg1State *st, double *d, double *sv, double *x, double *g, const double *z, const double *w, const DiagVec &dv, int64_t n) {
  const DiagRef D{dv.full, dv.cls, dv.tab, dv.ncomp, nullptr};
  switch (dv.z1_out ? diag_nc(dv, n) : 0) {
    case 2: hipLaunchKernelGGL(k_cg1_update<2>, grid_for(n), kBlock, 0, s, st, d, sv, x, g, z, w, dv.inert, n, D, dv.z1_out, dv.z1_scale); break;
    case 3: hipLaunchKernelGGL(k_cg1_update<3>, grid_for(n), kBlock, 0, s, st, d, sv, x, g, z, w, dv.inert, n, D, dv.z1_out, dv.z1_scale); break;
    default: hipLaunchKernelGGL(k_cg1_update<0>, grid_for(n), kBlock, 0, s, st, d, sv, x, g, z, w, dv.inert, n, D, dv.z1_out, dv.z1_scale);
  }
}
void pcg_scalars_sum(hipStream_t s, const double *partials, int n_sets, double *red) {
  hipLaunchKernelGGL(k_scalars_sum, 1, kBlock, 0, s, (const PcgScalars *)nullptr, partials, n_sets, red);
}
void pcg_scalars_start(hipStream_t s, PcgScalars *sc, const double *red, double abs_tol, double rel_tol, int max_iter, int stop_rule) {
  hipLaunchKernelGGL(k_scalars_start, 1, 1, 0, s, sc, red, abs_tol, rel_tol, max_iter, stop_rule);
}
void pcg_update_g_fused(hipStream_t s, PcgScalars *sc, int parity, double *g, const double *h, const DiagVec &dv, int prec, int64_t n, int64_t n_owned,
                        const double *partials_dh, const double *red, double *partials_out) {
  const DiagRef D{dv.full, dv.cls, dv.tab, dv.ncomp, dv.z, dv.z1_out, dv.z1_scale};
  switch (diag_nc(dv, n)) {
    case 2: hipLaunchKernelGGL(k_pcg_update_g_fused<2>, reduce_grid(n), kBlock, 0, s, sc, parity, g, h, D, prec, n, n_owned, partials_dh, red, partials_out); break;
    case 3: hipLaunchKernelGGL(k_pcg_update_g_fused<3>, reduce_grid(n), kBlock, 0, s, sc, parity, g, h, D, prec, n, n_owned, partials_dh, red, partials_out); break;
    default: hipLaunchKernelGGL(k_pcg_update_g_fused<0>, reduce_grid(n), kBlock, 0, s, sc, parity, g, h, D, prec, n, n_owned, partials_dh, red, partials_out);
  }
}
void pcg_update_d_fused(hipStream_t s, PcgScalars *sc, int parity, int it, double *x, double *d, const double *g, const DiagVec &dv, int prec, int64_t n,
                        const double *partials_in, const double *red) {
  const DiagRef D{dv.full, dv.cls, dv.tab, dv.ncomp, dv.z};
  switch (diag_nc(dv, n)) {
    case 2: hipLaunchKernelGGL(k_pcg_update_d_fused<2>, reduce_grid(n), kBlock, 0, s, sc, parity, it, x, d, g, D, prec, n, partials_in, red); break;
    case 3: hipLaunchKernelGGL(k_pcg_update_d_fused<3>, reduce_grid(n), kBlock, 0, s, sc, parity, it, x, d, g, D, prec, n, partials_in, red); break;
    default: hipLaunchKernelGGL(k_pcg_update_d_fused<0>, reduce_grid(n), kBlock, 0, s, sc, parity, it, x, d, g, D, prec, n, partials_in, red);
  }
}

}  // namespace poro
