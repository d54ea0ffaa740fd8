// Krylov drivers: device-controlled PCG (SolverCG restated), its single-reduction form for partitioned runs, the host-driven form for SSOR / ILU(0), the lambda_max estimate.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <unordered_map>
#include "common.hpp"
#include "ctx_internal.hpp"

using namespace poro;
using namespace poro::ctx_detail;

namespace poro {
namespace ctx_detail {
// ---- single-reduction PCG for partitioned runs (Chronopoulos & Gear) ------------------------------------------------------------------
// Same Krylov space, same stopping test and same iteration count as SolverCG's recurrence in exact arithmetic, rearranged so that an iteration
// costs ONE all-reduce: z = P^-1 g, w = A z, then {g.z, w.z, g.g} in one reduction, then d = -z + beta d, s = -w + beta s (= A d), x += alpha d,
// g += alpha s.  Per iteration: 1 operator application (1 grouped neighbour exchange) + the exchanges inside P^-1 + 1 all-reduce of 4 doubles.
// The price is two more vector passes than pcg(), which is why single-rank runs keep the three-kernel recurrence.
int pcg_single_reduction(poro_ctx *c, const std::function<bool(const double *, double *, double *)> &apply, int64_t n, int64_t plane, double *x, const double *b,
                         const DiagVec &diag, double *g, double *d, double *sv, const poro_solver_opts *opts, poro_solve_info *info,
                         const std::function<bool(const double *, double *, double *)> *precond, int *its_hint) {
  hipStream_t s = c->stream;
  const int which = n == c->n_u ? 0 : 1;
  if (c->cg1_w[which].n < (size_t)n) { c->cg1_w[which].alloc(n); c->cg1_z[which].alloc(n); }
  if (!c->cg1_state.p) c->cg1_state.alloc(1);
  double *w = c->cg1_w[which].p, *z = (precond && diag.z) ? const_cast<double *>(diag.z) : c->cg1_z[which].p;
  if (precond && !diag.z) throw Error("pcg: explicit preconditioner without a z vector");
  const int64_t n_own = owned(c, n, plane);
  const bool jacobi = opts->preconditioner == PORO_PREC_JACOBI;
  Cg1State *st = c->cg1_state.p; double *part = c->partials.p, *red = c->red.p;
  EventPair ev(c); PORO_HIP(hipEventRecord(ev.e0, s));
  apply(x, w, nullptr);
  pcg_init_residual(s, g, w, b, diag.inert, n);        // g = A x - b, zero on the inert dofs
  la_fill(s, d, 0.0, n); la_fill(s, sv, 0.0, n);
  DiagVec upd = diag; if (jacobi && !precond) { upd.z1_out = z; upd.z1_scale = 1.0; }   // Jacobi: the update kernel also leaves z = D^-1 g_new for the next iteration (in place of z)
  Cg1State hs{};
  int expect = 0, enq = 0;
  if (its_hint && its_hint[0] > 0) { expect = its_hint[1] > 0 ? 2 * its_hint[0] - its_hint[1] : its_hint[0]; expect = std::min(std::max(expect, its_hint[0] / 2), its_hint[0]); }   // (never above the last count: nothing here is gated)
  auto next_batch = [&](int done_its) { const int left = expect - 4 - done_its; return left >= 4 ? std::min(32, left) : 2; };
  int batch = expect > 0 ? next_batch(0) : 1;
  while (true) {
    for (int k = 0; k < batch; ++k) {
      c->cheb_z1_ready = precond && enq > 0 && diag.z1_out != nullptr;      // stored by the previous cg1_update
      if (precond) (void)(*precond)(g, z, nullptr);
      else if (jacobi) { if (enq == 0) la_cheb_first(s, z, g, diag, 1.0, n); }   // z = D^-1 g (zero on the inert dofs); after the first iteration the update kernel stores it with the new residual
      else la_copy(s, z, g, n);
      apply(z, w, nullptr);
      cg1_dots(s, g, z, w, enq == 0 ? b : nullptr, n_own, part);
      pcg_scalars_sum(s, part, 4, red);
      allreduce_sum(c, red, 4);
      cg1_scalars(s, st, red, enq == 0 ? 1 : 0, opts->abs_tol, opts->rel_tol, opts->max_iter, opts->stop_rule);
      cg1_update(s, st, d, sv, x, g, z, w, upd, n);
      ++enq;
    }
    PORO_HIP(hipMemcpyAsync(&hs, st, sizeof(hs), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s));
    if (hs.done) break;
    // (the preconditioner and operator launches of an iteration are not gated by the device-side `done` flag: without a hint poll at least every 8 iterations)
    if (expect > 0) batch = next_batch(enq); else if (batch < 8) batch *= 2;
  }
  if (its_hint) { its_hint[1] = its_hint[0]; its_hint[0] = hs.it; }
  PORO_HIP(hipEventRecord(ev.e1, s)); PORO_HIP(hipEventSynchronize(ev.e1));
  float ms = 0; PORO_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  if (info) { info->iterations = hs.it; info->converged = hs.converged; info->initial_residual = hs.res0; info->final_residual = hs.res; info->seconds = ms * 1e-3;
              info->operator_applications = hs.it + 2; }   // initial residual + one per iteration + the one that found the converged residual
  return hs.converged ? 0 : 1;
}

// ---- PCG with device-side control: SolverCG<>::solve restated, Jacobi instead of SSOR ---------------
// apply(x, y, dot_partials) as apply_A_u.  The vector kernels compute alpha / beta / the stopping test in their prologues: from the
// block partials (single rank, 3 launches per iteration incl. the operator) or from the all-reduced scalars (partitioned).
// precond != null: explicit preconditioner z = P^-1 g (a sequence of launches on the stream, e.g. the fast diagonalisation) written into
// diag.z between the two update kernels; the scalars stay on the device exactly as in the Jacobi case.  precond(g, z, gz_partials) returns true
// when it has already left the block partials of g . z (over the owned rows) in gz_partials.
int pcg(poro_ctx *c, const std::function<bool(const double *, double *, double *)> &apply, int64_t n, int64_t plane, double *x, const double *b,
        const DiagVec &diag, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info,
        const std::function<bool(const double *, double *, double *)> *precond, int *its_hint, bool precond_gated,
        const FdmOct *oct /* single rank, explicit preconditioner: the residual and z = P^-1 g live in octant form (kernels_fdmo.hip), `g` is unused */) {
  static const bool two_reductions = std::getenv("PORO_TWO_REDUCTION_CG") != nullptr;    // A/B hook: the three-kernel recurrence on partitioned runs too
  if (c->comm.multi() && !two_reductions && !oct) return pcg_single_reduction(c, apply, n, plane, x, b, diag, g, d, h, opts, info, precond, its_hint);
  hipStream_t s = c->stream;
  const int prec = opts->preconditioner == PORO_PREC_JACOBI ? 1 : 0;
  double *zbuf = const_cast<double *>(diag.z);
  if (oct) { if (!precond || c->comm.multi() != oct->slab.on) throw Error("pcg: the octant form needs an explicit preconditioner (one rank: octants, slab partition: quadrants)"); g = oct->g.p; zbuf = oct->z.p; }
  if (precond && !zbuf) throw Error("pcg: explicit preconditioner without a z vector");
  const int64_t n_own = owned(c, n, plane);
  const bool multi = c->comm.multi();
  double *part = c->partials.p, *red = c->red.p; PcgScalars *sc = c->scal.p;
  double *part_dh = part + 3 * (size_t)kMaxPartials;      // slots of the fused / separate d.h partials
  const auto t_start = std::chrono::steady_clock::now();
  int64_t applies = 0;
  // g = A x - b ; d = -P^-1 g ; gh = g.P^-1 g
  apply(x, h, nullptr); ++applies;
  if (oct) fdmo_init_residual(s, *oct, g, h, b, diag.inert); else pcg_init_residual(s, g, h, b, diag.inert, n);
  if (opts->stop_rule != PORO_STOP_REDUCTION) la_dot_partials(s, b, b, n_own, part);     // ||b||^2 only enters the ||b||-relative stopping rule (the slot keeps an older, finite value otherwise)
  if (precond) (void)(*precond)(g, zbuf, nullptr);
  if (oct) fdmo_first_direction(s, *oct, d, g, zbuf, part + kMaxPartials); else pcg_first_direction(s, d, g, diag, prec, n, n_own, part + kMaxPartials);
  pcg_scalars_sum(s, part, 3, red);
  allreduce_sum(c, red, 3);
  pcg_scalars_start(s, sc, red, opts->abs_tol, opts->rel_tol, opts->max_iter, opts->stop_rule);
  PORO_HIP(hipMemsetAsync(part_dh, 0, kMaxPartials * sizeof(double), s));
  PcgScalars hs{};
  int it = 0;
  // Iterations are enqueued in batches, THEN the device-side state is polled (a host round trip idles the GPU for ~50 us).  Launches behind the finishing
  // iteration are no-ops (the vector kernels, the structured operator and the fused Chebyshev kernels test the device-side flag; ~1 us each), so where
  // everything is gated an overshoot is cheaper than a poll; an ungated explicit preconditioner (fast diagonalisation) is not, so its batches stop short.
  // Expected iteration count: linear extrapolation of the last two solves of this system (a transient's warm-started counts drift steadily).
  int expect = 0;
  if (its_hint && its_hint[0] > 0) { expect = its_hint[1] > 0 ? 2 * its_hint[0] - its_hint[1] : its_hint[0]; expect = std::max(expect, its_hint[0] / 2); }
  const bool cheap_overshoot = !precond || precond_gated;
  // the extrapolation must not run away after an atypical solve (a warm restart that took 3 iterations, followed by a real step): never expect more than a quarter
  // above the last count, and nothing above it where an overshoot is expensive
  if (expect > 0) expect = std::min(expect, cheap_overshoot ? its_hint[0] + std::max(2, its_hint[0] / 4) : its_hint[0]);
  const bool small = n <= 400000;     // launch-bound sizes: an ungated preconditioner application costs less than the idle time of a poll
  int batch = expect > 0 ? (cheap_overshoot ? std::min(expect, 256) : std::max(1, expect - 1)) : (precond && !cheap_overshoot && !small ? 1 : 4);   // no history: a poll (~15 us through the mailbox) every 4 iterations
  while (true) {
    for (int k = 0; k < batch; ++k) {
      ++it;
      // operator (+ fused or separate d.h partials).  A fused dot runs over ALL local rows of the pre-exchange partial product, which
      // sums to the global d.Ad over the ranks; the separate kernel sees the exchanged h and therefore skips the upper shared plane.
      if (!apply(d, h, part_dh)) pcg_dot_dh(s, sc, d, h, n_own, part_dh);
      ++applies;
      if (multi) { pcg_scalars_sum(s, part_dh, 1, red); allreduce_sum(c, red, 1); }
      if (oct) fdmo_update_g(s, *oct, sc, (it - 1) & 1, g, h, diag.inert, part_dh, part, multi ? red : nullptr);
      else pcg_update_g_fused(s, sc, (it - 1) & 1, g, h, diag, prec, n, n_own, part_dh, multi ? red : nullptr, part);
      if (precond && !(*precond)(g, zbuf, part + kMaxPartials)) {
        if (oct && multi) fdmo_dot_owned(s, *oct, g, zbuf, part + kMaxPartials, precond_gated ? sc : nullptr);
        else la_dot_partials(s, g, zbuf, oct ? oct->n_oct : n_own, part + kMaxPartials, precond_gated ? sc : nullptr);
      }
      if (multi) { pcg_scalars_sum(s, part, 2, red + 1); allreduce_sum(c, red + 1, 2); }
      if (oct) fdmo_update_d(s, *oct, sc, (it - 1) & 1, it, x, d, zbuf, part, multi ? red + 1 : nullptr);
      else pcg_update_d_fused(s, sc, (it - 1) & 1, it, x, d, g, diag, prec, n, part, multi ? red + 1 : nullptr);
    }
    post_and_wait(c, nullptr, 0, sc); hs = c->mailbox->sc;
    if (hs.done || hs.finishing) break;
    if (expect > 0) batch = cheap_overshoot ? 3 : small ? 2 : 1;
    else if (cheap_overshoot && precond) batch = 4;          // (an explicit preconditioner: a no-op iteration still costs ~8 launches)
    else if (batch < 32) batch *= 2;
  }
  if (its_hint) { its_hint[1] = its_hint[0]; its_hint[0] = hs.it; }
  // (the last poll returned after the finishing iteration: the solve is complete on the device; wall time of the solve on the host clock)
  if (info) { info->iterations = hs.it; info->converged = hs.converged; info->initial_residual = hs.res0; info->final_residual = hs.res;
              info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
              info->operator_applications = hs.it + 1;   // initial residual + one per iteration (launches enqueued behind the finishing iteration are no-ops and are not counted)
              (void)applies; }
  return hs.converged ? 0 : 1;
}

// ---- PreconditionSSOR fidelity mode: SolverCG with the reference's SSOR(omega) in natural row order ---------------------------
void build_ssor_levels(poro_ctx *c, CsrDev &A) {
  if (A.ssor.built) return;
  std::vector<int64_t> rp(A.n + 1); std::vector<int32_t> col(A.nnz);
  PORO_HIP(hipMemcpy(rp.data(), A.rp.p, (A.n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost)); PORO_HIP(hipMemcpy(col.data(), A.col.p, A.nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  auto levels = [&](bool fwd, DevBuf<int32_t> &rows_dev, std::vector<int64_t> &off) {
    std::vector<int32_t> lvl(A.n, 0); int maxl = 0;
    if (fwd) for (int64_t r = 0; r < A.n; ++r) { int l = 0; for (int64_t j = rp[r]; j < rp[r + 1] && col[j] < r; ++j) l = std::max(l, lvl[col[j]] + 1); lvl[r] = l; maxl = std::max(maxl, l); }
    else for (int64_t r = A.n - 1; r >= 0; --r) { int l = 0; for (int64_t j = rp[r + 1] - 1; j >= rp[r] && col[j] > r; --j) l = std::max(l, lvl[col[j]] + 1); lvl[r] = l; maxl = std::max(maxl, l); }
    off.assign(maxl + 2, 0);
    for (int64_t r = 0; r < A.n; ++r) off[lvl[r] + 1]++;
    for (int l = 0; l <= maxl; ++l) off[l + 1] += off[l];
    std::vector<int32_t> rows(A.n); std::vector<int64_t> pos(off.begin(), off.end() - 1);
    for (int64_t r = 0; r < A.n; ++r) rows[pos[lvl[r]]++] = (int32_t)r;
    rows_dev.upload(rows);
  };
  levels(true, A.ssor.fwd_rows, A.ssor.fwd_off); levels(false, A.ssor.bwd_rows, A.ssor.bwd_off);
  A.ssor.built = true;
}
// global dot product on the host; n = rows this rank owns (the upper shared plane belongs to the neighbour)
double dot_host(poro_ctx *c, const double *a, const double *b, int64_t n) {
  la_dot_partials(c->stream, a, b, n, c->partials.p); la_reduce_finish(c->stream, c->partials.p, 1, c->red.p, 0);
  allreduce_sum(c, c->red.p, 1);
  post_and_wait(c, c->red.p, 1);
  return c->mailbox->vals[0];
}
// SolverCG<>::solve with an explicit preconditioner z = P^-1 g, host-driven scalars.  Used where an application of P^-1 is many
// launches anyway (SSOR sweeps) or where only a handful of iterations happen (fast diagonalisation).  Partitioned runs: `apply` and
// `precond` return vectors that are consistent on the shared planes; dots run over the `n_own` owned rows and are all-reduced.
int pcg_host(poro_ctx *c, int64_t n, int64_t n_own, const std::function<void(const double *, double *)> &apply, const std::function<void(const double *, double *)> &precond,
             double *x, const double *b, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info) {
  hipStream_t s = c->stream;
  const auto t0 = std::chrono::steady_clock::now();
  int64_t applies = 0; int it = 0, conv = 0;
  apply(x, g); ++applies;
  la_axpy(s, g, -1.0, b, n);                                     // g = A x - b
  double res = std::sqrt(dot_host(c, g, g, n_own)); const double res0 = res;
  const double tol = std::max(opts->abs_tol, opts->rel_tol * (opts->stop_rule == PORO_STOP_REDUCTION ? res0 : std::sqrt(dot_host(c, b, b, n_own))));
  if (res <= tol) conv = 1;
  else {
    precond(g, h);
    la_fill(s, d, 0.0, n); la_axpy(s, d, -1.0, h, n);          // d = -h
    double gh = dot_host(c, g, h, n_own);
    while (true) {
      ++it;
      apply(d, h); ++applies;
      const double alpha = gh / dot_host(c, d, h, n_own);
      la_axpy(s, g, alpha, h, n); la_axpy(s, x, alpha, d, n);
      res = std::sqrt(dot_host(c, g, g, n_own));
      if (res <= tol) { conv = 1; break; }
      if (it >= opts->max_iter) break;
      precond(g, h);
      const double beta_old = gh; gh = dot_host(c, g, h, n_own);
      la_xpby(s, d, gh / beta_old, -1.0, h, n);                   // d = beta d - h
    }
  }
  PORO_HIP(hipStreamSynchronize(s));
  if (info) { info->iterations = it; info->converged = conv; info->initial_residual = res0; info->final_residual = res; info->operator_applications = applies;
              info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  return conv ? 0 : 1;
}
int pcg_ssor(poro_ctx *c, CsrDev &A, const double *val, double *x, const double *b, double *g, double *d, double *h, const poro_solver_opts *opts, poro_solve_info *info) {
  if (c->comm.multi()) throw Error("PORO_PREC_SSOR is a single-rank fidelity mode (the sweeps are order dependent)");
  build_ssor_levels(c, A);
  const double om = opts->omega > 0 ? opts->omega : 1.0;
  return pcg_host(c, A.n, A.n, [&](const double *v, double *y) { la_csr_spmv(c->stream, A, val, v, y); },
                  [&](const double *gg, double *z) { la_ssor_apply(c->stream, A, val, A.ssor, om, gg, z); }, x, b, g, d, h, opts, info);
}

// ---- ILU(0): factorisation and solves on the device, both level-scheduled in the natural row order (la_ilu0_factor, la_ilu_apply) -----------------
void ilu0_factor(poro_ctx *c, const CsrDev &A, const double *val, DevBuf<double> &lu_dev) {
  std::vector<int64_t> rp(A.n + 1);
  PORO_HIP(hipMemcpy(rp.data(), A.rp.p, (A.n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  int64_t longest = 0; for (int64_t i = 0; i < A.n; ++i) longest = std::max(longest, rp[i + 1] - rp[i]);
  if (longest > 512) throw Error("ILU(0): rows longer than 512 entries are not supported by the device factorisation");
  if (lu_dev.n < (size_t)A.nnz) lu_dev.alloc(A.nnz);
  DevBuf<int> flag; flag.alloc(1); flag.zero(c->stream);
  PORO_HIP(hipMemcpyAsync(lu_dev.p, val, A.nnz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  la_ilu0_factor(c->stream, A, A.ssor, lu_dev.p, flag.p);
  int h = 0; PORO_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream)); PORO_HIP(hipStreamSynchronize(c->stream));
  if (h) throw Error("ILU(0): zero pivot in row " + std::to_string(h - 1));
}
int pcg_ilu0(poro_ctx *c, CsrDev &A, const double *val, DevBuf<double> &lu, bool &valid, double *x, const double *b, double *g, double *d, double *h,
             const poro_solver_opts *opts, poro_solve_info *info) {
  if (c->comm.multi()) throw Error("PORO_PREC_ILU0 is implemented for one rank (the factorisation is sequential in the row order)");
  build_ssor_levels(c, A);
  if (!valid) { ilu0_factor(c, A, val, lu); valid = true; }
  return pcg_host(c, A.n, A.n, [&](const double *v, double *y) { la_csr_spmv(c->stream, A, val, v, y); },
                  [&](const double *gg, double *z) { la_ilu_apply(c->stream, A, lu.p, A.ssor, gg, z); }, x, b, g, d, h, opts, info);
}

// lambda_max(D^-1 A_u) from the Lanczos tridiagonal of 25 Jacobi-preconditioned CG steps on a synthetic right-hand side (the constrained rows are
// inert): the largest Ritz value approaches lambda_max from below within a fraction of a percent, far faster than a power iteration
double estimate_lmax_u(poro_ctx *c, const std::function<bool(const double *, double *, double *)> &apply, const DiagVec &dj) {
  hipStream_t s = c->stream; const int64_t n = c->n_u, n_own = owned(c, n, c->comm.part.plane_u);
  std::vector<double> hv(n); for (int64_t i = 0; i < n; ++i) hv[i] = std::sin(0.731 * (double)i) + 0.3 * std::cos(0.013 * (double)i * (double)(i % 7));
  DevBuf<double> r, z, p, ap; r.upload(hv); z.alloc(n); p.alloc(n); ap.alloc(n);
  exchange_add(c, r.p, n, c->comm.part.plane_u);                      // partitioned runs: the start vector was filled by LOCAL index - make the copies of the shared dofs agree (any consistent vector will do)
  la_mask_zero(s, r.p, dj.inert, n);
  la_cheb_first(s, z.p, r.p, dj, 1.0, n);                            // z = D^-1 r
  la_copy(s, p.p, z.p, n);
  double rz = dot_host(c, r.p, z.p, n_own);
  const int K = 25; std::vector<double> al, be;
  for (int k = 0; k < K && rz > 0; ++k) {
    apply(p.p, ap.p, nullptr);
    la_mask_zero(s, ap.p, dj.inert, n);
    const double pap = dot_host(c, p.p, ap.p, n_own);
    if (!(pap > 0)) break;
    const double alpha = rz / pap;
    la_axpy(s, r.p, -alpha, ap.p, n);
    la_cheb_first(s, z.p, r.p, dj, 1.0, n);
    const double rz_new = dot_host(c, r.p, z.p, n_own), beta = rz_new / rz;
    al.push_back(alpha); be.push_back(beta);
    la_xpby(s, p.p, beta, 1.0, z.p, n);                               // p = beta p + z
    rz = rz_new;
  }
  const int m = (int)al.size();
  if (m == 0) return 4.0;
  std::vector<double> T((size_t)m * m, 0.0);
  for (int k = 0; k < m; ++k) {
    T[(size_t)k * m + k] = 1.0 / al[k] + (k > 0 ? be[k - 1] / al[k - 1] : 0.0);
    if (k + 1 < m) T[(size_t)k * m + k + 1] = T[(size_t)(k + 1) * m + k] = std::sqrt(be[k]) / al[k];
  }
  return 1.05 * sym_lambda_max(m, T);
}

}  // namespace ctx_detail
}  // namespace poro
