// =====================================================================================
// poro_oracle.cpp — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (fp64, single thread) of the per-timestep hot path of
// ishovkun/poroelasticity-dealii, used as the parity checker by tests/, by
// __graft_entry__.smoke() and as the `cpu_baseline` leg of bench.py.  Nothing in
// poroelasticity_dealii_amd/ may include, link or call this file.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or expected outputs
//  and deal.II 8.4 — which carries the arithmetic — is neither in
// /root/reference nor installable here, so the reference itself cannot be run.  The
// restatement is pinned instead by analytical known-answer tests (tests/test_oracle_kat.py,
// SURVEY.md §8c K1-K7) that do not depend on deal.II.
//
// Every function cites the reference lines it follows.  deal.II semantics relied upon
// (QGauss, FE_Q, MappingQ1, ConstraintMatrix::distribute_local_to_global, SolverCG,
// PreconditionSSOR) are restated from the library's documented behaviour.
// The mesh / DoF arrays come in through the same poro_desc the HIP library receives, so
// vectors are directly comparable; FE tables are recomputed here independently
// (Newton-iterated Gauss points, product-form Lagrange polynomials) and cross-checked
// against the host provider's tables by the tests.
// =====================================================================================
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <thread>
#include <functional>
#include <chrono>
#include "../include/poroel_hip.h"

namespace {

typedef std::vector<double> Vec;

int ipow(int b, int e) { int r = 1; while (e--) r *= b; return r; }

// ---- QGauss<1>(n) on [0,1]: roots of the Legendre polynomial by Newton iteration -------------
void qgauss01(int n, Vec &x, Vec &w) {
  x.assign(n, 0); w.assign(n, 0);
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < n; ++i) {
    long double z = cosl(pi * (i + 0.75L) / (n + 0.5L)), pp = 0;
    for (int it = 0; it < 100; ++it) {
      long double p1 = 1, p2 = 0;
      for (int j = 0; j < n; ++j) { long double p3 = p2; p2 = p1; p1 = ((2 * j + 1) * z * p2 - j * p3) / (j + 1); }
      pp = n * (z * p1 - p2) / (z * z - 1);
      long double z1 = z; z = z1 - p1 / pp;
      if (fabsl(z - z1) < 1e-19L) break;
    }
    // ascending order on [0,1]
    x[n - 1 - i] = (double)(0.5L * (1 + z));
    w[n - 1 - i] = (double)(1.0L / ((1 - z * z) * pp * pp));
  }
}

// ---- FE_Q(k) 1D Lagrange basis on equidistant support points, product form -----------------
void lagrange(int k, double x, double *v, double *d) {
  for (int i = 0; i <= k; ++i) {
    double val = 1, der = 0;
    for (int j = 0; j <= k; ++j) if (j != i) val *= (x - (double)j / k) / ((double)i / k - (double)j / k);
    for (int m = 0; m <= k; ++m) if (m != i) {
      double t = 1.0 / ((double)i / k - (double)m / k);
      for (int j = 0; j <= k; ++j) if (j != i && j != m) t *= (x - (double)j / k) / ((double)i / k - (double)j / k);
      der += t;
    }
    v[i] = val; d[i] = der;
  }
}

void shapes(int dim, int k, const double *xi, double *val, double *grad) {
  double v1[3][4], d1[3][4];
  for (int d = 0; d < dim; ++d) lagrange(k, xi[d], v1[d], d1[d]);
  const int n1 = k + 1, ns = ipow(n1, dim);
  for (int s = 0; s < ns; ++s) {
    const int idx[3] = {s % n1, (s / n1) % n1, s / (n1 * n1)};
    double v = 1;
    for (int d = 0; d < dim; ++d) v *= v1[d][idx[d]];
    val[s] = v;
    for (int g = 0; g < dim; ++g) {
      double t = 1;
      for (int d = 0; d < dim; ++d) t *= (d == g ? d1[d][idx[d]] : v1[d][idx[d]]);
      grad[s * dim + g] = t;
    }
  }
}

// ---- quadrature on the reference cell ---------------------------------------------------
struct Quad { int n = 0; Vec xi, w; };
Quad make_quad(int dim, int n1) {
  Vec x1, w1; qgauss01(n1, x1, w1);
  Quad Q; Q.n = ipow(n1, dim); Q.xi.assign(Q.n * dim, 0); Q.w.assign(Q.n, 1);
  for (int q = 0; q < Q.n; ++q) {
    const int idx[3] = {q % n1, (q / n1) % n1, q / (n1 * n1)};
    for (int d = 0; d < dim; ++d) { Q.xi[q * dim + d] = x1[idx[d]]; Q.w[q] *= w1[idx[d]]; }
  }
  return Q;
}

// ---- FEValues<dim> restatement: reference tables + reinit(cell) with MappingQ1 -------------
struct FEValues {
  int dim, k, ns, nq;
  Quad quad;
  Vec ref_val, ref_grad, map_val, map_grad;  // [q][s], [q][s][d]; mapping = Q1 on the vertices
  Vec jxw, grad, qpoint;                     // per cell: [q], [q][s][d], [q][d]
  FEValues(int dim_, int k_, const Quad &Q) : dim(dim_), k(k_), ns(ipow(k_ + 1, dim_)), nq(Q.n), quad(Q) {
    const int nv = 1 << dim;
    ref_val.resize(nq * ns); ref_grad.resize(nq * ns * dim); map_val.resize(nq * nv); map_grad.resize(nq * nv * dim);
    for (int q = 0; q < nq; ++q) {
      shapes(dim, k, &quad.xi[q * dim], &ref_val[q * ns], &ref_grad[q * ns * dim]);
      shapes(dim, 1, &quad.xi[q * dim], &map_val[q * nv], &map_grad[q * nv * dim]);
    }
    jxw.resize(nq); grad.resize(nq * ns * dim); qpoint.resize(nq * dim);
  }
  void reinit(const double *X /*[nv][dim] vertex coords of the cell*/) {
    const int nv = 1 << dim;
    for (int q = 0; q < nq; ++q) {
      double J[3][3] = {{0}}, Ji[3][3];
      for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) {
        double s = 0;
        for (int v = 0; v < nv; ++v) s += X[v * dim + a] * map_grad[(q * nv + v) * dim + b];
        J[a][b] = s;
      }
      double det;
      if (dim == 2) {
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        Ji[0][0] = J[1][1] / det; Ji[0][1] = -J[0][1] / det; Ji[1][0] = -J[1][0] / det; Ji[1][1] = J[0][0] / det;
      } else {
        det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
              J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det; Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det; Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
        Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det; Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det; Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det; Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
      }
      jxw[q] = det * quad.w[q];
      for (int s = 0; s < ns; ++s) for (int a = 0; a < dim; ++a) {
        double g = 0;
        for (int b = 0; b < dim; ++b) g += Ji[b][a] * ref_grad[(q * ns + s) * dim + b];
        grad[(q * ns + s) * dim + a] = g;
      }
      for (int a = 0; a < dim; ++a) {
        double s = 0;
        for (int v = 0; v < nv; ++v) s += X[v * dim + a] * map_val[q * nv + v];
        qpoint[q * dim + a] = s;
      }
    }
  }
  double shape_value(int s, int q) const { return ref_val[q * ns + s]; }
  const double *shape_grad(int s, int q) const { return &grad[(q * ns + s) * dim]; }
};

// ---- SparseMatrix<double> restatement (sorted CSR) ---------------------------------------
struct Csr {
  int64_t n = 0;
  std::vector<int64_t> rp; std::vector<int32_t> col; Vec val; std::vector<int64_t> diag;
  void pattern(int64_t n_, int64_t n_cells, int dpc, const int32_t *cell_dofs) {
    // DoFTools::make_sparsity_pattern with keep_constrained_dofs = true: all dofs of a cell couple
    n = n_;
    std::vector<int64_t> cnt(n + 1, 0);
    for (int64_t i = 0; i < n_cells * dpc; ++i) cnt[cell_dofs[i] + 1]++;
    for (int64_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    std::vector<int32_t> adj(cnt[n]); std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int64_t c = 0; c < n_cells; ++c) for (int i = 0; i < dpc; ++i) adj[pos[cell_dofs[c * dpc + i]]++] = (int32_t)c;
    rp.assign(n + 1, 0);
    std::vector<int32_t> row;
    std::vector<std::vector<int32_t>> rows(n);
    for (int64_t r = 0; r < n; ++r) {
      row.clear();
      for (int64_t a = cnt[r]; a < cnt[r + 1]; ++a) { const int32_t *cd = cell_dofs + (int64_t)adj[a] * dpc; row.insert(row.end(), cd, cd + dpc); }
      std::sort(row.begin(), row.end()); row.erase(std::unique(row.begin(), row.end()), row.end());
      rows[r] = row; rp[r + 1] = rp[r] + (int64_t)row.size();
    }
    col.resize(rp[n]); val.assign(rp[n], 0.0); diag.resize(n);
    for (int64_t r = 0; r < n; ++r) {
      std::copy(rows[r].begin(), rows[r].end(), col.begin() + rp[r]);
      diag[r] = find(r, (int32_t)r);
    }
  }
  int64_t find(int64_t r, int32_t c) const {
    auto b = col.begin() + rp[r], e = col.begin() + rp[r + 1];
    auto it = std::lower_bound(b, e, c);
    if (it == e || *it != c) throw std::runtime_error("csr: entry not in pattern");
    return it - col.begin();
  }
  void add(int64_t r, int32_t c, double v) { val[find(r, c)] += v; }
  void vmult(Vec &y, const Vec &x) const {
    for (int64_t r = 0; r < n; ++r) { double s = 0; for (int64_t j = rp[r]; j < rp[r + 1]; ++j) s += val[j] * x[col[j]]; y[r] = s; }
  }
  // SparseMatrix::precondition_SSOR: dst = (D+wU)^-1 w(2-w) D (D+wL)^-1 src, rows in index order
  void precondition_ssor(Vec &dst, const Vec &src, double om) const {
    for (int64_t r = 0; r < n; ++r) {
      double s = 0;
      for (int64_t j = rp[r]; j < diag[r]; ++j) s += val[j] * dst[col[j]];
      dst[r] = (src[r] - s * om) / val[diag[r]];
    }
    for (int64_t r = 0; r < n; ++r) dst[r] *= om * (2. - om) * val[diag[r]];
    for (int64_t r = n - 1; r >= 0; --r) {
      double s = 0;
      for (int64_t j = diag[r] + 1; j < rp[r + 1]; ++j) s += val[j] * dst[col[j]];
      dst[r] = (dst[r] - s * om) / val[diag[r]];
    }
  }
};

// communication hooks: single rank = no-ops; the multi-rank CPU tests plug gloo in here
struct Comm {
  poro_allreduce_fn ar = nullptr; poro_sendrecv_fn sr = nullptr; void *user = nullptr;
  poro_partition part{};
  int64_t n_u = 0;       // length of displacement vectors: tells the two spaces of a general partition apart
  bool multi() const { return part.n_ranks > 1; }
  bool general() const { return part.n_neighbours > 0; }
  // add the neighbours' partial sums on the shared dofs (rows assembled from cells of several ranks)
  void exchange_add(Vec &v, int64_t plane) const {
    if (!multi()) return;
    const int64_t n = (int64_t)v.size();
    if (general()) {
      // pairwise exchange with every neighbour, then the sum in ascending rank order (own contribution at its place): bitwise the same on every
      // rank holding the dof (poroel_hip.h, poro_partition)
      const int64_t *ptr = n == n_u ? part.shared_ptr_u : part.shared_ptr_p; const int32_t *dof = n == n_u ? part.shared_dof_u : part.shared_dof_p;
      const int nn = part.n_neighbours; Vec sbuf(ptr[nn]), rbuf(ptr[nn]);
      for (int64_t j = 0; j < ptr[nn]; ++j) sbuf[j] = v[dof[j]];
      for (int k = 0; k < nn; ++k) if (ptr[k + 1] > ptr[k]) sr(&sbuf[ptr[k]], &rbuf[ptr[k]], ptr[k + 1] - ptr[k], part.neighbour_rank[k], user);
      // per shared dof, walk the neighbours in ascending rank order; the own value joins before the first higher-ranked neighbour
      Vec acc(n, 0.0); std::vector<char> seen(n, 0), own_added(n, 0);
      const Vec v0 = v;
      for (int k = 0; k < nn; ++k) {
        const bool self_before = part.rank < part.neighbour_rank[k];
        for (int64_t j = ptr[k]; j < ptr[k + 1]; ++j) { const int32_t i = dof[j];
          if (self_before && !own_added[i]) { acc[i] += v0[i]; own_added[i] = 1; }
          acc[i] += rbuf[j]; seen[i] = 1; }
      }
      for (int64_t i = 0; i < n; ++i) if (seen[i]) { if (!own_added[i]) acc[i] += v0[i]; v[i] = acc[i]; }
      return;
    }
    Vec rbuf(plane);
    if (part.has_upper) { sr(&v[n - plane], rbuf.data(), plane, part.rank + 1, user); for (int64_t i = 0; i < plane; ++i) v[n - plane + i] += rbuf[i]; }
    if (part.has_lower) { sr(&v[0], rbuf.data(), plane, part.rank - 1, user); for (int64_t i = 0; i < plane; ++i) v[i] += rbuf[i]; }
  }
  int64_t owned(int64_t n, int64_t plane) const {
    if (!multi()) return n;
    if (general()) return n == n_u ? part.n_owned_u : part.n_owned_p;
    return part.has_upper ? n - plane : n;                               // a shared plane is counted by its upper owner
  }
  double dot(const Vec &a, const Vec &b, int64_t plane) const {
    const int64_t n = owned((int64_t)a.size(), plane);
    double s = 0; for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    if (multi()) ar(&s, 1, user);
    return s;
  }
};

struct SolveInfo { int iterations = 0; int converged = 0; double r0 = 0, r = 0; };

// Closed ConstraintMatrix beyond the Dirichlet list: hanging nodes (DoFTools::make_hanging_node_constraints, PoroElasticDisplacementSolver.h:112-113,
// PoroElasticPressureSolver.h:72-75).  deal.II condenses the assembled system (condense / distribute_local_to_global, PoroElasticPressureSolver.h:153,168,
// PoroElasticDisplacementSolver.h:280-286); on the unconstrained dofs that system is C^T A C x = C^T (b - A x_inh), x = C x_free + x_inh, which is
// what expand / reduce below apply around the unconstrained operator; distribute (:180, :306) fills the constrained entries afterwards.
struct Cons {
  std::vector<int32_t> dof, master; std::vector<int64_t> ptr; Vec weight, inhom;
  void init(const poro_constraints &c) {
    if (c.n <= 0) return;
    dof.assign(c.dof, c.dof + c.n); ptr.assign(c.ptr, c.ptr + c.n + 1); inhom.assign(c.inhomogeneity, c.inhomogeneity + c.n);
    if (ptr[c.n]) { master.assign(c.master, c.master + ptr[c.n]); weight.assign(c.weight, c.weight + ptr[c.n]); }
  }
  bool any() const { return !dof.empty(); }
  void expand(Vec &x, bool with_inhom) const {           // ConstraintMatrix::distribute
    for (size_t i = 0; i < dof.size(); ++i) { double s = with_inhom ? inhom[i] : 0.0; for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) s += weight[k] * x[master[k]]; x[dof[i]] = s; }
  }
  void reduce(Vec &y) const {                            // ConstraintMatrix::condense(vector)
    for (size_t i = 0; i < dof.size(); ++i) { for (int64_t k = ptr[i]; k < ptr[i + 1]; ++k) y[master[k]] += weight[k] * y[dof[i]]; }
    for (size_t i = 0; i < dof.size(); ++i) y[dof[i]] = 0.0;
  }
};

struct Oracle {
  poro_desc d; int dim, k_u, ns_u, ns_p, dpc_u, dpc_p, nv;
  std::vector<double> vx; std::vector<int32_t> cv, cdu, cdp, bfc, bfl, bfi, ddof, nlab, ncomp; Vec dval, nval;
  std::vector<char> is_dir; Vec dir_val;
  poro_material mat;
  Comm comm;
  bool hoisted = false;  // false: the reference's naive i x q x j loop (Q6); true: same numbers, C:eps_i hoisted
  // displacement solver state (PoroElasticDisplacementSolver.h:41-56)
  Csr A; Vec rhs_u, u; bool rebuild_system_matrix = true;
  // pressure solver state (PoroElasticPressureSolver.h:36-45)
  Csr Mp, Kp, Jp; Vec p, dp, p_old, residual, tmp1, tmp2, source;
  // projector (StrainProjector.h:42-43)
  Csr Pm; std::vector<Vec> proj_rhs, strains;
  Vec eps_v, eps_v0;
  int tensor_to_entry[9];
  int64_t work[6] = {0, 0, 0, 0, 0, 0};
  int64_t work_init[6] = {0, 0, 0, 0, 0, 0}; double seconds_init = 0, seconds_steps = 0;   // oracle_run: work counters and wall time of the initialisation (:308-317) / of the time steps   // apply_u, apply_p, asm_rhs_u, residual_p, jacobian_p, proj_rhs (same units as the host driver's counters)
  Cons cons_u, cons_p;      // hanging-node constraints of the two spaces (empty on uniform meshes)
  std::vector<char> is_pdir; Vec pdir_val; bool any_pdir = false;   // EXTENSION (not in the reference): prescribed pressures, e.g. a drained boundary (include/poroel_hip.h)
  int stop_rule_u = 0;      // stopping rule of the displacement solve (see cg)
  int n_noconvergence = 0;  // SolverControl::NoConvergence would have been thrown this many times

  explicit Oracle(const poro_desc *dd) : d(*dd) {
    dim = d.dim; k_u = d.degree_u; nv = 1 << dim;
    ns_u = ipow(k_u + 1, dim); ns_p = nv; dpc_u = ns_u * dim; dpc_p = ns_p;
    vx.assign(d.vertex_coords, d.vertex_coords + d.n_vertices * dim);
    cv.assign(d.cell_vertices, d.cell_vertices + d.n_cells * nv);
    cdu.assign(d.cell_dofs_u, d.cell_dofs_u + d.n_cells * dpc_u);
    cdp.assign(d.cell_dofs_p, d.cell_dofs_p + d.n_cells * dpc_p);
    bfc.assign(d.bface_cell, d.bface_cell + d.n_bfaces); bfl.assign(d.bface_local, d.bface_local + d.n_bfaces); bfi.assign(d.bface_id, d.bface_id + d.n_bfaces);
    ddof.assign(d.dirichlet_dof, d.dirichlet_dof + d.n_dirichlet); dval.assign(d.dirichlet_value, d.dirichlet_value + d.n_dirichlet);
    nlab.assign(d.neumann_label, d.neumann_label + d.n_neumann); ncomp.assign(d.neumann_component, d.neumann_component + d.n_neumann); nval.assign(d.neumann_value, d.neumann_value + d.n_neumann);
    mat = d.mat; comm.part = d.part; comm.n_u = d.n_dofs_u;
    cons_u.init(d.cons_u); cons_p.init(d.cons_p);
    is_pdir.assign(d.n_dofs_p, 0); pdir_val.assign(d.n_dofs_p, 0.0); any_pdir = d.n_dirichlet_p > 0;
    for (int64_t i = 0; i < d.n_dirichlet_p; ++i) { is_pdir[d.dirichlet_dof_p[i]] = 1; pdir_val[d.dirichlet_dof_p[i]] = d.dirichlet_value_p[i]; }
    // TensorIndexer.h:18-35
    if (dim == 2) { const int t[4] = {0, 1, 1, 2}; std::copy(t, t + 4, tensor_to_entry); }
    else { const int t[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5}; std::copy(t, t + 9, tensor_to_entry); }
    setup_dofs();
  }

  void cell_coords(int64_t c, double *X) const {
    for (int v = 0; v < nv; ++v) for (int a = 0; a < dim; ++a) X[v * dim + a] = vx[(int64_t)cv[c * nv + v] * dim + a];
  }

  // PoroElasticDisplacementSolver::setup_dofs :106-153, PoroElasticPressureSolver::setup_dofs :68-111,
  // StrainProjector::setup_dofs :82-98, PoroElasticProblem::setup_dofs PoroelasticityFSS.h:131-151
  void setup_dofs() {
    is_dir.assign(d.n_dofs_u, 0); dir_val.assign(d.n_dofs_u, 0);
    for (size_t i = 0; i < ddof.size(); ++i) { is_dir[ddof[i]] = 1; dir_val[ddof[i]] = dval[i]; }
    rebuild_system_matrix = true;                                    // :137
    A.pattern(d.n_dofs_u, d.n_cells, dpc_u, cdu.data());              // :142-149
    rhs_u.assign(d.n_dofs_u, 0); u.assign(d.n_dofs_u, 0);             // :150-151
    Mp.pattern(d.n_dofs_p, d.n_cells, dpc_p, cdp.data()); Kp = Mp; Jp = Mp;   // :80-94
    create_mass_and_laplace();                                       // :96-101
    for (Vec *v : {&p, &dp, &p_old, &residual, &tmp1, &tmp2, &source, &eps_v, &eps_v0}) v->assign(d.n_dofs_p, 0);
    const int n_sym = dim * (dim + 1) / 2;
    proj_rhs.assign(n_sym, Vec(d.n_dofs_p, 0)); strains.assign(n_sym, Vec(d.n_dofs_p, 0));
    Pm = Mp;
  }

  // MatrixCreator::create_mass_matrix / create_laplace_matrix with QGauss(fe.degree+1) (:96-101)
  void create_mass_and_laplace() {
    FEValues fv(dim, 1, make_quad(dim, 2));
    std::vector<double> X(nv * dim);
    for (int64_t c = 0; c < d.n_cells; ++c) {
      cell_coords(c, X.data()); fv.reinit(X.data());
      for (int i = 0; i < dpc_p; ++i) for (int j = 0; j < dpc_p; ++j) {
        double m = 0, k = 0;
        for (int q = 0; q < fv.nq; ++q) {
          m += fv.shape_value(i, q) * fv.shape_value(j, q) * fv.jxw[q];
          double g = 0; for (int a = 0; a < dim; ++a) g += fv.shape_grad(i, q)[a] * fv.shape_grad(j, q)[a];
          k += g * fv.jxw[q];
        }
        Mp.add(cdp[c * dpc_p + i], cdp[c * dpc_p + j], m); Kp.add(cdp[c * dpc_p + i], cdp[c * dpc_p + j], k);
      }
    }
  }

  // constitutive_model::get_strain_tensor(FEValues&, i, q) ConstitutiveModel.h:9-24 — symmetric gradient of
  // vector shape function i (= scalar node s, component comp); full dim x dim storage
  void strain_of_shape(const FEValues &fv, int s, int comp, int q, double e[3][3]) const {
    for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) {
      const double gab = (a == comp) ? fv.shape_grad(s, q)[b] : 0.0;  // shape_grad_component(i,q,a)[b]
      const double gba = (b == comp) ? fv.shape_grad(s, q)[a] : 0.0;
      e[a][b] = (gab + gba) / 2;
    }
  }

  // PoroElasticDisplacementSolver::assemble_system :155-291
  void assemble_system() {
    FEValues fv(dim, k_u, make_quad(dim, k_u + 1));                  // :159,162
    FEValues pfv(dim, 1, make_quad(dim, k_u + 1));                   // :167 (pressure FE on the u quadrature)
    const int nq = fv.nq;
    // isotropic_gassman_tensor ConstitutiveModel.h:45-57
    double C[3][3][3][3];
    for (int i = 0; i < dim; ++i) for (int j = 0; j < dim; ++j) for (int k = 0; k < dim; ++k) for (int l = 0; l < dim; ++l)
      C[i][j][k][l] = ((i == k && j == l) ? mat.shear_G : 0.0) + ((i == l && j == k) ? mat.shear_G : 0.0) + ((i == j && k == l) ? mat.lame_lambda : 0.0);
    Vec cell_matrix(dpc_u * dpc_u), cell_rhs(dpc_u), pressure_values(nq);
    std::vector<double> X(nv * dim);
    // face data (FEFaceValues :169-173)
    Vec x1, w1; qgauss01(k_u + 1, x1, w1);
    const int nqf = ipow(k_u + 1, dim - 1);
    std::vector<std::vector<int>> faces_of(d.n_cells);
    for (int64_t b = 0; b < d.n_bfaces; ++b) faces_of[bfc[b]].push_back((int)b);

    std::fill(rhs_u.begin(), rhs_u.end(), 0.0);                      // :204 (matrix is not zeroed here)
    for (int64_t c = 0; c < d.n_cells; ++c) {                        // :206
      std::fill(cell_matrix.begin(), cell_matrix.end(), 0.0); std::fill(cell_rhs.begin(), cell_rhs.end(), 0.0);
      cell_coords(c, X.data()); fv.reinit(X.data());                 // :209-210
      for (int q = 0; q < nq; ++q) {                                 // get_function_values :211-212
        double s = 0; for (int k = 0; k < ns_p; ++k) s += p[cdp[c * dpc_p + k]] * pfv.shape_value(k, q);
        pressure_values[q] = s;
      }
      // body force: BodyForces(rho, d=3) => zero in 2D, out of bounds in 3D (right_hand_side.h:69-71): parity = 0
      for (int i = 0; i < dpc_u; ++i) {                              // :216
        const int si = i / dim, ci = i % dim;                        // system_to_component_index :218
        for (int q = 0; q < nq; ++q) {                               // :220
          const double jxw = fv.jxw[q];
          cell_rhs[i] += (fv.shape_value(si, q) * 0.0) * jxw;        // :223-225
          double ei[3][3]; strain_of_shape(fv, si, ci, q, ei);       // :230-231
          double tr = 0; for (int a = 0; a < dim; ++a) tr += ei[a][a];
          cell_rhs[i] += (mat.biot_alpha * pressure_values[q] * tr) * jxw;   // :232-234
          if (hoisted) {
            double sig[3][3];
            for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) { double s = 0; for (int k = 0; k < dim; ++k) for (int l = 0; l < dim; ++l) s += C[a][b][k][l] * ei[k][l]; sig[a][b] = s; }
            for (int j = 0; j < dpc_u; ++j) {
              double ej[3][3]; strain_of_shape(fv, j / dim, j % dim, q, ej);
              double s = 0; for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) s += sig[a][b] * ej[a][b];
              cell_matrix[i * dpc_u + j] += s * jxw;
            }
          } else {
            for (int j = 0; j < dpc_u; ++j) {                        // :237-242
              double ej[3][3]; strain_of_shape(fv, j / dim, j % dim, q, ej);
              double sig[3][3];
              for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) { double s = 0; for (int k = 0; k < dim; ++k) for (int l = 0; l < dim; ++l) s += C[a][b][k][l] * ei[k][l]; sig[a][b] = s; }
              double s = 0; for (int a = 0; a < dim; ++a) for (int b = 0; b < dim; ++b) s += sig[a][b] * ej[a][b];
              cell_matrix[i * dpc_u + j] += s * jxw;
            }
          }
        }
      }
      // Neumann: :249-277
      for (int b : faces_of[c]) {
        const int f = bfl[b], nd = f / 2, side = f % 2;
        for (size_t l = 0; l < nlab.size(); ++l) {
          if (bfi[b] != nlab[l]) continue;
          for (int qf = 0; qf < nqf; ++qf) {
            double xi[3] = {0, 0, 0}; int rem = qf; double w = 1;
            for (int a = 0; a < dim; ++a) { if (a == nd) { xi[a] = side; continue; } const int i1 = rem % (k_u + 1); rem /= (k_u + 1); xi[a] = x1[i1]; w *= w1[i1]; }
            Vec sv(ns_u), sg(ns_u * dim), mv(nv), mg(nv * dim);
            shapes(dim, k_u, xi, sv.data(), sg.data()); shapes(dim, 1, xi, mv.data(), mg.data());
            double J[3][3] = {{0}};
            for (int a = 0; a < dim; ++a) for (int bb = 0; bb < dim; ++bb) for (int v = 0; v < nv; ++v) J[a][bb] += X[v * dim + a] * mg[v * dim + bb];
            // n dS = det(J) J^{-T} n_ref = cofactor column nd of J (times the side sign)
            double cof[3];
            if (dim == 2) { if (nd == 0) { cof[0] = J[1][1]; cof[1] = -J[0][1]; } else { cof[0] = -J[1][0]; cof[1] = J[0][0]; } }
            else { const int a1 = (nd + 1) % 3, a2 = (nd + 2) % 3; for (int r = 0; r < 3; ++r) { const int r1 = (r + 1) % 3, r2 = (r + 2) % 3; cof[r] = J[r1][a1] * J[r2][a2] - J[r1][a2] * J[r2][a1]; } }
            double len = 0; for (int a = 0; a < dim; ++a) len += cof[a] * cof[a]; len = std::sqrt(len);
            const double sgn = side ? 1.0 : -1.0, jxwf = len * w;
            for (int i = 0; i < dpc_u; ++i) {
              const int ci = i % dim; if (ci != ncomp[l]) continue;           // :263
              const double neumann_value = nval[l] * (sgn * cof[ci] / len);   // :265-267
              cell_rhs[i] += sv[i / dim] * neumann_value * jxwf;              // :269-272
            }
          }
        }
      }
      // constraints.distribute_local_to_global :279-286 (ConstraintMatrix semantics)
      const int32_t *idx = &cdu[c * dpc_u];
      for (int i = 0; i < dpc_u; ++i) {
        if (is_dir[idx[i]]) {
          if (rebuild_system_matrix) A.add(idx[i], idx[i], std::fabs(cell_matrix[i * dpc_u + i]));
          continue;
        }
        double r = cell_rhs[i];
        for (int j = 0; j < dpc_u; ++j) {
          if (is_dir[idx[j]]) r -= cell_matrix[i * dpc_u + j] * dir_val[idx[j]];
          else if (rebuild_system_matrix) A.add(idx[i], idx[j], cell_matrix[i * dpc_u + j]);
        }
        rhs_u[idx[i]] += r;
      }
    }
    rebuild_system_matrix = false;                                   // :290
    work[2]++;
    comm.exchange_add(rhs_u, d.part.plane_u);
    if (cons_u.any()) {                                              // hanging-node part of distribute_local_to_global (:280-286): C^T (b - A x_inh)
      Vec xin(d.n_dofs_u, 0.0), t(d.n_dofs_u, 0.0); cons_u.expand(xin, true);
      A.vmult(t, xin);
      for (int64_t i = 0; i < d.n_dofs_u; ++i) if (!is_dir[i]) rhs_u[i] -= t[i];
      cons_u.reduce(rhs_u);
    }
  }

  // SolverCG<>::solve with PreconditionSSOR (deal.II 8.4 semantics); prec: 0 none, 1 Jacobi, 2 SSOR(omega).
  // Multi-rank: the local matrix holds this slab's partial rows; shared-plane rows are completed by exchange_add.
  // stop_rule 0: ||g|| <= max(abs_tol, rel_tol ||b||) (the reference's SolverControl objects); 1: rel_tol against the residual of the warm
  // start (deal.II ReductionControl) - the stated rule of the transient benchmark, mirrored from include/poroel_hip.h PORO_STOP_REDUCTION
  SolveInfo cg(const Csr &M, Vec &x, const Vec &b, double abs_tol, double rel_tol, int max_iter, int prec, double omega, int64_t plane, int stop_rule = 0, const Cons *cons = nullptr,
               const std::vector<char> *frozen = nullptr) {
    const int64_t n = M.n; Vec g(n), dvec(n), h(n), diagv, tmp;
    if (cons && !cons->any()) cons = nullptr;
    if ((cons || frozen) && prec == 2) prec = 1;   // the condensed matrix exists at operator level only: Jacobi instead of the SSOR sweeps
    SolveInfo info;
    int64_t &napply = work[&M == &A ? 0 : 1];
    if (prec == 1 || comm.multi()) { diagv.resize(n); for (int64_t r = 0; r < n; ++r) diagv[r] = M.val[M.diag[r]]; comm.exchange_add(diagv, plane); }
    if (comm.multi() && prec == 2) prec = 1;  // SSOR is rank-local-order dependent; multi-rank oracle uses Jacobi
    auto apply = [&](Vec &y, const Vec &v) {
      if (cons) { tmp = v; cons->expand(tmp, false); M.vmult(y, tmp); cons->reduce(y); }
      else M.vmult(y, v);
      if (frozen) for (int64_t i = 0; i < n; ++i) if ((*frozen)[i]) y[i] = 0.0;   // rows taken out of the system (their unknowns stay at the warm start's value)
      comm.exchange_add(y, plane); ++napply;
    };
    auto precond = [&](Vec &y, const Vec &v) {
      if (prec == 2) M.precondition_ssor(y, v, omega);
      else if (prec == 1) for (int64_t r = 0; r < n; ++r) y[r] = v[r] / diagv[r];
      else y = v;
    };
    apply(g, x); for (int64_t i = 0; i < n; ++i) g[i] -= b[i];
    double res = std::sqrt(comm.dot(g, g, plane));
    const double tol = std::max(abs_tol, rel_tol * (stop_rule == 1 ? res : std::sqrt(comm.dot(b, b, plane))));
    info.r0 = res; info.r = res;
    if (res <= tol) { info.converged = 1; return info; }
    precond(h, g); for (int64_t i = 0; i < n; ++i) dvec[i] = -h[i];
    double gh = comm.dot(g, h, plane);
    int it = 0;
    while (true) {
      ++it;
      apply(h, dvec);
      double alpha = gh / comm.dot(dvec, h, plane);
      for (int64_t i = 0; i < n; ++i) { g[i] += alpha * h[i]; x[i] += alpha * dvec[i]; }
      res = std::sqrt(comm.dot(g, g, plane));
      info.iterations = it; info.r = res;
      if (res <= tol) { info.converged = 1; break; }
      if (it >= max_iter) { info.converged = 0; ++n_noconvergence; break; }   // SolverControl::NoConvergence
      precond(h, g);
      const double beta_old = gh; gh = comm.dot(g, h, plane);
      const double beta = gh / beta_old;
      for (int64_t i = 0; i < n; ++i) dvec[i] = beta * dvec[i] - h[i];
    }
    return info;
  }

  // PoroElasticDisplacementSolver::solve :294-307
  SolveInfo disp_solve(double abs_tol, double rel_tol, int max_iter, int prec, double omega) {
    SolveInfo s = cg(A, u, rhs_u, abs_tol, rel_tol, max_iter, prec, omega, d.part.plane_u, stop_rule_u, &cons_u);
    for (size_t i = 0; i < ddof.size(); ++i) u[ddof[i]] = dval[i];    // constraints.distribute :306
    cons_u.expand(u, true);
    return s;
  }

  // VectorTools::create_right_hand_side with SinglePhaseWell (PoroElasticPressureSolver.h:142-147, right_hand_side.h:99-116)
  void well_source(Vec &out) {
    FEValues fv(dim, 1, make_quad(dim, 2));
    std::vector<double> X(nv * dim);
    std::fill(out.begin(), out.end(), 0.0);
    for (int64_t c = 0; c < d.n_cells; ++c) {
      cell_coords(c, X.data()); fv.reinit(X.data());
      for (int q = 0; q < fv.nq; ++q) {
        const double *xq = &fv.qpoint[q * dim];
        const double r2 = xq[0] * xq[0] + xq[1] * xq[1];
        const double s = (r2 <= mat.r_well * mat.r_well) ? -mat.flow_rate / (3.1415926 * mat.r_well * mat.r_well) : 0.0;
        for (int i = 0; i < dpc_p; ++i) out[cdp[c * dpc_p + i]] += fv.shape_value(i, q) * s * fv.jxw[q];
      }
    }
    comm.exchange_add(out, d.part.plane_p);
  }

  // PoroElasticPressureSolver::assemble_residual :113-155
  double assemble_residual(double dt) {
    const int64_t n = d.n_dofs_p;
    for (int64_t i = 0; i < n; ++i) tmp1[i] = (eps_v[i] - eps_v0[i]) * (mat.biot_alpha / dt);          // :122-124
    for (int64_t i = 0; i < n; ++i) { tmp2[i] = (p[i] - p_old[i]) * (1. / mat.biot_M / dt); tmp1[i] += tmp2[i]; }  // :128-132
    Mp.vmult(residual, tmp1);                                                                       // :133
    Kp.vmult(tmp1, p); for (int64_t i = 0; i < n; ++i) { tmp1[i] *= mat.k_over_mu; residual[i] += tmp1[i]; }   // :136-139
    comm.exchange_add(residual, d.part.plane_p);
    well_source(source);                                                                            // :142-147
    for (int64_t i = 0; i < n; ++i) { residual[i] += source[i]; residual[i] *= -1; }                 // :148,152
    cons_p.reduce(residual);                                                                        // constraints.condense(residual) :153
    if (any_pdir) for (int64_t i = 0; i < n; ++i) if (is_pdir[i]) residual[i] = 0.0;                  // extension: prescribed-pressure rows
    work[3]++;
    return std::sqrt(comm.dot(residual, residual, d.part.plane_p));                                 // PoroelasticityFSS.h:364
  }
  // PoroElasticPressureSolver::assemble_jacobian :158-169
  void assemble_jacobian(double dt) {
    work[4]++;
    for (size_t j = 0; j < Jp.val.size(); ++j) Jp.val[j] = Mp.val[j] * (1. / mat.biot_M / dt) + (mat.k_over_mu) * Kp.val[j];
  }
  // PoroElasticPressureSolver::solve :172-185
  SolveInfo pres_solve(double abs_tol, double rel_tol, int max_iter, int prec, double omega) {
    SolveInfo s = cg(Jp, dp, residual, abs_tol, rel_tol, max_iter, prec, omega, d.part.plane_p, 0, &cons_p, any_pdir ? &is_pdir : nullptr);   // condensed Jacobian :168
    cons_p.expand(dp, true);                                                                        // constraints.distribute :180
    return s;
  }
  // PoroElasticPressureSolver::update_volumetric_strain :187-194
  void update_volumetric_strain() {
    for (int64_t i = 0; i < d.n_dofs_p; ++i) eps_v[i] += dp[i] * (mat.biot_alpha / mat.bulk_K);
  }

  // StrainProjector::assemble_projection_matrix :101-106
  void assemble_projection_matrix() { Pm.val = Mp.val; }
  // StrainProjector::assemble_projection_rhs :109-198
  void assemble_projection_rhs(const int32_t *comps, int ncomp_) {
    const Quad Q = make_quad(dim, 2);                                // :126
    FEValues pfv(dim, 1, Q), ufv(dim, k_u, Q);                        // :127-134
    std::vector<double> X(nv * dim);
    for (int c = 0; c < ncomp_; ++c) std::fill(proj_rhs[tensor_to_entry[comps[c]]].begin(), proj_rhs[tensor_to_entry[comps[c]]].end(), 0.0);   // :146-147
    std::vector<Vec> cell_rhs(ncomp_, Vec(dpc_p));
    for (int64_t cell = 0; cell < d.n_cells; ++cell) {               // :159
      for (auto &v : cell_rhs) std::fill(v.begin(), v.end(), 0.0);
      cell_coords(cell, X.data()); pfv.reinit(X.data()); ufv.reinit(X.data());
      for (int q = 0; q < Q.n; ++q) {                                // :168
        double grads[3][3] = {{0}};                                  // get_function_gradients :164-165
        for (int i = 0; i < dpc_u; ++i) { const double ui = u[cdu[cell * dpc_u + i]]; const double *g = ufv.shape_grad(i / dim, q); for (int a = 0; a < dim; ++a) grads[i % dim][a] += ui * g[a]; }
        double strain[3][3];                                         // get_strain_tensor(grad) ConstitutiveModel.h:27-42
        for (int a = 0; a < dim; ++a) strain[a][a] = grads[a][a];
        for (int a = 0; a < dim; ++a) for (int b = a + 1; b < dim; ++b) strain[a][b] = strain[b][a] = (grads[a][b] + grads[b][a]) / 2;
        const double jxw = pfv.jxw[q];                               // :170
        for (int i = 0; i < dpc_p; ++i) {                            // :173
          const double phi_i = pfv.shape_value(i, q);
          for (int c = 0; c < ncomp_; ++c) cell_rhs[c][i] += (phi_i * strain[comps[c] / dim][comps[c] % dim] * jxw);   // :176-185
        }
      }
      for (int c = 0; c < ncomp_; ++c) for (int i = 0; i < dpc_p; ++i) proj_rhs[tensor_to_entry[comps[c]]][cdp[cell * dpc_p + i]] += cell_rhs[c][i];   // :191-194
    }
    work[5]++;
    for (int c = 0; c < ncomp_; ++c) { comm.exchange_add(proj_rhs[tensor_to_entry[comps[c]]], d.part.plane_p); cons_p.reduce(proj_rhs[tensor_to_entry[comps[c]]]); }   // :191-194 through the pressure constraints
  }
  // StrainProjector::solve_projection_system :201-232
  SolveInfo proj_solve(int entry, double abs_tol, double rel_tol, int max_iter, int prec, double omega) {
    SolveInfo s = cg(Pm, strains[entry], proj_rhs[entry], abs_tol, rel_tol, max_iter, prec, omega, d.part.plane_p, 0, &cons_p);
    cons_p.expand(strains[entry], true);                                                            // constraints.distribute :216
    return s;
  }
  // PoroElasticProblem::get_volumetric_strain PoroelasticityFSS.h:179-186
  void get_volumetric_strain() {
    std::fill(eps_v.begin(), eps_v.end(), 0.0);
    for (int a = 0; a < dim; ++a) { const Vec &s = strains[tensor_to_entry[a * dim + a]]; for (int64_t i = 0; i < d.n_dofs_p; ++i) eps_v[i] += s[i]; }
  }
};

Vec *vec_of(Oracle *o, int which) {
  switch (which) {
    case PORO_VEC_U: return &o->u; case PORO_VEC_RHS_U: return &o->rhs_u; case PORO_VEC_P: return &o->p; case PORO_VEC_P_OLD: return &o->p_old;
    case PORO_VEC_DP: return &o->dp; case PORO_VEC_RESIDUAL_P: return &o->residual; case PORO_VEC_EPSV: return &o->eps_v; case PORO_VEC_EPSV0: return &o->eps_v0;
    case PORO_VEC_SOURCE_P: return &o->source;
  }
  const int n_sym = o->dim * (o->dim + 1) / 2;
  if (which >= PORO_VEC_STRAIN0 && which < PORO_VEC_STRAIN0 + n_sym) return &o->strains[which - PORO_VEC_STRAIN0];
  if (which >= PORO_VEC_PROJ_RHS0 && which < PORO_VEC_PROJ_RHS0 + n_sym) return &o->proj_rhs[which - PORO_VEC_PROJ_RHS0];
  return nullptr;
}
Csr *mat_of(Oracle *o, int which) {
  switch (which) { case PORO_MAT_A_U: return &o->A; case PORO_MAT_MASS_P: return &o->Mp; case PORO_MAT_LAPLACE_P: return &o->Kp; case PORO_MAT_JACOBIAN_P: return &o->Jp; }
  return nullptr;
}
void fill_info(const SolveInfo &s, poro_solve_info *info) {
  if (!info) return;
  info->iterations = s.iterations; info->converged = s.converged; info->initial_residual = s.r0; info->final_residual = s.r;
  info->operator_applications = s.iterations + 1; info->seconds = 0;
}

}  // namespace

// ------------------------------- C API for the tests -----------------------------------------
extern "C" {

typedef struct oracle_ctx oracle_ctx;
enum { ORACLE_PREC_NONE = 0, ORACLE_PREC_JACOBI = 1, ORACLE_PREC_SSOR = 2 };

int oracle_create(const poro_desc *d, oracle_ctx **out) {
  try { *out = reinterpret_cast<oracle_ctx *>(new Oracle(d)); return 0; }
  catch (const std::exception &e) { std::fprintf(stderr, "oracle_create: %s\n", e.what()); return -1; }
}
void oracle_destroy(oracle_ctx *c) { delete reinterpret_cast<Oracle *>(c); }
void oracle_set_hoisted(oracle_ctx *c, int on) { reinterpret_cast<Oracle *>(c)->hoisted = on != 0; }
void oracle_set_comm(oracle_ctx *c, poro_allreduce_fn ar, poro_sendrecv_fn sr, void *user) {
  Oracle *o = reinterpret_cast<Oracle *>(c); o->comm.ar = ar; o->comm.sr = sr; o->comm.user = user;
}
int oracle_vec_set(oracle_ctx *c, int which, const double *h, int64_t n) {
  Vec *v = vec_of(reinterpret_cast<Oracle *>(c), which); if (!v || (int64_t)v->size() != n) return -1; std::copy(h, h + n, v->begin()); return 0;
}
int oracle_vec_get(oracle_ctx *c, int which, double *h, int64_t n) {
  Vec *v = vec_of(reinterpret_cast<Oracle *>(c), which); if (!v || (int64_t)v->size() != n) return -1; std::copy(v->begin(), v->end(), h); return 0;
}
int oracle_vec_fill(oracle_ctx *c, int which, double val) { Vec *v = vec_of(reinterpret_cast<Oracle *>(c), which); if (!v) return -1; std::fill(v->begin(), v->end(), val); return 0; }
int oracle_disp_assemble_system(oracle_ctx *c, int rebuild) {
  Oracle *o = reinterpret_cast<Oracle *>(c);
  if (rebuild) { std::fill(o->A.val.begin(), o->A.val.end(), 0.0); o->rebuild_system_matrix = true; }
  o->assemble_system(); return 0;
}
int oracle_disp_solve(oracle_ctx *c, double abs_tol, double rel_tol, int max_iter, int prec, double omega, poro_solve_info *info) {
  SolveInfo s = reinterpret_cast<Oracle *>(c)->disp_solve(abs_tol, rel_tol, max_iter, prec, omega); fill_info(s, info); return s.converged ? 0 : 1;
}
int oracle_pres_assemble_residual(oracle_ctx *c, double dt, double *l2) { double r = reinterpret_cast<Oracle *>(c)->assemble_residual(dt); if (l2) *l2 = r; return 0; }
int oracle_pres_assemble_jacobian(oracle_ctx *c, double dt) { reinterpret_cast<Oracle *>(c)->assemble_jacobian(dt); return 0; }
int oracle_pres_solve(oracle_ctx *c, double abs_tol, double rel_tol, int max_iter, int prec, double omega, poro_solve_info *info) {
  SolveInfo s = reinterpret_cast<Oracle *>(c)->pres_solve(abs_tol, rel_tol, max_iter, prec, omega); fill_info(s, info); return s.converged ? 0 : 1;
}
int oracle_pres_update_volumetric_strain(oracle_ctx *c) { reinterpret_cast<Oracle *>(c)->update_volumetric_strain(); return 0; }
int oracle_proj_assemble_matrix(oracle_ctx *c) { reinterpret_cast<Oracle *>(c)->assemble_projection_matrix(); return 0; }
int oracle_proj_assemble_rhs(oracle_ctx *c, const int32_t *comps, int32_t n) { reinterpret_cast<Oracle *>(c)->assemble_projection_rhs(comps, n); return 0; }
int oracle_proj_solve(oracle_ctx *c, int32_t entry, double abs_tol, double rel_tol, int max_iter, int prec, double omega, poro_solve_info *info) {
  SolveInfo s = reinterpret_cast<Oracle *>(c)->proj_solve(entry, abs_tol, rel_tol, max_iter, prec, omega); fill_info(s, info); return s.converged ? 0 : 1;
}
void oracle_work_counts(oracle_ctx *c, int64_t *out, int reset) { Oracle *o = reinterpret_cast<Oracle *>(c); std::copy(o->work, o->work + 6, out); if (reset) std::fill(o->work, o->work + 6, 0); }
void oracle_set_stop_rule(oracle_ctx *c, int rule) { reinterpret_cast<Oracle *>(c)->stop_rule_u = rule; }
int oracle_noconvergence_count(oracle_ctx *c) { return reinterpret_cast<Oracle *>(c)->n_noconvergence; }
int oracle_get_volumetric_strain(oracle_ctx *c) { reinterpret_cast<Oracle *>(c)->get_volumetric_strain(); return 0; }
int oracle_export_csr_size(oracle_ctx *c, int which, int64_t *n, int64_t *nnz) {
  Csr *m = mat_of(reinterpret_cast<Oracle *>(c), which); if (!m) return -1; *n = m->n; *nnz = (int64_t)m->col.size(); return 0;
}
int oracle_export_csr(oracle_ctx *c, int which, int64_t *rp, int32_t *col, double *val) {
  Csr *m = mat_of(reinterpret_cast<Oracle *>(c), which); if (!m) return -1;
  std::copy(m->rp.begin(), m->rp.end(), rp); std::copy(m->col.begin(), m->col.end(), col); std::copy(m->val.begin(), m->val.end(), val); return 0;
}
int oracle_apply_operator(oracle_ctx *c, int which, const double *x, double *y) {
  Oracle *o = reinterpret_cast<Oracle *>(c); Csr *m = mat_of(o, which); if (!m) return -1;
  Vec xv(x, x + m->n), yv(m->n); m->vmult(yv, xv);
  o->comm.exchange_add(yv, which == PORO_MAT_A_U ? o->d.part.plane_u : o->d.part.plane_p);
  std::copy(yv.begin(), yv.end(), y); return 0;
}
// "All host cores" context figure for the CPU baseline: CSR SpMV of the assembled A_u and Jacobi-CG iterations on it, rows split
// over `threads` std::threads (the reference itself is serial; this is not its algorithm, only what the same data structure gives on
// every core of the box).  out = {seconds per SpMV, seconds per CG iteration}.
// benchmark helper: synthetic symmetric, strictly diagonally dominant (hence SPD) values on the REAL sparsity pattern of A_u, so that the threaded SpMV / CG timings below can
// be taken at sizes whose assembly by the reference's serial cell loop would take minutes (throughput of a CSR SpMV depends on the pattern, not on the values)
int oracle_fill_synthetic_matrix(oracle_ctx *c) {
  Oracle *o = reinterpret_cast<Oracle *>(c); Csr &A = o->A;
  for (int64_t r = 0; r < A.n; ++r) for (int64_t j = A.rp[r]; j < A.rp[r + 1]; ++j) A.val[j] = A.col[j] == r ? (double)(A.rp[r + 1] - A.rp[r]) : -0.5;
  return 0;
}
int oracle_bench_spmv_threads(oracle_ctx *c, int threads, int reps, double *out) {
  Oracle *o = reinterpret_cast<Oracle *>(c); const Csr &A = o->A; const int64_t n = A.n;
  if (n == 0 || threads < 1 || reps < 1) return -1;
  Vec x(n), y(n), g(n), d(n), h(n), dinv(n);
  for (int64_t i = 0; i < n; ++i) { x[i] = std::sin(0.37 * (double)i); dinv[i] = 1.0 / A.val[A.diag[i]]; }
  std::vector<double> part(64 * (size_t)threads);                // one cache line per thread
  std::atomic<int> arrived{0}; std::atomic<int> phase{0};
  auto barrier = [&](int &local_phase) {                          // sense-reversing spin barrier; threads are spawned once
    local_phase ^= 1;
    if (arrived.fetch_add(1) == threads - 1) { arrived.store(0); phase.store(local_phase); }
    else while (phase.load() != local_phase) std::this_thread::yield();
  };
  double t_spmv = 0, t_cg = 0; double gh_shared = 0;
  auto worker = [&](int t) {
    const int64_t a = n * t / threads, b = n * (t + 1) / threads; int lp = 0;
    auto spmv = [&](const Vec &in, Vec &outv) { for (int64_t r = a; r < b; ++r) { double s = 0; for (int64_t j = A.rp[r]; j < A.rp[r + 1]; ++j) s += A.val[j] * in[A.col[j]]; outv[r] = s; } };
    auto total = [&]() { double s = 0; for (int q = 0; q < threads; ++q) s += part[64 * (size_t)q]; return s; };
    spmv(x, y); barrier(lp);                                      // warm-up (first touch of y by its owner)
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; ++k) { spmv(x, y); barrier(lp); }
    if (t == 0) t_spmv = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    // Jacobi-CG on A v = y from v = 0: same recurrences as the device kernels
    { double s = 0; for (int64_t i = a; i < b; ++i) { g[i] = -y[i]; d[i] = dinv[i] * y[i]; x[i] = 0; s += g[i] * g[i] * dinv[i]; } part[64 * (size_t)t] = s; }
    barrier(lp);
    double gh = total(); barrier(lp);
    t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; ++k) {
      spmv(d, h);
      { double s = 0; for (int64_t i = a; i < b; ++i) s += d[i] * h[i]; part[64 * (size_t)t] = s; }
      barrier(lp);
      const double alpha = gh / total(); barrier(lp);
      { double s = 0; for (int64_t i = a; i < b; ++i) { g[i] += alpha * h[i]; x[i] += alpha * d[i]; s += g[i] * g[i] * dinv[i]; } part[64 * (size_t)t] = s; }
      barrier(lp);
      const double gz = total(), beta = gz / gh; gh = gz; barrier(lp);
      for (int64_t i = a; i < b; ++i) d[i] = beta * d[i] - dinv[i] * g[i];
      barrier(lp);
    }
    if (t == 0) { t_cg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps; gh_shared = gh; }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t) th.emplace_back(worker, t);
  for (auto &q : th) q.join();
  out[0] = t_spmv; out[1] = t_cg; (void)gh_shared;
  return 0;
}
// the oracle's own FE tables, for cross-checking the host provider (same layout as poro_fe_tables)
int oracle_fe_table(int dim, int k, int n1d_quad, int what /*0 val,1 grad,2 weights*/, double *out) {
  Quad Q = make_quad(dim, n1d_quad); const int ns = ipow(k + 1, dim);
  Vec v(ns), g(ns * dim);
  for (int q = 0; q < Q.n; ++q) {
    shapes(dim, k, &Q.xi[q * dim], v.data(), g.data());
    if (what == 0) std::copy(v.begin(), v.end(), out + q * ns);
    else if (what == 1) std::copy(g.begin(), g.end(), out + q * ns * dim);
    else out[q] = Q.w[q];
  }
  return Q.n;
}
// InputDataPoroel::compute_derived_parameters InputDataPoroel.h:213-222 (+ mD -> m^2 :162,168)
void oracle_derived_parameters(double E, double nu, double alpha, double poro, double f_comp, double perm_mD, double visc, double *out /*lambda,G,K,Ks,N,M,k_over_mu*/) {
  const double lambda = E * nu / ((1. + nu) * (1. - 2. * nu)), G = 0.5 * E / (1 + nu), K = lambda + 2. / 3. * G;
  const double Ks = K / (1. - alpha), N = Ks / (alpha - poro), M = (N / f_comp) / (N * poro + 1. / f_comp);
  out[0] = lambda; out[1] = G; out[2] = K; out[3] = Ks; out[4] = N; out[5] = M; out[6] = perm_mD * 9.869233e-16 / visc;
}

// PoroElasticProblem<dim>::run() PoroelasticityFSS.h:294-415 without mesh creation / AMR / output.
// trace rows (per FSS iteration): [step, fss_iteration, pressure_iterations, pressure_error_inner, |p|_inf, error_after_disp, disp_cg_its, pres_cg_its_total]
// solver controls are the reference's unless overridden (abs_u, rel_u, max_it, prec, omega_u).
int oracle_run(oracle_ctx *c, double p_init, double dt, int n_steps, double fss_tol, double pressure_tol, int max_fss, int max_pres,
               double abs_u, double rel_u, int max_it, int prec, double *trace, int max_rows, double *seconds_per_phase /*[4]: assemble_u, solve_u, projection, pressure*/,
               int coupled_fss /* bit 0: 0 = the reference (get_volumetric_strain() commented out at :399), 1 = that call restored: a real fixed-stress iteration;
                                   bit 1: strain increment taken against the PREVIOUS step (eps_v^n) instead of the initial state (:317, :361-363);
                                   bit 2: displacement solve stops on the reduction of its initial residual (stop rule 1 of cg) */) {
  Oracle *o = reinterpret_cast<Oracle *>(c);
  o->stop_rule_u = (coupled_fss & 4) ? 1 : 0;
  const int dim = o->dim; int rows = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tph[4] = {0, 0, 0, 0}; const double t_run0 = now();
  std::vector<int32_t> vol(dim); for (int a = 0; a < dim; ++a) vol[a] = a * dim + a;   // strain_tensor_volumetric_components PoroelasticityFSS.h:99-114
  const double om_u = prec == ORACLE_PREC_SSOR ? 1.2 : 1.0;
  auto normal_strains = [&] {                                      // get_normal_strain_components :153-164
    double t0 = now();
    o->assemble_projection_rhs(vol.data(), dim);
    for (int a = 0; a < dim; ++a) o->proj_solve(o->tensor_to_entry[vol[a]], 0.0, 1e-8, max_it, prec, 1.0);
    tph[2] += now() - t0;
  };
  // setup_dofs() reinit()s every vector to zero (:150-151, PoroElasticPressureSolver.h:103-108, StrainProjector.h:93-96, PoroelasticityFSS.h:145-146)
  for (Vec *v : {&o->u, &o->rhs_u, &o->dp, &o->p_old, &o->residual, &o->eps_v, &o->eps_v0}) std::fill(v->begin(), v->end(), 0.0);
  for (Vec &v : o->strains) std::fill(v.begin(), v.end(), 0.0);
  for (Vec &v : o->proj_rhs) std::fill(v.begin(), v.end(), 0.0);
  std::fill(o->p.begin(), o->p.end(), p_init);                     // :311
  if (o->any_pdir) for (int64_t i = 0; i < o->d.n_dofs_p; ++i) if (o->is_pdir[i]) o->p[i] = o->pdir_val[i];   // extension: prescribed pressures
  double t0 = now(); std::fill(o->A.val.begin(), o->A.val.end(), 0.0); o->rebuild_system_matrix = true; o->assemble_system(); tph[0] += now() - t0;   // :312
  t0 = now(); SolveInfo su = o->disp_solve(abs_u, rel_u, max_it, prec, om_u); tph[1] += now() - t0;   // :313
  o->assemble_projection_matrix();                                 // :314
  normal_strains();                                                // :315
  o->get_volumetric_strain(); o->eps_v0 = o->eps_v;                // :316-317
  if (rows < max_rows) { double *r = trace + 8 * rows++; r[0] = 0; r[1] = 0; r[2] = 0; r[3] = 0; r[4] = 0; r[5] = 0; r[6] = su.iterations; r[7] = 0; }
  std::copy(o->work, o->work + 6, o->work_init); o->seconds_init = now() - t_run0; const double t_steps0 = now();
  for (int step = 1; step <= n_steps; ++step) {                    // :327 (AMR branch :333-340 out of scope)
    o->p_old = o->p;                                               // :342
    if ((coupled_fss & 2) && step > 1) o->eps_v0 = o->eps_v;       // corrected storage term: alpha (eps_v^{n+1} - eps_v^n) / dt
    double pressure_error = pressure_tol * 2; int fss = 0;         // :345-346
    while (fss < max_fss && pressure_error > fss_tol) {            // :347-348
      ++fss; int pit = 0; int pcg = 0; double inner_err = 0;
      std::fill(o->dp.begin(), o->dp.end(), 0.0);                  // :356
      t0 = now();
      while (pit < max_pres) {                                     // :358
        ++pit;
        o->update_volumetric_strain();                             // :360
        pressure_error = o->assemble_residual(dt);                 // :361-364
        inner_err = pressure_error;
        if (pressure_error < pressure_tol) break;                  // :366-371
        o->assemble_jacobian(dt);                                  // :377
        SolveInfo sp = o->pres_solve(0.0, 1e-8, max_it, prec, 1.0); pcg += sp.iterations;   // :378
        for (int64_t i = 0; i < o->d.n_dofs_p; ++i) o->p[i] += o->dp[i];   // :379
      }
      tph[3] += now() - t0;
      double pinf = 0; for (double v : o->p) pinf = std::max(pinf, std::fabs(v));   // :387-389
      if (o->comm.multi()) { /* max over ranks is not needed for control flow */ }
      t0 = now(); o->assemble_system(); tph[0] += now() - t0;      // :395
      t0 = now(); su = o->disp_solve(abs_u, rel_u, max_it, prec, om_u); tph[1] += now() - t0;   // :396
      normal_strains();                                            // :398   (get_volumetric_strain() is commented out, :399)
      if (coupled_fss & 1) o->get_volumetric_strain();
      t0 = now(); pressure_error = o->assemble_residual(dt); tph[3] += now() - t0;   // :402-405
      if (rows < max_rows) { double *r = trace + 8 * rows++; r[0] = step; r[1] = fss; r[2] = pit - 1; r[3] = inner_err; r[4] = pinf; r[5] = pressure_error; r[6] = su.iterations; r[7] = pcg; }
    }
  }
  o->seconds_steps = now() - t_steps0;
  if (seconds_per_phase) std::copy(tph, tph + 4, seconds_per_phase);
  return rows;
}
// split of the last oracle_run: work counters at the end of the initialisation, wall seconds of the initialisation and of the time steps
void oracle_last_run_split(oracle_ctx *c, int64_t *work_init, double *seconds /*[2]*/) { Oracle *o = reinterpret_cast<Oracle *>(c); std::copy(o->work_init, o->work_init + 6, work_init); seconds[0] = o->seconds_init; seconds[1] = o->seconds_steps; }

}  // extern "C"
