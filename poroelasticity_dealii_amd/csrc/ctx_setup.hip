// Context set-up: setup_dofs of the three solvers (sparsity patterns, colouring, constraint lists, device vectors, mass / Laplace matrices, the structured-kernel self checks).
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
template <class F> void parallel_for(int64_t n, F &&f) {
  unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 32) nt = 32;
  if (n < 20000 || nt == 1) { f(0, n); return; }
  std::vector<std::thread> th; const int64_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) { const int64_t b = t * chunk, e = std::min<int64_t>(n, b + chunk); if (b < e) th.emplace_back([=, &f] { f(b, e); }); }
  for (auto &t : th) t.join();
}

// DoFTools::make_sparsity_pattern(keep_constrained_dofs = true): every dof couples with all dofs of its cells
void build_pattern(int64_t n, int64_t n_cells, int dpc, const int32_t *cell_dofs, std::vector<int64_t> &rp, std::vector<int32_t> &col, std::vector<int64_t> &diag) {
  std::vector<int64_t> cnt(n + 1, 0);
  for (int64_t i = 0; i < n_cells * dpc; ++i) cnt[cell_dofs[i] + 1]++;
  for (int64_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
  std::vector<int32_t> adj(cnt[n]);
  { std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int64_t c = 0; c < n_cells; ++c) for (int i = 0; i < dpc; ++i) adj[pos[cell_dofs[c * dpc + i]]++] = (int32_t)c; }
  rp.assign(n + 1, 0);
  auto row_cols = [&](int64_t r, std::vector<int32_t> &row) {
    row.clear();
    for (int64_t a = cnt[r]; a < cnt[r + 1]; ++a) { const int32_t *cd = cell_dofs + (int64_t)adj[a] * dpc; row.insert(row.end(), cd, cd + dpc); }
    std::sort(row.begin(), row.end()); row.erase(std::unique(row.begin(), row.end()), row.end());
  };
  parallel_for(n, [&](int64_t b, int64_t e) { std::vector<int32_t> row; for (int64_t r = b; r < e; ++r) { row_cols(r, row); rp[r + 1] = (int64_t)row.size(); } });
  for (int64_t r = 0; r < n; ++r) rp[r + 1] += rp[r];
  col.resize(rp[n]); diag.resize(n);
  parallel_for(n, [&](int64_t b, int64_t e) {
    std::vector<int32_t> row;
    for (int64_t r = b; r < e; ++r) {
      row_cols(r, row); std::copy(row.begin(), row.end(), col.begin() + rp[r]);
      diag[r] = rp[r] + (std::lower_bound(row.begin(), row.end(), (int32_t)r) - row.begin());
    }
  });
}

void upload_csr(CsrDev &A, int64_t n, const std::vector<int64_t> &rp, const std::vector<int32_t> &col, const std::vector<int64_t> &diag) {
  A.n = n; A.nnz = (int64_t)col.size(); A.rp.upload(rp); A.col.upload(col); A.diag_pos.upload(diag);
  const double avg = n ? (double)A.nnz / n : 1; int L = 2;
  while (L < 64 && L * 4 < avg) L *= 2;
  A.lanes_per_row = L;
}

// greedy colouring: cells of one colour share no vertex, hence no dof
void colour_cells(int64_t n_cells, int64_t n_vertices, int nv, const int32_t *cv, std::vector<int32_t> &cells_sorted, std::vector<int64_t> &off) {
  std::vector<int64_t> vp(n_vertices + 1, 0);
  for (int64_t i = 0; i < n_cells * nv; ++i) vp[cv[i] + 1]++;
  for (int64_t i = 0; i < n_vertices; ++i) vp[i + 1] += vp[i];
  std::vector<int32_t> vc(vp[n_vertices]);
  { std::vector<int64_t> pos(vp.begin(), vp.end() - 1); for (int64_t c = 0; c < n_cells; ++c) for (int v = 0; v < nv; ++v) vc[pos[cv[c * nv + v]]++] = (int32_t)c; }
  std::vector<int> colour(n_cells, -1); int ncol = 0;
  for (int64_t c = 0; c < n_cells; ++c) {
    uint64_t used = 0;
    for (int v = 0; v < nv; ++v) { const int32_t vx = cv[c * nv + v]; for (int64_t a = vp[vx]; a < vp[vx + 1]; ++a) { const int k = colour[vc[a]]; if (k >= 0) used |= (1ull << k); } }
    int k = 0; while (used & (1ull << k)) ++k;
    if (k >= 63) throw Error("colouring needs more than 63 colours");
    colour[c] = k; ncol = std::max(ncol, k + 1);
  }
  off.assign(ncol + 1, 0);
  for (int64_t c = 0; c < n_cells; ++c) off[colour[c] + 1]++;
  for (int k = 0; k < ncol; ++k) off[k + 1] += off[k];
  cells_sorted.resize(n_cells);
  { std::vector<int64_t> pos(off.begin(), off.end() - 1); for (int64_t c = 0; c < n_cells; ++c) cells_sorted[pos[colour[c]]++] = (int32_t)c; }
}

// poro_constraints -> device lists (+ transposed lists for the gather form of C^T y); `fixed` = byte mask of the Dirichlet dofs of the same space (or null)
void upload_constraints(ConsDev &C, const poro_constraints &h, int64_t n_dofs, const std::vector<uint8_t> *fixed, const char *what) {
  C.n = h.n;
  std::vector<uint8_t> inert(n_dofs, 0);
  if (fixed) inert = *fixed;
  if (h.n < 0) throw Error(std::string(what) + ": negative constraint count");
  if (h.n == 0) { if (fixed) C.inert.upload(inert); return; }
  if (!h.dof || !h.ptr || !h.inhomogeneity) throw Error(std::string(what) + ": null constraint arrays");
  std::vector<uint8_t> hanging(n_dofs, 0);
  for (int64_t i = 0; i < h.n; ++i) {
    const int32_t dof = h.dof[i];
    if (dof < 0 || dof >= n_dofs) throw Error(std::string(what) + ": constrained dof out of range");
    if (hanging[dof]) throw Error(std::string(what) + ": dof constrained twice");
    if (fixed && (*fixed)[dof]) throw Error(std::string(what) + ": dof is both in the Dirichlet list and in the constraint list");
    hanging[dof] = 1;
  }
  const int64_t nm = h.ptr[h.n];
  if (h.ptr[0] != 0 || nm < 0 || (nm && (!h.master || !h.weight))) throw Error(std::string(what) + ": bad constraint offsets");
  std::map<int32_t, std::vector<std::pair<int32_t, double>>> tr;
  for (int64_t i = 0; i < h.n; ++i) {
    if (h.ptr[i + 1] < h.ptr[i]) throw Error(std::string(what) + ": constraint offsets not ascending");
    for (int64_t k = h.ptr[i]; k < h.ptr[i + 1]; ++k) {
      const int32_t m = h.master[k];
      if (m < 0 || m >= n_dofs) throw Error(std::string(what) + ": master dof out of range");
      if (hanging[m] || (fixed && (*fixed)[m])) throw Error(std::string(what) + ": constraints are not closed (a master is itself constrained)");
      tr[m].emplace_back(h.dof[i], h.weight[k]);
    }
    if (h.inhomogeneity[i] != 0.0) C.any_inhom = true;
  }
  C.dof.upload(h.dof, h.n); C.ptr.upload(h.ptr, h.n + 1); C.inhom.upload(h.inhomogeneity, h.n);
  if (nm) { C.master.upload(h.master, nm); C.weight.upload(h.weight, nm); }
  std::vector<int32_t> tm, td; std::vector<int64_t> tp{0}; std::vector<double> tw;
  for (auto &kv : tr) { tm.push_back(kv.first); for (auto &e : kv.second) { td.push_back(e.first); tw.push_back(e.second); } tp.push_back((int64_t)td.size()); }
  C.n_masters = (int64_t)tm.size();
  if (C.n_masters) { C.t_master.upload(tm); C.t_dof.upload(td); C.t_ptr.upload(tp); C.t_weight.upload(tw); }
  for (int64_t i = 0; i < n_dofs; ++i) inert[i] = inert[i] | hanging[i];
  C.inert.upload(inert);
}

void setup(poro_ctx *c, const poro_desc *d) {
  if (d->abi_version != PORO_ABI_VERSION) throw Error("poro_desc.abi_version mismatch");
  if (d->dim != 2 && d->dim != 3) throw Error("dim must be 2 or 3");
  if (d->degree_u != 1 && d->degree_u != 2) throw Error("degree_u must be 1 or 2");
  if (d->degree_p != 1) throw Error("degree_p must be 1 (PoroElasticPressureSolver.h:20)");
  c->dim = d->dim; c->k_u = d->degree_u; c->nv = 1 << d->dim; c->ns_u = ipow(c->k_u + 1, c->dim); c->ns_p = c->nv; c->dpc_u = c->ns_u * c->dim; c->dpc_p = c->ns_p;
  c->n_cells = d->n_cells; c->n_u = d->n_dofs_u; c->n_p = d->n_dofs_p; c->mat = d->mat; c->comm.part = d->part;
  if (c->comm.part.n_ranks < 1) { c->comm.part.n_ranks = 1; c->comm.part.rank = 0; }
  c->comm.force_multi = std::getenv("PORO_FORCE_PARTITIONED_PATH") != nullptr;   // test hook: run the partitioned code path on one rank
  const poro_fe_tables &f = d->fe;
  if (f.nq_u != ipow(c->k_u + 1, c->dim) || f.nq_p != c->nv || f.ns_u != c->ns_u || f.ns_p != c->ns_p || f.nq_f != ipow(c->k_u + 1, c->dim - 1)) throw Error("poro_fe_tables sizes do not match dim / degree");
  if (c->n_cells <= 0 || c->n_u <= 0 || c->n_p <= 0) throw Error("empty mesh");
  for (int64_t i = 0; i < c->n_cells * c->dpc_u; ++i) if (d->cell_dofs_u[i] < 0 || d->cell_dofs_u[i] >= c->n_u) throw Error("cell_dofs_u out of range");
  for (int64_t i = 0; i < c->n_cells * c->dpc_p; ++i) if (d->cell_dofs_p[i] < 0 || d->cell_dofs_p[i] >= c->n_p) throw Error("cell_dofs_p out of range");
  for (int64_t i = 0; i < c->n_cells * c->nv; ++i) if (d->cell_vertices[i] < 0 || d->cell_vertices[i] >= d->n_vertices) throw Error("cell_vertices out of range");
  { bool inter = true;     // node-interleaved displacement numbering?
    for (int64_t i = 0; i < c->n_cells * c->ns_u && inter; ++i) { const int32_t b = d->cell_dofs_u[i * c->dim]; if (b % c->dim) inter = false; for (int k = 1; k < c->dim && inter; ++k) if (d->cell_dofs_u[i * c->dim + k] != b + k) inter = false; }
    c->interleaved_u = inter ? 1 : 0; }
  if (d->part.n_neighbours > 0 && d->box.enabled) throw Error("a general partition (poro_partition.n_neighbours > 0) carries no box tag: pieces are not boxes");
  if (!d->box.enabled && d->part.n_ranks > 1 && d->part.n_neighbours <= 0) throw Error("a partitioned general mesh needs the interface lists of poro_partition (n_neighbours > 0)");
  setup_general_partition(c, d);
  if (d->tensor.enabled && !d->box.enabled && (d->part.n_ranks > 1 || d->cons_u.n || d->cons_p.n)) throw Error("poro_desc.tensor: one rank, no constraint lists");
  if (d->box.enabled || d->tensor.enabled) {
    // the lexicographic numbering the structured kernels / the fast diagonalisation assume must be the caller's numbering (spot-checked on three cells)
    const int32_t *bn = d->box.enabled ? d->box.n : d->tensor.n;
    int64_t nn[3] = {1, 1, 1}, np[3] = {1, 1, 1}, ncells = 1;
    for (int k = 0; k < c->dim; ++k) { if (bn[k] < 1) throw Error("box.n must be positive"); nn[k] = (int64_t)c->k_u * bn[k] + 1; np[k] = (int64_t)bn[k] + 1; ncells *= bn[k]; }
    if (nn[0] * nn[1] * nn[2] * c->dim != c->n_u || np[0] * np[1] * np[2] != c->n_p || ncells != c->n_cells) throw Error("box does not match n_dofs_u / n_dofs_p / n_cells");
    const int n1 = c->k_u + 1;
    const int64_t ncx = bn[0], ncy = bn[1];
    for (int64_t cell : {(int64_t)0, c->n_cells / 2, c->n_cells - 1}) {
      const int64_t ci = cell % ncx, cj = (cell / ncx) % ncy, ck = cell / (ncx * ncy);
      for (int sidx = 0; sidx < c->ns_u; ++sidx) {
        const int a = sidx % n1, b = (sidx / n1) % n1, cc = sidx / (n1 * n1);
        const int64_t node = ((ck * c->k_u + cc) * nn[1] + (cj * c->k_u + b)) * nn[0] + (ci * c->k_u + a);
        for (int k = 0; k < c->dim; ++k) if (d->cell_dofs_u[(cell * c->ns_u + sidx) * c->dim + k] != node * c->dim + k) throw Error("cell_dofs_u is not the lexicographic box numbering");
      }
      for (int v = 0; v < c->nv; ++v) {
        const int a = v & 1, b = (v >> 1) & 1, cc = v >> 2;
        if (d->cell_dofs_p[cell * c->nv + v] != ((ck + cc) * np[1] + (cj + b)) * np[0] + (ci + a)) throw Error("cell_dofs_p is not the lexicographic box numbering");
      }
    }
  }
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount; }
  if (const char *v = std::getenv("PORO_MF_VARIANT")) c->mf_variant = (std::string(v) == "gather" || std::string(v) == "0") ? 0 : 1;
  c->box.enabled = d->box.enabled;
  for (int k = 0; k < 3; ++k) { c->box.n[k] = d->box.enabled && k < c->dim ? d->box.n[k] : 1; c->box.h[k] = d->box.h[k]; c->box.nn[k] = c->k_u * c->box.n[k] + 1; }
  if (d->box.enabled || d->tensor.enabled) {
    c->lines.on = true; c->lines.uniform = d->box.enabled != 0;
    for (int k = 0; k < 3; ++k) {
      const int n = k < c->dim ? (d->box.enabled ? d->box.n[k] : d->tensor.n[k]) : 1;
      c->lines.n[k] = n; c->lines.nn[k] = c->k_u * n + 1; c->lines.hcell[k].assign((size_t)n, d->box.enabled ? d->box.h[k] : 1.0);
      if (!d->box.enabled && k < c->dim) {
        if (!d->tensor.grid[k]) throw Error("poro_desc.tensor.grid missing");
        for (int i = 0; i < n; ++i) { const double h = d->tensor.grid[k][i + 1] - d->tensor.grid[k][i]; if (!(h > 0)) throw Error("poro_desc.tensor.grid must be strictly ascending"); c->lines.hcell[k][i] = h; }
      }
    }
  }

  // tables -> one device buffer
  std::vector<double> T; std::vector<size_t> off;
  auto push = [&](const double *p, size_t n) { off.push_back(T.size()); T.insert(T.end(), p, p + n); };
  const int dim = c->dim, nf = 2 * dim;
  push(f.w_qu, f.nq_u); push(f.w_qp, f.nq_p); push(f.w_qf, f.nq_f); push(f.u_qu, (size_t)f.nq_u * f.ns_u); push(f.du_qu, (size_t)f.nq_u * f.ns_u * dim);
  push(f.du_qp, (size_t)f.nq_p * f.ns_u * dim); push(f.q1_qu, (size_t)f.nq_u * f.ns_p); push(f.dq1_qu, (size_t)f.nq_u * f.ns_p * dim);
  push(f.q1_qp, (size_t)f.nq_p * f.ns_p); push(f.dq1_qp, (size_t)f.nq_p * f.ns_p * dim); push(f.u_qf, (size_t)nf * f.nq_f * f.ns_u); push(f.dq1_qf, (size_t)nf * f.nq_f * f.ns_p * dim);
  c->tables.upload(T);
  const double *tb = c->tables.p;
  c->fe = FeTablesDev{f.nq_u, f.nq_p, f.nq_f, f.ns_u, f.ns_p, tb + off[0], tb + off[1], tb + off[2], tb + off[3], tb + off[4], tb + off[5], tb + off[6], tb + off[7], tb + off[8], tb + off[9], tb + off[10], tb + off[11]};

  c->cell_dofs_u.upload(d->cell_dofs_u, c->n_cells * c->dpc_u); c->cell_dofs_p.upload(d->cell_dofs_p, c->n_cells * c->dpc_p);
  { std::vector<double> X((size_t)c->n_cells * c->nv * dim);
    for (int64_t i = 0; i < c->n_cells * c->nv; ++i) for (int k = 0; k < dim; ++k) X[i * dim + k] = d->vertex_coords[(int64_t)d->cell_vertices[i] * dim + k];
    c->cell_X.upload(X);
    // affine cells (graded / locally refined boxes, lattice-like Gmsh grids): MappingQ1's Jacobian is one matrix per cell - the sum-factorised operator kernel reads
    // its inverse instead of forming it from the eight vertices at every quadrature point
    if (dim == 3 && c->n_cells > 0) {
      std::vector<double> geo((size_t)c->n_cells * 10); bool affine = true;
      for (int64_t e = 0; e < c->n_cells && affine; ++e) {
        const double *x = X.data() + e * 24; double J[3][3], scale = 0;
        for (int r = 0; r < 3; ++r) { J[r][0] = x[3 + r] - x[r]; J[r][1] = x[6 + r] - x[r]; J[r][2] = x[12 + r] - x[r]; scale = std::max({scale, std::fabs(J[r][0]), std::fabs(J[r][1]), std::fabs(J[r][2])}); }
        for (int v = 0; v < 8 && affine; ++v) for (int r = 0; r < 3; ++r) {
          const double want = x[r] + (v & 1) * J[r][0] + ((v >> 1) & 1) * J[r][1] + (v >> 2) * J[r][2];
          if (std::fabs(x[v * 3 + r] - want) > 1e-13 * scale) affine = false;
        }
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2], c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det; double *g = geo.data() + e * 10;
        g[0] = c00 * id; g[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; g[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
        g[3] = c01 * id; g[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; g[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
        g[6] = c02 * id; g[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; g[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id; g[9] = det;
        if (!(det > 0)) affine = false;
      }
      static const bool no_affine = std::getenv("PORO_MFG_NO_AFFINE") != nullptr;
      if (affine && !no_affine) c->cell_geo.upload(geo);
    } }
  { std::vector<int32_t> cells; colour_cells(c->n_cells, d->n_vertices, c->nv, d->cell_vertices, cells, c->color_off); c->color_cells.upload(cells); }
  { std::vector<uint8_t> m(c->n_u, 0); std::vector<double> v(c->n_u, 0.0);
    for (int64_t i = 0; i < d->n_dirichlet; ++i) { const int32_t dof = d->dirichlet_dof[i]; if (dof < 0 || dof >= c->n_u) throw Error("dirichlet_dof out of range"); m[dof] = 1; v[dof] = d->dirichlet_value[i]; }
    c->dir_mask.upload(m); c->dir_val.upload(v);
    // hanging-node constraints (locally refined meshes): operator-level condensation, see include/poroel_hip.h poro_constraints
    if (d->cons_u.n || d->cons_p.n) {
      if (d->box.enabled) throw Error("constraint lists belong to general (non-box) meshes: assembled-CSR operator or the general matrix-free one");
      if (c->comm.part.n_ranks > 1 && d->part.n_neighbours <= 0) throw Error("constraint lists on partitioned meshes need the general form of poro_partition (interface lists), with every master of a local constrained dof local as well");
    }
    upload_constraints(c->cons_u, d->cons_u, c->n_u, &m, "cons_u");
    { // extension: prescribed pressures (drained boundaries); the rows leave the pressure Newton system exactly like hanging rows do
      std::vector<uint8_t> pm(c->n_p, 0); std::vector<double> pv(c->n_p, 0.0);
      if (d->n_dirichlet_p < 0 || (d->n_dirichlet_p && (!d->dirichlet_dof_p || !d->dirichlet_value_p))) throw Error("bad prescribed-pressure list");
      for (int64_t i = 0; i < d->n_dirichlet_p; ++i) { const int32_t dof = d->dirichlet_dof_p[i]; if (dof < 0 || dof >= c->n_p) throw Error("dirichlet_dof_p out of range"); pm[dof] = 1; pv[dof] = d->dirichlet_value_p[i]; }
      c->n_pdir = d->n_dirichlet_p;
      if (c->n_pdir) { if (c->comm.part.n_ranks > 1) throw Error("prescribed pressures are implemented for one rank"); c->pdir_mask.upload(pm); c->pdir_val.upload(pv); }
      upload_constraints(c->cons_p, d->cons_p, c->n_p, &pm, "cons_p");
      if (c->cons_p.n && c->n_pdir) throw Error("prescribed pressures together with hanging pressure nodes are not supported"); }
    { std::vector<uint8_t> nm((size_t)(c->n_u / c->dim), 0); for (int64_t i = 0; i < d->n_dirichlet; ++i) nm[d->dirichlet_dof[i] / c->dim] |= (uint8_t)(1u << (d->dirichlet_dof[i] % c->dim)); c->node_mask.upload(nm); c->h_node_mask = std::move(nm); }
    if (d->n_dirichlet) c->dir_dofs.upload(d->dirichlet_dof, d->n_dirichlet);
    if (d->box.enabled) {   // are all constrained dofs on the box boundary?  (lets the matrix-free kernels skip mask loads in the interior)
      int64_t nn[3] = {1, 1, 1}; for (int k = 0; k < c->dim; ++k) nn[k] = (int64_t)c->k_u * d->box.n[k] + 1;
      for (int64_t i = 0; i < d->n_dirichlet; ++i) {
        const int64_t node = d->dirichlet_dof[i] / c->dim; const int64_t ci = node % nn[0], cj = (node / nn[0]) % nn[1], ck = node / (nn[0] * nn[1]);
        if (!(ci == 0 || ci == nn[0] - 1 || cj == 0 || cj == nn[1] - 1 || (c->dim == 3 && (ck == 0 || ck == nn[2] - 1)))) { c->mask_anywhere = 1; break; }
      }
    }
    c->h_dir_dof.assign(d->dirichlet_dof, d->dirichlet_dof + d->n_dirichlet); c->h_dir_val.assign(d->dirichlet_value, d->dirichlet_value + d->n_dirichlet); }
  c->n_bfaces = d->n_bfaces; c->n_neumann = d->n_neumann;
  if (d->n_bfaces) { c->bface_cell.upload(d->bface_cell, d->n_bfaces); c->bface_local.upload(d->bface_local, d->n_bfaces); c->bface_id.upload(d->bface_id, d->n_bfaces); }
  if (d->n_neumann) { c->neu_label.upload(d->neumann_label, d->n_neumann); c->neu_comp.upload(d->neumann_component, d->n_neumann); c->neu_val.upload(d->neumann_value, d->n_neumann); }

  // sparsity patterns (PoroElasticPressureSolver.h:80-94, PoroElasticDisplacementSolver.h:140-149)
  { std::vector<int64_t> rp, diag; std::vector<int32_t> col; build_pattern(c->n_p, c->n_cells, c->dpc_p, d->cell_dofs_p, rp, col, diag); upload_csr(c->Ap, c->n_p, rp, col, diag); }
  c->Mp.alloc(c->Ap.nnz); c->Kp.alloc(c->Ap.nnz); c->Jp.alloc(c->Ap.nnz);
  if (c->operator_mode == PORO_OP_CSR) {
    std::vector<int64_t> rp, diag; std::vector<int32_t> col; build_pattern(c->n_u, c->n_cells, c->dpc_u, d->cell_dofs_u, rp, col, diag); upload_csr(c->Au, c->n_u, rp, col, diag);
    c->Au_val.alloc(c->Au.nnz);
  } else if (d->box.enabled) c->Ke.alloc((size_t)c->dpc_u * c->dpc_u);

  hipStream_t s = c->stream;
  const int n_sym = dim * (dim + 1) / 2;
  for (int id : {PORO_VEC_U, PORO_VEC_RHS_U, PORO_VEC_DIAG_U}) { c->vec[id].alloc(c->n_u); c->vec[id].zero(s); }
  for (int id : {PORO_VEC_P, PORO_VEC_P_OLD, PORO_VEC_DP, PORO_VEC_RESIDUAL_P, PORO_VEC_EPSV, PORO_VEC_EPSV0, PORO_VEC_SOURCE_P}) { c->vec[id].alloc(c->n_p); c->vec[id].zero(s); }
  for (int e = 0; e < n_sym; ++e) { c->vec[PORO_VEC_STRAIN0 + e].alloc(c->n_p); c->vec[PORO_VEC_STRAIN0 + e].zero(s); c->vec[PORO_VEC_PROJ_RHS0 + e].alloc(c->n_p); c->vec[PORO_VEC_PROJ_RHS0 + e].zero(s);
                                     c->vec[PORO_VEC_STRESS0 + e].alloc(c->n_p); c->vec[PORO_VEC_STRESS0 + e].zero(s); }
  for (DevBuf<double> *b : {&c->lift_u, &c->neumann_u, &c->diag_u_local, &c->wg_u, &c->wd_u, &c->wh_u}) { b->alloc(c->n_u); b->zero(s); }
  for (DevBuf<double> *b : {&c->diag_J, &c->diag_M, &c->src_local, &c->wg_p, &c->wd_p, &c->wh_p, &c->tmp_p}) { b->alloc(c->n_p); b->zero(s); }
  c->partials.alloc((size_t)6 * kMaxPartials); c->partials.zero(s); c->scal.alloc(1); c->scal.zero(s); c->red.alloc(kScalarSlots); c->red.zero(s);

  // MatrixCreator::create_mass_matrix / create_laplace_matrix (:96-101) + the time-independent well integral (:142-147)
  c->Mp.zero(s); c->Kp.zero(s); c->Jp.zero(s);
  const AsmArgs a = asm_args(c);
  for (size_t k = 0; k + 1 < c->color_off.size(); ++k)
    asm_p_matrices(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], c->Ap.rp.p, c->Ap.col.p, c->Mp.p, c->Kp.p, c->src_local.p);
  PORO_HIP(hipStreamSynchronize(s));

  // uniform box: the coupling / projection right-hand sides have a structured form (kernels_box.hip); check it once against the
  // per-cell kernels on synthetic vectors before it replaces them
  const char *ba = std::getenv("PORO_BOX_ASM");
  if (c->box.enabled && !(ba && std::string(ba) == "0")) {
    c->box_cpl = box_coupling(c->dim, c->k_u, c->box);
    if (!std::getenv("PORO_DIAG_SKIP_SELFCHECK")) {
      std::vector<double> hp(c->n_p), hu(c->n_u);
      for (int64_t i = 0; i < c->n_p; ++i) hp[i] = 1e7 * (1 + 0.3 * std::sin(0.37 * (double)i));
      for (int64_t i = 0; i < c->n_u; ++i) hu[i] = 1e-5 * std::sin(0.11 * (double)i);
      DevBuf<double> tp, tu, zero_u, r1, r2; tp.upload(hp); tu.upload(hu); zero_u.alloc(c->n_u); zero_u.zero(s); r1.alloc(c->n_u); r2.alloc(c->n_u);
      auto compare = [&](const char *what, double *x1, const double *x2, int64_t n) {
        la_axpy(s, x1, -1.0, x2, n);
        la_norm_partials(s, x1, n, c->partials.p, c->partials.p + kMaxPartials); la_norm_partials(s, x2, n, c->partials.p + 2 * kMaxPartials, c->partials.p + 3 * kMaxPartials);
        la_reduce_finish(s, c->partials.p, 4, c->red.p, 2 | 8);
        double h[4]; PORO_HIP(hipMemcpyAsync(h, c->red.p, sizeof(h), hipMemcpyDeviceToHost, s)); PORO_HIP(hipStreamSynchronize(s));
        if (!(h[1] <= 1e-11 * h[3])) throw Error(std::string("structured ") + what + " disagrees with the per-cell kernel: max diff " + std::to_string(h[1]) + " vs max " + std::to_string(h[3]));
      };
      r1.zero(s);
      for (size_t k = 0; k + 1 < c->color_off.size(); ++k) asm_u_rhs(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], tp.p, r1.p);
      la_rhs_u_finish(s, r1.p, zero_u.p, zero_u.p, c->dir_mask.p, c->n_u);
      box_rhs_u(s, c->dim, c->box_cpl, c->mat.biot_alpha, tp.p, zero_u.p, zero_u.p, c->dir_mask.p, r2.p);
      compare("coupling right-hand side", r1.p, r2.p, c->n_u);
      const int ncomp = c->dim * c->dim; int32_t comps[9]; double *o1[9], *o2[9];
      DevBuf<double> q1, q2; q1.alloc((size_t)ncomp * c->n_p); q2.alloc((size_t)ncomp * c->n_p); q1.zero(s);
      for (int e = 0; e < ncomp; ++e) { comps[e] = e; o1[e] = q1.p + (size_t)e * c->n_p; o2[e] = q2.p + (size_t)e * c->n_p; }
      for (int e0 = 0; e0 < ncomp; e0 += 6) {
        const int ne = std::min(6, ncomp - e0);
        for (size_t k = 0; k + 1 < c->color_off.size(); ++k) asm_proj_rhs(s, a, c->color_cells.p + c->color_off[k], c->color_off[k + 1] - c->color_off[k], tu.p, ne, comps + e0, o1 + e0);
        box_proj_rhs(s, c->dim, c->box_cpl, tu.p, ne, comps + e0, o2 + e0);
      }
      compare("projection right-hand side", q1.p, q2.p, (int64_t)ncomp * c->n_p);
    }
    c->box_asm = 1;
  }
}

void sync_source_vector(poro_ctx *c) {   // PORO_VEC_SOURCE_P = the assembled (rank-summed) well integral
  la_copy(c->stream, vec(c, PORO_VEC_SOURCE_P), c->src_local.p, c->n_p);
  exchange_add(c, vec(c, PORO_VEC_SOURCE_P), c->n_p, c->comm.part.plane_p);
}

}  // namespace ctx_detail
}  // namespace poro
