// Minimal mesh + DoF provider feeding the hot path without deal.II:
//  - structured box = GridGenerator::hyper_rectangle(colorize) + refine_global
//    (PoroelasticityFSS.h:418-435), optionally one z-slab (y-slab in 2D) of it;
//  - Gmsh 2.2 ASCII reader = GridIn::read_msh (PoroelasticityFSS.h:438-445) for 2D quads;
//  - DoF numbering, boundary faces and the closed Dirichlet constraint list
//    (PoroElasticDisplacementSolver.h:106-137).
// Produces the flat arrays of include/poroel_hip.h.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "fe_tables.hpp"

namespace poro_host {

struct Mesh {
  int dim = 2;
  std::vector<double>  vertices;       // [nv][dim]
  std::vector<int32_t> cells;          // [nc][2^dim] lexicographic vertices
  std::vector<int32_t> bface_cell, bface_local, bface_id;
  poro_structured box{};               // enabled for make_box meshes
  int64_t n_cells() const { return (int64_t)cells.size() >> dim; }
  int64_t n_vertices() const { return (int64_t)vertices.size() / dim; }
};

// Uniform box of n[d] cells, lower corner `origin`, spacing h.  Boundary ids follow
// hyper_rectangle(colorize=true): 2*d (low) / 2*d+1 (high).  low_face/high_face switch the two faces
// normal to the slowest direction off for interior slab interfaces of a partitioned box.
inline Mesh make_box(int dim, const int n[3], const double origin[3], const double h[3],
                     bool low_face = true, bool high_face = true) {
  Mesh m; m.dim = dim;
  int nn[3] = {n[0] + 1, n[1] + 1, dim == 3 ? n[2] + 1 : 1};
  int nc[3] = {n[0], n[1], dim == 3 ? n[2] : 1};
  const int64_t nv = (int64_t)nn[0] * nn[1] * nn[2];
  m.vertices.resize(nv * dim);
  for (int k = 0; k < nn[2]; ++k) for (int j = 0; j < nn[1]; ++j) for (int i = 0; i < nn[0]; ++i) {
    const int64_t v = ((int64_t)k * nn[1] + j) * nn[0] + i;
    const int idx[3] = {i, j, k};
    for (int d = 0; d < dim; ++d) m.vertices[v * dim + d] = origin[d] + h[d] * idx[d];
  }
  const int nvc = 1 << dim;
  m.cells.resize((int64_t)nc[0] * nc[1] * nc[2] * nvc);
  for (int k = 0; k < nc[2]; ++k) for (int j = 0; j < nc[1]; ++j) for (int i = 0; i < nc[0]; ++i) {
    const int64_t c = ((int64_t)k * nc[1] + j) * nc[0] + i;
    for (int v = 0; v < nvc; ++v) {
      const int di = v & 1, dj = (v >> 1) & 1, dk = (v >> 2) & 1;
      m.cells[c * nvc + v] = (int32_t)((((int64_t)(k + dk)) * nn[1] + (j + dj)) * nn[0] + (i + di));
    }
    const int idx[3] = {i, j, k};
    for (int d = 0; d < dim; ++d) {
      const bool slow = (d == dim - 1);
      if (idx[d] == 0 && (!slow || low_face)) { m.bface_cell.push_back((int32_t)c); m.bface_local.push_back(2 * d); m.bface_id.push_back(2 * d); }
      if (idx[d] == nc[d] - 1 && (!slow || high_face)) { m.bface_cell.push_back((int32_t)c); m.bface_local.push_back(2 * d + 1); m.bface_id.push_back(2 * d + 1); }
    }
  }
  m.box.enabled = 1;
  for (int d = 0; d < 3; ++d) { m.box.n[d] = d < dim ? n[d] : 1; m.box.origin[d] = d < dim ? origin[d] : 0; m.box.h[d] = d < dim ? h[d] : 1; }
  return m;
}

// Gmsh 2.2 ASCII: element type 1 = 2-node line (first tag = physical id -> boundary id),
// type 3 = 4-node quad (counter-clockwise -> lexicographic: v2 and v3 swapped).
inline Mesh read_gmsh22(const std::string &path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open " + path);
  Mesh m; m.dim = 2;
  std::string line;
  std::unordered_map<int64_t, int32_t> node_of;
  std::map<std::pair<int32_t, int32_t>, int32_t> line_id;
  while (std::getline(in, line)) {
    if (line.rfind("$Nodes", 0) == 0) {
      int64_t n; in >> n;
      for (int64_t i = 0; i < n; ++i) {
        int64_t id; double x, y, z; in >> id >> x >> y >> z;
        node_of[id] = (int32_t)i; m.vertices.push_back(x); m.vertices.push_back(y);
      }
    } else if (line.rfind("$Elements", 0) == 0) {
      int64_t n; in >> n;
      for (int64_t e = 0; e < n; ++e) {
        int64_t id; int type, ntags; in >> id >> type >> ntags;
        std::vector<int> tags(ntags); for (auto &t : tags) in >> t;
        if (type == 1) {
          int64_t a, b; in >> a >> b;
          int32_t va = node_of.at(a), vb = node_of.at(b);
          line_id[{std::min(va, vb), std::max(va, vb)}] = ntags ? tags[0] : 0;
        } else if (type == 3) {
          int64_t q[4]; in >> q[0] >> q[1] >> q[2] >> q[3];
          int32_t v[4]; for (int i = 0; i < 4; ++i) v[i] = node_of.at(q[i]);
          // enforce positive orientation, then lexicographic
          const double *p0 = &m.vertices[2 * v[0]], *p1 = &m.vertices[2 * v[1]], *p3 = &m.vertices[2 * v[3]];
          const double cross = (p1[0] - p0[0]) * (p3[1] - p0[1]) - (p1[1] - p0[1]) * (p3[0] - p0[0]);
          if (cross < 0) std::swap(v[1], v[3]);
          m.cells.insert(m.cells.end(), {v[0], v[1], v[3], v[2]});
        } else if (type == 15) { int64_t a; in >> a; }
        else throw std::runtime_error("read_gmsh22: unsupported element type");
      }
    }
  }
  // boundary faces = cell edges owned by exactly one cell; id from the line elements (default 0)
  static const int fv[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};
  std::map<std::pair<int32_t, int32_t>, int> count;
  const int64_t nc = m.n_cells();
  for (int64_t c = 0; c < nc; ++c) for (int f = 0; f < 4; ++f) {
    int32_t a = m.cells[c * 4 + fv[f][0]], b = m.cells[c * 4 + fv[f][1]];
    count[{std::min(a, b), std::max(a, b)}]++;
  }
  for (int64_t c = 0; c < nc; ++c) for (int f = 0; f < 4; ++f) {
    int32_t a = m.cells[c * 4 + fv[f][0]], b = m.cells[c * 4 + fv[f][1]];
    auto key = std::make_pair(std::min(a, b), std::max(a, b));
    if (count[key] == 1) {
      auto it = line_id.find(key);
      m.bface_cell.push_back((int32_t)c); m.bface_local.push_back(f); m.bface_id.push_back(it == line_id.end() ? 0 : it->second);
    }
  }
  return m;
}

// One uniform refinement of a 2D quadrilateral mesh (Triangulation::refine_global(1) for read_mesh()'s grid): every cell becomes its four children around the cell
// centre, edge midpoints are shared, boundary edges pass their id to both halves
inline Mesh refine_quads(const Mesh &m) {
  if (m.dim != 2) throw std::runtime_error("refine_quads: 2D meshes only");
  Mesh r; r.dim = 2; r.vertices = m.vertices;
  std::map<std::pair<int32_t, int32_t>, int32_t> mid;
  auto midpoint = [&](int32_t a, int32_t b) {
    auto key = std::make_pair(std::min(a, b), std::max(a, b));
    auto it = mid.find(key); if (it != mid.end()) return it->second;
    const int32_t id = (int32_t)(r.vertices.size() / 2);
    for (int d = 0; d < 2; ++d) r.vertices.push_back(0.5 * (m.vertices[2 * a + d] + m.vertices[2 * b + d]));
    mid[key] = id; return id;
  };
  const int64_t nc = m.n_cells();
  for (int64_t c = 0; c < nc; ++c) {
    const int32_t *v = &m.cells[4 * c];
    const int32_t e02 = midpoint(v[0], v[2]), e13 = midpoint(v[1], v[3]), e01 = midpoint(v[0], v[1]), e23 = midpoint(v[2], v[3]);
    const int32_t cc = (int32_t)(r.vertices.size() / 2);
    for (int d = 0; d < 2; ++d) r.vertices.push_back(0.25 * (m.vertices[2 * v[0] + d] + m.vertices[2 * v[1] + d] + m.vertices[2 * v[2] + d] + m.vertices[2 * v[3] + d]));
    const int32_t ch[4][4] = {{v[0], e01, e02, cc}, {e01, v[1], cc, e13}, {e02, cc, v[2], e23}, {cc, e13, e23, v[3]}};
    for (auto &q : ch) r.cells.insert(r.cells.end(), q, q + 4);
  }
  static const int kids[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};     // children of a parent along its local face 0..3
  for (size_t f = 0; f < m.bface_cell.size(); ++f) for (int h = 0; h < 2; ++h) {
    r.bface_cell.push_back(4 * m.bface_cell[f] + kids[m.bface_local[f]][h]); r.bface_local.push_back(m.bface_local[f]); r.bface_id.push_back(m.bface_id[f]);
  }
  return r;
}

// DoFHandler::distribute_dofs + cell->get_dof_indices for FESystem(FE_Q(k_u),dim) and FE_Q(1).
struct DoFs {
  int k_u = 2;
  int64_t n_u = 0, n_p = 0;
  std::vector<int32_t> cell_u, cell_p;
};

inline DoFs distribute_dofs(const Mesh &m, int k_u) {
  DoFs D; D.k_u = k_u;
  const int dim = m.dim, nvc = 1 << dim, n1 = k_u + 1, ns = ipow(n1, dim);
  const int64_t nc = m.n_cells();
  D.cell_p.assign(m.cells.begin(), m.cells.end());
  D.n_p = m.n_vertices();
  D.cell_u.resize(nc * ns * dim);
  if (m.box.enabled) {
    const int nn[3] = {k_u * m.box.n[0] + 1, k_u * m.box.n[1] + 1, dim == 3 ? k_u * m.box.n[2] + 1 : 1};
    D.n_u = (int64_t)nn[0] * nn[1] * nn[2] * dim;
    const int ncd[3] = {m.box.n[0], m.box.n[1], dim == 3 ? m.box.n[2] : 1};
    for (int k = 0; k < ncd[2]; ++k) for (int j = 0; j < ncd[1]; ++j) for (int i = 0; i < ncd[0]; ++i) {
      const int64_t c = ((int64_t)k * ncd[1] + j) * ncd[0] + i;
      for (int s = 0; s < ns; ++s) {
        const int a = s % n1, b = (s / n1) % n1, cc = s / (n1 * n1);
        const int64_t node = (((int64_t)(k * k_u + cc)) * nn[1] + (j * k_u + b)) * nn[0] + (i * k_u + a);
        for (int d = 0; d < dim; ++d) D.cell_u[(c * ns + s) * dim + d] = (int32_t)(node * dim + d);
      }
    }
    return D;
  }
  if (dim != 2) throw std::runtime_error("unstructured meshes: 2D only");
  if (k_u == 1) {
    D.n_u = m.n_vertices() * dim;
    for (int64_t c = 0; c < nc; ++c) for (int s = 0; s < ns; ++s) for (int d = 0; d < dim; ++d)
      D.cell_u[(c * ns + s) * dim + d] = m.cells[c * nvc + s] * dim + d;
    return D;
  }
  // Q2 on unstructured quads: vertex nodes, then unique edge midpoints, then cell centres
  std::map<std::pair<int32_t, int32_t>, int32_t> edge_node;
  int32_t next = (int32_t)m.n_vertices();
  auto edge = [&](int32_t a, int32_t b) {
    auto key = std::make_pair(std::min(a, b), std::max(a, b));
    auto it = edge_node.find(key);
    if (it != edge_node.end()) return it->second;
    edge_node[key] = next; return next++;
  };
  std::vector<int32_t> node(nc * 9);
  for (int64_t c = 0; c < nc; ++c) {
    const int32_t *v = &m.cells[c * 4];
    int32_t *nd = &node[c * 9];
    nd[0] = v[0]; nd[2] = v[1]; nd[6] = v[2]; nd[8] = v[3];
    nd[1] = edge(v[0], v[1]); nd[3] = edge(v[0], v[2]); nd[5] = edge(v[1], v[3]); nd[7] = edge(v[2], v[3]);
  }
  for (int64_t c = 0; c < nc; ++c) node[c * 9 + 4] = next++;
  D.n_u = (int64_t)next * dim;
  for (int64_t c = 0; c < nc; ++c) for (int s = 0; s < 9; ++s) for (int d = 0; d < dim; ++d)
    D.cell_u[(c * 9 + s) * dim + d] = node[c * 9 + s] * dim + d;
  return D;
}

struct BoundaryConditions {  // BoundaryConditions.h:6-62
  std::vector<int32_t> dirichlet_labels, dirichlet_components, neumann_labels, neumann_components;
  std::vector<double>  dirichlet_values, neumann_values;
  std::vector<int32_t> pressure_labels; std::vector<double> pressure_values;   // extension: prescribed pressure on whole boundary faces (drained boundary)
  std::vector<int32_t> tie_labels, tie_components;   // extension: displacement component tied to ONE value on a boundary (rigid frictionless plate): x[dof] = x[master]
};

// VectorTools::interpolate_boundary_values(label, ConstantFunction(value), constraints, mask[component])
// in the order of the conditions; an already constrained dof keeps its first value
// (PoroElasticDisplacementSolver.h:117-136).
inline void make_dirichlet(const Mesh &m, const DoFs &D, const BoundaryConditions &bc,
                           std::vector<int32_t> &dofs, std::vector<double> &values) {
  const int dim = m.dim, n1 = D.k_u + 1, ns = ipow(n1, dim);
  std::map<int32_t, double> cons;
  for (size_t cond = 0; cond < bc.dirichlet_labels.size(); ++cond) {
    const int comp = bc.dirichlet_components[cond];
    for (size_t bf = 0; bf < m.bface_cell.size(); ++bf) {
      if (m.bface_id[bf] != bc.dirichlet_labels[cond]) continue;
      const int64_t c = m.bface_cell[bf];
      const int f = m.bface_local[bf], nd = f / 2, side = f % 2;
      for (int s = 0; s < ns; ++s) {
        const int idx[3] = {s % n1, (s / n1) % n1, s / (n1 * n1)};
        if (idx[nd] != side * D.k_u) continue;
        const int32_t dof = D.cell_u[(c * ns + s) * dim + comp];
        cons.emplace(dof, bc.dirichlet_values[cond]);
      }
    }
  }
  dofs.clear(); values.clear();
  for (auto &kv : cons) { dofs.push_back(kv.first); values.push_back(kv.second); }
}

// ---- locally refined box: hanging nodes ---------------------------------------------------------------------------------------------
// x[dof] = sum w x[master] + inhomogeneity, closed (ConstraintMatrix::close()): masters are unconstrained
struct ConstraintList {
  std::vector<int32_t> dof, master; std::vector<int64_t> ptr{0}; std::vector<double> weight, inhom;
  int64_t n() const { return (int64_t)dof.size(); }
  poro_constraints c_view() const { return poro_constraints{n(), dof.data(), ptr.data(), master.data(), weight.data(), inhom.data()}; }
};

// Box of n[d] coarse cells; the coarse cells with index lo[d] <= i_d < hi[d] are replaced by their 2^dim children (one refinement level,
// the 2:1 situation deal.II allows).  Nodes live on an integer lattice of spacing h / (2 k) per direction, so shared nodes are found by
// their lattice coordinates.  A node of a fine cell that lies in the closure of an UNREFINED coarse cell without being one of its nodes
// is a hanging node, constrained by that cell's shape functions evaluated there (what DoFTools::make_hanging_node_constraints builds
// from the face interpolation matrices of FE_Q).
struct RefinedBox {
  Mesh mesh; DoFs dofs; ConstraintList cons_u, cons_p;
  // interpolation from the underlying uniform box: displacement node i = sum of weight * box node (lexicographic box numbering), rows by prol_ptr
  std::vector<int64_t> prol_ptr; std::vector<int32_t> prol_node; std::vector<double> prol_w;
  std::vector<int64_t> prol_ptr_p; std::vector<int32_t> prol_node_p; std::vector<double> prol_w_p;     // the same for the pressure space (vertices)
};
inline RefinedBox make_refined_box(int dim, const int n[3], const double origin[3], const double h[3], int k_u, const int lo[3], const int hi[3]) {
  RefinedBox R; Mesh &m = R.mesh; m.dim = dim;
  const int nc[3] = {n[0], n[1], dim == 3 ? n[2] : 1};
  auto refined = [&](const int c[3]) { for (int d = 0; d < dim; ++d) if (c[d] < lo[d] || c[d] >= hi[d]) return false; return true; };
  struct Cell { int c[3]; int child[3]; bool fine; };
  std::vector<Cell> cells;
  for (int k = 0; k < nc[2]; ++k) for (int j = 0; j < nc[1]; ++j) for (int i = 0; i < nc[0]; ++i) {
    const int c[3] = {i, j, k};
    if (!refined(c)) { cells.push_back(Cell{{i, j, k}, {0, 0, 0}, false}); continue; }
    for (int cz = 0; cz < (dim == 3 ? 2 : 1); ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) cells.push_back(Cell{{i, j, k}, {cx, cy, cz}, true});
  }
  // lattice numbering of one scalar space of degree k: key = lattice coordinates in units h / (2k)
  auto key_of = [&](int k, const int X[3]) { const int64_t L1 = 2 * k * nc[0] + 1, L2 = 2 * k * nc[1] + 1; return ((int64_t)(dim == 3 ? X[2] : 0) * L2 + X[1]) * L1 + X[0]; };
  auto number_space = [&](int k, std::vector<int32_t> &cell_nodes, std::map<int64_t, int32_t> &id, std::vector<std::array<int, 3>> &coord) {
    const int n1 = k + 1, ns = ipow(n1, dim);
    cell_nodes.assign(cells.size() * ns, -1);
    for (size_t ci = 0; ci < cells.size(); ++ci) for (int s = 0; s < ns; ++s) {
      const Cell &C = cells[ci]; const int a[3] = {s % n1, (s / n1) % n1, s / (n1 * n1)}; int X[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) X[d] = 2 * k * C.c[d] + (C.fine ? k * C.child[d] + a[d] : 2 * a[d]);
      const int64_t key = key_of(k, X);
      auto it = id.find(key);
      if (it == id.end()) { it = id.emplace(key, (int32_t)coord.size()).first; coord.push_back({X[0], X[1], X[2]}); }
      cell_nodes[ci * ns + s] = it->second;
    }
  };
  // hanging nodes of one scalar space: list of (node, masters, weights)
  auto hanging = [&](int k, const std::map<int64_t, int32_t> &id, const std::vector<std::array<int, 3>> &coord, std::vector<int32_t> &hn, std::vector<std::vector<std::pair<int32_t, double>>> &hw) {
    const int n1 = k + 1, ns = ipow(n1, dim);
    for (size_t nd = 0; nd < coord.size(); ++nd) {
      const int *X = coord[nd].data();
      bool odd = false; for (int d = 0; d < dim; ++d) odd = odd || (X[d] & 1);
      if (!odd) continue;                                    // a node of the coarse lattice can never hang
      // unrefined coarse cells whose closure contains X
      int c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) { const int q = X[d] / (2 * k), r = X[d] % (2 * k); c1[d] = std::min(q, nc[d] - 1); c0[d] = (r == 0 && q > 0) ? q - 1 : c1[d]; }
      bool done = false;
      for (int ck = c0[2]; ck <= c1[2] && !done; ++ck) for (int cj = c0[1]; cj <= c1[1] && !done; ++cj) for (int ci = c0[0]; ci <= c1[0] && !done; ++ci) {
        const int c[3] = {ci, cj, ck};
        if (refined(c)) continue;
        double xi[3] = {0, 0, 0}; for (int d = 0; d < dim; ++d) xi[d] = (double)(X[d] - 2 * k * c[d]) / (2 * k);
        std::vector<double> val(ns), grad((size_t)ns * dim); shape_at(dim, k, xi, val.data(), grad.data());
        std::vector<std::pair<int32_t, double>> w;
        for (int s = 0; s < ns; ++s) if (std::fabs(val[s]) > 1e-13) {
          const int a[3] = {s % n1, (s / n1) % n1, s / (n1 * n1)}; int Y[3] = {0, 0, 0};
          for (int d = 0; d < dim; ++d) Y[d] = 2 * k * c[d] + 2 * a[d];
          w.emplace_back(id.at(key_of(k, Y)), val[s]);
        }
        hn.push_back((int32_t)nd); hw.push_back(w); done = true;
      }
    }
  };
  // pressure space = vertices (k = 1)
  std::map<int64_t, int32_t> idp, idu; std::vector<std::array<int, 3>> cp, cu; std::vector<int32_t> nodes_p, nodes_u;
  number_space(1, nodes_p, idp, cp);
  number_space(k_u, nodes_u, idu, cu);
  m.vertices.resize(cp.size() * dim);
  for (size_t v = 0; v < cp.size(); ++v) for (int d = 0; d < dim; ++d) m.vertices[v * dim + d] = origin[d] + 0.5 * h[d] * cp[v][d];
  m.cells = nodes_p;
  const int nvc = 1 << dim;
  for (size_t ci = 0; ci < cells.size(); ++ci) for (int d = 0; d < dim; ++d) for (int side = 0; side < 2; ++side) {
    const Cell &C = cells[ci];
    const bool at = side ? (C.c[d] == nc[d] - 1 && (!C.fine || C.child[d] == 1)) : (C.c[d] == 0 && (!C.fine || C.child[d] == 0));
    if (at) { m.bface_cell.push_back((int32_t)ci); m.bface_local.push_back(2 * d + side); m.bface_id.push_back(2 * d + side); }
  }
  (void)nvc;
  DoFs &D = R.dofs; D.k_u = k_u; D.n_p = (int64_t)cp.size(); D.n_u = (int64_t)cu.size() * dim; D.cell_p = nodes_p;
  D.cell_u.resize(nodes_u.size() * dim);
  for (size_t i = 0; i < nodes_u.size(); ++i) for (int d = 0; d < dim; ++d) D.cell_u[i * dim + d] = nodes_u[i] * dim + d;
  std::vector<int32_t> hn; std::vector<std::vector<std::pair<int32_t, double>>> hw;
  hanging(1, idp, cp, hn, hw);
  for (size_t i = 0; i < hn.size(); ++i) {
    R.cons_p.dof.push_back(hn[i]); R.cons_p.inhom.push_back(0.0);
    for (auto &mw : hw[i]) { R.cons_p.master.push_back(mw.first); R.cons_p.weight.push_back(mw.second); }
    R.cons_p.ptr.push_back((int64_t)R.cons_p.master.size());
  }
  hn.clear(); hw.clear();
  hanging(k_u, idu, cu, hn, hw);
  for (size_t i = 0; i < hn.size(); ++i) for (int d = 0; d < dim; ++d) {
    R.cons_u.dof.push_back(hn[i] * dim + d); R.cons_u.inhom.push_back(0.0);
    for (auto &mw : hw[i]) { R.cons_u.master.push_back(mw.first * dim + d); R.cons_u.weight.push_back(mw.second); }
    R.cons_u.ptr.push_back((int64_t)R.cons_u.master.size());
  }
  // the FE functions of the unrefined box evaluated at every displacement node (lattice coordinates are in units h / (2 k_u))
  {
    const int k = k_u, n1 = k + 1, ns = ipow(n1, dim); const int64_t nn0 = (int64_t)k * nc[0] + 1, nn1 = (int64_t)k * nc[1] + 1;
    std::vector<double> val(ns), grad((size_t)ns * dim);
    R.prol_ptr.assign(1, 0);
    for (size_t nd = 0; nd < cu.size(); ++nd) {
      int c[3] = {0, 0, 0}; double xi[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) { c[d] = std::min(cu[nd][d] / (2 * k), nc[d] - 1); xi[d] = (double)(cu[nd][d] - 2 * k * c[d]) / (2 * k); }
      shape_at(dim, k, xi, val.data(), grad.data());
      for (int s = 0; s < ns; ++s) if (std::fabs(val[s]) > 1e-13) {
        const int a[3] = {s % n1, (s / n1) % n1, s / (n1 * n1)};
        const int64_t node = ((int64_t)(dim == 3 ? k * c[2] + a[2] : 0) * nn1 + (k * c[1] + a[1])) * nn0 + (k * c[0] + a[0]);
        R.prol_node.push_back((int32_t)node); R.prol_w.push_back(val[s]);
      }
      R.prol_ptr.push_back((int64_t)R.prol_node.size());
    }
    // pressure space: vertices, lattice units h / 2, box vertices lexicographic
    const int nsp = 1 << dim; const int64_t np0 = (int64_t)nc[0] + 1, np1 = (int64_t)nc[1] + 1;
    std::vector<double> valp(nsp), gradp((size_t)nsp * dim);
    R.prol_ptr_p.assign(1, 0);
    for (size_t nd = 0; nd < cp.size(); ++nd) {
      int c[3] = {0, 0, 0}; double xi[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) { c[d] = std::min(cp[nd][d] / 2, nc[d] - 1); xi[d] = (double)(cp[nd][d] - 2 * c[d]) / 2; }
      shape_at(dim, 1, xi, valp.data(), gradp.data());
      for (int s = 0; s < nsp; ++s) if (std::fabs(valp[s]) > 1e-13) {
        const int a[3] = {s & 1, (s >> 1) & 1, s >> 2};
        R.prol_node_p.push_back((int32_t)(((int64_t)(dim == 3 ? c[2] + a[2] : 0) * np1 + (c[1] + a[1])) * np0 + (c[0] + a[0]))); R.prol_w_p.push_back(valp[s]);
      }
      R.prol_ptr_p.push_back((int64_t)R.prol_node_p.size());
    }
  }
  return R;
}

// extension: prescribed pressure on the faces carrying the given labels (first condition wins, like the displacement conditions)
inline void make_dirichlet_p(const Mesh &m, const DoFs &D, const BoundaryConditions &bc, std::vector<int32_t> &dofs, std::vector<double> &values) {
  const int dim = m.dim, nv = 1 << dim;
  std::map<int32_t, double> cons;
  for (size_t cond = 0; cond < bc.pressure_labels.size(); ++cond)
    for (size_t bf = 0; bf < m.bface_cell.size(); ++bf) {
      if (m.bface_id[bf] != bc.pressure_labels[cond]) continue;
      const int64_t c = m.bface_cell[bf]; const int f = m.bface_local[bf], nd = f / 2, side = f % 2;
      for (int v = 0; v < nv; ++v) if (((v >> nd) & 1) == side) cons.emplace(D.cell_p[c * nv + v], bc.pressure_values[cond]);
    }
  dofs.clear(); values.clear();
  for (auto &kv : cons) { dofs.push_back(kv.first); values.push_back(kv.second); }
}

// Everything poro_desc points at, owned in one place.
struct ProblemData {
  Mesh mesh; DoFs dofs; FETables fe; BoundaryConditions bc;
  std::vector<int32_t> dirichlet_dof; std::vector<double> dirichlet_value;
  std::vector<int32_t> dirichlet_dof_p; std::vector<double> dirichlet_value_p;
  poro_material mat{}; poro_partition part{};
  ConstraintList cons_u, cons_p;      // hanging-node constraints (locally refined meshes)
  // general partition (partition_problem): interface lists and the local -> global maps of the piece
  std::vector<int32_t> part_neighbours, part_shared_u, part_shared_p, local_to_global_u, local_to_global_p;
  std::vector<int64_t> part_ptr_u, part_ptr_p;
  bool ties_added = false;
  bool dirichlet_given = false;       // the Dirichlet list was filled by the caller (pieces of a partition: from the global list)
  std::vector<double> tensor_grid[3]; // vertex planes per direction of a tensor-product grid without the box tag (graded boxes)
  // locally refined boxes: the underlying uniform box as a problem of its own + the interpolation from it (poro_coarse_space)
  std::unique_ptr<ProblemData> coarse; std::vector<int64_t> prol_ptr; std::vector<int32_t> prol_node; std::vector<double> prol_w;
  std::vector<int64_t> prol_ptr_p; std::vector<int32_t> prol_node_p; std::vector<double> prol_w_p;
  poro_desc d{};

  // ConstraintMatrix semantics of PoroElasticDisplacementSolver.h:112-136: hanging-node constraints first, boundary values only for dofs that are
  // not constrained yet, then close(): a hanging node whose master carries a boundary value gets it as an inhomogeneity
  void close_constraints() {
    if (!cons_u.n()) return;
    std::map<int32_t, double> dir; for (size_t i = 0; i < dirichlet_dof.size(); ++i) dir[dirichlet_dof[i]] = dirichlet_value[i];
    for (int32_t hd : cons_u.dof) dir.erase(hd);
    ConstraintList out;
    for (int64_t i = 0; i < cons_u.n(); ++i) {
      double b = cons_u.inhom[i];
      out.dof.push_back(cons_u.dof[i]);
      for (int64_t k = cons_u.ptr[i]; k < cons_u.ptr[i + 1]; ++k) {
        auto it = dir.find(cons_u.master[k]);
        if (it != dir.end()) b += cons_u.weight[k] * it->second; else { out.master.push_back(cons_u.master[k]); out.weight.push_back(cons_u.weight[k]); }
      }
      out.inhom.push_back(b); out.ptr.push_back((int64_t)out.master.size());
    }
    cons_u = out;
    dirichlet_dof.clear(); dirichlet_value.clear();
    for (auto &kv : dir) { dirichlet_dof.push_back(kv.first); dirichlet_value.push_back(kv.second); }
  }
  // extension: rigid-plate conditions.  All dofs of component comp on the faces labelled `label` that carry no boundary value are tied to the first of them
  // (an ordinary entry of the constraint list: x[dof] = 1 * x[master]); the tractions of the tied rows then add up on the master (C^T b)
  void add_ties() {
    const int dim = mesh.dim, n1 = dofs.k_u + 1, ns = ipow(n1, dim);
    for (size_t t = 0; t < bc.tie_labels.size(); ++t) {
      std::vector<int32_t> on;
      for (size_t bf = 0; bf < mesh.bface_cell.size(); ++bf) {
        if (mesh.bface_id[bf] != bc.tie_labels[t]) continue;
        const int64_t c = mesh.bface_cell[bf]; const int f = mesh.bface_local[bf], nd = f / 2, side = f % 2;
        for (int s = 0; s < ns; ++s) { const int idx[3] = {s % n1, (s / n1) % n1, s / (n1 * n1)}; if (idx[nd] == side * dofs.k_u) on.push_back(dofs.cell_u[(c * ns + s) * dim + bc.tie_components[t]]); }
      }
      std::sort(on.begin(), on.end()); on.erase(std::unique(on.begin(), on.end()), on.end());
      std::vector<int32_t> freed;
      for (int32_t d : on) if (!std::binary_search(dirichlet_dof.begin(), dirichlet_dof.end(), d) && std::find(cons_u.dof.begin(), cons_u.dof.end(), d) == cons_u.dof.end()) freed.push_back(d);
      for (size_t i = 1; i < freed.size(); ++i) {
        cons_u.dof.push_back(freed[i]); cons_u.master.push_back(freed[0]); cons_u.weight.push_back(1.0); cons_u.inhom.push_back(0.0); cons_u.ptr.push_back((int64_t)cons_u.master.size());
      }
    }
  }
  void finalize(int k_u, bool have_dofs = false) {
    if (!have_dofs) dofs = distribute_dofs(mesh, k_u);
    fe.build(mesh.dim, k_u);
    if (!dirichlet_given) make_dirichlet(mesh, dofs, bc, dirichlet_dof, dirichlet_value);
    if (!ties_added && !bc.tie_labels.empty()) { add_ties(); ties_added = true; }
    close_constraints();
    d = poro_desc{};
    d.abi_version = PORO_ABI_VERSION; d.dim = mesh.dim; d.degree_u = k_u; d.degree_p = 1;
    d.n_cells = mesh.n_cells(); d.n_vertices = mesh.n_vertices(); d.n_dofs_u = dofs.n_u; d.n_dofs_p = dofs.n_p;
    d.vertex_coords = mesh.vertices.data(); d.cell_vertices = mesh.cells.data();
    d.cell_dofs_u = dofs.cell_u.data(); d.cell_dofs_p = dofs.cell_p.data();
    d.fe = fe.c;
    d.n_bfaces = (int64_t)mesh.bface_cell.size();
    d.bface_cell = mesh.bface_cell.data(); d.bface_local = mesh.bface_local.data(); d.bface_id = mesh.bface_id.data();
    d.n_dirichlet = (int64_t)dirichlet_dof.size(); d.dirichlet_dof = dirichlet_dof.data(); d.dirichlet_value = dirichlet_value.data();
    d.n_neumann = (int32_t)bc.neumann_labels.size();
    d.neumann_label = bc.neumann_labels.data(); d.neumann_component = bc.neumann_components.data(); d.neumann_value = bc.neumann_values.data();
    d.mat = mat; d.box = mesh.box;
    if (cons_u.n() || cons_p.n()) d.box.enabled = 0;   // constraint lists run on the general operators (the structured kernels know Dirichlet conditions only)
    if (part.n_ranks == 0) { part.n_ranks = 1; part.rank = 0; }
    part.n_neighbours = (int32_t)part_neighbours.size();
    part.neighbour_rank = part_neighbours.data(); part.shared_ptr_u = part_ptr_u.data(); part.shared_dof_u = part_shared_u.data();
    part.shared_ptr_p = part_ptr_p.data(); part.shared_dof_p = part_shared_p.data();
    d.part = part;
    d.cons_u = cons_u.c_view(); d.cons_p = cons_p.c_view();
    set_pressure_bc();
  }
  void set_pressure_bc() {
    if (!dirichlet_given) make_dirichlet_p(mesh, dofs, bc, dirichlet_dof_p, dirichlet_value_p);
    d.n_dirichlet_p = (int64_t)dirichlet_dof_p.size(); d.dirichlet_dof_p = dirichlet_dof_p.data(); d.dirichlet_value_p = dirichlet_value_p.data();
  }
};

// Slab `rank` of `n_ranks` of the global box: whole cell layers in the slowest direction.
inline void slab_range(int n_layers, int rank, int n_ranks, int &c0, int &c1) {
  const int base = n_layers / n_ranks, rem = n_layers % n_ranks;
  c0 = rank * base + std::min(rank, rem);
  c1 = c0 + base + (rank < rem ? 1 : 0);
}

inline void build_box_problem(ProblemData &P, int dim, const int n[3], const double size[3], int k_u,
                              int rank = 0, int n_ranks = 1) {
  // hyper_rectangle(p1=+size/2, p2=-size/2): the box [-size/2, size/2]^dim (PoroelasticityFSS.h:423-432)
  double origin[3] = {0, 0, 0}, h[3] = {1, 1, 1};
  int nl[3] = {n[0], n[1], dim == 3 ? n[2] : 1};
  for (int d = 0; d < dim; ++d) { h[d] = size[d] / n[d]; origin[d] = -size[d] / 2; }
  const int sd = dim - 1;
  int c0 = 0, c1 = n[sd];
  if (n_ranks > 1) slab_range(n[sd], rank, n_ranks, c0, c1);
  nl[sd] = c1 - c0; origin[sd] += h[sd] * c0;
  P.mesh = make_box(dim, nl, origin, h, rank == 0, rank == n_ranks - 1);
  P.part.rank = rank; P.part.n_ranks = n_ranks;
  P.part.has_lower = rank > 0; P.part.has_upper = rank < n_ranks - 1;
  int64_t pu = dim, pp = 1;
  for (int d = 0; d < sd; ++d) { pu *= (k_u * n[d] + 1); pp *= (n[d] + 1); }
  P.part.plane_u = pu; P.part.plane_p = pp;
  P.finalize(k_u);
}

// Auxiliary coarse space for a general 2D mesh that fills a rectangle whose sides carry one boundary id each, in the colorized order (x low, x high, y low, y high = 0..3):
// a uniform box of about as many cells with the same boundary conditions, and the box's FE functions evaluated at every displacement node of the mesh (poro_coarse_space).
// The spaces are not nested; that is what the two-level preconditioner's smoother is for (auxiliary-space preconditioning).  Returns false (and attaches nothing) when
// the geometry or the labels do not fit.
inline bool attach_auxiliary_box(ProblemData &P, int k_u) {
  const Mesh &m = P.mesh; const int dim = m.dim;
  if (dim != 2 || m.box.enabled || P.cons_u.n() || P.part.n_ranks > 1) return false;
  double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
  for (int64_t v = 0; v < m.n_vertices(); ++v) for (int d = 0; d < 2; ++d) { lo[d] = std::min(lo[d], m.vertices[2 * v + d]); hi[d] = std::max(hi[d], m.vertices[2 * v + d]); }
  const double size[2] = {hi[0] - lo[0], hi[1] - lo[1]};
  if (!(size[0] > 0 && size[1] > 0)) return false;
  static const int fv[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};
  // every boundary edge lies on one side of the rectangle, and a boundary id names ONE side (Gmsh numbers the sides of domain.msh counter-clockwise from the bottom,
  // hyper_rectangle(colorize) by direction): the box gets the mesh's conditions with the ids translated to its own
  std::map<int, int> side_of_id;
  for (size_t f = 0; f < m.bface_cell.size(); ++f) {
    const int32_t va = m.cells[4 * m.bface_cell[f] + fv[m.bface_local[f]][0]], vb = m.cells[4 * m.bface_cell[f] + fv[m.bface_local[f]][1]];
    int side = -1;
    for (int d = 0; d < 2 && side < 0; ++d) for (int hs = 0; hs < 2; ++hs) {
      const double want = hs ? hi[d] : lo[d];
      if (std::fabs(m.vertices[2 * va + d] - want) <= 1e-9 * size[d] && std::fabs(m.vertices[2 * vb + d] - want) <= 1e-9 * size[d]) { side = 2 * d + hs; break; }
    }
    if (side < 0) return false;                                 // (a hole, a curved or slanted boundary: not a rectangle)
    auto it = side_of_id.find(m.bface_id[f]);
    if (it == side_of_id.end()) side_of_id[m.bface_id[f]] = side; else if (it->second != side) return false;
  }
  auto translate = [&](std::vector<int32_t> &labels) { for (auto &l : labels) { auto it = side_of_id.find(l); l = it == side_of_id.end() ? -1 - l : it->second; } };   // (ids no edge carries: to ids the box does not have)
  // resolution of the auxiliary box: as many cells as the mesh, divided by coarsen^2 (the spaces need not be nested; a coarser box makes the coarse solve cheaper
  // - its transforms cost O(n^3) per application - at the price of a few CG iterations; PORO_AUX_BOX_COARSEN overrides the default)
  // measured on the bundled grid (profiles/r03_gmsh_step.json): half the resolution costs 5 of 51 CG iterations per step and saves 12 % of the step at 102 400 cells;
  // a third of it costs 25 iterations.  Small meshes keep the full resolution (the coarse solve is launch-bound there anyway).
  static const double coarsen_env = [] { const char *e = std::getenv("PORO_AUX_BOX_COARSEN"); const double v = e ? std::atof(e) : 0.0; return v >= 1.0 ? v : 0.0; }();
  const double coarsen = coarsen_env > 0 ? coarsen_env : (m.n_cells() >= 20000 ? 2.0 : 1.0);
  const double target = (double)m.n_cells() / (coarsen * coarsen);
  const int nside = std::max(2, (int)std::lround(std::sqrt(target * size[0] / size[1])));
  int n[3] = {nside, std::max(2, (int)std::lround(target / nside)), 1};
  double origin[3] = {lo[0], lo[1], 0}, h[3] = {size[0] / n[0], size[1] / n[1], 1};
  P.coarse.reset(new ProblemData()); ProblemData &C = *P.coarse;
  C.bc = P.bc; C.mat = P.mat;
  translate(C.bc.dirichlet_labels); translate(C.bc.neumann_labels); translate(C.bc.pressure_labels); translate(C.bc.tie_labels);
  C.mesh = make_box(dim, n, origin, h, true, true);
  C.part = poro_partition{}; C.part.n_ranks = 1;
  C.finalize(k_u);
  // displacement nodes of the mesh: positions by the cells' Q1 maps of the reference nodes
  const int n1 = k_u + 1, ns = ipow(n1, dim); const int64_t nn = P.dofs.n_u / dim;
  std::vector<double> X((size_t)nn * 2, std::numeric_limits<double>::quiet_NaN());
  for (int64_t c = 0; c < m.n_cells(); ++c) for (int s = 0; s < ns; ++s) {
    const int64_t node = P.dofs.cell_u[(c * ns + s) * dim] / dim;
    if (X[2 * node] == X[2 * node]) continue;
    const double a = (double)(s % n1) / k_u, b = (double)(s / n1) / k_u, w[4] = {(1 - a) * (1 - b), a * (1 - b), (1 - a) * b, a * b};
    for (int d = 0; d < 2; ++d) { double x = 0; for (int v = 0; v < 4; ++v) x += w[v] * m.vertices[2 * m.cells[4 * c + v] + d]; X[2 * node + d] = x; }
  }
  const int64_t nn0 = (int64_t)k_u * n[0] + 1;
  std::vector<double> val(ns), grad((size_t)ns * dim);
  P.prol_ptr.assign(1, 0); P.prol_node.clear(); P.prol_w.clear();
  for (int64_t i = 0; i < nn; ++i) {
    int c[2]; double xi[2];
    for (int d = 0; d < 2; ++d) { const double t = (X[2 * i + d] - origin[d]) / h[d]; c[d] = std::min(std::max((int)std::floor(t), 0), n[d] - 1); xi[d] = std::min(std::max(t - c[d], 0.0), 1.0); }
    shape_at(dim, k_u, xi, val.data(), grad.data());
    for (int s = 0; s < ns; ++s) if (std::fabs(val[s]) > 1e-13) {
      P.prol_node.push_back((int32_t)((int64_t)(k_u * c[1] + s / n1) * nn0 + (k_u * c[0] + s % n1))); P.prol_w.push_back(val[s]);
    }
    P.prol_ptr.push_back((int64_t)P.prol_node.size());
  }
  // pressure space: the mesh's vertices in the box's Q1 space
  { const int64_t np0 = (int64_t)n[0] + 1; double v4[4], g8[8];
    P.prol_ptr_p.assign(1, 0); P.prol_node_p.clear(); P.prol_w_p.clear();
    for (int64_t v = 0; v < m.n_vertices(); ++v) {
      int c[2]; double xi[2];
      for (int d = 0; d < 2; ++d) { const double t = (m.vertices[2 * v + d] - origin[d]) / h[d]; c[d] = std::min(std::max((int)std::floor(t), 0), n[d] - 1); xi[d] = std::min(std::max(t - c[d], 0.0), 1.0); }
      shape_at(dim, 1, xi, v4, g8);
      for (int s = 0; s < 4; ++s) if (std::fabs(v4[s]) > 1e-13) { P.prol_node_p.push_back((int32_t)((int64_t)(c[1] + (s >> 1)) * np0 + (c[0] + (s & 1)))); P.prol_w_p.push_back(v4[s]); }
      P.prol_ptr_p.push_back((int64_t)P.prol_node_p.size());
    } }
  P.d.coarse = poro_coarse_space{}; P.d.coarse.enabled = 1; P.d.coarse.box_problem = &C.d;
  P.d.coarse.ptr = P.prol_ptr.data(); P.d.coarse.node = P.prol_node.data(); P.d.coarse.weight = P.prol_w.data();
  P.d.coarse.ptr_p = P.prol_ptr_p.data(); P.d.coarse.node_p = P.prol_node_p.data(); P.d.coarse.weight_p = P.prol_w_p.data();
  return true;
}

// Graded box: the colorized box with its vertices moved by x -> origin + size * (exp(g t) - 1) / (exp(g) - 1), t in [0, 1], per direction (g = 0: unchanged).
// Cells stay rectilinear but differ in size, so the mesh carries NO box tag: it runs through the general kernels (the mesh class a graded or r-adapted
// hyper_rectangle of the reference would give; used to measure the general matrix-free operator at scale)
inline void build_graded_box_problem(ProblemData &P, int dim, const int n[3], const double size[3], int k_u, const double grading[3]) {
  double origin[3] = {0, 0, 0}, h[3] = {1, 1, 1};
  int nl[3] = {n[0], n[1], dim == 3 ? n[2] : 1};
  for (int d = 0; d < dim; ++d) { h[d] = size[d] / n[d]; origin[d] = -size[d] / 2; }
  P.mesh = make_box(dim, nl, origin, h, true, true);
  for (int64_t v = 0; v < P.mesh.n_vertices(); ++v)
    for (int d = 0; d < dim; ++d) {
      const double g = grading[d]; double &x = P.mesh.vertices[(size_t)v * dim + d];
      if (g != 0.0) { const double t = (x - origin[d]) / size[d]; x = origin[d] + size[d] * std::expm1(g * t) / std::expm1(g); }
    }
  P.part.rank = 0; P.part.n_ranks = 1; P.part.has_lower = 0; P.part.has_upper = 0; P.part.plane_u = 0; P.part.plane_p = 0;
  P.finalize(k_u);            // (the dofs keep the lexicographic numbering of the box)
  P.d.box.enabled = 0;        // ... but the descriptor carries no box tag: the cells are not congruent
  // what is left of the structure: a tensor-product grid (the fast-diagonalisation preconditioners stay exact on it)
  P.d.tensor = poro_tensor_grid{}; P.d.tensor.enabled = 1;
  for (int d = 0; d < 3; ++d) {
    P.d.tensor.n[d] = d < dim ? n[d] : 1; P.tensor_grid[d].assign((size_t)P.d.tensor.n[d] + 1, 0.0);
    if (d < dim) for (int i = 0; i <= n[d]; ++i) { const double g = grading[d], t = (double)i / n[d]; P.tensor_grid[d][i] = g != 0.0 ? origin[d] + size[d] * std::expm1(g * t) / std::expm1(g) : origin[d] + size[d] * t; }
    P.d.tensor.grid[d] = P.tensor_grid[d].data();
  }
}

inline void build_refined_box_problem(ProblemData &P, int dim, const int n[3], const double size[3], int k_u, const int lo[3], const int hi[3]) {
  double origin[3] = {0, 0, 0}, h[3] = {1, 1, 1};
  for (int d = 0; d < dim; ++d) { h[d] = size[d] / n[d]; origin[d] = -size[d] / 2; }
  RefinedBox R = make_refined_box(dim, n, origin, h, k_u, lo, hi);
  P.mesh = std::move(R.mesh); P.dofs = std::move(R.dofs); P.cons_u = std::move(R.cons_u); P.cons_p = std::move(R.cons_p);
  P.prol_ptr = std::move(R.prol_ptr); P.prol_node = std::move(R.prol_node); P.prol_w = std::move(R.prol_w);
  P.prol_ptr_p = std::move(R.prol_ptr_p); P.prol_node_p = std::move(R.prol_node_p); P.prol_w_p = std::move(R.prol_w_p);
  P.part = poro_partition{}; P.part.n_ranks = 1;
  P.finalize(k_u, true);
  // coarse space of the two-level preconditioner: the unrefined box with the same material and boundary conditions
  P.coarse.reset(new ProblemData()); P.coarse->bc = P.bc; P.coarse->mat = P.mat;
  build_box_problem(*P.coarse, dim, n, size, k_u);
  P.d.coarse = poro_coarse_space{}; P.d.coarse.enabled = 1; P.d.coarse.box_problem = &P.coarse->d;
  P.d.coarse.ptr = P.prol_ptr.data(); P.d.coarse.node = P.prol_node.data(); P.d.coarse.weight = P.prol_w.data();
  P.d.coarse.ptr_p = P.prol_ptr_p.data(); P.d.coarse.node_p = P.prol_node_p.data(); P.d.coarse.weight_p = P.prol_w_p.data();
}

// ---- general partition (last sentence): contiguous ranges of the cells in Morton order + indexed interface lists --------------
// Works on any global ProblemData (Gmsh mesh, box, graded box): rank r gets the cells of its range as a standalone mesh.  Local numbering: the
// dofs this rank owns first (owner = the highest rank touching the dof, as the upper slab owns a shared plane), each group in ascending global
// order, which keeps the components of a displacement node adjacent.  Pressure dofs and vertices coincide in this provider, so the local vertex
// numbering is the local pressure numbering.
inline std::vector<int64_t> morton_cell_order(const Mesh &m) {
  const int dim = m.dim, nv = 1 << dim; const int64_t nc = m.n_cells();
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t v = 0; v < m.n_vertices(); ++v) for (int d = 0; d < dim; ++d) { lo[d] = std::min(lo[d], m.vertices[v * dim + d]); hi[d] = std::max(hi[d], m.vertices[v * dim + d]); }
  std::vector<std::pair<uint64_t, int64_t>> key(nc);
  for (int64_t c = 0; c < nc; ++c) {
    uint64_t q[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
      double x = 0; for (int v = 0; v < nv; ++v) x += m.vertices[(int64_t)m.cells[c * nv + v] * dim + d];
      x = (x / nv - lo[d]) / std::max(hi[d] - lo[d], 1e-300);
      q[d] = (uint64_t)std::min(1023.0, std::max(0.0, std::floor(x * 1024.0)));
    }
    uint64_t k = 0;
    for (int b = 9; b >= 0; --b) for (int d = dim - 1; d >= 0; --d) k = (k << 1) | ((q[d] >> b) & 1);
    key[c] = {k, c};
  }
  std::sort(key.begin(), key.end());
  std::vector<int64_t> order(nc); for (int64_t i = 0; i < nc; ++i) order[i] = key[i].second;
  return order;
}

inline void partition_problem(const ProblemData &G, int rank, int n_ranks, ProblemData &L) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::runtime_error("partition_problem: bad rank / n_ranks");
  if (G.part.n_ranks > 1) throw std::runtime_error("partition_problem: the problem is already a piece of a partition");
  if (!G.dirichlet_dof_p.empty()) throw std::runtime_error("partition_problem: prescribed pressures are implemented for one rank");
  const Mesh &gm = G.mesh; const int dim = gm.dim, nv = 1 << dim, k_u = G.dofs.k_u, dpc = ipow(k_u + 1, dim) * dim; const int64_t nc = gm.n_cells();
  if (nc < n_ranks) throw std::runtime_error("partition_problem: fewer cells than ranks");
  const std::vector<int64_t> order = morton_cell_order(gm);
  std::vector<int32_t> cell_rank(nc);
  for (int r = 0; r < n_ranks; ++r) { int c0, c1; slab_range((int)nc, r, n_ranks, c0, c1); for (int i = c0; i < c1; ++i) cell_rank[order[i]] = r; }
  // ranks touching every dof (sorted, unique)
  auto touching = [&](const std::vector<int32_t> &cell_dofs, int per_cell, int64_t n) {
    std::vector<std::vector<int32_t>> t(n);
    for (int64_t c = 0; c < nc; ++c) for (int k = 0; k < per_cell; ++k) { auto &v = t[cell_dofs[c * per_cell + k]]; if (std::find(v.begin(), v.end(), cell_rank[c]) == v.end()) v.push_back(cell_rank[c]); }
    for (auto &v : t) std::sort(v.begin(), v.end());
    return t;
  };
  auto tu = touching(G.dofs.cell_u, dpc, G.dofs.n_u), tp = touching(G.dofs.cell_p, nv, G.dofs.n_p);
  // constraint lists (hanging nodes, ties; closed): a rank that holds a constrained dof must hold all of its masters, or it could neither expand x nor fold the row.  Masters that
  // none of the rank's cells touch become GHOST dofs of the piece: local, shared with the ranks that do touch them (so the interface exchange keeps their values and row sums
  // consistent), part of no local cell.  Each rank then condenses with its own complete slice of the list; partial rows are folded BEFORE the interface sums (the library's order)
  auto add_ghost_masters = [&](std::vector<std::vector<int32_t>> &t, const ConstraintList &cl) {
    for (int64_t i = 0; i < cl.n(); ++i) for (int32_t r : std::vector<int32_t>(t[cl.dof[i]])) for (int64_t k = cl.ptr[i]; k < cl.ptr[i + 1]; ++k) {
      auto &v = t[cl.master[k]]; auto it = std::lower_bound(v.begin(), v.end(), r); if (it == v.end() || *it != r) v.insert(it, r);
    }
  };
  add_ghost_masters(tu, G.cons_u); add_ghost_masters(tp, G.cons_p);
  auto has = [&](const std::vector<int32_t> &v) { return std::binary_search(v.begin(), v.end(), (int32_t)rank); };
  // local numbering: owned first, each group ascending in the global index
  auto number = [&](const std::vector<std::vector<int32_t>> &t, std::vector<int32_t> &l2g, std::vector<int32_t> &g2l, int64_t &n_owned) {
    g2l.assign(t.size(), -1); l2g.clear();
    for (size_t g = 0; g < t.size(); ++g) if (has(t[g]) && t[g].back() == rank) { g2l[g] = (int32_t)l2g.size(); l2g.push_back((int32_t)g); }
    n_owned = (int64_t)l2g.size();
    for (size_t g = 0; g < t.size(); ++g) if (has(t[g]) && t[g].back() != rank) { g2l[g] = (int32_t)l2g.size(); l2g.push_back((int32_t)g); }
  };
  std::vector<int32_t> g2l_u, g2l_p; int64_t own_u = 0, own_p = 0;
  number(tu, L.local_to_global_u, g2l_u, own_u); number(tp, L.local_to_global_p, g2l_p, own_p);
  // interface lists: for every other rank q the dofs both touch, ascending global index
  std::map<int32_t, std::vector<int32_t>> su, sp;
  for (size_t g = 0; g < tu.size(); ++g) if (has(tu[g])) for (int32_t q : tu[g]) if (q != rank) su[q].push_back(g2l_u[g]);
  for (size_t g = 0; g < tp.size(); ++g) if (has(tp[g])) for (int32_t q : tp[g]) if (q != rank) sp[q].push_back(g2l_p[g]);
  for (auto &kv : su) sp[kv.first];       // same neighbour set for both spaces (pressure lists may be empty)
  for (auto &kv : sp) su[kv.first];
  L.part_neighbours.clear(); L.part_ptr_u.assign(1, 0); L.part_ptr_p.assign(1, 0); L.part_shared_u.clear(); L.part_shared_p.clear();
  for (auto &kv : su) {
    L.part_neighbours.push_back(kv.first);
    L.part_shared_u.insert(L.part_shared_u.end(), kv.second.begin(), kv.second.end()); L.part_ptr_u.push_back((int64_t)L.part_shared_u.size());
    const auto &vp = sp[kv.first]; L.part_shared_p.insert(L.part_shared_p.end(), vp.begin(), vp.end()); L.part_ptr_p.push_back((int64_t)L.part_shared_p.size());
  }
  // the piece as a standalone mesh
  Mesh &m = L.mesh; m = Mesh{}; m.dim = dim;
  m.vertices.resize(L.local_to_global_p.size() * dim);
  for (size_t v = 0; v < L.local_to_global_p.size(); ++v) for (int d = 0; d < dim; ++d) m.vertices[v * dim + d] = gm.vertices[(int64_t)L.local_to_global_p[v] * dim + d];
  std::vector<int32_t> local_cell(nc, -1); int32_t nl = 0;
  L.dofs = DoFs{}; L.dofs.k_u = k_u; L.dofs.n_u = (int64_t)L.local_to_global_u.size(); L.dofs.n_p = (int64_t)L.local_to_global_p.size();
  for (int64_t i = 0; i < nc; ++i) { const int64_t c = order[i]; if (cell_rank[c] != rank) continue;    // local cells keep the Morton order
    local_cell[c] = nl++;
    for (int v = 0; v < nv; ++v) { m.cells.push_back(g2l_p[gm.cells[c * nv + v]]); L.dofs.cell_p.push_back(g2l_p[G.dofs.cell_p[c * nv + v]]); }
    for (int k = 0; k < dpc; ++k) L.dofs.cell_u.push_back(g2l_u[G.dofs.cell_u[c * dpc + k]]);
  }
  for (size_t f = 0; f < gm.bface_cell.size(); ++f) if (cell_rank[gm.bface_cell[f]] == rank) { m.bface_cell.push_back(local_cell[gm.bface_cell[f]]); m.bface_local.push_back(gm.bface_local[f]); m.bface_id.push_back(gm.bface_id[f]); }
  // boundary values from the GLOBAL closed list: a local dof can sit on the boundary without any local boundary face
  L.dirichlet_dof.clear(); L.dirichlet_value.clear(); L.dirichlet_given = true;
  { std::vector<std::pair<int32_t, double>> dl;
    for (size_t i = 0; i < G.dirichlet_dof.size(); ++i) if (g2l_u[G.dirichlet_dof[i]] >= 0) dl.emplace_back(g2l_u[G.dirichlet_dof[i]], G.dirichlet_value[i]);
    std::sort(dl.begin(), dl.end());
    for (auto &kv : dl) { L.dirichlet_dof.push_back(kv.first); L.dirichlet_value.push_back(kv.second); } }
  L.bc = G.bc; L.mat = G.mat;
  // the rank's slice of the (closed) constraint lists in local numbering; ties were entered on the global problem already
  auto localise = [&](const ConstraintList &cl, const std::vector<int32_t> &g2l, ConstraintList &out) {
    out = ConstraintList{};
    std::vector<std::pair<int32_t, int64_t>> mine;
    for (int64_t i = 0; i < cl.n(); ++i) if (g2l[cl.dof[i]] >= 0) mine.emplace_back(g2l[cl.dof[i]], i);
    std::sort(mine.begin(), mine.end());
    for (auto &kv : mine) {
      const int64_t i = kv.second; out.dof.push_back(kv.first); out.inhom.push_back(cl.inhom[i]);
      for (int64_t k = cl.ptr[i]; k < cl.ptr[i + 1]; ++k) { if (g2l[cl.master[k]] < 0) throw std::runtime_error("partition_problem: a master is not local (internal error)"); out.master.push_back(g2l[cl.master[k]]); out.weight.push_back(cl.weight[k]); }
      out.ptr.push_back((int64_t)out.master.size());
    }
  };
  localise(G.cons_u, g2l_u, L.cons_u); localise(G.cons_p, g2l_p, L.cons_p);
  L.ties_added = true;
  L.part = poro_partition{}; L.part.rank = rank; L.part.n_ranks = n_ranks; L.part.n_owned_u = own_u; L.part.n_owned_p = own_p;
  L.finalize(k_u, true);
}

}  // namespace poro_host
