// Minimal mesh + DoF provider feeding the hot path without deal.II:
//  - structured box = GridGenerator::hyper_rectangle(colorize) + refine_global
//    (PoroelasticityFSS.h:418-435), optionally one z-slab (y-slab in 2D) of it;
//  - Gmsh 2.2 ASCII reader = GridIn::read_msh (PoroelasticityFSS.h:438-445) for 2D quads;
//  - DoF numbering, boundary faces and the closed Dirichlet constraint list
//    (PoroElasticDisplacementSolver.h:106-137).
// Produces the flat arrays of include/poroel_hip.h.
#pragma once
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <map>
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

// Everything poro_desc points at, owned in one place.
struct ProblemData {
  Mesh mesh; DoFs dofs; FETables fe; BoundaryConditions bc;
  std::vector<int32_t> dirichlet_dof; std::vector<double> dirichlet_value;
  poro_material mat{}; poro_partition part{};
  poro_desc d{};

  void finalize(int k_u) {
    dofs = distribute_dofs(mesh, k_u);
    fe.build(mesh.dim, k_u);
    make_dirichlet(mesh, dofs, bc, dirichlet_dof, dirichlet_value);
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
    if (part.n_ranks == 0) { part.n_ranks = 1; part.rank = 0; }
    d.part = part;
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

}  // namespace poro_host
