// C entry points of the host layer (mesh / DoF / FE-table provider, parameter file front end and the
// run() driver), so tests and bench.py can reach the C++ host code through ctypes.
#include <cstring>
#include <string>
#include "input_data.hpp"
#include "mesh.hpp"
#include "problem.hpp"

using namespace poro_host;

namespace {
thread_local std::string g_err;
void fill_bc(BoundaryConditions &bc, int n_dir, const int32_t *dl, const int32_t *dc, const double *dv,
             int n_neu, const int32_t *nl, const int32_t *nc, const double *nv) {
  bc.dirichlet_labels.assign(dl, dl + n_dir); bc.dirichlet_components.assign(dc, dc + n_dir); bc.dirichlet_values.assign(dv, dv + n_dir);
  bc.neumann_labels.assign(nl, nl + n_neu); bc.neumann_components.assign(nc, nc + n_neu); bc.neumann_values.assign(nv, nv + n_neu);
}
}  // namespace

extern "C" {

const char *poro_host_last_error() { return g_err.c_str(); }

// structured box (or one slab of it): GridGenerator::hyper_rectangle(colorize) + uniform refinement
void *poro_host_build_box(int dim, const int32_t *n, const double *size, int k_u, int rank, int n_ranks,
                          int n_dir, const int32_t *dl, const int32_t *dc, const double *dv,
                          int n_neu, const int32_t *nl, const int32_t *nc, const double *nv, const poro_material *mat) {
  try {
    auto *P = new ProblemData();
    fill_bc(P->bc, n_dir, dl, dc, dv, n_neu, nl, nc, nv);
    P->mat = *mat;
    int nn[3] = {n[0], n[1], dim == 3 ? n[2] : 1}; double sz[3] = {size[0], size[1], dim == 3 ? size[2] : 1};
    build_box_problem(*P, dim, nn, sz, k_u, rank, n_ranks);
    return P;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

// graded box (no box tag: general kernels), see build_graded_box_problem
void *poro_host_build_graded_box(int dim, const int32_t *n, const double *size, int k_u, const double *grading,
                                 int n_dir, const int32_t *dl, const int32_t *dc, const double *dv,
                                 int n_neu, const int32_t *nl, const int32_t *nc, const double *nv, const poro_material *mat) {
  try {
    auto *P = new ProblemData();
    fill_bc(P->bc, n_dir, dl, dc, dv, n_neu, nl, nc, nv);
    P->mat = *mat;
    int nn[3] = {n[0], n[1], dim == 3 ? n[2] : 1}; double sz[3] = {size[0], size[1], dim == 3 ? size[2] : 1}, gr[3] = {grading[0], grading[1], dim == 3 ? grading[2] : 0.0};
    build_graded_box_problem(*P, dim, nn, sz, k_u, gr);
    return P;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

// box with one locally refined block of cells (hanging nodes): the mesh class one refine_mesh() pass produces (PoroelasticityFSS.h:447-498)
void *poro_host_build_refined_box(int dim, const int32_t *n, const double *size, int k_u, const int32_t *lo, const int32_t *hi,
                                  int n_dir, const int32_t *dl, const int32_t *dc, const double *dv,
                                  int n_neu, const int32_t *nl, const int32_t *nc, const double *nv, const poro_material *mat) {
  try {
    auto *P = new ProblemData();
    fill_bc(P->bc, n_dir, dl, dc, dv, n_neu, nl, nc, nv);
    P->mat = *mat;
    int nn[3] = {n[0], n[1], dim == 3 ? n[2] : 1}; double sz[3] = {size[0], size[1], dim == 3 ? size[2] : 1};
    int l3[3] = {lo[0], lo[1], dim == 3 ? lo[2] : 0}, h3[3] = {hi[0], hi[1], dim == 3 ? hi[2] : 1};
    build_refined_box_problem(*P, dim, nn, sz, k_u, l3, h3);
    return P;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

// Gmsh 2.2 mesh (GridIn::read_msh, PoroelasticityFSS.h:438-445)
void *poro_host_build_gmsh(const char *path, int k_u,
                           int n_dir, const int32_t *dl, const int32_t *dc, const double *dv,
                           int n_neu, const int32_t *nl, const int32_t *nc, const double *nv, const poro_material *mat) {
  try {
    auto *P = new ProblemData();
    fill_bc(P->bc, n_dir, dl, dc, dv, n_neu, nl, nc, nv);
    P->mat = *mat;
    P->mesh = read_gmsh22(path);
    P->finalize(k_u);
    attach_auxiliary_box(*P, k_u);     // (where the mesh fills a rectangle with colorized side ids: coarse space of the two-level preconditioner)
    return P;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
// the same mesh after `refine` uniform refinements (refine_global of read_mesh()'s grid)
void *poro_host_build_gmsh_refined(const char *path, int k_u, int refine,
                                   int n_dir, const int32_t *dl, const int32_t *dc, const double *dv,
                                   int n_neu, const int32_t *nl, const int32_t *nc, const double *nv, const poro_material *mat) {
  try {
    auto *P = new ProblemData();
    fill_bc(P->bc, n_dir, dl, dc, dv, n_neu, nl, nc, nv);
    P->mat = *mat;
    P->mesh = read_gmsh22(path);
    for (int r = 0; r < refine; ++r) P->mesh = refine_quads(P->mesh);
    P->finalize(k_u);
    attach_auxiliary_box(*P, k_u);
    return P;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}

// piece `rank` of a general n_ranks partition of a global problem (Morton cell ranges + interface lists); the global problem stays valid
void *poro_host_partition(void *global, int rank, int n_ranks) {
  try { auto *L = new ProblemData(); try { partition_problem(*static_cast<ProblemData *>(global), rank, n_ranks, *L); } catch (...) { delete L; throw; } return L; }
  catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
// local -> global dof map of a piece (space 0: displacement, 1: pressure); returns the length, copies when out != NULL
int64_t poro_host_local_to_global(void *h, int space, int32_t *out) {
  auto &v = space == 0 ? static_cast<ProblemData *>(h)->local_to_global_u : static_cast<ProblemData *>(h)->local_to_global_p;
  if (out) std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
  return (int64_t)v.size();
}

// extension: prescribed pressures on boundary labels (before the context is created)
int poro_host_set_pressure_bc(void *h, int n, const int32_t *labels, const double *values) {
  try {
    auto *P = static_cast<ProblemData *>(h);
    P->bc.pressure_labels.assign(labels, labels + n); P->bc.pressure_values.assign(values, values + n);
    P->set_pressure_bc();
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// extension: rigid frictionless plates - component `components[i]` of the displacement takes ONE (unknown) value on the boundary `labels[i]` (before the context is created)
int poro_host_tie_boundary(void *h, int n, const int32_t *labels, const int32_t *components) {
  try {
    auto *P = static_cast<ProblemData *>(h);
    if (P->part.n_ranks > 1) throw std::runtime_error("tie_boundary: implemented for one rank");
    if (P->ties_added) throw std::runtime_error("tie_boundary: already applied");
    P->bc.tie_labels.assign(labels, labels + n); P->bc.tie_components.assign(components, components + n);
    for (int i = 0; i < n; ++i) if (components[i] < 0 || components[i] >= P->mesh.dim) throw std::runtime_error("tie_boundary: component out of range");
    P->finalize(P->dofs.k_u, true);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
const poro_desc *poro_host_desc(void *h) { return &static_cast<ProblemData *>(h)->d; }
void poro_host_free(void *h) { delete static_cast<ProblemData *>(h); }

// InputDataPoroel::read_input_file flattened for ctypes
typedef struct poro_input_flat {
  int32_t dim; double domain_size[3]; int32_t initial_refinement_level, max_refinement_level;
  double youngs_modulus, poisson_ratio, biot_coef, perm, poro, visc, bulk_density, f_comp, r_well, flow_rate;
  double p_init, time_step, t_max, fss_tol, pressure_tol; int32_t max_fss_iterations, max_pressure_iterations;
  int32_t n_dirichlet, dirichlet_labels[16], dirichlet_components[16]; double dirichlet_values[16];
  int32_t n_neumann, neumann_labels[16], neumann_components[16]; double neumann_values[16];
  double lame_constant, shear_modulus, bulk_modulus, grain_bulk_modulus, n_modulus, m_modulus;
  poro_material material;
} poro_input_flat;

int poro_host_read_input(const char *path /* NULL = declared defaults */, poro_input_flat *o) {
  try {
    input_data::InputDataPoroel in;
    if (path) in.read_input_file(path);
    std::memset(o, 0, sizeof(*o));
    o->dim = in.dim;
    for (size_t i = 0; i < in.domain_size.size() && i < 3; ++i) o->domain_size[i] = in.domain_size[i];
    o->initial_refinement_level = in.initial_refinement_level; o->max_refinement_level = in.max_refinement_level;
    o->youngs_modulus = in.youngs_modulus; o->poisson_ratio = in.poisson_ratio; o->biot_coef = in.biot_coef; o->perm = in.perm; o->poro = in.poro;
    o->visc = in.visc; o->bulk_density = in.bulk_density; o->f_comp = in.f_comp; o->r_well = in.r_well; o->flow_rate = in.flow_rate;
    o->p_init = in.p_init; o->time_step = in.time_step; o->t_max = in.t_max; o->fss_tol = in.fss_tol; o->pressure_tol = in.pressure_tol;
    o->max_fss_iterations = in.max_fss_iterations; o->max_pressure_iterations = in.max_pressure_iterations;
    if (in.displacement_boundary_labels.size() > 16 || in.stress_boundary_labels.size() > 16) throw std::runtime_error("more than 16 boundary conditions");
    if (in.displacement_boundary_labels.size() != in.displacement_boundary_components.size() || in.displacement_boundary_labels.size() != in.displacement_boundary_values.size() ||
        in.stress_boundary_labels.size() != in.stress_boundary_components.size() || in.stress_boundary_labels.size() != in.stress_boundary_values.size())
      throw std::runtime_error("boundary label / component / value lists differ in length");
    o->n_dirichlet = (int32_t)in.displacement_boundary_labels.size();
    for (int i = 0; i < o->n_dirichlet; ++i) { o->dirichlet_labels[i] = in.displacement_boundary_labels[i]; o->dirichlet_components[i] = in.displacement_boundary_components[i]; o->dirichlet_values[i] = in.displacement_boundary_values[i]; }
    o->n_neumann = (int32_t)in.stress_boundary_labels.size();
    for (int i = 0; i < o->n_neumann; ++i) { o->neumann_labels[i] = in.stress_boundary_labels[i]; o->neumann_components[i] = in.stress_boundary_components[i]; o->neumann_values[i] = in.stress_boundary_values[i]; }
    o->lame_constant = in.lame_constant; o->shear_modulus = in.shear_modulus; o->bulk_modulus = in.bulk_modulus;
    o->grain_bulk_modulus = in.grain_bulk_modulus; o->n_modulus = in.n_modulus; o->m_modulus = in.m_modulus;
    o->material = in.material();
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// PoroElasticProblem<dim>::run() on the HIP back end (problem.hpp).  trace rows: see PoroElasticProblem::time_step.
int poro_host_run(void *problem_data, int device, int operator_mode, double p_init, double dt, int n_steps,
                  double fss_tol, double pressure_tol, int max_fss, int max_pres,
                  double abs_u, double rel_u, int max_it, int preconditioner, int flags /* bit 0: coupled_fss, bit 1: incremental_strain, bit 2: PORO_STOP_REDUCTION for the displacement solve, bit 3: Jacobi for the pressure / projection solves (default: the strongest form the mesh supports), bit 4: PORO_PREC_TWO_LEVEL for them, bits 8-15: Chebyshev degree, bits 16-30: Chebyshev interval ratio (0 = defaults) */, double *trace, int max_rows, poro_ctx **ctx_out) {
  try {
    auto *P = static_cast<ProblemData *>(problem_data);
    RunControls rc; rc.p_init = p_init; rc.time_step = dt; rc.n_steps = n_steps; rc.fss_tol = fss_tol; rc.pressure_tol = pressure_tol;
    rc.max_fss_iterations = max_fss; rc.max_pressure_iterations = max_pres; rc.abs_tol_u = abs_u; rc.rel_tol_u = rel_u; rc.max_iter = max_it; rc.preconditioner = preconditioner; rc.coupled_fss = (flags & 1) != 0; rc.incremental_strain = (flags & 2) != 0; rc.stop_rule_u = (flags & 4) ? PORO_STOP_REDUCTION : PORO_STOP_RHS; if (flags & 8) rc.preconditioner_p = PORO_PREC_JACOBI; if (flags & 16) rc.preconditioner_p = PORO_PREC_TWO_LEVEL; rc.chebyshev_degree = (flags >> 8) & 0xff; rc.chebyshev_ratio = (double)((flags >> 16) & 0x7fff);
    int rows = 0;
    if (P->mesh.dim == 2) { PoroElasticProblem<2> prob(*P, device, operator_mode); rows = prob.run(rc, trace, max_rows); if (ctx_out) *ctx_out = prob.release(); }
    else { PoroElasticProblem<3> prob(*P, device, operator_mode); rows = prob.run(rc, trace, max_rows); if (ctx_out) *ctx_out = prob.release(); }
    return rows;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// Steppable driver for bench.py: the same PoroElasticProblem object, one time step per call.
struct HostRunner {
  int dim; RunControls rc; PoroElasticProblem<2> *p2 = nullptr; PoroElasticProblem<3> *p3 = nullptr;
  ~HostRunner() { delete p2; delete p3; }
};
void *poro_host_runner_create(void *problem_data, int device, int operator_mode, double p_init, double dt, double fss_tol, double pressure_tol,
                              int max_fss, int max_pres, double abs_u, double rel_u, int max_it, int preconditioner, int flags /* bit 0: coupled_fss, bit 1: incremental_strain, bit 2: PORO_STOP_REDUCTION for the displacement solve, bit 3: Jacobi for the pressure / projection solves (default: the strongest form the mesh supports), bit 4: PORO_PREC_TWO_LEVEL for them, bits 8-15: Chebyshev degree, bits 16-30: Chebyshev interval ratio (0 = defaults) */) {
  try {
    auto *P = static_cast<ProblemData *>(problem_data);
    auto *R = new HostRunner(); R->dim = P->mesh.dim;
    R->rc.p_init = p_init; R->rc.time_step = dt; R->rc.fss_tol = fss_tol; R->rc.pressure_tol = pressure_tol; R->rc.max_fss_iterations = max_fss;
    R->rc.max_pressure_iterations = max_pres; R->rc.abs_tol_u = abs_u; R->rc.rel_tol_u = rel_u; R->rc.max_iter = max_it; R->rc.preconditioner = preconditioner; R->rc.coupled_fss = (flags & 1) != 0; R->rc.incremental_strain = (flags & 2) != 0; R->rc.stop_rule_u = (flags & 4) ? PORO_STOP_REDUCTION : PORO_STOP_RHS; if (flags & 8) R->rc.preconditioner_p = PORO_PREC_JACOBI; if (flags & 16) R->rc.preconditioner_p = PORO_PREC_TWO_LEVEL; R->rc.chebyshev_degree = (flags >> 8) & 0xff; R->rc.chebyshev_ratio = (double)((flags >> 16) & 0x7fff);
    if (R->dim == 2) R->p2 = new PoroElasticProblem<2>(*P, device, operator_mode); else R->p3 = new PoroElasticProblem<3>(*P, device, operator_mode);
    return R;
  } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
poro_ctx *poro_host_runner_ctx(void *r) { auto *R = static_cast<HostRunner *>(r); return R->dim == 2 ? R->p2->context() : R->p3->context(); }
int poro_host_runner_initialize(void *r) {
  try { auto *R = static_cast<HostRunner *>(r); if (R->dim == 2) R->p2->initialize(R->rc); else R->p3->initialize(R->rc); return 0; }
  catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// one time step; trace rows as poro_host_run; work[11] = apply_u, apply_p, asm_rhs_u, asm_matrix_u, residual_p, jacobian_p, proj_rhs, cg_u, cg_p, cg_proj, seconds_solve_u*1e6
int poro_host_runner_step(void *r, double *trace, int max_rows, int64_t *work) {
  try {
    auto *R = static_cast<HostRunner *>(r);
    auto fill = [&](auto &w) { work[0] = w.apply_u; work[1] = w.apply_p; work[2] = w.asm_rhs_u; work[3] = w.asm_matrix_u; work[4] = w.residual_p; work[5] = w.jacobian_p;
                               work[6] = w.proj_rhs; work[7] = w.cg_u; work[8] = w.cg_p; work[9] = w.cg_proj; work[10] = (int64_t)(w.seconds_solve_u * 1e6); };
    int rows;
    if (R->dim == 2) { rows = R->p2->time_step(R->rc, trace, max_rows); if (work) fill(R->p2->work); }
    else { rows = R->p3->time_step(R->rc, trace, max_rows); if (work) fill(R->p3->work); }
    return rows;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// roll back to the snapshot, then one time step, in ONE call (a benchmark that repeats a step pays one host-language round trip instead of two)
int poro_host_runner_restore_and_step(void *r, double *trace, int max_rows, int64_t *work) {
  try { auto *R = static_cast<HostRunner *>(r); if (R->dim == 2) R->p2->restore_state(); else R->p3->restore_state(); }
  catch (const std::exception &e) { g_err = e.what(); return -1; }
  return poro_host_runner_step(r, trace, max_rows, work);
}
// action 0: snapshot the device state, 1: roll back to it
int poro_host_runner_state(void *r, int action) {
  try {
    auto *R = static_cast<HostRunner *>(r);
    if (R->dim == 2) { if (action) R->p2->restore_state(); else R->p2->save_state(); }
    else { if (action) R->p3->restore_state(); else R->p3->save_state(); }
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// tail of the time loop body (PoroelasticityFSS.h:409-411): shear strains, effective stresses and, with a directory, solution-NNNN.vtk
int poro_host_runner_postprocess(void *r, const char *output_dir, int corrected) {
  try {
    auto *R = static_cast<HostRunner *>(r);
    RunControls rc = R->rc; rc.output_dir = output_dir ? output_dir : ""; rc.corrected_postprocessing = corrected != 0;
    if (R->dim == 2) R->p2->postprocess(rc); else R->p3->postprocess(rc);
    return 0;
  } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// cumulative work counters since creation (same layout as poro_host_runner_step's `work`)
void poro_host_runner_work(void *r, int64_t *work) {
  auto *R = static_cast<HostRunner *>(r);
  auto fill = [&](auto &w) { work[0] = w.apply_u; work[1] = w.apply_p; work[2] = w.asm_rhs_u; work[3] = w.asm_matrix_u; work[4] = w.residual_p; work[5] = w.jacobian_p;
                             work[6] = w.proj_rhs; work[7] = w.cg_u; work[8] = w.cg_p; work[9] = w.cg_proj; work[10] = (int64_t)(w.seconds_solve_u * 1e6); };
  if (R->dim == 2) fill(R->p2->work); else fill(R->p3->work);
}
void poro_host_runner_free(void *r) { delete static_cast<HostRunner *>(r); }

}  // extern "C"
