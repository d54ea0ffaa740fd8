// Host-side mirror of the reference's operator interface for the hot path: the same class and member
// names, argument meaning and error behaviour as lib/include/PoroElasticDisplacementSolver.h,
// PoroElasticPressureSolver.h, StrainProjector.h and the FSS loop of PoroelasticityFSS.h:294-415,
// forwarding to the C-ABI of include/poroel_hip.h.  deal.II Vector<double> members become handles to
// device-resident vectors; nothing here computes on the CPU.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>
#include <fstream>
#include <cstdio>
#include "../../include/poroel_hip.h"
#include "mesh.hpp"

namespace poro_host {

// analogue of dealii::SolverControl::NoConvergence (thrown by cg.solve, e.g. PoroElasticDisplacementSolver.h:305)
struct NoConvergence : std::runtime_error {
  int last_step; double last_residual;
  NoConvergence(const std::string &who, int it, double r)
      : std::runtime_error(who + ": iterative method did not converge in " + std::to_string(it) + " steps, residual " + std::to_string(r)),
        last_step(it), last_residual(r) {}
};

inline void check(int rc, const char *what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + poro_last_error());
}

// stand-in for dealii::Vector<double> members that now live in HBM
struct DeviceVector {
  poro_ctx *ctx = nullptr; int id = -1;
  DeviceVector &operator=(double v) { check(poro_vec_fill(ctx, id, v), "vec_fill"); return *this; }
  DeviceVector &operator=(const DeviceVector &o) { if (o.id != id || o.ctx != ctx) check(poro_vec_copy(ctx, id, o.id), "vec_copy"); return *this; }
  DeviceVector &operator+=(const DeviceVector &o) { check(poro_vec_axpy(ctx, id, 1.0, o.id), "vec_axpy"); return *this; }
  double l2_norm() const { double a, b; check(poro_vec_norm(ctx, id, &a, &b), "vec_norm"); return a; }
  double linfty_norm() const { double a, b; check(poro_vec_norm(ctx, id, &a, &b), "vec_norm"); return b; }
  void get(std::vector<double> &h, int64_t n) const { h.resize(n); check(poro_vec_get(ctx, id, h.data(), n), "vec_get"); }
};

struct RunControls {
  double p_init = 10e6, time_step = 60; int n_steps = 1;
  double fss_tol = 1e-8, pressure_tol = 1e-8; int max_fss_iterations = 50, max_pressure_iterations = 50;
  double abs_tol_u = 1e-12, rel_tol_u = 0.0; int max_iter = 1000;   // PoroElasticDisplacementSolver.h:298-299
  int stop_rule_u = PORO_STOP_RHS;                                  // PORO_STOP_REDUCTION: rel_tol_u is taken against the residual of the warm start (every step of a transient solves)
  int preconditioner = PORO_PREC_JACOBI;                            // displacement solve; PORO_PREC_SSOR = the reference's PreconditionSSOR (CSR operator, one rank) for all three systems
  bool coupled_fss = false;                                         // true = restore the get_volumetric_strain() call the reference commented out (:399): eps_v follows the new
                                                                    // displacement inside the fixed-stress loop, which then really iterates (SURVEY 8f-4 "corrected physics", first half)
  bool incremental_strain = false;                                  // true = storage term against the previous step, alpha (eps_v - eps_v^n) / dt, instead of the initial state eps_v0 (:317, :361-363)
  bool corrected_postprocessing = false;                            // false = the reference's output (shear RHS never assembled, 2D "sigma_yy" shows sigma_xx); true = both fixed
  std::string output_dir;                                           // "" = no files; the reference always writes ./solution/solution-NNNN.vtk (:285-290)
  int chebyshev_degree = 0; double chebyshev_ratio = 0.0;           // PORO_PREC_CHEBYSHEV: 0 = the library's defaults
  int preconditioner_p = -1;                                        // pressure / projection solves; -1 = fast diagonalisation where the context supports it, else the two-level form, else Jacobi
};

}  // namespace poro_host

namespace solvers {
using poro_host::DeviceVector;

template <int dim> class PoroElasticDisplacementSolver {
 public:
  DeviceVector solution;                                   // PoroElasticDisplacementSolver.h:47
  poro_solver_opts control{1e-12, 0.0, 1000, PORO_PREC_JACOBI, 1.2, PORO_STOP_RHS, 0};   // :298-299, omega :303 (Jacobi is the fast default; PORO_PREC_SSOR = the reference's)
  poro_solve_info  last{};
  explicit PoroElasticDisplacementSolver(poro_ctx *c) : ctx(c) { solution.ctx = c; solution.id = PORO_VEC_U; }
  void setup_dofs() { rebuild_system_matrix = true; }      // :106-153 (pattern / constraints are built by poro_ctx_create)
  // :155-291 — pressure_solution must be the pressure solver's `solution`
  void assemble_system(DeviceVector &pressure_solution) {
    if (pressure_solution.id != PORO_VEC_P) throw std::invalid_argument("assemble_system expects pressure_solver.solution");
    poro_host::check(poro_disp_assemble_system(ctx, rebuild_system_matrix ? 1 : 0), "disp_assemble_system");
    rebuild_system_matrix = false;                         // :290
  }
  void solve() {                                           // :294-307
    const int rc = poro_disp_solve(ctx, &control, &last);
    poro_host::check(rc, "disp_solve");
    if (rc > 0) throw poro_host::NoConvergence("PoroElasticDisplacementSolver::solve", last.iterations, last.final_residual);
  }
 private:
  poro_ctx *ctx; bool rebuild_system_matrix = true;        // :55
};

template <int dim> class PoroElasticPressureSolver {
 public:
  DeviceVector solution, solution_update, old_solution, residual;   // PoroElasticPressureSolver.h:38-40
  poro_solver_opts control{0.0, 1e-8, 1000, PORO_PREC_JACOBI, 1.0, PORO_STOP_RHS, 0};       // :175, omega :178
  poro_solve_info  last{};
  explicit PoroElasticPressureSolver(poro_ctx *c) : ctx(c) {
    solution.ctx = solution_update.ctx = old_solution.ctx = residual.ctx = c;
    solution.id = PORO_VEC_P; solution_update.id = PORO_VEC_DP; old_solution.id = PORO_VEC_P_OLD; residual.id = PORO_VEC_RESIDUAL_P;
  }
  void setup_dofs() {}                                     // :68-111 (mass / Laplace matrices are built by poro_ctx_create)
  void assemble_jacobian(double time_step) { poro_host::check(poro_pres_assemble_jacobian(ctx, time_step), "pres_assemble_jacobian"); }   // :158-169
  // :113-155 — the strains must be the problem's volumetric_strain / initial_volumetric_strain
  void assemble_residual(double time_step, DeviceVector &volumetric_strain, DeviceVector &initial_volumetric_strain) {
    if (volumetric_strain.id != PORO_VEC_EPSV || initial_volumetric_strain.id != PORO_VEC_EPSV0) throw std::invalid_argument("assemble_residual expects the problem's strain vectors");
    poro_host::check(poro_pres_assemble_residual(ctx, time_step, &residual_l2), "pres_assemble_residual");
  }
  void update_volumetric_strain(DeviceVector &volumetric_strain) {   // :187-194
    if (volumetric_strain.id != PORO_VEC_EPSV) throw std::invalid_argument("update_volumetric_strain expects the problem's volumetric_strain");
    poro_host::check(poro_pres_update_volumetric_strain(ctx), "pres_update_volumetric_strain");
  }
  void solve() {                                           // :172-185
    const int rc = poro_pres_solve(ctx, &control, &last);
    poro_host::check(rc, "pres_solve");
    if (rc > 0) throw poro_host::NoConvergence("PoroElasticPressureSolver::solve", last.iterations, last.final_residual);
  }
  double residual_l2 = 0;   // residual.l2_norm() of the last assemble_residual, computed on device in the same pass
 private:
  poro_ctx *ctx;
};
}  // namespace solvers

namespace projection {
template <int dim> class StrainProjector {
 public:
  poro_solver_opts control{0.0, 1e-8, 1000, PORO_PREC_JACOBI, 1.0, PORO_STOP_RHS, 0};       // StrainProjector.h:209, omega :212
  poro_solve_info  last{};
  StrainProjector() {}
  void set_solvers(poro_ctx *c) { ctx = c; }               // :73-79
  void setup_dofs() {}                                     // :82-98
  void assemble_projection_matrix() { poro_host::check(poro_proj_assemble_matrix(ctx), "proj_assemble_matrix"); }   // :101-106
  void assemble_projection_rhs(std::vector<int> tensor_components) {   // :109-198
    std::vector<int32_t> tc(tensor_components.begin(), tensor_components.end());
    poro_host::check(poro_proj_assemble_rhs(ctx, tc.data(), (int32_t)tc.size()), "proj_assemble_rhs");
  }
  void solve_projection_systems(const std::vector<int32_t> &rhs_entries, std::vector<poro_solve_info> &infos) {
    infos.assign(rhs_entries.size(), poro_solve_info{});
    const int rc = poro_proj_solve_many(ctx, rhs_entries.data(), (int32_t)rhs_entries.size(), &control, infos.data());
    poro_host::check(rc, "proj_solve_many");
    if (!infos.empty()) last = infos.back();
    if (rc > 0) throw poro_host::NoConvergence("StrainProjector::solve_projection_systems", last.iterations, last.final_residual);
  }
  void solve_projection_system(int rhs_entry) {            // :201-232
    const int rc = poro_proj_solve(ctx, rhs_entry, &control, &last);
    poro_host::check(rc, "proj_solve");
    if (rc > 0) throw poro_host::NoConvergence("StrainProjector::solve_projection_system", last.iterations, last.final_residual);
  }
 private:
  poro_ctx *ctx = nullptr;
};
}  // namespace projection

namespace indexing {
// TensorIndexer.h:6-52
template <int dim> class TensorIndexer {
 public:
  int entryIndex(int tensor_index) const {
    static const int m2[4] = {0, 1, 1, 2}, m3[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    return dim == 2 ? m2[tensor_index] : m3[tensor_index];
  }
};
}  // namespace indexing

namespace poro_host {

// PoroElasticProblem<dim> (PoroelasticityFSS.h:42-90): owns the context and reproduces run()'s call sequence
// (:294-415) with mesh creation, AMR (:333-340) and output (:409-411) removed.
template <int dim> class PoroElasticProblem {
  poro_ctx *ctx;   // declared first: the solver members below are constructed from it
 public:
  PoroElasticProblem(ProblemData &P, int device, int operator_mode) : ctx(make_ctx(P, device, operator_mode)), pressure_solver(ctx), displacement_solver(ctx), pd(&P) {
    volumetric_strain.ctx = initial_volumetric_strain.ctx = ctx;
    volumetric_strain.id = PORO_VEC_EPSV; initial_volumetric_strain.id = PORO_VEC_EPSV0;
    if (dim == 2) { strain_tensor_volumetric_components = {0, 3}; strain_tensor_shear_components = {1}; }
    else { strain_tensor_volumetric_components = {0, 4, 8}; strain_tensor_shear_components = {1, 2, 5}; }   // :104-111
  }
  ~PoroElasticProblem() { if (ctx) poro_ctx_destroy(ctx); }
  poro_ctx *context() { return ctx; }
  poro_ctx *release() { poro_ctx *c = ctx; ctx = nullptr; return c; }

  void setup_dofs() {                                      // :131-151
    pressure_solver.setup_dofs(); displacement_solver.setup_dofs();
    strain_projector.set_solvers(ctx); strain_projector.setup_dofs();
    // every reinit() of the reference's setup_dofs leaves a zero vector (PoroElasticDisplacementSolver.h:150-151,
    // PoroElasticPressureSolver.h:103-108, StrainProjector.h:93-96, PoroelasticityFSS.h:145-146)
    const int n_sym = dim * (dim + 1) / 2;
    for (int id : {PORO_VEC_U, PORO_VEC_RHS_U, PORO_VEC_P, PORO_VEC_P_OLD, PORO_VEC_DP, PORO_VEC_RESIDUAL_P, PORO_VEC_EPSV, PORO_VEC_EPSV0})
      check(poro_vec_fill(ctx, id, 0.0), "vec_fill");
    for (int e = 0; e < n_sym; ++e) { check(poro_vec_fill(ctx, PORO_VEC_STRAIN0 + e, 0.0), "vec_fill"); check(poro_vec_fill(ctx, PORO_VEC_PROJ_RHS0 + e, 0.0), "vec_fill"); }
  }
  void get_normal_strain_components() {                    // :153-164
    strain_projector.assemble_projection_rhs(strain_tensor_volumetric_components);
    // the loop of :157-163 over the volumetric components in one library call (solved together where the library can, entry by entry otherwise)
    std::vector<int32_t> entries; for (const auto &comp : strain_tensor_volumetric_components) entries.push_back(tensor_indexer.entryIndex(comp));
    std::vector<poro_solve_info> infos(entries.size());
    strain_projector.solve_projection_systems(entries, infos);
    for (const auto &li : infos) { work.cg_proj += li.iterations; work.apply_p += li.operator_applications; }
  }
  void get_volumetric_strain() { check(poro_get_volumetric_strain(ctx), "get_volumetric_strain"); }   // :179-186
  // :167-176.  The reference never assembles the shear right-hand sides (assemble_projection_rhs is only called with the volumetric
  // components, :157), so these solves see rhs = 0 and return eps_ij = 0; `corrected` assembles them first.
  void get_shear_strain_components(bool corrected = false) {
    if (corrected) strain_projector.assemble_projection_rhs(strain_tensor_shear_components);
    for (const auto &comp : strain_tensor_shear_components) strain_projector.solve_projection_system(tensor_indexer.entryIndex(comp));
  }
  void get_effective_stresses() { check(poro_get_effective_stresses(ctx), "get_effective_stresses"); }   // :189-224
  // :227-291: legacy-VTK file of u, p, strains and stresses, one patch per cell with its 2^dim vertices (build_patches(min degree) = 1
  // subdivision), fields in the reference's order.  2D quirk: the reference writes stresses[0] under the name "sigma_yy" (:257-258).
  void output_results(unsigned int step, const std::string &dir, bool corrected = false) {
    const ProblemData &P = *pd; const int nv = 1 << dim, k = P.d.degree_u, n1 = k + 1, ns = P.d.fe.ns_u;
    const int64_t nc = P.d.n_cells, npts = nc * nv;
    std::vector<double> u, p; displacement_solver.solution.get(u, P.d.n_dofs_u); pressure_solver.solution.get(p, P.d.n_dofs_p);
    const int n_sym = dim * (dim + 1) / 2;
    std::vector<std::vector<double>> eps(n_sym), sig(n_sym);
    for (int e = 0; e < n_sym; ++e) { eps[e].resize(P.d.n_dofs_p); sig[e].resize(P.d.n_dofs_p);
      check(poro_vec_get(ctx, PORO_VEC_STRAIN0 + e, eps[e].data(), P.d.n_dofs_p), "vec_get"); check(poro_vec_get(ctx, PORO_VEC_STRESS0 + e, sig[e].data(), P.d.n_dofs_p), "vec_get"); }
    char name[64]; std::snprintf(name, sizeof name, "/solution-%04u", step);
    std::string file = dir + name + (P.part.n_ranks > 1 ? ".r" + std::to_string(P.part.rank) : std::string()) + ".vtk";
    std::ofstream out(file);
    if (!out) throw std::runtime_error("cannot write " + file);
    out.precision(12);
    out << "# vtk DataFile Version 3.0\n#This file was generated by poroelasticity_dealii_amd (layout of deal.II DataOut::write_vtk)\nASCII\nDATASET UNSTRUCTURED_GRID\n\n";
    out << "POINTS " << npts << " double\n";
    for (int64_t c = 0; c < nc; ++c) for (int v = 0; v < nv; ++v) {
      const double *x = &P.mesh.vertices[(size_t)P.mesh.cells[c * nv + v] * dim];
      out << x[0] << ' ' << x[1] << ' ' << (dim == 3 ? x[2] : 0.0) << '\n';
    }
    static const int vtk2[4] = {0, 1, 3, 2}, vtk3[8] = {0, 1, 3, 2, 4, 5, 7, 6};   // lexicographic -> VTK quad / hexahedron order
    out << "\nCELLS " << nc << ' ' << nc * (nv + 1) << '\n';
    for (int64_t c = 0; c < nc; ++c) { out << nv; for (int v = 0; v < nv; ++v) out << '\t' << c * nv + (dim == 2 ? vtk2[v] : vtk3[v]); out << '\n'; }
    out << "\nCELL_TYPES " << nc << '\n';
    for (int64_t c = 0; c < nc; ++c) out << (dim == 2 ? 9 : 12) << (c + 1 < nc ? ' ' : '\n');
    out << "POINT_DATA " << npts << '\n' << "VECTORS u double\n";
    for (int64_t c = 0; c < nc; ++c) for (int v = 0; v < nv; ++v) {
      const int a = (v & 1) * k, b = ((v >> 1) & 1) * k, cc = (v >> 2) * k, sidx = a + n1 * (b + n1 * cc);
      double w[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) w[d] = u[P.dofs.cell_u[((size_t)c * ns + sidx) * dim + d]];
      out << w[0] << ' ' << w[1] << ' ' << w[2] << '\n';
    }
    auto scalar = [&](const char *nm, const std::vector<double> &f) {
      out << "SCALARS " << nm << " double 1\nLOOKUP_TABLE default\n";
      for (int64_t c = 0; c < nc; ++c) { for (int v = 0; v < nv; ++v) out << f[P.dofs.cell_p[c * nv + v]] << ' '; }
      out << '\n';
    };
    scalar("p", p); scalar("eps_xx", eps[0]); scalar("sigma_xx", sig[0]);
    if (dim == 2) {
      scalar("eps_xy", eps[1]); scalar("eps_yy", eps[2]); scalar("sigma_xy", sig[1]); scalar("sigma_yy", corrected ? sig[2] : sig[0]);
    } else {
      scalar("eps_xy", eps[1]); scalar("eps_xz", eps[2]); scalar("eps_yy", eps[3]); scalar("eps_yz", eps[4]); scalar("eps_zz", eps[5]);
      scalar("sigma_xy", sig[1]); scalar("sigma_xz", sig[2]); scalar("sigma_yy", sig[3]); scalar("sigma_yz", sig[4]); scalar("sigma_zz", sig[5]);
    }
  }
  // the tail of the time loop body (:409-411)
  void postprocess(const RunControls &rc) {
    get_shear_strain_components(rc.corrected_postprocessing);
    get_effective_stresses();
    if (!rc.output_dir.empty()) output_results((unsigned)time_step_number, rc.output_dir, rc.corrected_postprocessing);
  }

  // work counters of the metric "DoF-updates in assemble + SpMV" (SURVEY 8d): operator applications and assembly passes
  struct Work { int64_t apply_u = 0, apply_p = 0, asm_rhs_u = 0, asm_matrix_u = 0, residual_p = 0, jacobian_p = 0, proj_rhs = 0;
                int64_t cg_u = 0, cg_p = 0, cg_proj = 0; double seconds_solve_u = 0; } work;

  // trace rows: [step, fss_iteration, pressure_iterations, inner pressure error, |p|_inf, error after displacement, u CG its, p CG its]
  void initialize(const RunControls &rc) {
    displacement_solver.control.abs_tol = rc.abs_tol_u; displacement_solver.control.rel_tol = rc.rel_tol_u; displacement_solver.control.stop_rule = rc.stop_rule_u;
    displacement_solver.control.max_iter = pressure_solver.control.max_iter = strain_projector.control.max_iter = rc.max_iter;
    // rc.preconditioner < 0: the strongest displacement preconditioner this mesh supports
    displacement_solver.control.preconditioner = rc.preconditioner >= 0 ? rc.preconditioner
        : poro_supports_preconditioner(context(), 0, PORO_PREC_FDM) ? PORO_PREC_FDM : poro_supports_preconditioner(context(), 0, PORO_PREC_TWO_LEVEL) ? PORO_PREC_TWO_LEVEL : PORO_PREC_CHEBYSHEV;
    if (displacement_solver.control.preconditioner == PORO_PREC_CHEBYSHEV) { displacement_solver.control.omega = rc.chebyshev_ratio; displacement_solver.control.poly_degree = rc.chebyshev_degree; }
    pressure_solver.control.preconditioner = strain_projector.control.preconditioner =
        rc.preconditioner == PORO_PREC_SSOR ? PORO_PREC_SSOR : rc.preconditioner_p >= 0 ? rc.preconditioner_p
        : poro_supports_preconditioner(context(), 1, PORO_PREC_FDM) ? PORO_PREC_FDM : PORO_PREC_JACOBI;
    // the pressure Jacobian's stiffness part makes Jacobi-CG grow with 1/h: the two-level form where the mesh carries a coarse space.  (The projection's mass matrix
    // is well conditioned under Jacobi on any mesh: 12-15 iterations, fewer than the additive two-level form needs; below ~4k pressure dofs a CG iteration is
    // launch-bound and the two-level form's extra launches cost more than the iterations it saves - profiles/r03_refined_box_step.json.)
    if (rc.preconditioner != PORO_PREC_SSOR && rc.preconditioner_p < 0 && pressure_solver.control.preconditioner == PORO_PREC_JACOBI && pd->d.n_dofs_p >= 4096 && poro_supports_preconditioner(context(), 1, PORO_PREC_TWO_LEVEL))
      pressure_solver.control.preconditioner = PORO_PREC_TWO_LEVEL;
    setup_dofs();                                          // :308
    pressure_solver.solution = rc.p_init;                  // :311
    check(poro_pres_apply_boundary_values(ctx), "pres_apply_boundary_values");   // (extension: prescribed pressures; no-op for the reference's problems)
    assemble_displacement();                               // :312
    solve_displacement();                                  // :313
    strain_projector.assemble_projection_matrix();         // :314
    normal_strains();                                      // :315
    get_volumetric_strain();                               // :316
    initial_volumetric_strain = volumetric_strain;         // :317
    time_step_number = 0;
  }
  // one pass of the time loop body (:328-407); returns the number of trace rows written (= FSS iterations)
  int time_step(const RunControls &rc, double *trace, int max_rows) {
    int rows = 0;
    time_step_number++;                                    // :329
    pressure_solver.old_solution = pressure_solver.solution;   // :342
    if (rc.incremental_strain && time_step_number > 1) initial_volumetric_strain = volumetric_strain;
    double pressure_error = rc.pressure_tol * 2; int fss_iteration = 0;   // :345-346
    while (fss_iteration < rc.max_fss_iterations && pressure_error > rc.fss_tol) {   // :347-348
      fss_iteration++;
      int pressure_iteration = 0, pcg = 0; double inner = 0;
      pressure_solver.solution_update = 0.0;               // :356
      while (pressure_iteration < rc.max_pressure_iterations) {   // :358
        pressure_iteration++;
        pressure_solver.update_volumetric_strain(volumetric_strain);   // :360
        pressure_solver.assemble_residual(rc.time_step, volumetric_strain, initial_volumetric_strain); work.residual_p++;   // :361-363
        pressure_error = pressure_solver.residual_l2;      // :364
        inner = pressure_error;
        if (pressure_error < rc.pressure_tol) break;       // :366-371
        pressure_solver.assemble_jacobian(rc.time_step);                       // :377
        if (rc.time_step != jacobian_dt) { work.jacobian_p++; jacobian_dt = rc.time_step; }   // the library re-forms J only when dt changed
        pressure_solver.solve();                           // :378
        pcg += pressure_solver.last.iterations; work.cg_p += pressure_solver.last.iterations; work.apply_p += pressure_solver.last.operator_applications;
        pressure_solver.solution += pressure_solver.solution_update;   // :379
      }
      const double pinf = pressure_solver.solution.linfty_norm();   // :387-389
      assemble_displacement();                             // :395
      solve_displacement();                                // :396
      normal_strains();                                    // :398  (get_volumetric_strain() stays commented out, :399)
      if (rc.coupled_fss) get_volumetric_strain();
      pressure_solver.assemble_residual(rc.time_step, volumetric_strain, initial_volumetric_strain); work.residual_p++;   // :402-404
      pressure_error = pressure_solver.residual_l2;        // :405
      if (rows < max_rows) { double *r = trace + 8 * rows++; r[0] = time_step_number; r[1] = fss_iteration; r[2] = pressure_iteration - 1; r[3] = inner; r[4] = pinf; r[5] = pressure_error; r[6] = displacement_solver.last.iterations; r[7] = pcg; }
    }
    return rows;
  }
  // device-side snapshot / rollback of the whole solver state (retry or repeat a time step)
  void save_state() { check(poro_state_save(ctx), "state_save"); saved_step_number = time_step_number; }
  void restore_state() { check(poro_state_restore(ctx), "state_restore"); time_step_number = saved_step_number; }
  int run(const RunControls &rc, double *trace, int max_rows) {
    int rows = 0;
    initialize(rc);
    if (rows < max_rows) { double *r = trace + 8 * rows++; for (int i = 0; i < 8; ++i) r[i] = 0; r[6] = displacement_solver.last.iterations; }
    for (int s = 1; s <= rc.n_steps; ++s) { rows += time_step(rc, trace + 8 * rows, max_rows - rows); if (!rc.output_dir.empty()) postprocess(rc); }   // :327, :409-411
    return rows;
  }

  solvers::PoroElasticPressureSolver<dim>     pressure_solver;     // :77
  solvers::PoroElasticDisplacementSolver<dim> displacement_solver; // :78
  projection::StrainProjector<dim>            strain_projector;    // :79
  indexing::TensorIndexer<dim>                tensor_indexer;      // :81
  DeviceVector volumetric_strain, initial_volumetric_strain;       // :83
  std::vector<int> strain_tensor_volumetric_components, strain_tensor_shear_components;   // :86-87

 private:
  void assemble_displacement() { const bool rebuild = first_assembly; displacement_solver.assemble_system(pressure_solver.solution); first_assembly = false; work.asm_rhs_u++; if (rebuild) work.asm_matrix_u++; }
  void solve_displacement() {
    displacement_solver.solve();
    work.cg_u += displacement_solver.last.iterations; work.apply_u += displacement_solver.last.operator_applications; work.seconds_solve_u += displacement_solver.last.seconds;
  }
  void normal_strains() {
    get_normal_strain_components(); work.proj_rhs++;
  }
  bool first_assembly = true; int time_step_number = 0, saved_step_number = 0; double jacobian_dt = -1;
  const ProblemData *pd;
  static poro_ctx *make_ctx(ProblemData &P, int device, int operator_mode) {
    poro_ctx *c = nullptr;
    check(poro_ctx_create(&P.d, device, operator_mode, &c), "poro_ctx_create");
    return c;
  }
};

}  // namespace poro_host
