// Driver executable: `poro_run input.data [--mesh domain.msh] [--degree 1|2] [--matrix-free] [--ssor | --chebyshev | --block-fdm] [--steps N] [--output DIR] [--corrected-output] [--coupled-fss] [--incremental-strain]`.
// Stands in for the reference's missing code/source/Runner.cpp (code/CMakeLists.txt:8): argv[1] is the
// parameter file (parse_command_line.h:5-27); the mesh is create_mesh()'s colorized box refined
// `Initial refinement level` times (PoroelasticityFSS.h:418-435) unless --mesh names a Gmsh file
// (read_mesh, :438-445).  --output DIR writes DIR/solution-NNNN.vtk after every step like output_results (:227-291); AMR is out of scope.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include "input_data.hpp"
#include "mesh.hpp"
#include "problem.hpp"

using namespace poro_host;

int main(int argc, char **argv) {
  if (argc < 2) { std::cerr << "specify the file name" << std::endl; return 1; }   // parse_command_line.h:9-13
  std::string mesh_file; int degree = 2, op = PORO_OP_CSR, steps = -1, device = 0, prec = PORO_PREC_JACOBI; std::string output_dir; bool corrected = false, coupled = false, incremental = false;
  for (int i = 2; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--mesh") && i + 1 < argc) mesh_file = argv[++i];
    else if (!std::strcmp(argv[i], "--degree") && i + 1 < argc) degree = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--matrix-free")) op = PORO_OP_MATRIX_FREE;
    else if (!std::strcmp(argv[i], "--output") && i + 1 < argc) output_dir = argv[++i];
    else if (!std::strcmp(argv[i], "--corrected-output")) corrected = true;
    else if (!std::strcmp(argv[i], "--incremental-strain")) incremental = true;   // storage term against the previous step instead of the initial state
    else if (!std::strcmp(argv[i], "--coupled-fss")) coupled = true;    // restore get_volumetric_strain() inside the fixed-stress loop (:399)
    else if (!std::strcmp(argv[i], "--ssor")) prec = PORO_PREC_SSOR;   // the reference's PreconditionSSOR instead of Jacobi
    else if (!std::strcmp(argv[i], "--chebyshev")) prec = PORO_PREC_CHEBYSHEV;   // polynomial preconditioner (any mesh / operator)
    else if (!std::strcmp(argv[i], "--block-fdm")) prec = PORO_PREC_FDM;         // block fast diagonalisation (uniform boxes with face-wise Dirichlet data)
    else if (!std::strcmp(argv[i], "--two-level")) prec = PORO_PREC_TWO_LEVEL;   // Jacobi + block fast diagonalisation of the underlying / auxiliary box (refined boxes, rectangle-filling Gmsh meshes)
    else if (!std::strcmp(argv[i], "--fastest")) prec = -1;                      // the strongest preconditioner the mesh supports: block FDM, else two-level, else Chebyshev
    else { std::cerr << "unknown option " << argv[i] << std::endl; return 1; }
  }
  try {
    input_data::InputDataPoroel data;
    data.read_input_file(argv[1]);
    ProblemData P;
    P.bc.dirichlet_labels.assign(data.displacement_boundary_labels.begin(), data.displacement_boundary_labels.end());
    P.bc.dirichlet_components.assign(data.displacement_boundary_components.begin(), data.displacement_boundary_components.end());
    P.bc.dirichlet_values = data.displacement_boundary_values;
    P.bc.neumann_labels.assign(data.stress_boundary_labels.begin(), data.stress_boundary_labels.end());
    P.bc.neumann_components.assign(data.stress_boundary_components.begin(), data.stress_boundary_components.end());
    P.bc.neumann_values = data.stress_boundary_values;
    P.mat = data.material();
    if (!mesh_file.empty()) { P.mesh = read_gmsh22(mesh_file); P.finalize(degree); attach_auxiliary_box(P, degree); }   // (coarse space of --two-level / --fastest where the mesh fills a rectangle)
    else {
      int n[3] = {1, 1, 1}; double size[3] = {1, 1, 1};
      for (int d = 0; d < data.dim; ++d) { n[d] = 1 << data.initial_refinement_level; size[d] = data.domain_size.at(d); }
      build_box_problem(P, data.dim, n, size, degree);
    }
    RunControls rc; rc.preconditioner = prec; rc.output_dir = output_dir; rc.corrected_postprocessing = corrected; rc.coupled_fss = coupled; rc.incremental_strain = incremental;
    rc.p_init = data.p_init; rc.time_step = data.time_step; rc.fss_tol = data.fss_tol; rc.pressure_tol = data.pressure_tol;
    rc.max_fss_iterations = data.max_fss_iterations; rc.max_pressure_iterations = data.max_pressure_iterations;
    int n_steps = 0; for (double t = 0; t < data.t_max; t += data.time_step) ++n_steps;   // while (time < t_max) (:327)
    rc.n_steps = steps >= 0 ? steps : n_steps;
    std::vector<double> trace(8 * (size_t)(1 + rc.n_steps * rc.max_fss_iterations));
    int rows;
    std::cout << "starting time loop" << std::endl << "time max " << data.t_max << std::endl;   // :325-326
    if (data.dim == 2) { PoroElasticProblem<2> prob(P, device, op); rows = prob.run(rc, trace.data(), (int)trace.size() / 8); }
    else { PoroElasticProblem<3> prob(P, device, op); rows = prob.run(rc, trace.data(), (int)trace.size() / 8); }
    for (int r = 1; r < rows; ++r) {
      const double *t = &trace[8 * r];
      if (t[1] == 1) std::cout << "Time: " << t[0] * rc.time_step << std::endl;                  // :330
      std::cout << "    Coupling iteration: " << (int)t[1] << std::endl;                         // :352
      std::cout << "        pressure converged; iterations: " << (int)t[2] << std::endl;        // :367-369
      std::cout << "Solution limits: " << t[4] << "\t" << std::endl;                             // :387-389
      std::cout << "        Error: " << t[5] << std::endl;                                       // :406
    }
  } catch (std::exception &exc) {   // PoroelasticityFSS.h:512-523
    std::cerr << std::endl << "----------------------------------------------------" << std::endl
              << "Exception on processing: " << std::endl << exc.what() << std::endl << "Aborting!" << std::endl
              << "----------------------------------------------------" << std::endl;
    return 1;
  }
  return 0;
}
