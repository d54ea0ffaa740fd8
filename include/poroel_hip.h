/*
 * poroel_hip.h — C-ABI of the MI355X-native fixed-stress Biot hot path.
 *
 * This is the drop-in boundary for the per-timestep "assemble + Krylov solve"
 * path of ishovkun/poroelasticity-dealii.  The reference has no FFI; the path
 * sits behind public member functions of three class templates that are only
 * called from PoroElasticProblem<dim>::run() (lib/include/PoroelasticityFSS.h:294-415).
 * Every entry point below names the reference member (file:line) it replaces.
 * Signatures use plain pointers and sizes only (no C++/torch types); the
 * library owns all device memory behind the opaque handle, the caller owns the
 * host arrays it passes in.  Entry points that hand something back to the host (a norm, a
 * poro_solve_info, a vector, a matrix) return when the device has produced it; entry points that
 * only transform device-resident state (assembly, vector updates, the `distribute` at the end of a
 * solve) are ordered on the context's stream and may return before the device has finished - exactly
 * what lets the fixed-stress loop run without idling the GPU between calls.  poro_ctx_synchronize()
 * waits for everything enqueued so far; errors of asynchronous work surface at the next
 * synchronising call.
 *
 * Conventions fixed by this ABI (the reference leaves them to deal.II):
 *   - local scalar nodes of a cell are lexicographic on the (k+1)^dim tensor
 *     grid, x fastest; local vector dof i = scalar_node*dim + component
 *     (FESystem::system_to_component_index, PoroElasticDisplacementSolver.h:218);
 *   - cell vertices are lexicographic (v = ix + 2*iy + 4*iz), as deal.II;
 *   - faces: f = 2*normal_direction + (0 low side | 1 high side), as deal.II
 *     GeometryInfo; hyper_rectangle(colorize) boundary ids equal f
 *     (PoroelasticityFSS.h:430-432, input.data:8-11);
 *   - quadrature points are tensorised Gauss-Legendre on [0,1], x fastest
 *     (QGauss<dim>, PoroElasticDisplacementSolver.h:159);
 *   - all floating point data is IEEE double; dof/cell indices are int32,
 *     CSR row pointers int64.
 *
 * Return codes: 0 ok; >0 a Krylov solve hit its iteration cap (the analogue of
 * deal.II SolverControl::NoConvergence, info is filled); <0 invalid argument,
 * HIP or RCCL failure (message via poro_last_error()).
 */
#ifndef POROEL_HIP_H
#define POROEL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PORO_ABI_VERSION 4   /* 4: poro_desc.tensor (tensor-product grids keep the fast-diagonalisation preconditioners), poro_desc.coarse + PORO_PREC_TWO_LEVEL;  2: poro_solver_opts.omega, PORO_PREC_SSOR / FDM / ILU0, PORO_VEC_STRESS0; 3: poro_solver_opts.stop_rule / poly_degree, poro_constraints, PORO_PREC_FDM and PORO_PREC_CHEBYSHEV for the displacement system, general form of poro_partition, prescribed pressures */

/* Reference-cell tables: exactly the numbers the reference pulls out of
 * FEValues / FEFaceValues (PoroElasticDisplacementSolver.h:162-173,
 * StrainProjector.h:127-134, MatrixCreator calls PoroElasticPressureSolver.h:96-101).
 * "q1" is the Q1 Lagrange basis on the cell vertices: it is both MappingQ1 and
 * the pressure element FE_Q(1) (PoroElasticPressureSolver.h:20,57). */
typedef struct poro_fe_tables {
  int32_t nq_u;  /* (k_u+1)^dim  volume points, QGauss(fe.degree+1) of u      */
  int32_t nq_p;  /* 2^dim        volume points, QGauss(2) of p                */
  int32_t nq_f;  /* (k_u+1)^(dim-1) face points                               */
  int32_t ns_u;  /* scalar nodes per cell of u: (k_u+1)^dim                   */
  int32_t ns_p;  /* = number of vertices 2^dim                                */
  const double *w_qu;      /* [nq_u]                                          */
  const double *w_qp;      /* [nq_p]                                          */
  const double *w_qf;      /* [nq_f]                                          */
  const double *u_qu;      /* [nq_u][ns_u]        phi^u_s(xi_q)               */
  const double *du_qu;     /* [nq_u][ns_u][dim]   d phi^u_s / d xi_d          */
  const double *du_qp;     /* [nq_p][ns_u][dim]                               */
  const double *q1_qu;     /* [nq_u][ns_p]                                    */
  const double *dq1_qu;    /* [nq_u][ns_p][dim]                               */
  const double *q1_qp;     /* [nq_p][ns_p]                                    */
  const double *dq1_qp;    /* [nq_p][ns_p][dim]                               */
  const double *u_qf;      /* [2*dim][nq_f][ns_u] u shape values on face f    */
  const double *dq1_qf;    /* [2*dim][nq_f][ns_p][dim]                        */
} poro_fe_tables;

/* Constant coefficients, = InputDataPoroel after compute_derived_parameters
 * (InputDataPoroel.h:213-222; units :162,168). */
typedef struct poro_material {
  double lame_lambda;   /* data.lame_constant  */
  double shear_G;       /* data.shear_modulus  */
  double biot_alpha;    /* data.biot_coef      */
  double bulk_K;        /* data.bulk_modulus (drained)  */
  double biot_M;        /* data.m_modulus      */
  double k_over_mu;     /* data.perm/data.visc */
  double r_well;        /* data.r_well         */
  double flow_rate;     /* data.flow_rate      */
} poro_material;

/* Optional: the mesh is a uniform box n[0] x n[1] (x n[2]) with lexicographic
 * node numbering and dof = node*dim + comp (u), dof = vertex (p).  Enables the
 * matrix-free operator.  hyper_rectangle + refine_global meshes
 * (PoroelasticityFSS.h:418-435) are of this kind. */
typedef struct poro_structured {
  int32_t enabled;
  int32_t n[3];
  double  origin[3];
  double  h[3];
} poro_structured;

/* Optional: the mesh is a TENSOR-PRODUCT grid - topology and numbering as for a poro_structured box with n[] cells per direction, vertex planes of direction d at
 * grid[d][0 .. n[d]] (ascending, any spacing), e.g. a graded hyper_rectangle.  The cells are not congruent, so the structured operator kernels do not
 * apply (the general matrix-free operator runs), but the fast-diagonalisation preconditioners stay exact: their 1D FE matrices are assembled on the
 * given 1D grids (PORO_PREC_FDM for all three systems; one rank).  Ignored when box.enabled. */
typedef struct poro_tensor_grid {
  int32_t enabled;
  int32_t n[3];
  const double *grid[3];   /* [n[d] + 1] each */
} poro_tensor_grid;

/* Optional coarse space for the two-level preconditioner of the displacement system (PORO_PREC_TWO_LEVEL): the mesh is a refinement of a uniform box - a locally
 * refined hyper_rectangle (refine_mesh, PoroelasticityFSS.h:447-498) - whose own description is `box_problem` (same material and boundary conditions, box tag
 * set; it must stay alive until poro_ctx_create has returned), and every displacement NODE i of this mesh (= dof / dim: the numbering must be node-interleaved)
 * interpolates the box's FE functions: v_h(node i) = sum_k weight[k] * v_H(node[k]), k in ptr[i] .. ptr[i+1].  One rank. */
struct poro_desc;
typedef struct poro_coarse_space {
  int32_t enabled;
  const struct poro_desc *box_problem;
  const int64_t *ptr;       /* [n_dofs_u / dim + 1] */
  const int32_t *node;      /* [ptr[last]] coarse node numbers */
  const double  *weight;
  /* optional: the same for the pressure space (dof i = sum weight_p[k] * box pressure dof node_p[k]): PORO_PREC_TWO_LEVEL for the pressure and projection solves */
  const int64_t *ptr_p;     /* [n_dofs_p + 1] or NULL */
  const int32_t *node_p;
  const double  *weight_p;
} poro_coarse_space;

/* Partition over ranks.  Slab form for a structured box:  The local mesh is
 * the rank's slab as a standalone box; node planes at the low / high end in the
 * slowest direction are shared with the neighbour rank when has_lower/has_upper. */
typedef struct poro_partition {
  int32_t rank, n_ranks;
  int32_t has_lower, has_upper;
  int64_t plane_u;   /* u dofs on one interface plane  */
  int64_t plane_p;   /* p dofs on one interface plane  */
  /* General partition (any mesh; SURVEY 8e: contiguous cell ranges + an indexed interface list), used when n_neighbours > 0 (then has_lower /
   * has_upper / plane_* are ignored).  The local mesh is the rank's cells as a standalone mesh; a dof touched by cells of several ranks is
   * local to each of them.  Local dofs [0, n_owned) are the ones this rank owns (every shared dof has exactly one owner; dots run over owned
   * dofs), the others follow.  shared_dof_*[shared_ptr_*[k] .. shared_ptr_*[k+1]) are the local dofs shared with rank neighbour_rank[k], listed
   * in the SAME order on both sides (e.g. ascending global index); a dof shared by three ranks appears in two lists on each of them.  Partial
   * sums are exchanged pairwise and added in ascending rank order, so every rank holding a dof computes bitwise the same value. */
  int32_t n_neighbours;
  const int32_t *neighbour_rank;   /* [n_neighbours], ascending, without `rank` */
  const int64_t *shared_ptr_u;     /* [n_neighbours + 1] */
  const int32_t *shared_dof_u;
  const int64_t *shared_ptr_p;     /* [n_neighbours + 1] */
  const int32_t *shared_dof_p;
  int64_t n_owned_u, n_owned_p;
} poro_partition;

/* Closed affine constraints beyond the Dirichlet list: the hanging nodes of a locally refined mesh (DoFTools::make_hanging_node_constraints,
 * PoroElasticDisplacementSolver.h:112-113, PoroElasticPressureSolver.h:72-75; refine_mesh, PoroelasticityFSS.h:447-498).
 *   x[dof[i]] = sum_{k in ptr[i]..ptr[i+1]} weight[k] * x[master[k]] + inhomogeneity[i]
 * (Partitioned runs, general form of poro_partition: every rank passes the entries whose constrained dof is local to it, and every master of such an entry must be
 * local too - where none of the rank's cells touches a master it becomes a local dof of no local cell, listed in the interface lists of all ranks that hold it.)
 * exactly what ConstraintMatrix holds after close(): masters are unconstrained dofs, a constrained dof appears once and is not in the
 * Dirichlet list.  The library condenses at operator level (C^T A C on the free dofs, C^T b), which gives the same solution as
 * ConstraintMatrix::condense / distribute_local_to_global (:153, :168, PoroElasticDisplacementSolver.h:280-286), and distributes after each solve (:180, :306). */
typedef struct poro_constraints {
  int64_t n;
  const int32_t *dof;            /* [n]      */
  const int64_t *ptr;            /* [n+1]    */
  const int32_t *master;         /* [ptr[n]] */
  const double  *weight;         /* [ptr[n]] */
  const double  *inhomogeneity;  /* [n]      */
} poro_constraints;

typedef struct poro_desc {
  int32_t abi_version;
  int32_t dim;        /* 2 | 3 */
  int32_t degree_u;   /* 1 | 2  (the reference hard-wires 2, PoroElasticDisplacementSolver.h:67) */
  int32_t degree_p;   /* 1 */
  int64_t n_cells, n_vertices, n_dofs_u, n_dofs_p;
  const double  *vertex_coords;  /* [n_vertices][dim]                       */
  const int32_t *cell_vertices;  /* [n_cells][2^dim]                        */
  const int32_t *cell_dofs_u;    /* [n_cells][dim*ns_u]  cell->get_dof_indices (:279) */
  const int32_t *cell_dofs_p;    /* [n_cells][ns_p]      (StrainProjector.h:191)      */
  poro_fe_tables fe;
  /* boundary faces, for the Neumann term (:249-277) */
  int64_t n_bfaces;
  const int32_t *bface_cell, *bface_local, *bface_id;
  /* closed Dirichlet constraint list = `constraints` after :117-136 */
  int64_t n_dirichlet;
  const int32_t *dirichlet_dof;
  const double  *dirichlet_value;
  /* Neumann conditions (BoundaryConditions.h:45-62) */
  int32_t n_neumann;
  const int32_t *neumann_label, *neumann_component;
  const double  *neumann_value;
  poro_material   mat;
  poro_structured box;
  poro_partition  part;
  poro_constraints cons_u;   /* displacement space (n = 0: none) */
  poro_constraints cons_p;   /* pressure space; also used by the strain projection (StrainProjector.h:191-194) */
  /* EXTENSION (not in the reference, whose pressure space has "no dirichlet pressure BC's", PoroElasticPressureSolver.h:69-70): prescribed pressures,
   * e.g. a drained boundary p = 0.  Needed to validate the corrected-physics switches on Terzaghi's consolidation problem.  The rows are
   * taken out of the pressure Newton system (residual 0, update 0); poro_pres_apply_boundary_values writes the values into PORO_VEC_P. */
  int64_t n_dirichlet_p;
  const int32_t *dirichlet_dof_p;
  const double  *dirichlet_value_p;
  poro_tensor_grid tensor;
  poro_coarse_space coarse;
} poro_desc;

/* Krylov controls.  Reference values: displacement abs 1e-12, 1000 its
 * (PoroElasticDisplacementSolver.h:298-299); pressure / projection
 * rel 1e-8*||rhs||, 1000 its (PoroElasticPressureSolver.h:175, StrainProjector.h:209).
 * Stopping test is on the recursively updated residual: ||g||_2 <= max(abs_tol, rel_tol*||b||_2) (PORO_STOP_RHS, what the
 * reference's three SolverControl objects express), or ||g||_2 <= max(abs_tol, rel_tol*||g_0||_2) with g_0 the residual of the
 * warm start (PORO_STOP_REDUCTION = deal.II ReductionControl): a transient whose right-hand side barely changes from step to step
 * still solves every step to the same relative accuracy instead of accepting the warm start. */
typedef struct poro_solver_opts {
  double  abs_tol;
  double  rel_tol;
  int32_t max_iter;
  int32_t preconditioner;  /* PORO_PREC_* */
  double  omega;           /* relaxation of PORO_PREC_SSOR: 1.2 displacement (:303), 1.0 pressure / projection (:178, StrainProjector.h:212) */
  int32_t stop_rule;       /* PORO_STOP_* */
  int32_t poly_degree;     /* PORO_PREC_CHEBYSHEV: operator applications per preconditioner call (<= 0: 6; odd values are rounded up); `omega` then holds the ratio lambda_max / a of
                              the interval [a, lambda_max] the polynomial is built for (<= 0: a mesh-size based default) */
} poro_solver_opts;
enum { PORO_STOP_RHS = 0, PORO_STOP_REDUCTION = 1 };

/* PORO_PREC_SSOR = PreconditionSSOR in the matrix's natural row order (level-scheduled sweeps; assembled-CSR operators only):
 * reproduces the reference's Krylov iterates, at many small launches per application - a fidelity mode, not the fast path.
 * PORO_PREC_FDM = fast diagonalisation: on a uniform box (poro_desc.box.enabled; slab-partitioned runs included) the pressure Jacobian and the projection
 * mass matrix are sums of Kronecker products of 1D matrices and are inverted exactly by 2*dim batched dense transforms (fp64 MFMA);
 * CG keeps the reference's stopping rule and needs 1-2 iterations.  poro_supports_preconditioner() tells whether a context can.
 * For the displacement system PORO_PREC_FDM is the BLOCK fast diagonalisation: the diagonal blocks A_cc of the elasticity operator (one per
 * displacement component) are Kronecker sums of 1D FE_Q(k) matrices whenever every Dirichlet condition covers whole faces, and are inverted
 * exactly the same way (per-component 1D eigenvectors, fp64 MFMA transforms); CG on a 10 M-dof box then takes ~20 iterations instead of ~250.
 * PORO_PREC_ILU0 = incomplete LU on the pattern of the assembled CSR matrix (factorised once per matrix ON THE DEVICE, level-scheduled in the natural
 * row order like the triangular solves, so the factors equal those of a sequential IKJ sweep; one rank, moderate sizes).
 * PORO_PREC_CHEBYSHEV (displacement system) = Chebyshev polynomial in D^-1 A of degree poly_degree around the Jacobi preconditioner: the CG iteration
 * count drops by about the degree + 1 while the operator applications of the polynomial need no dot products and, on 3D boxes, no vector kernels either
 * (the update z_{j+1} = z_j + D^-1 (g - A z_j) / r_j - roots r_j of the shifted Chebyshev polynomial - is applied inside the structured operator kernel where the product leaves the registers): fewer bytes and far fewer reductions per
 * operator application than Jacobi-CG.  lambda_max(D^-1 A) comes from the Lanczos tridiagonal of 25 Jacobi-CG steps when the matrix is (re)built (+5 %,
 * capped on uniform boxes by the rigorous element bound lambda_max(diag(K_e)^-1 K_e)).
 * PORO_PREC_TWO_LEVEL (displacement system; meshes with poro_desc.coarse, i.e. locally refined boxes with their hanging-node constraints) = additive two-level
 * preconditioner z = omega D^-1 g + P B_H^-1 P^T g: Jacobi on the refined mesh plus the BLOCK fast diagonalisation of the underlying uniform box as coarse solve (P = the
 * FE interpolation of poro_coarse_space).  The CG iteration count stays bounded under uniform refinement of the whole configuration.  With poro_coarse_space.ptr_p ...
 * also for poro_pres_solve (a M + kappa K: Jacobi + the box's exact scalar fast diagonalisation through the vertex interpolation; hanging nodes allowed, prescribed
 * pressures not) and poro_proj_solve (accepted; Jacobi alone is already mesh-independent on the mass matrix and needs fewer iterations). */
enum { PORO_PREC_NONE = 0, PORO_PREC_JACOBI = 1, PORO_PREC_SSOR = 2, PORO_PREC_FDM = 3, PORO_PREC_ILU0 = 4, PORO_PREC_CHEBYSHEV = 5, PORO_PREC_TWO_LEVEL = 6 };
enum { PORO_OP_CSR = 0, PORO_OP_MATRIX_FREE = 1 };
enum { PORO_MAT_A_U = 0, PORO_MAT_MASS_P = 1, PORO_MAT_LAPLACE_P = 2, PORO_MAT_JACOBIAN_P = 3 };
enum { PORO_VEC_U = 0, PORO_VEC_RHS_U = 1, PORO_VEC_P = 2, PORO_VEC_P_OLD = 3, PORO_VEC_DP = 4,
       PORO_VEC_RESIDUAL_P = 5, PORO_VEC_EPSV = 6, PORO_VEC_EPSV0 = 7, PORO_VEC_SOURCE_P = 8,
       PORO_VEC_STRAIN0 = 16 /* + packed symmetric entry (TensorIndexer.h:24-31) */,
       PORO_VEC_PROJ_RHS0 = 32 /* + entry */, PORO_VEC_DIAG_U = 48,
       PORO_VEC_STRESS0 = 64 /* + packed symmetric entry: nodal effective stress (PoroelasticityFSS.h:189-224) */ };

typedef struct poro_solve_info {
  int32_t iterations;
  int32_t converged;
  double  initial_residual;
  double  final_residual;
  double  seconds;       /* wall time of the solve on device */
  int64_t operator_applications;
} poro_solve_info;

typedef struct poro_ctx poro_ctx;

const char *poro_last_error(void);
int  poro_abi_version(void);

/* setup_dofs of the three solvers (PoroElasticDisplacementSolver.h:106-153,
 * PoroElasticPressureSolver.h:68-111, StrainProjector.h:82-98): builds sparsity,
 * mass / Laplace matrices, constraint tables, device vectors.  operator_mode
 * selects assembled CSR or the matrix-free A_u (needs desc->box.enabled). */
int  poro_ctx_create(const poro_desc *desc, int device, int operator_mode, poro_ctx **out);
void poro_ctx_destroy(poro_ctx *ctx);
/* wait until the device has finished everything the context has enqueued (see the note on synchronisation at the top) */
int  poro_ctx_synchronize(poro_ctx *ctx);

/* multi-GPU wiring.  id is the 128-byte ncclUniqueId from rank 0. */
int  poro_comm_unique_id(void *id128);
int  poro_ctx_comm_init_rccl(poro_ctx *ctx, const void *id128);
/* host-staged exchange through caller callbacks (tests: 2 ranks on one GPU, gloo). */
typedef void (*poro_allreduce_fn)(double *buf, int32_t n, void *user);
typedef void (*poro_sendrecv_fn)(const double *send, double *recv, int64_t n, int32_t peer, void *user);
int  poro_ctx_comm_init_callbacks(poro_ctx *ctx, poro_allreduce_fn ar, poro_sendrecv_fn sr, void *user);

/* vectors live on the device between calls; these move them across the boundary */
int  poro_vec_set(poro_ctx *ctx, int which, const double *host, int64_t n);
int  poro_vec_get(poro_ctx *ctx, int which, double *host, int64_t n);
int  poro_vec_fill(poro_ctx *ctx, int which, double value);
int  poro_vec_copy(poro_ctx *ctx, int dst, int src);               /* e.g. old_solution = solution (PoroelasticityFSS.h:342) */
int  poro_vec_axpy(poro_ctx *ctx, int y, double a, int x);         /* solution += solution_update (:379)                    */
int  poro_vec_norm(poro_ctx *ctx, int which, double *l2, double *linf);

/* device-side snapshot of every PORO_VEC_* vector (one slot) and its restoration: lets a caller retry a time step (e.g. with another dt) or
 * repeat one for measurements without moving the state across PCIe; the reference has no counterpart (its only persistence is the per-step VTK dump,
 * PoroelasticityFSS.h:286-290). */
int  poro_state_save(poro_ctx *ctx);
int  poro_state_restore(poro_ctx *ctx);

/* PoroElasticDisplacementSolver<dim>::assemble_system (:155-291); pressure taken from PORO_VEC_P.
 * rebuild_matrix mirrors `rebuild_system_matrix` (:280): 1 = (re)build A_u (CSR values, or the
 * matrix-free operator data + diagonal) and the constant Dirichlet lifting, 0 = RHS only. */
int  poro_disp_assemble_system(poro_ctx *ctx, int rebuild_matrix);
/* PoroElasticDisplacementSolver<dim>::solve (:294-307): PCG, warm start from PORO_VEC_U, then constraints.distribute. */
int  poro_disp_solve(poro_ctx *ctx, const poro_solver_opts *opts, poro_solve_info *info);
/* 1 if `preconditioner` can be used by poro_pres_solve / poro_proj_solve (which_system = 1) or poro_disp_solve (which_system = 0) on this context, else 0 */
int  poro_supports_preconditioner(poro_ctx *ctx, int32_t which_system, int32_t preconditioner);

/* PoroElasticPressureSolver<dim>::assemble_residual (:113-155) from PORO_VEC_{P,P_OLD,EPSV,EPSV0}; l2 = residual.l2_norm() (PoroelasticityFSS.h:364) */
int  poro_pres_assemble_residual(poro_ctx *ctx, double time_step, double *l2);
/* extension: PORO_VEC_P[dof] = value on the prescribed-pressure dofs of the descriptor; a no-op without any; call after setting the initial pressure */
int  poro_pres_apply_boundary_values(poro_ctx *ctx);
/* PoroElasticPressureSolver<dim>::assemble_jacobian (:158-169) */
int  poro_pres_assemble_jacobian(poro_ctx *ctx, double time_step);
/* PoroElasticPressureSolver<dim>::solve (:172-185): J dp = R into PORO_VEC_DP (warm start) */
int  poro_pres_solve(poro_ctx *ctx, const poro_solver_opts *opts, poro_solve_info *info);
/* PoroElasticPressureSolver<dim>::update_volumetric_strain (:187-194): eps_v += (alpha/K) dp */
int  poro_pres_update_volumetric_strain(poro_ctx *ctx);

/* StrainProjector<dim>::assemble_projection_matrix (:101-106) */
int  poro_proj_assemble_matrix(poro_ctx *ctx);
/* StrainProjector<dim>::assemble_projection_rhs (:109-198); tensor_components are full indices a*dim+b */
int  poro_proj_assemble_rhs(poro_ctx *ctx, const int32_t *tensor_components, int32_t n_comp);
/* StrainProjector<dim>::solve_projection_system (:201-232); rhs_entry = packed symmetric entry */
int  poro_proj_solve(poro_ctx *ctx, int32_t rhs_entry, const poro_solver_opts *opts, poro_solve_info *info);
/* the loop of StrainProjector::solve_projection_system over several entries (PoroelasticityFSS.h:157-163) in one call: where PORO_PREC_FDM is the exact inverse of the
 * projection mass matrix the systems are solved directly and together (info[e].iterations = 0, residual checked against the stopping rule), otherwise entry by entry */
int  poro_proj_solve_many(poro_ctx *ctx, const int32_t *rhs_entries, int32_t n_entries, const poro_solver_opts *opts, poro_solve_info *info /* [n_entries] */);
/* PoroElasticProblem<dim>::get_volumetric_strain (PoroelasticityFSS.h:179-186): eps_v = sum of normal strains */
int  poro_get_volumetric_strain(poro_ctx *ctx);
/* PoroElasticProblem<dim>::get_effective_stresses (PoroelasticityFSS.h:189-224): sigma' = C : eps at every pressure node, C = isotropic_gassman_tensor
 * (ConstitutiveModel.h:45-57); PORO_VEC_STRAIN0+e -> PORO_VEC_STRESS0+e */
int  poro_get_effective_stresses(poro_ctx *ctx);

/* parity / measurement hooks */
int  poro_export_csr_size(poro_ctx *ctx, int which, int64_t *n_rows, int64_t *nnz);
int  poro_export_csr(poro_ctx *ctx, int which, int64_t *row_ptr, int32_t *col, double *val);
/* y = A x with the operator `which` (A_U honours the ctx operator mode); host in/out */
int  poro_apply_operator(poro_ctx *ctx, int which, const double *x_host, double *y_host);
/* z = P^-1 g with the displacement preconditioner (PORO_PREC_JACOBI | PORO_PREC_FDM), host in/out: parity hook for the preconditioner
 * PoroElasticDisplacementSolver<dim>::solve hands to cg.solve (:302-305; the reference's is PreconditionSSOR).  reps > 0 additionally times
 * `reps` applications on the device (HIP events) into *seconds_per_apply. */
int  poro_apply_preconditioner_u(poro_ctx *ctx, int32_t preconditioner, const double *g_host, double *z_host, int32_t reps, double *seconds_per_apply);
/* repeat y = A_u x `reps` times on device-resident synthetic x; returns mean seconds per application (HIP events) */
int  poro_bench_operator(poro_ctx *ctx, int which, int operator_mode, int reps, double *seconds_per_apply);
/* accumulated HIP-event time (s) and launch count of the named kernel family since the last reset */
int  poro_timers_reset(poro_ctx *ctx);            /* clears the accumulators and switches event timing on */
int  poro_timers_enable(poro_ctx *ctx, int on);   /* 0: off; 1: events on every launch; k > 1: on every k-th launch of a family (the events cost ~2.5 us per launch), poro_timers_get scales the sampled time to all launches */
int  poro_timers_get(poro_ctx *ctx, const char *name, double *seconds, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* POROEL_HIP_H */
