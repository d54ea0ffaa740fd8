"""ctypes front end of the MI355X-native fixed-stress Biot hot path.

Thin plumbing only: the product is the C-ABI library ``lib/libporoel_hip.so`` (hand-written HIP kernels,
``include/poroel_hip.h``) and the C++ host layer ``lib/libporoel_host.so`` (mesh / DoF / FE-table provider,
parameter-file front end, ``PoroElasticProblem<dim>::run()`` mirror).  There is no CPU fallback: creating a
context without a HIP device fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

# ids of include/poroel_hip.h
PREC_NONE, PREC_JACOBI, PREC_SSOR, PREC_FDM, PREC_ILU0, PREC_CHEBYSHEV, PREC_TWO_LEVEL = 0, 1, 2, 3, 4, 5, 6
STOP_RHS, STOP_REDUCTION = 0, 1
OP_CSR, OP_MATRIX_FREE = 0, 1
MAT_A_U, MAT_MASS_P, MAT_LAPLACE_P, MAT_JACOBIAN_P = 0, 1, 2, 3
VEC_U, VEC_RHS_U, VEC_P, VEC_P_OLD, VEC_DP, VEC_RESIDUAL_P, VEC_EPSV, VEC_EPSV0, VEC_SOURCE_P = range(9)
VEC_STRAIN0, VEC_PROJ_RHS0, VEC_DIAG_U, VEC_STRESS0 = 16, 32, 48, 64

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class FeTables(C.Structure):
    _fields_ = [("nq_u", C.c_int32), ("nq_p", C.c_int32), ("nq_f", C.c_int32), ("ns_u", C.c_int32), ("ns_p", C.c_int32)] + [
        (n, _dp) for n in ("w_qu", "w_qp", "w_qf", "u_qu", "du_qu", "du_qp", "q1_qu", "dq1_qu", "q1_qp", "dq1_qp", "u_qf", "dq1_qf")]


class Material(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lame_lambda", "shear_G", "biot_alpha", "bulk_K", "biot_M", "k_over_mu", "r_well", "flow_rate")]


class Structured(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("n", C.c_int32 * 3), ("origin", C.c_double * 3), ("h", C.c_double * 3)]


class Partition(C.Structure):
    _fields_ = [("rank", C.c_int32), ("n_ranks", C.c_int32), ("has_lower", C.c_int32), ("has_upper", C.c_int32), ("plane_u", C.c_int64), ("plane_p", C.c_int64),
                ("n_neighbours", C.c_int32), ("neighbour_rank", _ip), ("shared_ptr_u", _lp), ("shared_dof_u", _ip), ("shared_ptr_p", _lp), ("shared_dof_p", _ip),
                ("n_owned_u", C.c_int64), ("n_owned_p", C.c_int64)]


class Constraints(C.Structure):
    _fields_ = [("n", C.c_int64), ("dof", _ip), ("ptr", C.POINTER(C.c_int64)), ("master", _ip), ("weight", _dp), ("inhomogeneity", _dp)]


class TensorGrid(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("n", C.c_int32 * 3), ("grid", _dp * 3)]


class CoarseSpace(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("box_problem", C.c_void_p), ("ptr", C.POINTER(C.c_int64)), ("node", _ip), ("weight", _dp),
                ("ptr_p", C.POINTER(C.c_int64)), ("node_p", _ip), ("weight_p", _dp)]


class Desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("dim", C.c_int32), ("degree_u", C.c_int32), ("degree_p", C.c_int32),
                ("n_cells", C.c_int64), ("n_vertices", C.c_int64), ("n_dofs_u", C.c_int64), ("n_dofs_p", C.c_int64),
                ("vertex_coords", _dp), ("cell_vertices", _ip), ("cell_dofs_u", _ip), ("cell_dofs_p", _ip),
                ("fe", FeTables),
                ("n_bfaces", C.c_int64), ("bface_cell", _ip), ("bface_local", _ip), ("bface_id", _ip),
                ("n_dirichlet", C.c_int64), ("dirichlet_dof", _ip), ("dirichlet_value", _dp),
                ("n_neumann", C.c_int32), ("neumann_label", _ip), ("neumann_component", _ip), ("neumann_value", _dp),
                ("mat", Material), ("box", Structured), ("part", Partition), ("cons_u", Constraints), ("cons_p", Constraints),
                ("n_dirichlet_p", C.c_int64), ("dirichlet_dof_p", _ip), ("dirichlet_value_p", _dp), ("tensor", TensorGrid), ("coarse", CoarseSpace)]


class SolverOpts(C.Structure):
    _fields_ = [("abs_tol", C.c_double), ("rel_tol", C.c_double), ("max_iter", C.c_int32), ("preconditioner", C.c_int32), ("omega", C.c_double),
                ("stop_rule", C.c_int32), ("poly_degree", C.c_int32)]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("initial_residual", C.c_double), ("final_residual", C.c_double),
                ("seconds", C.c_double), ("operator_applications", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class InputFlat(C.Structure):
    _fields_ = [("dim", C.c_int32), ("domain_size", C.c_double * 3), ("initial_refinement_level", C.c_int32), ("max_refinement_level", C.c_int32)] + [
        (n, C.c_double) for n in ("youngs_modulus", "poisson_ratio", "biot_coef", "perm", "poro", "visc", "bulk_density", "f_comp", "r_well", "flow_rate",
                                  "p_init", "time_step", "t_max", "fss_tol", "pressure_tol")] + [
        ("max_fss_iterations", C.c_int32), ("max_pressure_iterations", C.c_int32),
        ("n_dirichlet", C.c_int32), ("dirichlet_labels", C.c_int32 * 16), ("dirichlet_components", C.c_int32 * 16), ("dirichlet_values", C.c_double * 16),
        ("n_neumann", C.c_int32), ("neumann_labels", C.c_int32 * 16), ("neumann_components", C.c_int32 * 16), ("neumann_values", C.c_double * 16)] + [
        (n, C.c_double) for n in ("lame_constant", "shear_modulus", "bulk_modulus", "grain_bulk_modulus", "n_modulus", "m_modulus")] + [("material", Material)]


ALLREDUCE_FN = C.CFUNCTYPE(None, _dp, C.c_int32, C.c_void_p)
SENDRECV_FN = C.CFUNCTYPE(None, _dp, _dp, C.c_int64, C.c_int32, C.c_void_p)

# every symbol include/poroel_hip.h declares (checked by the CPU test-suite against the built library)
HIP_SYMBOLS = [
    "poro_last_error", "poro_abi_version", "poro_ctx_create", "poro_ctx_destroy", "poro_ctx_synchronize", "poro_comm_unique_id", "poro_ctx_comm_init_rccl",
    "poro_ctx_comm_init_callbacks", "poro_vec_set", "poro_vec_get", "poro_vec_fill", "poro_vec_copy", "poro_vec_axpy", "poro_vec_norm",
    "poro_state_save", "poro_state_restore", "poro_disp_assemble_system", "poro_disp_solve", "poro_supports_preconditioner", "poro_pres_assemble_residual", "poro_pres_apply_boundary_values", "poro_pres_assemble_jacobian", "poro_pres_solve",
    "poro_pres_update_volumetric_strain", "poro_proj_assemble_matrix", "poro_proj_assemble_rhs", "poro_proj_solve", "poro_proj_solve_many", "poro_get_volumetric_strain", "poro_get_effective_stresses",
    "poro_export_csr_size", "poro_export_csr", "poro_apply_operator", "poro_apply_preconditioner_u", "poro_bench_operator", "poro_timers_reset", "poro_timers_enable", "poro_timers_get"]

_hip = None
_host = None


def _need(path):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no fallback path")
    return path


def load_hip():
    """The C-ABI library (HIP kernels).  Loads without a GPU; compute calls need one."""
    global _hip
    if _hip is None:
        L = C.CDLL(_need(os.path.join(LIB_DIR, "libporoel_hip.so")), mode=C.RTLD_GLOBAL)
        L.poro_last_error.restype = C.c_char_p
        L.poro_ctx_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.poro_ctx_destroy.argtypes = [C.c_void_p]
        L.poro_ctx_destroy.restype = None
        L.poro_ctx_synchronize.argtypes = [C.c_void_p]
        L.poro_comm_unique_id.argtypes = [C.c_void_p]
        L.poro_ctx_comm_init_rccl.argtypes = [C.c_void_p, C.c_void_p]
        L.poro_ctx_comm_init_callbacks.argtypes = [C.c_void_p, ALLREDUCE_FN, SENDRECV_FN, C.c_void_p]
        L.poro_vec_set.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64]
        L.poro_vec_get.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64]
        L.poro_vec_fill.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.poro_vec_copy.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.poro_vec_axpy.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
        L.poro_vec_norm.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.poro_disp_assemble_system.argtypes = [C.c_void_p, C.c_int]
        L.poro_state_save.argtypes = [C.c_void_p]
        L.poro_state_restore.argtypes = [C.c_void_p]
        L.poro_disp_solve.argtypes = [C.c_void_p, C.POINTER(SolverOpts), C.POINTER(SolveInfo)]
        L.poro_supports_preconditioner.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.poro_pres_assemble_residual.argtypes = [C.c_void_p, C.c_double, _dp]
        L.poro_pres_assemble_jacobian.argtypes = [C.c_void_p, C.c_double]
        L.poro_pres_apply_boundary_values.argtypes = [C.c_void_p]
        L.poro_pres_solve.argtypes = [C.c_void_p, C.POINTER(SolverOpts), C.POINTER(SolveInfo)]
        L.poro_pres_update_volumetric_strain.argtypes = [C.c_void_p]
        L.poro_proj_assemble_matrix.argtypes = [C.c_void_p]
        L.poro_proj_assemble_rhs.argtypes = [C.c_void_p, _ip, C.c_int32]
        L.poro_proj_solve.argtypes = [C.c_void_p, C.c_int32, C.POINTER(SolverOpts), C.POINTER(SolveInfo)]
        L.poro_proj_solve_many.argtypes = [C.c_void_p, _ip, C.c_int32, C.POINTER(SolverOpts), C.POINTER(SolveInfo)]
        L.poro_get_volumetric_strain.argtypes = [C.c_void_p]
        L.poro_get_effective_stresses.argtypes = [C.c_void_p]
        L.poro_export_csr_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.poro_export_csr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), _ip, _dp]
        L.poro_apply_operator.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.poro_bench_operator.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp]
        L.poro_apply_preconditioner_u.argtypes = [C.c_void_p, C.c_int32, _dp, _dp, C.c_int32, _dp]
        L.poro_timers_reset.argtypes = [C.c_void_p]
        L.poro_timers_enable.argtypes = [C.c_void_p, C.c_int]
        L.poro_timers_get.argtypes = [C.c_void_p, C.c_char_p, _dp, C.POINTER(C.c_int64)]
        _hip = L
    return _hip


def load_host():
    """The C++ host layer (mesh / DoFs / FE tables / parameter file / run() driver)."""
    global _host
    if _host is None:
        load_hip()
        L = C.CDLL(_need(os.path.join(LIB_DIR, "libporoel_host.so")))
        L.poro_host_last_error.restype = C.c_char_p
        bc = [C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp, C.POINTER(Material)]
        L.poro_host_build_graded_box.restype = C.c_void_p
        L.poro_host_build_graded_box.argtypes = [C.c_int, _ip, _dp, C.c_int, _dp] + bc
        L.poro_host_build_box.restype = C.c_void_p
        L.poro_host_build_box.argtypes = [C.c_int, _ip, _dp, C.c_int, C.c_int, C.c_int] + bc
        L.poro_host_build_refined_box.restype = C.c_void_p
        L.poro_host_build_refined_box.argtypes = [C.c_int, _ip, _dp, C.c_int, _ip, _ip] + bc
        L.poro_host_build_gmsh.restype = C.c_void_p
        L.poro_host_build_gmsh.argtypes = [C.c_char_p, C.c_int] + bc
        L.poro_host_build_gmsh_refined.restype = C.c_void_p
        L.poro_host_build_gmsh_refined.argtypes = [C.c_char_p, C.c_int, C.c_int] + bc
        L.poro_host_set_pressure_bc.argtypes = [C.c_void_p, C.c_int, _ip, _dp]
        L.poro_host_tie_boundary.argtypes = [C.c_void_p, C.c_int, _ip, _ip]
        L.poro_host_partition.restype = C.c_void_p
        L.poro_host_partition.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.poro_host_local_to_global.restype = C.c_int64
        L.poro_host_local_to_global.argtypes = [C.c_void_p, C.c_int, _ip]
        L.poro_host_desc.restype = C.POINTER(Desc)
        L.poro_host_desc.argtypes = [C.c_void_p]
        L.poro_host_free.argtypes = [C.c_void_p]
        L.poro_host_free.restype = None
        L.poro_host_read_input.argtypes = [C.c_char_p, C.POINTER(InputFlat)]
        L.poro_host_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.POINTER(C.c_void_p)]
        L.poro_host_runner_create.restype = C.c_void_p
        L.poro_host_runner_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        L.poro_host_runner_ctx.restype = C.c_void_p
        L.poro_host_runner_ctx.argtypes = [C.c_void_p]
        L.poro_host_runner_initialize.argtypes = [C.c_void_p]
        L.poro_host_runner_step.argtypes = [C.c_void_p, _dp, C.c_int, C.POINTER(C.c_int64)]
        L.poro_host_runner_restore_and_step.argtypes = [C.c_void_p, _dp, C.c_int, C.POINTER(C.c_int64)]
        L.poro_host_runner_work.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.poro_host_runner_work.restype = None
        L.poro_host_runner_postprocess.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.poro_host_runner_state.argtypes = [C.c_void_p, C.c_int]
        L.poro_host_runner_free.argtypes = [C.c_void_p]
        L.poro_host_runner_free.restype = None
        _host = L
    return _host


def _arr_i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _arr_d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def read_input(path=None):
    """InputDataPoroel::read_input_file (InputDataPoroel.h:77-86); path=None gives the declared defaults."""
    out = InputFlat()
    if load_host().poro_host_read_input(path.encode() if path else None, C.byref(out)) != 0:
        raise RuntimeError(load_host().poro_host_last_error().decode())
    return out


class Problem:
    """Mesh + DoFs + FE tables + constraints = everything poro_desc points at (owned by the C++ host layer)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError(load_host().poro_host_last_error().decode())
        self.handle = C.c_void_p(handle)
        self.desc_ptr = load_host().poro_host_desc(self.handle)
        self.desc = self.desc_ptr.contents

    @staticmethod
    def _bc(dirichlet, neumann):
        dl, dc, dv = (list(x) for x in zip(*dirichlet)) if dirichlet else ([], [], [])
        nl, nc, nv = (list(x) for x in zip(*neumann)) if neumann else ([], [], [])
        keep = [_arr_i(dl), _arr_i(dc), _arr_d(dv), _arr_i(nl), _arr_i(nc), _arr_d(nv)]
        return keep, [len(dl), keep[0][1], keep[1][1], keep[2][1], len(nl), keep[3][1], keep[4][1], keep[5][1]]

    @classmethod
    def box(cls, dim, n, size, degree_u, material, dirichlet, neumann=(), rank=0, n_ranks=1):
        """hyper_rectangle(colorize) box of n[d] cells (PoroelasticityFSS.h:418-435); dirichlet / neumann = [(label, component, value)]."""
        keep, args = cls._bc(dirichlet, neumann)
        n3, pn = _arr_i(list(n) + [1] * (3 - len(n)))
        s3, ps = _arr_d(list(size) + [1.0] * (3 - len(size)))
        return cls(load_host().poro_host_build_box(dim, pn, ps, degree_u, rank, n_ranks, *args, C.byref(material)))

    @classmethod
    def graded_box(cls, dim, n, size, degree_u, material, dirichlet, grading, neumann=()):
        """the colorized box with exponentially graded vertex spacing (grading[d] = 0: uniform in that direction): rectilinear cells of different sizes, NO box tag -
        the general (unstructured) kernels run on it"""
        keep, args = cls._bc(dirichlet, neumann)
        n3, pn = _arr_i(list(n) + [1] * (3 - len(n)))
        s3, ps = _arr_d(list(size) + [1.0] * (3 - len(size)))
        g3, pg = _arr_d(list(grading) + [0.0] * (3 - len(grading)))
        return cls(load_host().poro_host_build_graded_box(dim, pn, ps, degree_u, pg, *args, C.byref(material)))

    @classmethod
    def refined_box(cls, dim, n, size, degree_u, material, dirichlet, refine_lo, refine_hi, neumann=()):
        """box of n[d] coarse cells whose cells with index in [refine_lo, refine_hi) are split once (2^dim children): a mesh with hanging nodes,
        as one refine_mesh() pass of the reference produces (PoroelasticityFSS.h:447-498); the constraint lists come with the descriptor"""
        keep, args = cls._bc(dirichlet, neumann)
        n3, pn = _arr_i(list(n) + [1] * (3 - len(n)))
        s3, ps = _arr_d(list(size) + [1.0] * (3 - len(size)))
        lo, plo = _arr_i(list(refine_lo) + [0] * (3 - len(refine_lo)))
        hi, phi = _arr_i(list(refine_hi) + [1] * (3 - len(refine_hi)))
        return cls(load_host().poro_host_build_refined_box(dim, pn, ps, degree_u, plo, phi, *args, C.byref(material)))

    @classmethod
    def gmsh(cls, path, degree_u, material, dirichlet, neumann=(), refine=0):
        """Gmsh 2.2 quadrilateral mesh (read_mesh, PoroelasticityFSS.h:438-445), optionally after `refine` uniform refinements.  Where the mesh fills a rectangle whose sides
        carry the colorized boundary ids the descriptor also gets an auxiliary uniform box as coarse space (poro_desc.coarse -> PREC_TWO_LEVEL)"""
        keep, args = cls._bc(dirichlet, neumann)
        if refine:
            return cls(load_host().poro_host_build_gmsh_refined(path.encode(), degree_u, int(refine), *args, C.byref(material)))
        return cls(load_host().poro_host_build_gmsh(path.encode(), degree_u, *args, C.byref(material)))

    def partition(self, rank, n_ranks):
        """piece `rank` of a general partition of this (global) problem: contiguous ranges of the cells in Morton order + interface lists
        (poro_partition.n_neighbours > 0); .local_to_global_u / _p map the piece's dofs back"""
        H = load_host()
        h = H.poro_host_partition(self.handle, rank, n_ranks)
        if not h:
            raise RuntimeError(H.poro_host_last_error().decode())
        P = type(self)(h)
        for name, space in (("local_to_global_u", 0), ("local_to_global_p", 1)):
            a = np.zeros(H.poro_host_local_to_global(h, space, None), dtype=np.int32)
            H.poro_host_local_to_global(h, space, a.ctypes.data_as(_ip))
            setattr(P, name, a)
        return P

    def tie_boundary(self, conditions):
        """extension: rigid frictionless plates [(boundary label, component)] - that displacement component takes one (unknown) value on the whole boundary;
        ordinary entries x[dof] = x[master] of the displacement constraint list (poro_desc.cons_u)"""
        lab, pl = _arr_i([c[0] for c in conditions]); comp, pc = _arr_i([c[1] for c in conditions])
        if load_host().poro_host_tie_boundary(self.handle, len(conditions), pl, pc) != 0:
            raise RuntimeError(load_host().poro_host_last_error().decode())
        self.desc = self.desc_ptr.contents
        return self

    def set_pressure_bc(self, conditions):
        """extension (the reference has no pressure boundary conditions): prescribed pressure [(boundary label, value)], e.g. a drained face p = 0"""
        lab, pl = _arr_i([c[0] for c in conditions]); val, pv = _arr_d([c[1] for c in conditions])
        if load_host().poro_host_set_pressure_bc(self.handle, len(conditions), pl, pv) != 0:
            raise RuntimeError(load_host().poro_host_last_error().decode())
        self.desc = self.desc_ptr.contents
        return self

    def close(self):
        if self.handle:
            load_host().poro_host_free(self.handle)
            self.handle = None

    def array(self, name, shape, dtype=np.float64):
        ptr = getattr(self.desc, name)
        return np.ctypeslib.as_array(ptr, shape=shape).copy() if int(np.prod(shape)) else np.zeros(shape, dtype)


class Context:
    """Handle to the device-resident solver state behind the C-ABI."""

    def __init__(self, problem, device=0, operator_mode=OP_CSR, ptr=None):
        self.L = load_hip()
        self.problem = problem
        self.n_u, self.n_p, self.dim = problem.desc.n_dofs_u, problem.desc.n_dofs_p, problem.desc.dim
        if ptr is None:
            p = C.c_void_p()
            self._chk(self.L.poro_ctx_create(problem.desc_ptr, device, operator_mode, C.byref(p)))
            ptr = p
        self.ptr = ptr
        self._cb = None

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self.L.poro_last_error().decode())
        return rc

    def close(self):
        if self.ptr:
            self.L.poro_ctx_destroy(self.ptr)
            self.ptr = None

    def synchronize(self):
        """wait for everything the context has enqueued (device-only entry points are stream-ordered)"""
        self._chk(self.L.poro_ctx_synchronize(self.ptr))

    def _len(self, which):
        return self.n_u if which in (VEC_U, VEC_RHS_U, VEC_DIAG_U) else self.n_p

    def set(self, which, arr):
        a, p = _arr_d(arr)
        self._chk(self.L.poro_vec_set(self.ptr, which, p, a.size))

    def get(self, which):
        out = np.empty(self._len(which))
        self._chk(self.L.poro_vec_get(self.ptr, which, out.ctypes.data_as(_dp), out.size))
        return out

    def fill(self, which, v):
        self._chk(self.L.poro_vec_fill(self.ptr, which, v))

    def copy(self, dst, src):
        self._chk(self.L.poro_vec_copy(self.ptr, dst, src))

    def axpy(self, y, a, x):
        self._chk(self.L.poro_vec_axpy(self.ptr, y, a, x))

    def norm(self, which):
        a, b = C.c_double(), C.c_double()
        self._chk(self.L.poro_vec_norm(self.ptr, which, C.byref(a), C.byref(b)))
        return a.value, b.value

    def state_save(self):
        self._chk(self.L.poro_state_save(self.ptr))

    def state_restore(self):
        self._chk(self.L.poro_state_restore(self.ptr))

    def disp_assemble_system(self, rebuild=True):
        self._chk(self.L.poro_disp_assemble_system(self.ptr, int(rebuild)))

    @staticmethod
    def _opts(abs_tol, rel_tol, max_iter, prec, omega=1.0, stop_rule=STOP_RHS, poly_degree=0):
        return SolverOpts(abs_tol, rel_tol, max_iter, prec, omega, stop_rule, poly_degree)

    def supports_preconditioner(self, which_system, prec):
        """which_system: 0 displacement, 1 pressure / projection."""
        return bool(self.L.poro_supports_preconditioner(self.ptr, which_system, prec))

    def disp_solve(self, abs_tol=1e-12, rel_tol=0.0, max_iter=1000, prec=PREC_JACOBI, omega=1.2, reduction=False, poly_degree=0):
        """omega: SSOR relaxation, or (PREC_CHEBYSHEV) the interval ratio lambda_max / a (<= 0: default); poly_degree: Chebyshev degree (0: default)"""
        info = SolveInfo()
        if prec == PREC_CHEBYSHEV and omega == 1.2:
            omega = 0.0
        rc = self._chk(self.L.poro_disp_solve(self.ptr, C.byref(self._opts(abs_tol, rel_tol, max_iter, prec, omega, STOP_REDUCTION if reduction else STOP_RHS, poly_degree)), C.byref(info)))
        return rc, info

    def pres_assemble_residual(self, dt):
        l2 = C.c_double()
        self._chk(self.L.poro_pres_assemble_residual(self.ptr, dt, C.byref(l2)))
        return l2.value

    def pres_assemble_jacobian(self, dt):
        self._chk(self.L.poro_pres_assemble_jacobian(self.ptr, dt))

    def pres_apply_boundary_values(self):
        self._chk(self.L.poro_pres_apply_boundary_values(self.ptr))

    def pres_solve(self, abs_tol=0.0, rel_tol=1e-8, max_iter=1000, prec=PREC_JACOBI, omega=1.0):
        info = SolveInfo()
        rc = self._chk(self.L.poro_pres_solve(self.ptr, C.byref(self._opts(abs_tol, rel_tol, max_iter, prec, omega)), C.byref(info)))
        return rc, info

    def pres_update_volumetric_strain(self):
        self._chk(self.L.poro_pres_update_volumetric_strain(self.ptr))

    def proj_assemble_matrix(self):
        self._chk(self.L.poro_proj_assemble_matrix(self.ptr))

    def proj_assemble_rhs(self, comps):
        a, p = _arr_i(comps)
        self._chk(self.L.poro_proj_assemble_rhs(self.ptr, p, a.size))

    def proj_solve(self, entry, abs_tol=0.0, rel_tol=1e-8, max_iter=1000, prec=PREC_JACOBI, omega=1.0):
        info = SolveInfo()
        rc = self._chk(self.L.poro_proj_solve(self.ptr, entry, C.byref(self._opts(abs_tol, rel_tol, max_iter, prec, omega)), C.byref(info)))
        return rc, info

    def proj_solve_many(self, entries, abs_tol=0.0, rel_tol=1e-8, max_iter=1000, prec=PREC_JACOBI, omega=1.0):
        """several projection systems in one call (solved directly and together where PREC_FDM is the exact inverse: info.iterations == 0)"""
        ent, pe = _arr_i(list(entries)); infos = (SolveInfo * len(entries))()
        rc = self._chk(self.L.poro_proj_solve_many(self.ptr, pe, len(entries), C.byref(self._opts(abs_tol, rel_tol, max_iter, prec, omega)), infos))
        return rc, list(infos)

    def get_volumetric_strain(self):
        self._chk(self.L.poro_get_volumetric_strain(self.ptr))

    def get_effective_stresses(self):
        self._chk(self.L.poro_get_effective_stresses(self.ptr))

    def export_csr(self, which):
        n, nnz = C.c_int64(), C.c_int64()
        self._chk(self.L.poro_export_csr_size(self.ptr, which, C.byref(n), C.byref(nnz)))
        rp, col, val = np.empty(n.value + 1, np.int64), np.empty(nnz.value, np.int32), np.empty(nnz.value)
        self._chk(self.L.poro_export_csr(self.ptr, which, rp.ctypes.data_as(C.POINTER(C.c_int64)), col.ctypes.data_as(_ip), val.ctypes.data_as(_dp)))
        return rp, col, val

    def apply(self, which, x):
        a, p = _arr_d(x)
        y = np.empty_like(a)
        self._chk(self.L.poro_apply_operator(self.ptr, which, p, y.ctypes.data_as(_dp)))
        return y

    def apply_preconditioner_u(self, prec, g, reps=0):
        """z = P^-1 g with the displacement preconditioner; returns z (and the mean device seconds per application when reps > 0)"""
        a, p = _arr_d(g)
        z = np.empty_like(a)
        t = C.c_double(0.0)
        self._chk(self.L.poro_apply_preconditioner_u(self.ptr, prec, p, z.ctypes.data_as(_dp), reps, C.byref(t)))
        return (z, t.value) if reps > 0 else z

    def bench_operator(self, operator_mode, reps):
        t = C.c_double()
        self._chk(self.L.poro_bench_operator(self.ptr, MAT_A_U, operator_mode, reps, C.byref(t)))
        return t.value

    def timers_reset(self):
        self._chk(self.L.poro_timers_reset(self.ptr))

    def timers_enable(self, on):
        self._chk(self.L.poro_timers_enable(self.ptr, int(on)))

    def timer(self, name):
        s, n = C.c_double(), C.c_int64()
        self._chk(self.L.poro_timers_get(self.ptr, name.encode(), C.byref(s), C.byref(n)))
        return s.value, n.value

    def comm_callbacks(self, allreduce, sendrecv):
        """host-staged communicator (tests); allreduce(np_view) sums in place, sendrecv(send_view, recv_view, peer)."""
        def _ar(buf, n, _u):
            allreduce(np.ctypeslib.as_array(buf, shape=(n,)))

        def _sr(send, recv, n, peer, _u):
            sendrecv(np.ctypeslib.as_array(send, shape=(n,)), np.ctypeslib.as_array(recv, shape=(n,)), peer)
        self._cb = (ALLREDUCE_FN(_ar), SENDRECV_FN(_sr))
        self._chk(self.L.poro_ctx_comm_init_callbacks(self.ptr, self._cb[0], self._cb[1], None))

    def comm_rccl(self, unique_id_bytes):
        buf = C.create_string_buffer(bytes(unique_id_bytes), 128)
        self._chk(self.L.poro_ctx_comm_init_rccl(self.ptr, buf))


def rccl_unique_id():
    buf = C.create_string_buffer(128)
    L = load_hip()
    if L.poro_comm_unique_id(buf) != 0:
        raise RuntimeError(L.poro_last_error().decode())
    return buf.raw


def run_problem(problem, n_steps, p_init, dt, device=0, operator_mode=OP_CSR, fss_tol=1e-8, pressure_tol=1e-8, max_fss=50, max_pres=50,
                abs_u=1e-12, rel_u=0.0, max_it=1000, prec=PREC_JACOBI, coupled_fss=False, incremental_strain=False, reduction=False, cheb_degree=0, cheb_ratio=0, jacobi_p=False, two_level_p=False):
    """PoroElasticProblem<dim>::run() (PoroelasticityFSS.h:294-415) through the C++ host driver; returns (trace, Context)."""
    H = load_host()
    max_rows = 1 + n_steps * max_fss
    trace = np.zeros((max_rows, 8))
    ctx = C.c_void_p()
    rows = H.poro_host_run(problem.handle, device, operator_mode, p_init, dt, n_steps, fss_tol, pressure_tol, max_fss, max_pres, abs_u, rel_u, max_it, prec, int(bool(coupled_fss)) | (2 if incremental_strain else 0) | (4 if reduction else 0) | (8 if jacobi_p else 0) | (16 if two_level_p else 0) | (int(cheb_degree) << 8) | (int(cheb_ratio) << 16),
                           trace.ctypes.data_as(_dp), max_rows, C.byref(ctx))
    if rows < 0:
        raise RuntimeError(H.poro_host_last_error().decode())
    return trace[:rows], Context(problem, ptr=ctx)


WORK_FIELDS = ("apply_u", "apply_p", "asm_rhs_u", "asm_matrix_u", "residual_p", "jacobian_p", "proj_rhs", "cg_u", "cg_p", "cg_proj", "usec_solve_u")


class Runner:
    """Steppable PoroElasticProblem<dim> (C++ host driver): initialize() = PoroelasticityFSS.h:308-317, step() = one pass of :328-407."""

    def __init__(self, problem, device=0, operator_mode=OP_MATRIX_FREE, p_init=10e6, dt=60.0, fss_tol=1e-8, pressure_tol=1e-8, max_fss=50, max_pres=50,
                 abs_u=1e-12, rel_u=0.0, max_it=1000, prec=PREC_JACOBI, coupled_fss=False, incremental_strain=False, reduction=False, cheb_degree=0, cheb_ratio=0, jacobi_p=False, two_level_p=False):
        self.H = load_host()
        self.max_fss = max_fss
        h = self.H.poro_host_runner_create(problem.handle, device, operator_mode, p_init, dt, fss_tol, pressure_tol, max_fss, max_pres, abs_u, rel_u, max_it, prec, int(bool(coupled_fss)) | (2 if incremental_strain else 0) | (4 if reduction else 0) | (8 if jacobi_p else 0) | (16 if two_level_p else 0) | (int(cheb_degree) << 8) | (int(cheb_ratio) << 16))
        if not h:
            raise RuntimeError(self.H.poro_host_last_error().decode())
        self.h = C.c_void_p(h)
        self.ctx = Context(problem, ptr=C.c_void_p(self.H.poro_host_runner_ctx(self.h)))

    def initialize(self):
        if self.H.poro_host_runner_initialize(self.h) != 0:
            raise RuntimeError(self.H.poro_host_last_error().decode())

    def step(self, restore=False):
        """one time step; restore=True: roll the device state back to the snapshot first (same call)"""
        trace = np.zeros((self.max_fss, 8))
        work = (C.c_int64 * 11)()
        rows = (self.H.poro_host_runner_restore_and_step if restore else self.H.poro_host_runner_step)(self.h, trace.ctypes.data_as(_dp), self.max_fss, work)
        if rows < 0:
            raise RuntimeError(self.H.poro_host_last_error().decode())
        return trace[:rows], dict(zip(WORK_FIELDS, list(work)))

    def save_state(self):
        """device-side snapshot of every solver vector (and the step counter)"""
        if self.H.poro_host_runner_state(self.h, 0) != 0:
            raise RuntimeError(self.H.poro_host_last_error().decode())

    def restore_state(self):
        if self.H.poro_host_runner_state(self.h, 1) != 0:
            raise RuntimeError(self.H.poro_host_last_error().decode())

    def postprocess(self, output_dir=None, corrected=False):
        """PoroelasticityFSS.h:409-411: shear strains, effective stresses (PORO_VEC_STRESS0+e) and, with a directory, solution-NNNN.vtk"""
        if self.H.poro_host_runner_postprocess(self.h, output_dir.encode() if output_dir else None, int(corrected)) < 0:
            raise RuntimeError(self.H.poro_host_last_error().decode())

    def work(self):
        """cumulative work counters since creation"""
        work = (C.c_int64 * 11)()
        self.H.poro_host_runner_work(self.h, work)
        return dict(zip(WORK_FIELDS, list(work)))

    def close(self):
        if self.h:
            self.ctx.ptr = None          # owned by the runner
            self.H.poro_host_runner_free(self.h)
            self.h = None
