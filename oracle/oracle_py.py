"""ctypes loader of the CPU oracle (oracle/poro_oracle.cpp).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
PREC_NONE, PREC_JACOBI, PREC_SSOR = 0, 1, 2
_lib = None


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("initial_residual", C.c_double), ("final_residual", C.c_double),
                ("seconds", C.c_double), ("operator_applications", C.c_int64)]


ALLREDUCE_FN = C.CFUNCTYPE(None, _dp, C.c_int32, C.c_void_p)
SENDRECV_FN = C.CFUNCTYPE(None, _dp, _dp, C.c_int64, C.c_int32, C.c_void_p)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_destroy.restype = None
        L.oracle_set_hoisted.argtypes = [C.c_void_p, C.c_int]
        L.oracle_bench_spmv_threads.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.oracle_set_hoisted.restype = None
        L.oracle_set_comm.argtypes = [C.c_void_p, ALLREDUCE_FN, SENDRECV_FN, C.c_void_p]
        L.oracle_set_comm.restype = None
        L.oracle_vec_set.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64]
        L.oracle_vec_get.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64]
        L.oracle_vec_fill.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.oracle_disp_assemble_system.argtypes = [C.c_void_p, C.c_int]
        slv = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.POINTER(SolveInfo)]
        L.oracle_disp_solve.argtypes = [C.c_void_p] + slv
        L.oracle_pres_assemble_residual.argtypes = [C.c_void_p, C.c_double, _dp]
        L.oracle_pres_assemble_jacobian.argtypes = [C.c_void_p, C.c_double]
        L.oracle_pres_solve.argtypes = [C.c_void_p] + slv
        L.oracle_pres_update_volumetric_strain.argtypes = [C.c_void_p]
        L.oracle_proj_assemble_matrix.argtypes = [C.c_void_p]
        L.oracle_proj_assemble_rhs.argtypes = [C.c_void_p, _ip, C.c_int32]
        L.oracle_proj_solve.argtypes = [C.c_void_p, C.c_int32] + slv
        L.oracle_get_volumetric_strain.argtypes = [C.c_void_p]
        L.oracle_noconvergence_count.argtypes = [C.c_void_p]
        L.oracle_set_stop_rule.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_stop_rule.restype = None
        L.oracle_work_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        L.oracle_work_counts.restype = None
        L.oracle_last_run_split.argtypes = [C.c_void_p, C.POINTER(C.c_int64), _dp]
        L.oracle_last_run_split.restype = None
        L.oracle_export_csr_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.oracle_export_csr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), _ip, _dp]
        L.oracle_apply_operator.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.oracle_fe_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.oracle_derived_parameters.argtypes = [C.c_double] * 7 + [_dp]
        L.oracle_derived_parameters.restype = None
        L.oracle_run.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                 _dp, C.c_int, _dp, C.c_int]
        _lib = L
    return _lib


def derived_parameters(E, nu, alpha, poro, f_comp, perm_mD, visc):
    out = np.zeros(7)
    load().oracle_derived_parameters(E, nu, alpha, poro, f_comp, perm_mD, visc, out.ctypes.data_as(_dp))
    return dict(zip(["lambda", "G", "K", "Ks", "N", "M", "k_over_mu"], out))


def fe_table(dim, k, n1d_quad, what):
    nq, ns = n1d_quad ** dim, (k + 1) ** dim
    out = np.zeros({0: nq * ns, 1: nq * ns * dim, 2: nq}[what])
    load().oracle_fe_table(dim, k, n1d_quad, what, out.ctypes.data_as(_dp))
    return out


class Oracle:
    def __init__(self, problem, hoisted=False):
        self.L = load()
        self.problem = problem
        self.n_u, self.n_p, self.dim = problem.desc.n_dofs_u, problem.desc.n_dofs_p, problem.desc.dim
        p = C.c_void_p()
        if self.L.oracle_create(problem.desc_ptr, C.byref(p)) != 0:
            raise RuntimeError("oracle_create failed")
        self.ptr = p
        self.L.oracle_set_hoisted(self.ptr, int(hoisted))
        self._cb = None

    def close(self):
        if self.ptr:
            self.L.oracle_destroy(self.ptr)
            self.ptr = None

    def _len(self, which):
        return self.n_u if which in (0, 1) else self.n_p

    def set(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert self.L.oracle_vec_set(self.ptr, which, a.ctypes.data_as(_dp), a.size) == 0

    def get(self, which):
        out = np.empty(self._len(which))
        assert self.L.oracle_vec_get(self.ptr, which, out.ctypes.data_as(_dp), out.size) == 0
        return out

    def fill(self, which, v):
        assert self.L.oracle_vec_fill(self.ptr, which, v) == 0

    def disp_assemble_system(self, rebuild=True):
        self.L.oracle_disp_assemble_system(self.ptr, int(rebuild))

    def disp_solve(self, abs_tol=1e-12, rel_tol=0.0, max_iter=1000, prec=PREC_SSOR, omega=1.2, reduction=False):
        info = SolveInfo()
        self.L.oracle_set_stop_rule(self.ptr, int(bool(reduction)))
        rc = self.L.oracle_disp_solve(self.ptr, abs_tol, rel_tol, max_iter, prec, omega, C.byref(info))
        return rc, info

    def pres_assemble_residual(self, dt):
        l2 = C.c_double()
        self.L.oracle_pres_assemble_residual(self.ptr, dt, C.byref(l2))
        return l2.value

    def pres_assemble_jacobian(self, dt):
        self.L.oracle_pres_assemble_jacobian(self.ptr, dt)

    def pres_solve(self, abs_tol=0.0, rel_tol=1e-8, max_iter=1000, prec=PREC_SSOR, omega=1.0):
        info = SolveInfo()
        rc = self.L.oracle_pres_solve(self.ptr, abs_tol, rel_tol, max_iter, prec, omega, C.byref(info))
        return rc, info

    def pres_update_volumetric_strain(self):
        self.L.oracle_pres_update_volumetric_strain(self.ptr)

    def proj_assemble_matrix(self):
        self.L.oracle_proj_assemble_matrix(self.ptr)

    def proj_assemble_rhs(self, comps):
        a = np.ascontiguousarray(comps, dtype=np.int32)
        self.L.oracle_proj_assemble_rhs(self.ptr, a.ctypes.data_as(_ip), a.size)

    def proj_solve(self, entry, abs_tol=0.0, rel_tol=1e-8, max_iter=1000, prec=PREC_SSOR, omega=1.0):
        info = SolveInfo()
        rc = self.L.oracle_proj_solve(self.ptr, entry, abs_tol, rel_tol, max_iter, prec, omega, C.byref(info))
        return rc, info

    def get_volumetric_strain(self):
        self.L.oracle_get_volumetric_strain(self.ptr)

    def work_counts(self, reset=False):
        out = (C.c_int64 * 6)()
        self.L.oracle_work_counts(self.ptr, out, int(reset))
        return dict(zip(("apply_u", "apply_p", "asm_rhs_u", "residual_p", "jacobian_p", "proj_rhs"), list(out)))

    def last_run_split(self):
        """(work counters at the end of the last run()'s initialisation, seconds of the initialisation, seconds of its time steps)"""
        w, t = (C.c_int64 * 6)(), (C.c_double * 2)()
        self.L.oracle_last_run_split(self.ptr, w, t)
        return dict(zip(("apply_u", "apply_p", "asm_rhs_u", "residual_p", "jacobian_p", "proj_rhs"), list(w))), t[0], t[1]

    def fill_synthetic_matrix(self):
        """benchmark helper: SPD values on the real pattern of A_u (see oracle_fill_synthetic_matrix)"""
        assert self.L.oracle_fill_synthetic_matrix(self.ptr) == 0

    def bench_spmv_threads(self, threads, reps=20):
        """(seconds per CSR SpMV of A_u, seconds per Jacobi-CG iteration) with the rows split over `threads` host threads"""
        out = (C.c_double * 2)()
        assert self.L.oracle_bench_spmv_threads(self.ptr, int(threads), int(reps), out) == 0
        return out[0], out[1]

    def noconvergence_count(self):
        return self.L.oracle_noconvergence_count(self.ptr)

    def export_csr(self, which):
        n, nnz = C.c_int64(), C.c_int64()
        assert self.L.oracle_export_csr_size(self.ptr, which, C.byref(n), C.byref(nnz)) == 0
        rp, col, val = np.empty(n.value + 1, np.int64), np.empty(nnz.value, np.int32), np.empty(nnz.value)
        self.L.oracle_export_csr(self.ptr, which, rp.ctypes.data_as(C.POINTER(C.c_int64)), col.ctypes.data_as(_ip), val.ctypes.data_as(_dp))
        return rp, col, val

    def apply(self, which, x):
        a = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(a)
        assert self.L.oracle_apply_operator(self.ptr, which, a.ctypes.data_as(_dp), y.ctypes.data_as(_dp)) == 0
        return y

    def comm_callbacks(self, allreduce, sendrecv):
        def _ar(buf, n, _u):
            allreduce(np.ctypeslib.as_array(buf, shape=(n,)))

        def _sr(send, recv, n, peer, _u):
            sendrecv(np.ctypeslib.as_array(send, shape=(n,)), np.ctypeslib.as_array(recv, shape=(n,)), peer)
        self._cb = (ALLREDUCE_FN(_ar), SENDRECV_FN(_sr))
        self.L.oracle_set_comm(self.ptr, self._cb[0], self._cb[1], None)

    def run(self, n_steps, p_init, dt, fss_tol=1e-8, pressure_tol=1e-8, max_fss=50, max_pres=50, abs_u=1e-12, rel_u=0.0, max_it=1000, prec=PREC_SSOR, coupled_fss=False, incremental_strain=False, reduction=False):
        """PoroElasticProblem<dim>::run() restatement; returns (trace[rows,8], seconds_per_phase[4]).  reduction: the displacement solve stops at
        rel_u x its initial residual (PORO_STOP_REDUCTION) instead of rel_u x ||b||."""
        max_rows = 1 + n_steps * max_fss
        trace, tph = np.zeros((max_rows, 8)), np.zeros(4)
        rows = self.L.oracle_run(self.ptr, p_init, dt, n_steps, fss_tol, pressure_tol, max_fss, max_pres, abs_u, rel_u, max_it, prec,
                                 trace.ctypes.data_as(_dp), max_rows, tph.ctypes.data_as(_dp), int(bool(coupled_fss)) | (2 if incremental_strain else 0) | (4 if reduction else 0))
        return trace[:rows], tph
