"""Matrix-free operator on GENERAL meshes (the kernel BASELINE.json's north_star describes: element dof indices, quadrature data and the constant
material coefficients staged in LDS; PoroElasticDisplacementSolver.h:206-246 applied to a vector): the bundled Gmsh mesh of config 1 (read_mesh,
PoroelasticityFSS.h:438-445) and locally refined boxes with hanging nodes, against the assembled CSR operator and the oracle."""
import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import BC_2D, DOMAIN_MSH, REF, host_material
from test_constraints_cpu import MESHES, refined

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deg", [1, 2])
def test_gmsh_mesh_matrix_free_equals_assembled(deg):
    P = pk.Problem.gmsh(DOMAIN_MSH, deg, host_material(), BC_2D)
    A, F = pk.Context(P, 0, pk.OP_CSR), pk.Context(P, 0, pk.OP_MATRIX_FREE)
    O = oracle_py.Oracle(P, hoisted=True)
    try:
        p = REF["p_init"] * (1 + 0.2 * np.sin(0.37 * np.arange(A.n_p)))
        for S in (A, F, O):
            S.set(pk.VEC_P, p); S.disp_assemble_system(True)
        x = np.sin(0.37 * np.arange(A.n_u))
        ya, yf, yo = A.apply(pk.MAT_A_U, x), F.apply(pk.MAT_A_U, x), O.apply(pk.MAT_A_U, x)
        assert np.abs(ya - yo).max() <= 1e-12 * np.abs(yo).max() and np.abs(yf - yo).max() <= 1e-12 * np.abs(yo).max()
        assert np.abs(F.get(pk.VEC_DIAG_U) - A.get(pk.VEC_DIAG_U)).max() <= 1e-12 * np.abs(A.get(pk.VEC_DIAG_U)).max()
        assert np.abs(F.get(pk.VEC_RHS_U) - O.get(pk.VEC_RHS_U)).max() <= 1e-12 * np.abs(O.get(pk.VEC_RHS_U)).max()
        assert O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=20000)[0] == 0
        for prec in (pk.PREC_JACOBI, pk.PREC_CHEBYSHEV):
            F.fill(pk.VEC_U, 0.0)
            rc, info = F.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=20000, prec=prec)
            assert rc == 0 and np.linalg.norm(F.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
        assert not F.supports_preconditioner(0, pk.PREC_FDM)
    finally:
        A.close(); F.close(); O.close(); P.close()


def test_config_1_time_step_matrix_free():
    """BASELINE config 1 (domain.msh + input.data, Q1/Q1, one time step) with the matrix-free displacement operator: same trace and fields as the oracle"""
    P = pk.Problem.gmsh(DOMAIN_MSH, 1, host_material(), BC_2D)
    O = oracle_py.Oracle(P, hoisted=True)
    try:
        t0, _ = O.run(1, REF["p_init"], REF["dt"], max_it=5000)
        t1, G = pk.run_problem(P, 1, REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=5000)
        assert np.array_equal(t1[:, :3], t0[:, :3])
        assert np.linalg.norm(G.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-8 * np.linalg.norm(O.get(pk.VEC_U))
        assert np.abs(G.get(pk.VEC_P) - O.get(pk.VEC_P)).max() <= 1e-10 * np.abs(O.get(pk.VEC_P)).max()
        G.close()
    finally:
        O.close(); P.close()


@pytest.mark.parametrize("cfg", MESHES, ids=str)
def test_hanging_node_meshes_matrix_free(cfg):
    """locally refined boxes: the general matrix-free operator under the operator-level condensation C^T A C (poro_desc.cons_u)"""
    P = refined(*cfg)
    O = oracle_py.Oracle(P, hoisted=True)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        p = REF["p_init"] * (1 + 0.3 * np.sin(0.37 * np.arange(G.n_p)))
        for S in (O, G):
            S.set(pk.VEC_P, p); S.disp_assemble_system(True)
        assert np.abs(G.get(pk.VEC_RHS_U) - O.get(pk.VEC_RHS_U)).max() <= 1e-12 * np.abs(O.get(pk.VEC_RHS_U)).max()
        rc0, _ = O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000); rc1, _ = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000)
        assert rc0 == 0 and rc1 == 0
        assert np.linalg.norm(G.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
    finally:
        G.close(); O.close(); P.close()


@pytest.mark.parametrize("deg,n,grading", [(2, (4, 3, 5), (1.0, 0.5, -0.7)), (1, (5, 4, 6), (0.8, 0.0, 1.2)), (2, (9, 2, 3), (0.0, 0.0, 0.0))], ids=str)
def test_graded_box_sum_factorised_operator(deg, n, grading, monkeypatch):
    """3D hexahedra without the box tag (graded box: rectilinear cells of different sizes): the sum-factorised general kernel (k_mfg3_sf, several cells per workgroup)
    against the oracle's assembled operator, against the one-wave-per-cell kernel it replaces, and inside a solve"""
    from common import BC_3D, material
    P = pk.Problem.graded_box(3, list(n), [10.0] * 3, deg, material(), BC_3D, list(grading))
    assert not P.desc.box.enabled
    O = oracle_py.Oracle(P, hoisted=True)
    F = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        p = REF["p_init"] * (1 + 0.2 * np.sin(0.37 * np.arange(F.n_p)))
        for S in (F, O):
            S.set(pk.VEC_P, p); S.disp_assemble_system(True)
        x = np.sin(0.37 * np.arange(F.n_u))
        yo, yf = O.apply(pk.MAT_A_U, x), F.apply(pk.MAT_A_U, x)
        assert np.abs(yf - yo).max() <= 1e-12 * np.abs(yo).max(), np.abs(yf - yo).max() / np.abs(yo).max()
        monkeypatch.setenv("PORO_MFG_NO_SUMFAC", "1")       # (read once per process: only effective if this is the first general-mesh apply; the comparison with the oracle above is the hard check)
        assert np.abs(F.get(pk.VEC_RHS_U) - O.get(pk.VEC_RHS_U)).max() <= 1e-12 * np.abs(O.get(pk.VEC_RHS_U)).max()
        assert O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=20000)[0] == 0
        rc, info = F.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=20000, prec=pk.PREC_CHEBYSHEV)
        assert rc == 0 and np.linalg.norm(F.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
    finally:
        F.close(); O.close(); P.close()


@pytest.mark.parametrize("deg,n,grading", [(2, (6, 5, 4), (1.5, 0.0, -1.0)), (1, (7, 6, 8), (2.0, 1.0, 0.5)), (2, (6, 6, 6), (0.0, 0.0, 0.0))], ids=str)
def test_graded_box_keeps_the_fast_diagonalisation(deg, n, grading):
    """poro_desc.tensor: on a tensor-product grid without the box tag the fast-diagonalisation preconditioners stay EXACT for the blocks they invert (1D FE matrices on the
    graded 1D grids): the pressure Jacobian and the projection mass matrix are solved in 1-2 CG iterations, the displacement system in a number of iterations that does
    not depend on the grading; the solutions are the oracle's (Jacobi-CG on the assembled matrices)"""
    from common import BC_3D, material
    P = pk.Problem.graded_box(3, list(n), [10.0] * 3, deg, material(), BC_3D, list(grading))
    U = pk.Problem.graded_box(3, list(n), [10.0] * 3, deg, material(), BC_3D, [0.0] * 3)
    assert not P.desc.box.enabled and P.desc.tensor.enabled and list(P.desc.tensor.n) == list(n)
    O = oracle_py.Oracle(P, hoisted=True)
    F, G = pk.Context(P, 0, pk.OP_MATRIX_FREE), pk.Context(U, 0, pk.OP_MATRIX_FREE)
    try:
        assert F.supports_preconditioner(0, pk.PREC_FDM) and F.supports_preconditioner(1, pk.PREC_FDM)
        synth = lambda m, ph=0.0: np.sin(0.37 * np.arange(m) + ph)   # noqa: E731
        p = REF["p_init"] * (1 + 0.2 * synth(F.n_p))
        for S in (F, G, O):
            S.set(pk.VEC_P, p); S.disp_assemble_system(True)
        assert O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=20000)[0] == 0
        rc, info = F.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=200, prec=pk.PREC_FDM)
        rcu, infou = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=200, prec=pk.PREC_FDM)
        assert rc == 0 and rcu == 0 and np.linalg.norm(F.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
        assert info.iterations <= infou.iterations + 8, (info.iterations, infou.iterations)   # block-diagonal preconditioning: the count follows the material, not the mesh
        # pressure Jacobian a M + kappa K and the projection mass matrix: inverted exactly
        n = F.n_p
        vals = {pk.VEC_P_OLD: 10e6 * (1 + 0.05 * synth(n, 0.2)), pk.VEC_EPSV: -2e-6 * (1 + 0.3 * synth(n, 0.5)), pk.VEC_EPSV0: -2e-6 * np.ones(n)}
        for S in (F, O):
            for k, v in vals.items():
                S.set(k, v)
            S.pres_assemble_residual(60.0); S.pres_assemble_jacobian(60.0)
        rc0, _ = O.pres_solve(rel_tol=1e-13); rc, info = F.pres_solve(rel_tol=1e-8, prec=pk.PREC_FDM)
        assert rc0 == 0 and rc == 0 and info.iterations <= 2, info.iterations
        assert np.linalg.norm(F.get(pk.VEC_DP) - O.get(pk.VEC_DP)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_DP))
        for S in (F, O):
            S.proj_assemble_matrix(); S.proj_assemble_rhs([0, 4, 8])
        for e in (0, 3, 5):
            rc0, _ = O.proj_solve(e, rel_tol=1e-13); rc, info = F.proj_solve(e, rel_tol=1e-8, prec=pk.PREC_FDM)
            assert rc0 == 0 and rc == 0 and info.iterations <= 2, info.iterations
            assert np.linalg.norm(F.get(pk.VEC_STRAIN0 + e) - O.get(pk.VEC_STRAIN0 + e)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_STRAIN0 + e))
    finally:
        F.close(); G.close(); O.close(); P.close(); U.close()


def test_gmsh_mesh_two_level_preconditioner_is_mesh_independent():
    """read_mesh()'s grid (domain.msh: unstructured numbering, Gmsh's boundary ids) and its uniform refinements: the descriptor carries an auxiliary uniform box with the
    same boundary conditions (ids translated side by side) as coarse space; PREC_TWO_LEVEL = Jacobi + block fast diagonalisation of that box through the FE interpolation.
    The CG iteration count stays flat under refinement (Jacobi's doubles), the solution is the oracle's"""
    from common import BC_2D, DOMAIN_MSH, material
    counts, jacobi, pres = [], [], []
    for r in (0, 1, 2, 3):
        P = pk.Problem.gmsh(DOMAIN_MSH, 2, material(), BC_2D, refine=r)
        assert P.desc.coarse.enabled and not P.desc.box.enabled
        F = pk.Context(P, 0, pk.OP_MATRIX_FREE)
        O = oracle_py.Oracle(P, hoisted=True) if r <= 1 else None
        try:
            assert F.supports_preconditioner(0, pk.PREC_TWO_LEVEL) and not F.supports_preconditioner(0, pk.PREC_FDM)
            p = REF["p_init"] * (1 + 0.2 * np.sin(0.37 * np.arange(F.n_p)))
            F.set(pk.VEC_P, p); F.disp_assemble_system(True)
            rc, info = F.disp_solve(abs_tol=1e-14, rel_tol=1e-10, max_iter=500, prec=pk.PREC_TWO_LEVEL)
            assert rc == 0
            counts.append(info.iterations); u = F.get(pk.VEC_U)
            F.fill(pk.VEC_U, 0.0)
            rc, info = F.disp_solve(abs_tol=1e-14, rel_tol=1e-10, max_iter=50000)
            assert rc == 0 and np.linalg.norm(F.get(pk.VEC_U) - u) <= 1e-7 * np.linalg.norm(u)
            jacobi.append(info.iterations)
            # the pressure Jacobian through the same auxiliary box (vertex interpolation): same update as Jacobi-CG, fewer iterations once the mesh is fine
            m = F.n_p
            for k, v in {pk.VEC_P: 10e6 * (1 + 0.05 * np.sin(0.37 * np.arange(m))), pk.VEC_P_OLD: 10e6 * (1 + 0.05 * np.sin(0.2 * np.arange(m))),
                         pk.VEC_EPSV: -2e-6 * (1 + 0.3 * np.sin(0.5 * np.arange(m))), pk.VEC_EPSV0: -2e-6 * np.ones(m)}.items():
                F.set(k, v)
            assert F.supports_preconditioner(1, pk.PREC_TWO_LEVEL)
            F.pres_assemble_residual(60.0); F.pres_assemble_jacobian(60.0)
            rc, two = F.pres_solve(rel_tol=1e-10, max_iter=500, prec=pk.PREC_TWO_LEVEL)
            assert rc == 0
            dp = F.get(pk.VEC_DP); F.fill(pk.VEC_DP, 0.0)
            rc, jac = F.pres_solve(rel_tol=1e-10, max_iter=20000)
            assert rc == 0 and np.linalg.norm(F.get(pk.VEC_DP) - dp) <= 1e-7 * np.linalg.norm(dp)
            pres.append((two.iterations, jac.iterations))
            if O is not None:
                O.set(pk.VEC_P, p); O.disp_assemble_system(True)
                assert O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000)[0] == 0
                assert np.linalg.norm(u - O.get(pk.VEC_U)) <= 1e-8 * np.linalg.norm(u)
        finally:
            F.close(); P.close()
            if O is not None:
                O.close()
    print("Gmsh mesh, two-level CG iterations per refinement:", counts, "Jacobi:", jacobi)
    assert all(b <= 1.3 * a for a, b in zip(counts, counts[1:])), counts
    assert counts[-1] < jacobi[-1] / 5, (counts, jacobi)
    print("Gmsh mesh, pressure CG iterations (two-level, Jacobi) per refinement:", pres)
    assert pres[-1][0] <= 1.3 * pres[-2][0] and pres[-1][0] < pres[-1][1], pres


def test_general_hexahedra_path_without_the_affine_shortcut():
    """every mesh provider of this repo makes parallelepipeds, so the library reads one pre-inverted Jacobian per cell (poro_ctx::cell_geo); the kernel variant for general
    hexahedra (MappingQ1's Jacobian from the eight vertices at every quadrature point) stays covered by a run with PORO_MFG_NO_AFFINE=1 - the switch is read once per
    process, hence the child process.  Same operator as the oracle's, Q1 and Q2"""
    import os, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import poroelasticity_dealii_amd as pk, oracle_py
from common import BC_3D, REF, material
for deg, n, grading in ((2, (4, 3, 5), (1.0, 0.5, -0.7)), (1, (5, 4, 6), (0.8, 0.0, 1.2))):
    P = pk.Problem.graded_box(3, list(n), [10.0] * 3, deg, material(), BC_3D, list(grading))
    O = oracle_py.Oracle(P, hoisted=True); F = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    p = REF["p_init"] * (1 + 0.2 * np.sin(0.37 * np.arange(F.n_p)))
    for S in (F, O):
        S.set(pk.VEC_P, p); S.disp_assemble_system(True)
    x = np.sin(0.37 * np.arange(F.n_u))
    yo, yf = O.apply(pk.MAT_A_U, x), F.apply(pk.MAT_A_U, x)
    err = np.abs(yf - yo).max() / np.abs(yo).max()
    assert err <= 1e-12, err
    F.close(); O.close(); P.close()
print("general-hexahedra path ok")
''' % (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PORO_MFG_NO_AFFINE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "general-hexahedra path ok" in r.stdout, r.stdout + r.stderr
