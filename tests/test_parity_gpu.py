"""Parity of the HIP path (through the C-ABI) against the CPU oracle on the same inputs — SURVEY 8c K8.
Tolerances (fp64, stated per check): assembled values 1e-13 relative to the matrix max (summation order
differs), residual / rhs vectors 1e-12 relative, converged solutions 1e-8 relative in l2 (both sides solve to
a recursive residual; the oracle with the reference's SSOR-CG, the device with Jacobi-CG)."""
import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import BC_2D, BC_3D, DOMAIN_MSH, INPUT_DATA, REF, box_problem, csr_to_scipy, host_material, material, node_coords_box

pytestmark = pytest.mark.gpu

CASES = [(2, 8, 1), (2, 8, 2), (2, (7, 5), 2), (3, 3, 1), (3, 3, 2), (3, (4, 2, 3), 2)]


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def rel2(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def synth(n, k=0.37):
    return np.sin(k * np.arange(n))


@pytest.fixture(params=CASES, ids=lambda c: f"{c[0]}d-n{c[1]}-Q{c[2]}")
def pair(request):
    dim, n, deg = request.param
    P = box_problem(dim, n, deg)
    O = oracle_py.Oracle(P, hoisted=True)
    G = pk.Context(P, 0, pk.OP_CSR)
    yield P, O, G
    G.close(); O.close(); P.close()


def test_pressure_matrices_and_source(pair):
    P, O, G = pair
    for which in (pk.MAT_MASS_P, pk.MAT_LAPLACE_P):
        rp, col, val = G.export_csr(which); rp0, col0, val0 = O.export_csr(which)
        assert np.array_equal(rp, rp0) and np.array_equal(col, col0)
        assert np.abs(val - val0).max() <= 1e-13 * np.abs(val0).max()
    O.fill(pk.VEC_P, 1.0); O.pres_assemble_residual(60.0)
    assert rel(G.get(pk.VEC_SOURCE_P), O.get(pk.VEC_SOURCE_P)) <= 1e-13


def test_displacement_matrix_and_rhs(pair):
    P, O, G = pair
    p = 10e6 * (1 + 0.1 * synth(G.n_p))
    O.set(pk.VEC_P, p); G.set(pk.VEC_P, p)
    O.disp_assemble_system(True); G.disp_assemble_system(True)
    rp, col, val = G.export_csr(pk.MAT_A_U); rp0, col0, val0 = O.export_csr(pk.MAT_A_U)
    assert np.array_equal(rp, rp0) and np.array_equal(col, col0)
    assert np.abs(val - val0).max() <= 1e-13 * np.abs(val0).max()
    A = csr_to_scipy(rp, col, val)
    assert abs(A - A.T).max() <= 1e-13 * np.abs(val).max()
    assert rel(G.get(pk.VEC_RHS_U), O.get(pk.VEC_RHS_U)) <= 1e-12
    # rhs-only path (rebuild_system_matrix == false, PoroElasticDisplacementSolver.h:284-286) with a new pressure
    p2 = p * 1.01
    O.set(pk.VEC_P, p2); G.set(pk.VEC_P, p2)
    O.disp_assemble_system(False); G.disp_assemble_system(False)
    assert rel(G.get(pk.VEC_RHS_U), O.get(pk.VEC_RHS_U)) <= 1e-12
    x = synth(G.n_u, 0.11)
    assert rel(G.apply(pk.MAT_A_U, x), O.apply(pk.MAT_A_U, x)) <= 1e-13


def test_matrix_free_operator_equals_assembled(pair):
    P, O, G = pair
    p = 10e6 * (1 + 0.1 * synth(G.n_p))
    O.set(pk.VEC_P, p); O.disp_assemble_system(True)
    F = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        F.set(pk.VEC_P, p); F.disp_assemble_system(True)
        x = synth(F.n_u, 0.11)
        y, y0 = F.apply(pk.MAT_A_U, x), O.apply(pk.MAT_A_U, x)
        assert rel(y, y0) <= 1e-12
        assert rel(F.get(pk.VEC_RHS_U), O.get(pk.VEC_RHS_U)) <= 1e-12
        rp, col, val = O.export_csr(pk.MAT_A_U)
        assert rel(F.get(pk.VEC_DIAG_U), csr_to_scipy(rp, col, val).diagonal()) <= 1e-13
    finally:
        F.close()


@pytest.mark.parametrize("mode", [pk.OP_CSR, pk.OP_MATRIX_FREE], ids=["csr", "matrix_free"])
def test_displacement_solve_and_projection(pair, mode):
    P, O, _ = pair
    G = pk.Context(P, 0, mode)
    try:
        p = 10e6 * (1 + 0.1 * synth(G.n_p))
        O.set(pk.VEC_P, p); G.set(pk.VEC_P, p)
        O.disp_assemble_system(True); G.disp_assemble_system(True)
        rc0, i0 = O.disp_solve(abs_tol=1e-12, max_iter=1000)                      # the reference's SSOR(1.2)-CG
        rc, i1 = G.disp_solve(abs_tol=1e-12, max_iter=5000)
        assert rc0 == 0 and rc == 0, (i0.iterations, i1.iterations, i1.final_residual)
        u, u0 = G.get(pk.VEC_U), O.get(pk.VEC_U)
        assert rel2(u, u0) <= 1e-9
        comps = [a * G.dim + a for a in range(G.dim)]
        O.proj_assemble_matrix(); G.proj_assemble_matrix()
        O.set(pk.VEC_U, u); O.proj_assemble_rhs(comps); G.proj_assemble_rhs(comps)
        entries = [0, 2] if G.dim == 2 else [0, 3, 5]
        for e in entries:
            assert rel(G.get(pk.VEC_PROJ_RHS0 + e), O.get(pk.VEC_PROJ_RHS0 + e)) <= 1e-11
            rc0, _ = O.proj_solve(e, rel_tol=1e-12); rc, _ = G.proj_solve(e, rel_tol=1e-12)
            assert rc0 == 0 and rc == 0
            assert rel2(G.get(pk.VEC_STRAIN0 + e), O.get(pk.VEC_STRAIN0 + e)) <= 1e-9
        O.get_volumetric_strain(); G.get_volumetric_strain()
        assert rel2(G.get(pk.VEC_EPSV), O.get(pk.VEC_EPSV)) <= 1e-9
    finally:
        G.close()


def test_ssor_cg_follows_the_reference_iteration(pair):
    """PORO_PREC_SSOR = PreconditionSSOR(omega) in natural row order (PoroElasticDisplacementSolver.h:300-307,
    PoroElasticPressureSolver.h:176-181, StrainProjector.h:210-215): the same Krylov sequence as the oracle's SSOR-CG,
    so the iteration counts agree, not only the converged fields."""
    P, O, G = pair
    p = 10e6 * (1 + 0.1 * synth(G.n_p))
    O.set(pk.VEC_P, p); G.set(pk.VEC_P, p)
    O.disp_assemble_system(True); G.disp_assemble_system(True)
    rc0, i0 = O.disp_solve(abs_tol=1e-12, max_iter=1000, omega=1.2)
    rc, i1 = G.disp_solve(abs_tol=1e-12, max_iter=1000, prec=pk.PREC_SSOR, omega=1.2)
    assert rc0 == 0 and rc == 0
    assert abs(i1.iterations - i0.iterations) <= 2, (i1.iterations, i0.iterations)
    assert abs(i1.initial_residual - i0.initial_residual) <= 1e-10 * i0.initial_residual
    assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-9
    # capped solve: after the same k iterations both sit on the same iterate (the sweeps are order dependent, so this pins the order)
    for S in (O, G):
        S.fill(pk.VEC_U, 0.0)
    k = max(3, i0.iterations // 3)
    rc0, j0 = O.disp_solve(abs_tol=1e-300, max_iter=k, omega=1.2)
    rc, j1 = G.disp_solve(abs_tol=1e-300, max_iter=k, prec=pk.PREC_SSOR, omega=1.2)
    assert rc0 == 1 and rc == 1 and j0.iterations == k and j1.iterations == k
    assert abs(j1.final_residual - j0.final_residual) <= 1e-6 * j0.final_residual
    # pressure Jacobian and projection mass matrix, omega = 1
    n = G.n_p
    for key, v in {pk.VEC_P_OLD: p * 0.99, pk.VEC_EPSV: -2e-6 * (1 + 0.3 * synth(n, 0.5)), pk.VEC_EPSV0: -2e-6 * np.ones(n)}.items():
        O.set(key, v); G.set(key, v)
    O.pres_assemble_residual(60.0); G.pres_assemble_residual(60.0)
    O.pres_assemble_jacobian(60.0); G.pres_assemble_jacobian(60.0)
    rc0, i0 = O.pres_solve(rel_tol=1e-8); rc, i1 = G.pres_solve(rel_tol=1e-8, prec=pk.PREC_SSOR)
    assert rc0 == 0 and rc == 0 and abs(i1.iterations - i0.iterations) <= 1, (i1.iterations, i0.iterations)
    assert rel2(G.get(pk.VEC_DP), O.get(pk.VEC_DP)) <= 1e-7
    F = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        F.set(pk.VEC_P, p); F.disp_assemble_system(True)
        with pytest.raises(RuntimeError, match="CSR"):
            F.disp_solve(prec=pk.PREC_SSOR)
    finally:
        F.close()


def test_fast_diagonalisation_preconditioner(pair):
    """PORO_PREC_FDM: on a box the pressure Jacobian (PoroElasticPressureSolver.h:158-169) and the projection mass matrix
    (StrainProjector.h:101-106) are Kronecker sums; the preconditioner is their exact inverse, so SolverCG stops after one or
    two iterations on the same solution the oracle's SSOR-CG converges to."""
    P, O, G = pair
    assert G.supports_preconditioner(1, pk.PREC_FDM) and G.supports_preconditioner(0, pk.PREC_FDM)
    n = G.n_p
    vals = {pk.VEC_P: 10e6 * (1 + 0.05 * synth(n)), pk.VEC_P_OLD: 10e6 * (1 + 0.05 * synth(n, 0.2)), pk.VEC_EPSV: -2e-6 * (1 + 0.3 * synth(n, 0.5)),
            pk.VEC_EPSV0: -2e-6 * np.ones(n)}
    for k, v in vals.items():
        O.set(k, v); G.set(k, v)
    O.pres_assemble_residual(60.0); G.pres_assemble_residual(60.0)
    O.pres_assemble_jacobian(60.0); G.pres_assemble_jacobian(60.0)
    rc0, _ = O.pres_solve(rel_tol=1e-13); rc, info = G.pres_solve(rel_tol=1e-8, prec=pk.PREC_FDM)
    assert rc0 == 0 and rc == 0 and info.iterations <= 2, info.iterations
    assert info.final_residual <= 1e-12 * info.initial_residual          # exact inverse: far below the requested 1e-8
    assert rel2(G.get(pk.VEC_DP), O.get(pk.VEC_DP)) <= 1e-9
    u = 1e-5 * synth(G.n_u, 0.05)
    O.set(pk.VEC_U, u); G.set(pk.VEC_U, u)
    comps = [a * G.dim + a for a in range(G.dim)]
    O.proj_assemble_matrix(); G.proj_assemble_matrix()
    O.proj_assemble_rhs(comps); G.proj_assemble_rhs(comps)
    for e in ([0, 2] if G.dim == 2 else [0, 3, 5]):
        rc0, _ = O.proj_solve(e, rel_tol=1e-13); rc, info = G.proj_solve(e, rel_tol=1e-8, prec=pk.PREC_FDM)
        assert rc0 == 0 and rc == 0 and info.iterations <= 2
        assert rel2(G.get(pk.VEC_STRAIN0 + e), O.get(pk.VEC_STRAIN0 + e)) <= 1e-9
    # all entries in one call (poro_proj_solve_many): solved directly and together where the library has the fused form (3D boxes of <= 80 vertices per line, matrix-free
    # context), entry by entry otherwise - the same strains either way
    ents = [0, 2] if G.dim == 2 else [0, 3, 5]
    for e in ents:
        G.fill(pk.VEC_STRAIN0 + e, 0.0)
    rc, infos = G.proj_solve_many(ents, rel_tol=1e-8, prec=pk.PREC_FDM)
    assert rc == 0 and all(i.converged and i.final_residual <= 1e-8 * max(i.initial_residual, 1e-300) for i in infos), [(i.iterations, i.final_residual) for i in infos]
    for e in ents:
        assert rel2(G.get(pk.VEC_STRAIN0 + e), O.get(pk.VEC_STRAIN0 + e)) <= 1e-9
    # the displacement system's block form (tests/test_fdm_u_gpu.py) also serves the assembled-CSR operator
    O.disp_assemble_system(True); G.disp_assemble_system(True)
    O.fill(pk.VEC_U, 0.0); G.fill(pk.VEC_U, 0.0)
    rc0, _ = O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=5000); rc, info = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=100, prec=pk.PREC_FDM)
    assert rc0 == 0 and rc == 0 and info.iterations <= 40
    assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-9


def test_fast_diagonalisation_needs_a_box():
    P = pk.Problem.gmsh(DOMAIN_MSH, 1, host_material(), BC_2D)
    G = pk.Context(P, 0, pk.OP_CSR)
    try:
        assert not G.supports_preconditioner(1, pk.PREC_FDM)
        G.fill(pk.VEC_P, 1e7); G.copy(pk.VEC_P_OLD, pk.VEC_P)
        G.pres_assemble_residual(60.0); G.pres_assemble_jacobian(60.0)
        with pytest.raises(RuntimeError, match="uniform box"):
            G.pres_solve(prec=pk.PREC_FDM)
    finally:
        G.close(); P.close()


def test_pressure_residual_jacobian_solve(pair):
    P, O, G = pair
    n = G.n_p
    vals = {pk.VEC_P: 10e6 * (1 + 0.05 * synth(n)), pk.VEC_P_OLD: 10e6 * (1 + 0.05 * synth(n, 0.2)), pk.VEC_EPSV: -2e-6 * (1 + 0.3 * synth(n, 0.5)),
            pk.VEC_EPSV0: -2e-6 * np.ones(n), pk.VEC_DP: 1e3 * synth(n, 0.7)}
    for k, v in vals.items():
        O.set(k, v); G.set(k, v)
    O.pres_update_volumetric_strain(); G.pres_update_volumetric_strain()
    assert rel(G.get(pk.VEC_EPSV), O.get(pk.VEC_EPSV)) <= 1e-15
    r0, r1 = O.pres_assemble_residual(60.0), G.pres_assemble_residual(60.0)
    assert abs(r1 - r0) <= 1e-12 * r0
    assert rel(G.get(pk.VEC_RESIDUAL_P), O.get(pk.VEC_RESIDUAL_P)) <= 1e-12
    O.pres_assemble_jacobian(60.0); G.pres_assemble_jacobian(60.0)
    _, _, v1 = G.export_csr(pk.MAT_JACOBIAN_P); _, _, v0 = O.export_csr(pk.MAT_JACOBIAN_P)
    assert np.abs(v1 - v0).max() <= 1e-13 * np.abs(v0).max()
    rc0, _ = O.pres_solve(rel_tol=1e-13); rc, _ = G.pres_solve(rel_tol=1e-13)
    assert rc0 == 0 and rc == 0
    assert rel2(G.get(pk.VEC_DP), O.get(pk.VEC_DP)) <= 1e-9


def test_patch_test_on_device():
    """K2 on the device: uniform p, input.data BCs => u linear, eps_xx = eps_yy = -1e-6 exactly (Q2)."""
    P = box_problem(2, 16, 2, mat=host_material())
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        G.fill(pk.VEC_P, REF["p_init"]); G.disp_assemble_system(True)
        rc, info = G.disp_solve(max_iter=5000)
        assert rc == 0
        X = node_coords_box(2, 16, 2); u = G.get(pk.VEC_U)
        assert np.abs(u[0::2] + 1e-5 * (X[:, 0] + 5) / 10).max() <= 1e-17
        assert np.abs(u[1::2] + 1e-5 * (X[:, 1] + 5) / 10).max() <= 1e-17
        G.proj_assemble_matrix(); G.proj_assemble_rhs([0, 3])
        for e in (0, 2):
            assert G.proj_solve(e, rel_tol=1e-12)[0] == 0
            assert np.abs(G.get(pk.VEC_STRAIN0 + e) + 1e-6).max() <= 1e-15
    finally:
        G.close(); P.close()


@pytest.mark.parametrize("cfg", [("box", 2, 16, 2), ("box", 3, 4, 1), ("msh", 2, 0, 1), ("msh", 2, 0, 2)], ids=["ref-default-2dQ2", "3dQ1", "domain.msh-Q1", "domain.msh-Q2"])
def test_run_trace_matches_oracle(cfg):
    """K7/K8: the FSS time loop (PoroelasticityFSS.h:294-415) through the C++ host driver vs the oracle's restatement:
    identical FSS / pressure iteration counts, matching residual norms and fields after 2 steps."""
    kind, dim, n, deg = cfg
    mat = host_material()
    P = box_problem(dim, n, deg, mat=mat) if kind == "box" else pk.Problem.gmsh(DOMAIN_MSH, deg, mat, BC_2D)
    O = oracle_py.Oracle(P)
    try:
        t0, _ = O.run(2, REF["p_init"], REF["dt"], max_it=1000)
        assert O.noconvergence_count() == 0
        t1, G = pk.run_problem(P, 2, REF["p_init"], REF["dt"], operator_mode=pk.OP_CSR, max_it=5000)
        try:
            assert t1.shape == t0.shape
            assert np.array_equal(t1[:, :3], t0[:, :3])                      # step, fss iteration, pressure iterations
            assert np.all(t1[1:, 3] < 1e-8) and np.all(t0[1:, 3] < 1e-8)      # inner residual under pressure_tol on both
            assert np.allclose(t1[:, 4], t0[:, 4], rtol=1e-10)               # |p|_inf
            assert rel2(G.get(pk.VEC_P), O.get(pk.VEC_P)) <= 1e-10
            assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-8
            # eps_v comes from projections both sides stop at the reference's 1e-8*||rhs|| (StrainProjector.h:209) with
            # different preconditioners, so they agree to ~cond(M)*1e-8, not to solver-independent precision
            assert rel2(G.get(pk.VEC_EPSV), O.get(pk.VEC_EPSV)) <= 1e-6
        finally:
            G.close()
    finally:
        O.close(); P.close()


def test_run_with_reference_preconditioner_reproduces_cg_counts():
    """The whole run() with PORO_PREC_SSOR (the reference's SolverCG + PreconditionSSOR): CG iteration counts of the
    displacement and pressure solves follow the oracle's SSOR-CG step by step, and eps_v agrees beyond solver tolerance
    because both sides now stop on the same iterate."""
    P = box_problem(2, 16, 2, mat=host_material())
    O = oracle_py.Oracle(P)
    try:
        t0, _ = O.run(2, REF["p_init"], REF["dt"], max_it=1000)
        t1, G = pk.run_problem(P, 2, REF["p_init"], REF["dt"], operator_mode=pk.OP_CSR, max_it=1000, prec=pk.PREC_SSOR)
        try:
            assert t1.shape == t0.shape and np.array_equal(t1[:, :3], t0[:, :3])
            assert np.abs(t1[:, 6] - t0[:, 6]).max() <= 2, (t1[:, 6], t0[:, 6])     # displacement CG iterations
            assert np.abs(t1[:, 7] - t0[:, 7]).max() <= 1, (t1[:, 7], t0[:, 7])     # pressure CG iterations
            assert rel2(G.get(pk.VEC_P), O.get(pk.VEC_P)) <= 1e-10
            assert rel2(G.get(pk.VEC_EPSV), O.get(pk.VEC_EPSV)) <= 1e-9
        finally:
            G.close()
    finally:
        O.close(); P.close()


def test_neumann_traction():
    """Neumann term (PoroElasticDisplacementSolver.h:249-277): traction on x-high / y-high instead of displacement."""
    bc = [(0, 0, 0.0), (2, 1, 0.0)]
    neu = [(1, 0, -2e6), (3, 1, -1e6)]
    for dim, n, deg in ((2, 6, 2), (3, 3, 1)):
        b = bc + ([(4, 2, 0.0)] if dim == 3 else [])
        nm = neu + ([(5, 2, -3e6)] if dim == 3 else [])
        P = box_problem(dim, n, deg, bc=b, neumann=nm)
        O = oracle_py.Oracle(P, hoisted=True)
        for mode in (pk.OP_CSR, pk.OP_MATRIX_FREE):
            G = pk.Context(P, 0, mode)
            O.fill(pk.VEC_P, 1e6); G.fill(pk.VEC_P, 1e6)
            O.disp_assemble_system(True); G.disp_assemble_system(True)
            assert rel(G.get(pk.VEC_RHS_U), O.get(pk.VEC_RHS_U)) <= 1e-12
            G.close()
        O.close(); P.close()


def test_error_behaviour():
    P = box_problem(2, 4, 2)
    G = pk.Context(P, 0, pk.OP_CSR)
    try:
        with pytest.raises(RuntimeError):
            G.disp_solve()                                  # solve before assemble
        G.fill(pk.VEC_P, 10e6); G.disp_assemble_system(True)
        rc, info = G.disp_solve(abs_tol=1e-30, max_iter=5)   # SolverControl::NoConvergence analogue
        assert rc == 1 and info.iterations == 5 and info.converged == 0
        with pytest.raises(RuntimeError):
            G.set(pk.VEC_P, np.zeros(3))
    finally:
        G.close(); P.close()
    Pm = pk.Problem.gmsh(DOMAIN_MSH, 2, material(), BC_2D)
    Gm = pk.Context(Pm, 0, pk.OP_MATRIX_FREE)               # unstructured meshes run the general matrix-free operator (tests/test_mfg_gpu.py) ...
    Gm.fill(pk.VEC_P, 10e6); Gm.disp_assemble_system(True)
    with pytest.raises(RuntimeError, match="PORO_PREC_FDM"):
        Gm.disp_solve(prec=pk.PREC_FDM)                     # ... but not the box-only preconditioner
    Gm.close(); Pm.close()


@pytest.mark.parametrize("deg", [2, 1])
@pytest.mark.parametrize("n", [(3, 3, 3), (31, 7, 9), (5, 40, 3), (64, 13, 40), (30, 6, 1), (33, 12, 24), (70, 31, 5), (91, 17, 33)], ids=str)
def test_sum_factorised_operator_equals_element_matrix_operator(n, deg, monkeypatch):
    """k_kron3_q2 / k_kron3_q1 (Kronecker sweeps) vs k_mf_apply (element-matrix gather, itself checked against the oracle's CSR above) on
    boxes spanning several x / y tiles of both shapes and z chunks, with and without Dirichlet rows; 1e-12 relative to the result's max."""
    P = box_problem(3, n, deg)
    x = synth(P.desc.n_dofs_u, 0.11) + 0.3
    ys = []
    for variant in ("kron", "gather"):
        monkeypatch.setenv("PORO_MF_VARIANT", variant)
        G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
        G.fill(pk.VEC_P, 1e6); G.disp_assemble_system(True)          # the build also self-checks the unconstrained operators against each other
        ys.append((G.apply(pk.MAT_A_U, x), G.get(pk.VEC_RHS_U)))
        G.close()
    assert rel(ys[0][0], ys[1][0]) <= 1e-12
    assert rel(ys[0][1], ys[1][1]) <= 1e-12
    P.close()


@pytest.mark.parametrize("dim,n", [(2, (9, 6)), (3, (4, 3, 5)), (3, 6)], ids=str)
def test_structured_pressure_operator(dim, n):
    """matrix-free contexts apply the pressure Jacobian / projection mass matrix as constant-coefficient stencils (k_p_stencil);
    the solves must land on the oracle's CSR solves (pressure increment and projected strains, 1e-9 in l2)."""
    P = box_problem(dim, n, 1)
    O = oracle_py.Oracle(P, hoisted=True)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        npp = G.n_p
        vals = {pk.VEC_P: 10e6 * (1 + 0.05 * synth(npp)), pk.VEC_P_OLD: 10e6 * (1 + 0.05 * synth(npp, 0.2)), pk.VEC_EPSV: -2e-6 * (1 + 0.3 * synth(npp, 0.5)),
                pk.VEC_EPSV0: -2e-6 * np.ones(npp), pk.VEC_DP: 1e3 * synth(npp, 0.7)}
        for k, v in vals.items():
            O.set(k, v); G.set(k, v)
        r0, r1 = O.pres_assemble_residual(60.0), G.pres_assemble_residual(60.0)
        assert abs(r1 - r0) <= 1e-12 * r0
        O.pres_assemble_jacobian(60.0); G.pres_assemble_jacobian(60.0)
        rc0, _ = O.pres_solve(rel_tol=1e-13); rc, _ = G.pres_solve(rel_tol=1e-13)
        assert rc0 == 0 and rc == 0
        assert rel2(G.get(pk.VEC_DP), O.get(pk.VEC_DP)) <= 1e-9
        u = 1e-5 * synth(G.n_u, 0.13)
        O.set(pk.VEC_U, u); G.set(pk.VEC_U, u)
        comps = [a * dim + a for a in range(dim)]
        O.proj_assemble_matrix(); G.proj_assemble_matrix(); O.proj_assemble_rhs(comps); G.proj_assemble_rhs(comps)
        for e in ([0, 2] if dim == 2 else [0, 3, 5]):
            assert O.proj_solve(e, rel_tol=1e-13)[0] == 0 and G.proj_solve(e, rel_tol=1e-13)[0] == 0
            assert rel2(G.get(pk.VEC_STRAIN0 + e), O.get(pk.VEC_STRAIN0 + e)) <= 1e-9
    finally:
        G.close(); O.close(); P.close()


@pytest.mark.parametrize("dim,n,deg", [(2, 6, 2), (3, 3, 2), (3, 4, 1)])
def test_postprocessing_stresses_and_vtk(tmp_path, dim, n, deg):
    """Tail of the time loop body (PoroelasticityFSS.h:409-411): get_shear_strain_components (:167-176), get_effective_stresses
    (:189-224, sigma' = C:eps with the isotropic tensor of ConstitutiveModel.h:45-57) and output_results (:227-291).
    Reference behaviour: the shear right-hand sides are never assembled, so eps_ij (i != j) stay 0, and the 2D file shows
    sigma_xx under the name sigma_yy; `corrected` fixes both."""
    mat = host_material()
    P = box_problem(dim, n, deg, mat=mat)
    R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=REF["p_init"], dt=REF["dt"], max_it=5000)
    try:
        R.initialize(); R.step()
        G = R.ctx
        n_sym = dim * (dim + 1) // 2
        shear = [1] if dim == 2 else [1, 2, 4]
        normal = [0, 2] if dim == 2 else [0, 3, 5]
        out = tmp_path / "solution"; out.mkdir()
        R.postprocess(str(out))
        eps = [G.get(pk.VEC_STRAIN0 + e) for e in range(n_sym)]; sig = [G.get(pk.VEC_STRESS0 + e) for e in range(n_sym)]
        for e in shear:
            assert np.all(eps[e] == 0.0) and np.all(sig[e] == 0.0)            # the reference's zero right-hand side
        tr = sum(eps[e] for e in normal)
        lam, mu = mat.lame_lambda, mat.shear_G
        for e in range(n_sym):
            want = 2 * mu * eps[e] + (lam * tr if e in normal else 0.0)
            assert np.abs(sig[e] - want).max() <= 1e-14 * max(np.abs(want).max(), 1.0)
        # the file: one patch per cell, fields in the reference's order
        txt = (out / "solution-0001.vtk").read_text().split("\n")
        nv, nc = 2 ** dim, n ** dim
        assert txt[0] == "# vtk DataFile Version 3.0" and txt[2] == "ASCII" and txt[3] == "DATASET UNSTRUCTURED_GRID"
        assert f"POINTS {nc * nv} double" in txt and f"CELLS {nc} {nc * (nv + 1)}" in txt and f"POINT_DATA {nc * nv}" in txt
        names = [l.split()[1] for l in txt if l.startswith("SCALARS") or l.startswith("VECTORS")]
        want_names = ["u", "p", "eps_xx", "sigma_xx"] + (["eps_xy", "eps_yy", "sigma_xy", "sigma_yy"] if dim == 2 else
                                                         ["eps_xy", "eps_xz", "eps_yy", "eps_yz", "eps_zz", "sigma_xy", "sigma_xz", "sigma_yy", "sigma_yz", "sigma_zz"])
        assert names == want_names

        def field(lines, nm):
            i = next(k for k, l in enumerate(lines) if l.startswith(f"SCALARS {nm} "))
            return np.array(lines[i + 2].split(), dtype=float)
        cell_p = np.ctypeslib.as_array(P.desc.cell_dofs_p, shape=(nc * nv,))
        assert np.allclose(field(txt, "p"), G.get(pk.VEC_P)[cell_p], rtol=1e-11)
        if dim == 2:
            assert np.allclose(field(txt, "sigma_yy"), sig[0][cell_p], rtol=1e-11)       # the label quirk (:257-258)
        # corrected mode: shear strains from their own right-hand sides, sigma_yy is sigma_yy
        R.postprocess(str(out), corrected=True)
        G.proj_assemble_rhs([1]); G.proj_solve(1, rel_tol=1e-8, prec=pk.PREC_FDM)
        eps_xy = G.get(pk.VEC_STRAIN0 + 1)
        assert np.abs(eps_xy).max() > 0
        txt2 = (out / "solution-0001.vtk").read_text().split("\n")
        assert np.allclose(field(txt2, "eps_xy"), eps_xy[cell_p], rtol=1e-9, atol=1e-22)
        if dim == 2:
            assert np.allclose(field(txt2, "sigma_yy"), G.get(pk.VEC_STRESS0 + 2)[cell_p], rtol=1e-11)
    finally:
        R.close(); P.close()


def test_ilu0_preconditioner(pair):
    """PORO_PREC_ILU0 (north-star "Jacobi/ILU(0)-preconditioned Krylov"): same converged solutions as the oracle, in fewer CG
    iterations than Jacobi; the factors live on the assembled CSR pattern, so the matrix-free mode refuses it."""
    P, O, G = pair
    p = 10e6 * (1 + 0.1 * synth(G.n_p))
    O.set(pk.VEC_P, p); G.set(pk.VEC_P, p)
    O.disp_assemble_system(True); G.disp_assemble_system(True)
    rc0, _ = O.disp_solve(abs_tol=1e-12, max_iter=1000)
    rcj, ij = G.disp_solve(abs_tol=1e-12, max_iter=5000)
    G.fill(pk.VEC_U, 0.0)
    rc, ii = G.disp_solve(abs_tol=1e-12, max_iter=5000, prec=pk.PREC_ILU0)
    assert rc0 == 0 and rcj == 0 and rc == 0
    assert ii.iterations < ij.iterations, (ii.iterations, ij.iterations)
    assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-9
    n = G.n_p
    for key, v in {pk.VEC_P_OLD: p * 0.99, pk.VEC_EPSV: -2e-6 * (1 + 0.3 * synth(n, 0.5)), pk.VEC_EPSV0: -2e-6 * np.ones(n)}.items():
        O.set(key, v); G.set(key, v)
    O.pres_assemble_residual(60.0); G.pres_assemble_residual(60.0)
    O.pres_assemble_jacobian(60.0); G.pres_assemble_jacobian(60.0)
    rc0, _ = O.pres_solve(rel_tol=1e-13); rcj, ij = G.pres_solve(rel_tol=1e-12)
    G.fill(pk.VEC_DP, 0.0)
    rc, ii = G.pres_solve(rel_tol=1e-12, prec=pk.PREC_ILU0)
    assert rc0 == 0 and rcj == 0 and rc == 0 and ii.iterations < ij.iterations
    assert rel2(G.get(pk.VEC_DP), O.get(pk.VEC_DP)) <= 1e-9
    assert G.supports_preconditioner(0, pk.PREC_ILU0)
    F = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        assert not F.supports_preconditioner(0, pk.PREC_ILU0)
        F.set(pk.VEC_P, p); F.disp_assemble_system(True)
        with pytest.raises(RuntimeError, match="CSR"):
            F.disp_solve(prec=pk.PREC_ILU0)
    finally:
        F.close()


def test_ragged_box_shapes():
    """42 box shapes from 1 cell to 30 x 15 x 16, both degrees, 2D and 3D: the structured kernels pass their set-up self-checks, the
    matrix-free operator equals the assembled one (1e-12) and both PCG paths converge to the same displacement."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "shape_sweep.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "failures: 0" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_transient_of_ten_steps_tracks_the_oracle():
    """BASELINE config 5 in miniature: 10 time steps (warm starts, pressure history, the fast-diagonalised pressure / projection solves, the
    structured right-hand sides) against the oracle's restatement of the same loop: identical FSS / pressure iteration counts per step
    and the same pressure and displacement at the end."""
    P = box_problem(3, 4, 2, mat=host_material())
    O = oracle_py.Oracle(P)
    try:
        t0, _ = O.run(10, REF["p_init"], REF["dt"], max_it=2000)
        assert O.noconvergence_count() == 0
        t1, G = pk.run_problem(P, 10, REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=5000)
        try:
            assert t1.shape == t0.shape and np.array_equal(t1[:, :3], t0[:, :3])
            assert np.allclose(t1[:, 4], t0[:, 4], rtol=1e-10)                       # |p|_inf after every step
            assert rel2(G.get(pk.VEC_P), O.get(pk.VEC_P)) <= 1e-10
            assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-8
        finally:
            G.close()
    finally:
        O.close(); P.close()


def test_transient_with_the_fast_solver_tracks_the_oracle():
    """the same loop on the path the benchmark times - matrix-free operator, block fast diagonalisation in octant form for the displacement system, pressure updates and
    the three projections of a step computed directly from the exact inverses - on 6^3 Q2/Q1 cells for 12 steps against the oracle's SSOR-CG restatement: identical
    fixed-stress rows, the same |p|_inf after every step, the same fields at the end (config 5 at a size the oracle finishes in seconds).  The pressure Newton loop takes
    FEWER passes than the oracle's (an exact inner solve instead of CG to 1e-8), never more"""
    P = box_problem(3, 6, 2, mat=host_material())
    O = oracle_py.Oracle(P, hoisted=True)
    try:
        t0, _ = O.run(12, REF["p_init"], REF["dt"], max_it=5000)
        assert O.noconvergence_count() == 0
        t1, G = pk.run_problem(P, 12, REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=5000, prec=pk.PREC_FDM)
        try:
            assert G.supports_preconditioner(0, pk.PREC_FDM) and G.supports_preconditioner(1, pk.PREC_FDM)
            assert t1.shape == t0.shape and np.array_equal(t1[:, :2], t0[:, :2]) and np.all(t1[:, 2] <= t0[:, 2])
            assert np.allclose(t1[:, 4], t0[:, 4], rtol=1e-8)
            assert t1[1:, 6].max() <= 40 and t1[1:, 7].max() == 0                     # block-FDM CG iterations per step; pressure systems solved directly
            assert rel2(G.get(pk.VEC_P), O.get(pk.VEC_P)) <= 1e-9
            assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-7
        finally:
            G.close()
    finally:
        O.close(); P.close()


@pytest.mark.parametrize("dim,n,deg,incremental", [(2, 8, 2, False), (3, 3, 2, False), (2, 8, 2, True)])
def test_coupled_fixed_stress_iteration(dim, n, deg, incremental):
    """`coupled_fss`: the get_volumetric_strain() call the reference commented out (PoroelasticityFSS.h:399) restored.  The fixed-stress loop
    (:347-407) then really iterates: several coupling iterations per step with a contracting error, the same counts as the oracle."""
    P = box_problem(dim, n, deg, mat=host_material())
    O = oracle_py.Oracle(P)
    try:
        # `incremental`: additionally the storage term against the previous step's strain instead of the initial one (3 steps, so it matters)
        t0, _ = O.run(3, REF["p_init"], REF["dt"], max_it=2000, coupled_fss=True, incremental_strain=incremental)
        t1, G = pk.run_problem(P, 3, REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=5000, coupled_fss=True, incremental_strain=incremental)
        try:
            assert t1.shape == t0.shape and np.array_equal(t1[:, :3], t0[:, :3])
            step1 = t1[t1[:, 0] == 1]
            if dim == 2:                                                             # (on the 3^3 mesh the coupling error is below the tolerance at once)
                assert len(step1) > 1                                                # more than one coupling iteration per step
            assert np.all(np.diff(step1[:, 5]) < 0) and step1[-1, 5] <= 1e-8      # contracting, converged
            assert np.allclose(t1[:, 4], t0[:, 4], rtol=1e-9)
            assert rel2(G.get(pk.VEC_P), O.get(pk.VEC_P)) <= 1e-9
            assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-7
        finally:
            G.close()
    finally:
        O.close(); P.close()


def test_box_regression_goldens_on_device():
    """The HIP path against the committed traces of tests/golden/box_traces.json (oracle-generated): iteration counts, |p|_inf per step, norms."""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "box_traces.json")) as f:
        gold = json.load(f)
    for name, g in gold.items():
        P = box_problem(g["dim"], g["n"], g["degree"], mat=host_material())
        kw = {"coupled_fss": True, "incremental_strain": True} if g["variant"] == "coupled" else {}
        t1, G = pk.run_problem(P, 2, REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=5000, **kw)
        try:
            rows = np.array(g["rows"])
            assert t1.shape[0] == rows.shape[0] and np.array_equal(t1[:, :3], rows[:, :3]), name
            assert np.allclose(t1[:, 4], rows[:, 4], rtol=1e-9), name
            assert abs(np.linalg.norm(G.get(pk.VEC_P)) - g["p_l2"]) <= 1e-9 * g["p_l2"], name
            assert abs(np.linalg.norm(G.get(pk.VEC_U)) - g["u_l2"]) <= 1e-7 * g["u_l2"], name
        finally:
            G.close(); P.close()


def test_device_ilu0_factors_equal_a_sequential_factorisation():
    """the level-scheduled device factorisation (k_ilu0_level) takes the pivots of every row in the same order as a sequential IKJ sweep and rounds the products the same
    way: ILU(0)-preconditioned CG, restated here in numpy on the exported matrix, takes the same number of iterations and lands on the same solution"""
    P = box_problem(3, 3, 2)
    G = pk.Context(P, 0, pk.OP_CSR)
    try:
        p = 10e6 * (1 + 0.1 * synth(G.n_p))
        G.set(pk.VEC_P, p); G.disp_assemble_system(True)
        rp, col, val = G.export_csr(pk.MAT_A_U)
        b = G.get(pk.VEC_RHS_U); n = len(b)
        lu = val.copy(); dpos = np.array([rp[i] + int(np.searchsorted(col[rp[i]:rp[i + 1]], i)) for i in range(n)])
        for i in range(n):                                           # row-wise IKJ on A's own pattern
            where = {int(c): j for j, c in zip(range(rp[i], rp[i + 1]), col[rp[i]:rp[i + 1]])}
            for kk in range(rp[i], dpos[i]):
                k = int(col[kk]); lik = lu[kk] / lu[dpos[k]]; lu[kk] = lik
                for jj in range(dpos[k] + 1, rp[k + 1]):
                    pos = where.get(int(col[jj]))
                    if pos is not None:
                        lu[pos] -= lik * lu[jj]
        A = csr_to_scipy(rp, col, val)

        def prec(g):
            z = g.copy()
            for i in range(n):
                z[i] -= lu[rp[i]:dpos[i]] @ z[col[rp[i]:dpos[i]]]
            for i in range(n - 1, -1, -1):
                z[i] = (z[i] - lu[dpos[i] + 1:rp[i + 1]] @ z[col[dpos[i] + 1:rp[i + 1]]]) / lu[dpos[i]]
            return z
        x = np.zeros(n); g = A @ x - b; tol = 1e-12; it = 0
        h = prec(g); d = -h; gh = g @ h
        while np.sqrt(g @ g) > tol and it < 1000:                    # SolverCG's recurrence (g = A x - b)
            it += 1
            Ad = A @ d; alpha = gh / (d @ Ad); g = g + alpha * Ad; x = x + alpha * d
            if np.sqrt(g @ g) <= tol:
                break
            h = prec(g); beta = gh; gh = g @ h; beta = gh / beta; d = -h + beta * d
        rc, info = G.disp_solve(abs_tol=1e-12, max_iter=1000, prec=pk.PREC_ILU0)
        assert rc == 0 and abs(info.iterations - it) <= 1, (info.iterations, it)
        free = np.ones(n, bool); free[np.ctypeslib.as_array(P.desc.dirichlet_dof, shape=(P.desc.n_dirichlet,))] = False
        assert rel2(G.get(pk.VEC_U)[free], x[free]) <= 1e-9
    finally:
        G.close(); P.close()
