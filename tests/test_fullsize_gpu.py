"""Parity at BASELINE.json's full sizes through size-independent properties of the operator and the solve (the oracle cannot
run 9 M dofs in seconds): rigid-body null space, symmetry, linearity, and the analytic patch-test solution."""
import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
from common import BC_2D, BC_3D, REF, box_problem, node_coords_box

pytestmark = pytest.mark.gpu

FULL = [(3, 72, 2), (3, 99, 1), (2, 336, 2)]     # BASELINE configs 4/5, 3, 2


@pytest.mark.parametrize("dim,n,deg", FULL, ids=lambda v: str(v))
def test_unconstrained_operator_annihilates_rigid_body_modes(dim, n, deg):
    P = box_problem(dim, n, deg, bc=[])
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        G.fill(pk.VEC_P, 0.0); G.disp_assemble_system(True)
        X = node_coords_box(dim, n, deg); N = X.shape[0]
        scale = G.get(pk.VEC_DIAG_U).max()
        modes = []
        for c in range(dim):
            t = np.zeros((N, dim)); t[:, c] = 1.0; modes.append(t.ravel())
        for a in range(dim):
            for b in range(a + 1, dim):
                r = np.zeros((N, dim)); r[:, a] = -X[:, b]; r[:, b] = X[:, a]; modes.append(r.ravel())
        for m in modes:
            assert np.abs(G.apply(pk.MAT_A_U, m)).max() <= 1e-12 * scale * np.abs(m).max()
        s = np.zeros((N, dim)); s[:, 0] = X[:, 0]                   # a stretch is not in the null space
        assert s.ravel() @ G.apply(pk.MAT_A_U, s.ravel()) > 0
    finally:
        G.close(); P.close()


@pytest.mark.parametrize("dim,n,deg", FULL, ids=lambda v: str(v))
def test_constrained_operator_is_symmetric_and_linear(dim, n, deg):
    P = box_problem(dim, n, deg)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        G.fill(pk.VEC_P, 0.0); G.disp_assemble_system(True)
        nu = G.n_u
        x, y = np.sin(0.37 * np.arange(nu)), np.cos(0.11 * np.arange(nu)) + 0.2
        Ax, Ay = G.apply(pk.MAT_A_U, x), G.apply(pk.MAT_A_U, y)
        assert abs(y @ Ax - x @ Ay) <= 1e-12 * (abs(y @ Ax) + np.linalg.norm(Ax) * np.linalg.norm(y))
        Az = G.apply(pk.MAT_A_U, 2.5 * x - 0.75 * y)
        assert np.abs(Az - (2.5 * Ax - 0.75 * Ay)).max() <= 1e-12 * np.abs(Az).max()
        assert x @ Ax > 0                                            # SPD
    finally:
        G.close(); P.close()


@pytest.mark.parametrize("dim,n,deg", FULL, ids=lambda v: str(v))
def test_patch_test_at_full_size(dim, n, deg):
    """uniform p + the input.data displacement BCs: u is the linear field of the Dirichlet data, projected normal strains are -1e-6."""
    P = box_problem(dim, n, deg)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        G.fill(pk.VEC_P, REF["p_init"]); G.disp_assemble_system(True)
        rc, info = G.disp_solve(abs_tol=1e-12, rel_tol=1e-12, max_iter=50000)
        assert rc == 0, (info.iterations, info.final_residual)
        X = node_coords_box(dim, n, deg); u = G.get(pk.VEC_U)
        for c in range(dim):
            assert np.abs(u[c::dim] + 1e-5 * (X[:, c] + 5) / 10).max() <= 5e-15     # |u| <= 1e-5: relative 5e-10
        comps = [a * dim + a for a in range(dim)]
        G.proj_assemble_matrix(); G.proj_assemble_rhs(comps)
        for e in ([0, 2] if dim == 2 else [0, 3, 5]):
            assert G.proj_solve(e, rel_tol=1e-12, max_iter=5000)[0] == 0
            assert np.abs(G.get(pk.VEC_STRAIN0 + e) + 1e-6).max() <= 1e-14
    finally:
        G.close(); P.close()


def test_matrix_free_equals_csr_at_medium_size():
    """3D Q2 24^3 (353 k dofs, 66 M non-zeros): the sum-factorised operator against the assembled CSR SpMV of the same context family"""
    P = box_problem(3, 24, 2)
    A, F = pk.Context(P, 0, pk.OP_CSR), pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        for G in (A, F):
            G.fill(pk.VEC_P, 3e6); G.disp_assemble_system(True)
        x = np.sin(0.37 * np.arange(A.n_u))
        ya, yf = A.apply(pk.MAT_A_U, x), F.apply(pk.MAT_A_U, x)
        assert np.abs(ya - yf).max() <= 1e-12 * np.abs(ya).max()
        assert np.abs(A.get(pk.VEC_RHS_U) - F.get(pk.VEC_RHS_U)).max() <= 1e-12 * np.abs(A.get(pk.VEC_RHS_U)).max()
    finally:
        A.close(); F.close(); P.close()


def test_preconditioners_agree_at_config_4_size_over_a_transient():
    """BASELINE config 5 (72^3 Q2/Q1, consecutive time steps) with the three displacement preconditioners: identical fixed-stress / pressure iteration
    counts in every step and the same fields after 5 steps (each CG stops at 1e-10 of its step's initial residual, so the fields agree far better
    than the 1e-6 asserted); every step does Krylov work (the reduction rule keeps the transient live)."""
    P = box_problem(3, 72, 2)
    runs = {}
    try:
        for name, prec in (("block_fdm", pk.PREC_FDM), ("chebyshev", pk.PREC_CHEBYSHEV), ("jacobi", pk.PREC_JACOBI)):
            R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=REF["p_init"], dt=REF["dt"], abs_u=1e-12, rel_u=1e-10, max_it=50000, prec=prec, reduction=True)
            R.initialize()
            tr = [R.step()[0] for _ in range(5 if name != "jacobi" else 2)]
            runs[name] = (tr, R.ctx.get(pk.VEC_U), R.ctx.get(pk.VEC_P))
            R.close()
        tf, uf, pf = runs["block_fdm"]; tc, uc, pc = runs["chebyshev"]; tj = runs["jacobi"][0]
        for a, b in zip(tf, tc):
            assert np.array_equal(a[:, :3], b[:, :3]) and a[0, 6] > 0 and b[0, 6] > 0
            assert a[0, 6] <= 40 and b[0, 6] < 120
        for a, b in zip(tf, tj):
            assert np.array_equal(a[:, :3], b[:, :3]) and b[0, 6] > a[0, 6]
        assert np.linalg.norm(uf - uc) <= 1e-6 * np.linalg.norm(uf)
        assert np.abs(pf - pc).max() <= 1e-10 * np.abs(pf).max()
    finally:
        P.close()


def test_config_5_hundred_time_steps():
    """BASELINE config 5 in full: 100 consecutive time steps on the 72^3 Q2/Q1 box ("100 timesteps transient, fixed-stress iteration count + wall-clock").
    Two independent preconditioners (block fast diagonalisation, Chebyshev) walk the same transient: identical fixed-stress / pressure iteration counts in all 100
    steps, every step does Krylov work, the CG counts stay bounded, and the final fields agree."""
    P = box_problem(3, 72, 2)
    runs = {}
    try:
        for name, prec in (("block_fdm", pk.PREC_FDM), ("chebyshev", pk.PREC_CHEBYSHEV)):
            R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=REF["p_init"], dt=REF["dt"], abs_u=1e-12, rel_u=1e-10, max_it=50000, prec=prec, reduction=True)
            R.initialize()
            tr = [R.step()[0] for _ in range(100)]
            runs[name] = (tr, R.ctx.get(pk.VEC_U), R.ctx.get(pk.VEC_P))
            R.close()
        (tf, uf, pf), (tc, uc, pc) = runs["block_fdm"], runs["chebyshev"]
        for a, b in zip(tf, tc):
            assert np.array_equal(a[:, :3], b[:, :3])
            assert 0 < a[0, 6] <= 40 and 0 < b[0, 6] < 120
        assert np.linalg.norm(uf - uc) <= 1e-6 * np.linalg.norm(uf)
        assert np.abs(pf - pc).max() <= 1e-9 * np.abs(pf).max()
        assert np.isfinite(uf).all() and np.isfinite(pf).all() and pf.max() > REF["p_init"]          # the injection well raises the pressure
    finally:
        P.close()
