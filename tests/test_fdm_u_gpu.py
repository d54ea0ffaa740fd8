"""Block fast-diagonalisation preconditioner of the displacement system (the preconditioner slot of
PoroElasticDisplacementSolver<dim>::solve, PoroElasticDisplacementSolver.h:302-305): the device result against an independent block
inverse built from the ORACLE's assembled matrix (sparse direct solves per displacement component), and the preconditioned solve
against the oracle's solution."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

import poroelasticity_dealii_amd as pk
import oracle_py
from common import BC_2D, BC_3D, REF, box_problem, csr_to_scipy

pytestmark = pytest.mark.gpu

# (dim, cells, degree, Dirichlet list): uneven boxes catch direction mix-ups; the mixed lists put one component on faces of several directions
MIXED_3D = [(0, 0, 0.0), (1, 0, -1e-5), (2, 1, 0.0), (3, 1, -1e-5), (4, 2, 0.0), (2, 0, 2e-6), (5, 1, 0.0)]
MIXED_2D = [(0, 0, 0.0), (2, 1, 0.0), (3, 0, 1e-6)]
# the last three have lines of more than 160 points in one or two directions: the blocked even / odd kernel (k_fdmu_blk), mixed with the register forms
CASES = [(3, (4, 3, 5), 2, BC_3D), (3, (5, 4, 3), 1, BC_3D), (3, (3, 4, 2), 2, MIXED_3D), (2, (6, 5), 2, BC_2D), (2, (7, 4), 1, MIXED_2D), (3, (9, 2, 3), 2, BC_3D),
         (2, (100, 90), 2, BC_2D), (3, (170, 3, 2), 1, BC_3D), (2, (5, 230), 1, BC_2D), (2, (168, 3), 2, BC_2D),
         (3, (100, 3, 2), 2, BC_3D), (3, (2, 120, 3), 2, BC_3D), (3, (3, 2, 110), 2, BC_3D), (3, (84, 3, 50), 2, BC_3D),
         (2, (16, 20), 2, MIXED_2D), (3, (16, 3, 20), 2, MIXED_3D), (2, (36, 5), 2, MIXED_2D), (2, (170, 3), 2, MIXED_2D)]   # (the last: 341 points per line with different end conditions - planar form without the parity split)   # (6 - 8 MFMA tiles per half line in x / y / z: the 6-, 7- and 8-wave transform workgroups with > 64 KB of LDS; the last three: DIFFERENT end conditions on lines of 33 / 41 / 73 points = an odd number of 16-point chunks in the full-length nodal kernels, whose fused last-direction pass wrote past a register array until round 3)   # (168 cells Q2: 336 / 337 modes per parity: the components differ in chunk count)


def block_inverse(A, mask, dim, g):
    """blockdiag(A_cc)^-1 g on the free dofs of every component, zero on the Dirichlet dofs"""
    z = np.zeros_like(g)
    for c in range(dim):
        idx = np.arange(c, A.shape[0], dim)
        idx = idx[~mask[idx]]
        z[idx] = spla.splu(A[idx][:, idx].tocsc()).solve(g[idx])
    return z


@pytest.mark.parametrize("dim,n,deg,bc", CASES, ids=lambda v: str(v) if not isinstance(v, list) else f"bc{len(v)}")
def test_block_fdm_equals_the_block_inverse(dim, n, deg, bc):
    P = box_problem(dim, n, deg, bc=bc)
    O = oracle_py.Oracle(P, hoisted=True)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        assert G.supports_preconditioner(0, pk.PREC_FDM)
        O.fill(pk.VEC_P, 0.0); O.disp_assemble_system(True)
        G.fill(pk.VEC_P, 0.0); G.disp_assemble_system(True)
        A = csr_to_scipy(*O.export_csr(pk.MAT_A_U))
        nd = P.desc.n_dirichlet
        mask = np.zeros(G.n_u, bool); mask[np.ctypeslib.as_array(P.desc.dirichlet_dof, shape=(nd,))] = True
        rng = np.random.default_rng(7)
        g = rng.standard_normal(G.n_u) * 1e3; g[mask] = 0.0
        z = G.apply_preconditioner_u(pk.PREC_FDM, g)
        z0 = block_inverse(A, mask, dim, g)
        assert np.abs(z[mask]).max() == 0.0
        assert np.abs(z - z0).max() <= 1e-10 * np.abs(z0).max(), np.abs(z - z0).max() / np.abs(z0).max()
        # symmetric positive definite as CG needs it
        g2 = rng.standard_normal(G.n_u); g2[mask] = 0.0
        z2 = G.apply_preconditioner_u(pk.PREC_FDM, g2)
        assert abs(g2 @ z - g @ z2) <= 1e-10 * abs(g2 @ z) + 1e-300 and g @ z > 0
    finally:
        G.close(); O.close(); P.close()


@pytest.mark.parametrize("dim,n,deg", [(3, 6, 2), (2, 12, 2), (3, 7, 1), (2, (120, 30), 2)], ids=str)
def test_block_fdm_cg_solves_like_the_oracle(dim, n, deg):
    """nonuniform pressure -> displacement solve: same u as the oracle's SSOR-CG (both to the recursive residual 1e-12 |b|), few iterations"""
    P = box_problem(dim, n, deg)
    O = oracle_py.Oracle(P, hoisted=True)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        p = REF["p_init"] * (1 + 0.3 * np.sin(0.37 * np.arange(G.n_p)))
        for S in (O, G):
            S.set(pk.VEC_P, p); S.disp_assemble_system(True)
        rc0, i0 = O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=5000)
        rc1, i1 = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=200, prec=pk.PREC_FDM)
        assert rc0 == 0 and rc1 == 0
        u0, u1 = O.get(pk.VEC_U), G.get(pk.VEC_U)
        assert np.linalg.norm(u1 - u0) <= 1e-9 * np.linalg.norm(u0)
        assert i1.iterations <= 40, i1.iterations
        G.fill(pk.VEC_U, 0.0)
        rc2, i2 = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=5000, prec=pk.PREC_JACOBI)
        assert rc2 == 0 and i1.iterations < i2.iterations
    finally:
        G.close(); O.close(); P.close()


def test_block_fdm_refuses_non_separable_constraints():
    """a pure-Neumann component makes its block singular: the context says so instead of dividing by zero"""
    P = box_problem(3, 3, 2, bc=[(0, 0, 0.0), (2, 1, 0.0)])
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        assert not G.supports_preconditioner(0, pk.PREC_FDM)
        G.fill(pk.VEC_P, 0.0); G.disp_assemble_system(True)
        with pytest.raises(RuntimeError, match="constrained face"):
            G.disp_solve(prec=pk.PREC_FDM)
    finally:
        G.close(); P.close()


def test_run_with_block_fdm_matches_the_oracle_trace():
    """whole time steps with the preconditioner in the loop: identical FSS / pressure iteration counts, same fields"""
    P = box_problem(3, 4, 2)
    O = oracle_py.Oracle(P, hoisted=True)
    try:
        t0, _ = O.run(3, REF["p_init"], REF["dt"], max_it=5000)
        t1, G = pk.run_problem(P, 3, REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=500, prec=pk.PREC_FDM)
        assert np.array_equal(t1[:, :3], t0[:, :3])
        assert np.linalg.norm(G.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-8 * np.linalg.norm(O.get(pk.VEC_U))
        assert np.abs(G.get(pk.VEC_P) - O.get(pk.VEC_P)).max() <= 1e-10 * np.abs(O.get(pk.VEC_P)).max()
        assert t1[:, 6].max() <= 40
        G.close()
    finally:
        O.close(); P.close()


def test_block_fdm_at_config_4_size():
    """BASELINE config 4 (72^3 Q2/Q1): first time step's displacement solve in <= 40 CG iterations, same u as the Jacobi-CG solve"""
    P = box_problem(3, 72, 2)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        p = REF["p_init"] * (1 + 0.3 * np.sin(0.37 * np.arange(G.n_p)))
        G.set(pk.VEC_P, p); G.disp_assemble_system(True)
        rc, info = G.disp_solve(abs_tol=1e-12, rel_tol=1e-10, max_iter=200, prec=pk.PREC_FDM)
        assert rc == 0 and info.iterations <= 40, (rc, info.iterations)
        u1 = G.get(pk.VEC_U)
        G.fill(pk.VEC_U, 0.0)
        rc, info2 = G.disp_solve(abs_tol=1e-12, rel_tol=1e-10, max_iter=20000, prec=pk.PREC_JACOBI)
        assert rc == 0
        u2 = G.get(pk.VEC_U)
        assert np.linalg.norm(u1 - u2) <= 1e-7 * np.linalg.norm(u2)
        print(f"config 4: block-FDM {info.iterations} its / {info.seconds * 1e3:.1f} ms, Jacobi {info2.iterations} its / {info2.seconds * 1e3:.1f} ms")
    finally:
        G.close(); P.close()
