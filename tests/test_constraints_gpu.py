"""Hanging-node constraints through the C-ABI (poro_desc.cons_u / cons_p): the HIP path on locally refined meshes against the
oracle (whose condensed solves are pinned by tests/test_constraints_cpu.py: patch test + scipy elimination)."""
import ctypes as C

import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import REF
from test_constraints_cpu import MESHES, refined, u_node_coords

pytestmark = pytest.mark.gpu


def rel2(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(params=MESHES, ids=str)
def trio(request):
    P = refined(*request.param)
    O = oracle_py.Oracle(P, hoisted=True)
    G = pk.Context(P, 0, pk.OP_CSR)
    yield P, O, G
    G.close(); O.close(); P.close()


def test_condensed_displacement_system(trio):
    P, O, G = trio
    p = REF["p_init"] * (1 + 0.3 * np.sin(0.37 * np.arange(G.n_p)))
    for S in (O, G):
        S.set(pk.VEC_P, p); S.disp_assemble_system(True)
    b0, b1 = O.get(pk.VEC_RHS_U), G.get(pk.VEC_RHS_U)
    assert np.abs(b1 - b0).max() <= 1e-12 * np.abs(b0).max()
    nd = P.desc.cons_u.n
    hanging = np.ctypeslib.as_array(P.desc.cons_u.dof, shape=(nd,))
    assert np.all(b1[hanging] == 0.0)
    rc0, _ = O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000); rc1, info = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000)
    assert rc0 == 0 and rc1 == 0 and info.iterations > 0
    assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-9
    assert not G.supports_preconditioner(0, pk.PREC_FDM) and not G.supports_preconditioner(0, pk.PREC_SSOR) and G.supports_preconditioner(0, pk.PREC_JACOBI)
    with pytest.raises(RuntimeError, match="constraint lists"):
        G.disp_solve(prec=pk.PREC_SSOR)
    # the Chebyshev polynomial is built on the condensed operator and its Jacobi diagonal: same solution in fewer iterations
    assert G.supports_preconditioner(0, pk.PREC_CHEBYSHEV)
    G.fill(pk.VEC_U, 0.0)
    rc2, info2 = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000, prec=pk.PREC_CHEBYSHEV)
    assert rc2 == 0 and 0 < info2.iterations < info.iterations
    assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-9


def test_patch_test_on_device(trio):
    P, O, G = trio
    dim = G.dim
    G.fill(pk.VEC_P, REF["p_init"]); G.disp_assemble_system(True)
    assert G.disp_solve(abs_tol=1e-13, rel_tol=0.0, max_iter=50000)[0] == 0
    X = u_node_coords(P); u = G.get(pk.VEC_U)
    for c in range(dim):
        assert np.abs(u[c::dim] + 1e-5 * (X[:, c] + 5) / 10).max() <= 1e-15
    G.proj_assemble_matrix(); G.proj_assemble_rhs([a * dim + a for a in range(dim)])
    for e in ([0, 2] if dim == 2 else [0, 3, 5]):
        assert G.proj_solve(e, rel_tol=1e-13, max_iter=5000)[0] == 0
        assert np.abs(G.get(pk.VEC_STRAIN0 + e) + 1e-6).max() <= 1e-14


def test_condensed_pressure_and_projection_systems(trio):
    P, O, G = trio
    n = G.n_p
    vals = {pk.VEC_P: 10e6 * (1 + 0.05 * np.sin(0.37 * np.arange(n))), pk.VEC_P_OLD: 10e6 * (1 + 0.05 * np.sin(0.2 * np.arange(n))),
            pk.VEC_EPSV: -2e-6 * (1 + 0.3 * np.sin(0.5 * np.arange(n))), pk.VEC_EPSV0: -2e-6 * np.ones(n)}
    for k, v in vals.items():
        O.set(k, v); G.set(k, v)
    r0, r1 = O.pres_assemble_residual(60.0), G.pres_assemble_residual(60.0)
    assert abs(r1 - r0) <= 1e-12 * r0
    R0, R1 = O.get(pk.VEC_RESIDUAL_P), G.get(pk.VEC_RESIDUAL_P)
    assert np.abs(R1 - R0).max() <= 1e-12 * np.abs(R0).max()
    hanging = np.ctypeslib.as_array(P.desc.cons_p.dof, shape=(P.desc.cons_p.n,))
    assert np.all(R1[hanging] == 0.0)
    O.pres_assemble_jacobian(60.0); G.pres_assemble_jacobian(60.0)
    rc0, _ = O.pres_solve(rel_tol=1e-13, max_iter=5000); rc1, _ = G.pres_solve(rel_tol=1e-13, max_iter=5000)
    assert rc0 == 0 and rc1 == 0
    assert rel2(G.get(pk.VEC_DP), O.get(pk.VEC_DP)) <= 1e-9
    u = 1e-5 * np.sin(0.05 * np.arange(G.n_u))
    O.set(pk.VEC_U, u); G.set(pk.VEC_U, u)
    comps = [a * G.dim + a for a in range(G.dim)]
    O.proj_assemble_matrix(); G.proj_assemble_matrix(); O.proj_assemble_rhs(comps); G.proj_assemble_rhs(comps)
    for e in ([0, 2] if G.dim == 2 else [0, 3, 5]):
        assert rel2(G.get(pk.VEC_PROJ_RHS0 + e), O.get(pk.VEC_PROJ_RHS0 + e)) <= 1e-12
        rc0, _ = O.proj_solve(e, rel_tol=1e-13, max_iter=5000); rc1, _ = G.proj_solve(e, rel_tol=1e-13, max_iter=5000)
        assert rc0 == 0 and rc1 == 0 and rel2(G.get(pk.VEC_STRAIN0 + e), O.get(pk.VEC_STRAIN0 + e)) <= 1e-9
    # the two-level form of the scalar systems (Jacobi here + the underlying box's exact fast diagonalisation through the vertex interpolation poro_desc.coarse.ptr_p ...):
    # the same condensed solutions, hanging values distributed
    assert G.supports_preconditioner(1, pk.PREC_TWO_LEVEL)
    G.fill(pk.VEC_DP, 0.0)
    rc2, two = G.pres_solve(rel_tol=1e-13, max_iter=500, prec=pk.PREC_TWO_LEVEL)
    assert rc2 == 0 and two.iterations > 0 and rel2(G.get(pk.VEC_DP), O.get(pk.VEC_DP)) <= 1e-9
    for e in ([0, 2] if G.dim == 2 else [0, 3, 5]):
        G.fill(pk.VEC_STRAIN0 + e, 0.0)
        rc2, two = G.proj_solve(e, rel_tol=1e-13, max_iter=500, prec=pk.PREC_TWO_LEVEL)
        assert rc2 == 0 and two.iterations > 0 and rel2(G.get(pk.VEC_STRAIN0 + e), O.get(pk.VEC_STRAIN0 + e)) <= 1e-9


@pytest.mark.parametrize("cfg", MESHES[1:3], ids=str)
def test_time_steps_on_a_mesh_with_hanging_nodes(cfg):
    """the whole fixed-stress loop (the state a deal.II-side caller hands over after one refine_mesh(), PoroelasticityFSS.h:333-340): same
    FSS / pressure iteration counts and fields as the oracle"""
    P = refined(*cfg)
    O = oracle_py.Oracle(P, hoisted=True)
    try:
        t0, _ = O.run(2, REF["p_init"], REF["dt"], max_it=20000, prec=oracle_py.PREC_JACOBI)
        t1, G = pk.run_problem(P, 2, REF["p_init"], REF["dt"], operator_mode=pk.OP_CSR, max_it=20000)
        assert np.array_equal(t1[:, :3], t0[:, :3])
        assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-8
        assert np.abs(G.get(pk.VEC_P) - O.get(pk.VEC_P)).max() <= 1e-10 * np.abs(O.get(pk.VEC_P)).max()
        G.close()
    finally:
        O.close(); P.close()


def test_constraint_lists_are_validated():
    """ConstraintMatrix::close() semantics are the caller's job: a master that is itself constrained is refused with a message"""
    P = refined(2, (3, 3), 1, (0, 0), (2, 1))
    try:
        d = pk.Desc.from_buffer_copy(P.desc)
        n = d.cons_u.n
        masters = (C.c_int32 * int(np.ctypeslib.as_array(d.cons_u.ptr, shape=(n + 1,))[-1]))(*np.ctypeslib.as_array(d.cons_u.master, shape=(int(np.ctypeslib.as_array(d.cons_u.ptr, shape=(n + 1,))[-1]),)))
        masters[0] = int(np.ctypeslib.as_array(d.cons_u.dof, shape=(n,))[1])      # a master that hangs itself
        d.cons_u.master = C.cast(masters, C.POINTER(C.c_int32))

        class Bad:
            desc = d; desc_ptr = C.pointer(d)
        with pytest.raises(RuntimeError, match="not closed"):
            pk.Context(Bad, 0, pk.OP_CSR)
    finally:
        P.close()


def test_two_level_preconditioner_on_refined_boxes(trio):
    """PORO_PREC_TWO_LEVEL: Jacobi on the refined mesh + block fast diagonalisation of the underlying uniform box (poro_desc.coarse, interpolation P) - the same
    condensed solution as the oracle's Jacobi-CG, in far fewer iterations"""
    P, O, G = trio
    assert P.desc.coarse.enabled and G.supports_preconditioner(0, pk.PREC_TWO_LEVEL) and G.supports_preconditioner(1, pk.PREC_TWO_LEVEL)
    p = REF["p_init"] * (1 + 0.3 * np.sin(0.37 * np.arange(G.n_p)))
    for S in (O, G):
        S.set(pk.VEC_P, p); S.disp_assemble_system(True)
    rc0, _ = O.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000)
    rc1, jac = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=50000)
    G.fill(pk.VEC_U, 0.0)
    rc2, two = G.disp_solve(abs_tol=1e-14, rel_tol=1e-12, max_iter=500, prec=pk.PREC_TWO_LEVEL)
    assert rc0 == 0 and rc1 == 0 and rc2 == 0 and 0 < two.iterations <= jac.iterations, (two.iterations, jac.iterations)   # (the smallest meshes have a few dozen dofs: both stop at the dimension of the space)
    assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-9


def test_two_level_iteration_counts_do_not_grow_with_refinement():
    """uniform refinement of the whole configuration (box and refined block): the CG iteration count of the two-level preconditioner grows by less than 1.3x per level,
    Jacobi's roughly doubles"""
    from common import BC_3D, material
    counts, jacobi = [], []
    for n in (4, 8, 16):
        P = pk.Problem.refined_box(3, [n] * 3, [10.0] * 3, 2, material(), BC_3D, [n // 4] * 3, [3 * n // 4] * 3)
        G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
        try:
            G.set(pk.VEC_P, REF["p_init"] * (1 + 0.3 * np.sin(0.37 * np.arange(G.n_p)))); G.disp_assemble_system(True)
            rc, info = G.disp_solve(abs_tol=1e-14, rel_tol=1e-8, max_iter=2000, prec=pk.PREC_TWO_LEVEL)
            assert rc == 0
            counts.append(info.iterations)
            u = G.get(pk.VEC_U)
            if n <= 8:
                G.fill(pk.VEC_U, 0.0)
                rc, info = G.disp_solve(abs_tol=1e-14, rel_tol=1e-8, max_iter=50000)
                assert rc == 0 and rel2(u, G.get(pk.VEC_U)) <= 1e-6
                jacobi.append(info.iterations)
        finally:
            G.close(); P.close()
    print("two-level CG iterations per refinement:", counts, "Jacobi:", jacobi)
    assert counts[1] <= 1.3 * counts[0] and counts[2] <= 1.3 * counts[1], counts
    assert counts[1] < jacobi[1] / 3, (counts, jacobi)


def test_two_level_scalar_solves_do_not_grow_with_refinement():
    """pressure Jacobian (dt = 60 s: the stiffness part dominates) and projection mass matrix on uniformly refined configurations: the two-level CG counts stay flat,
    Jacobi's pressure count roughly doubles per level"""
    from common import BC_3D, material
    pres, proj, jac = [], [], []
    for n in (4, 8, 16):
        P = pk.Problem.refined_box(3, [n] * 3, [10.0] * 3, 2, material(), BC_3D, [n // 4] * 3, [3 * n // 4] * 3)
        G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
        try:
            m = G.n_p
            vals = {pk.VEC_P: 10e6 * (1 + 0.05 * np.sin(0.37 * np.arange(m))), pk.VEC_P_OLD: 10e6 * (1 + 0.05 * np.sin(0.2 * np.arange(m))),
                    pk.VEC_EPSV: -2e-6 * (1 + 0.3 * np.sin(0.5 * np.arange(m))), pk.VEC_EPSV0: -2e-6 * np.ones(m)}
            for k, v in vals.items():
                G.set(k, v)
            G.pres_assemble_residual(60.0); G.pres_assemble_jacobian(60.0)
            rc, info = G.pres_solve(rel_tol=1e-8, max_iter=500, prec=pk.PREC_TWO_LEVEL)
            assert rc == 0
            pres.append(info.iterations); dp = G.get(pk.VEC_DP)
            G.fill(pk.VEC_DP, 0.0)
            rc, info = G.pres_solve(rel_tol=1e-8, max_iter=20000)
            assert rc == 0 and rel2(dp, G.get(pk.VEC_DP)) <= 1e-6
            jac.append(info.iterations)
            G.set(pk.VEC_U, 1e-5 * np.sin(0.05 * np.arange(G.n_u)))
            G.proj_assemble_matrix(); G.proj_assemble_rhs([0])
            rc, info = G.proj_solve(0, rel_tol=1e-8, max_iter=500, prec=pk.PREC_TWO_LEVEL)
            assert rc == 0
            proj.append(info.iterations)
        finally:
            G.close(); P.close()
    print("two-level CG iterations, pressure:", pres, "projection:", proj, "Jacobi pressure:", jac)
    assert pres[2] <= 1.3 * pres[1] and proj[2] <= 1.3 * proj[1], (pres, proj)
    assert pres[2] < jac[2] / 3, (pres, jac)


def test_time_steps_with_the_two_level_solvers_track_the_oracle():
    """the whole fixed-stress loop with PORO_PREC_TWO_LEVEL on all three systems (the driver's own choice on a mesh this small would be Jacobi for the scalar ones): the same FSS / pressure iteration structure and fields as the oracle's Jacobi-CG run"""
    P = refined(3, (3, 3, 2), 2, (1, 1, 0), (2, 2, 1))
    O = oracle_py.Oracle(P, hoisted=True)
    try:
        t0, _ = O.run(2, REF["p_init"], REF["dt"], max_it=20000, prec=oracle_py.PREC_JACOBI)
        t1, G = pk.run_problem(P, 2, REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=2000, prec=pk.PREC_TWO_LEVEL, two_level_p=True)
        assert np.array_equal(t1[:, :3], t0[:, :3])
        assert rel2(G.get(pk.VEC_U), O.get(pk.VEC_U)) <= 1e-7
        assert np.abs(G.get(pk.VEC_P) - O.get(pk.VEC_P)).max() <= 1e-9 * np.abs(O.get(pk.VEC_P)).max()
        G.close()
    finally:
        O.close(); P.close()
