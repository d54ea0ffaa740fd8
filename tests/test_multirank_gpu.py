"""The partitioned HIP path (slab partition + interface exchange + all-reduced dots inside the device PCG) on ONE GPU:
2 ranks share device 0 and exchange through the host-staged callback communicator (gloo); on a multi-GPU node the same
code runs with RCCL (poro_ctx_comm_init_rccl).  Compared with the single-rank oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import REF, box_problem
from test_multirank_cpu import HERE, free_port, stitch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,dim,n,deg,backend", [(2, 3, (3, 3, 6), 2, "hip_mf"), (2, 3, (4, 4, 6), 1, "hip_csr"), (2, 2, (8, 12), 2, "hip_mf"),
                                                      (3, 3, (4, 5, 7), 1, "hip_mf"), (3, 2, (9, 10), 2, "hip_mf"), (4, 3, (3, 3, 9), 2, "hip_mf"),
                                                      (2, 3, (3, 4, 6), 2, "hip_mf_fdm"), (3, 2, (9, 10), 2, "hip_mf_fdm"), (3, 3, (4, 5, 7), 1, "hip_mf_fdm"), (4, 3, (5, 3, 9), 2, "hip_mf_fdm"),
                                                      (2, 3, (3, 4, 6), 2, "hip_mf_cheb"), (3, 2, (9, 10), 2, "hip_mf_cheb"), (2, 2, (6, 170), 1, "hip_mf_fdm")])
def test_ranks_on_one_gpu(tmp_path, world, dim, n, deg, backend):
    """2 and 3 ranks (uneven slabs, column groups that do not divide evenly): halo exchange, all-reduced dots, and the distributed
    fast-diagonalisation solves of the pressure / projection systems (all-to-all of column groups)."""
    port = free_port()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mr_worker.py"), str(r), str(world), str(port), str(dim), ",".join(map(str, n)), str(deg), outs[r], backend])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    R = [np.load(o) for o in outs]
    P = box_problem(dim, n, deg)
    O = oracle_py.Oracle(P, hoisted=True)
    tr, _ = O.run(1, REF["p_init"], REF["dt"], max_it=20000, prec=oracle_py.PREC_JACOBI)
    nn = [deg * m + 1 for m in n]
    plane_u = dim * int(np.prod(nn[:-1])); plane_p = int(np.prod([m + 1 for m in n[:-1]]))
    off_u = [int(r["offset_u"][0]) for r in R]
    off_p = [0]
    for r in R[:-1]:
        off_p.append(off_p[-1] + r["p"].size - plane_p)
    for r in R:
        assert np.array_equal(r["trace"][1:, :3], tr[1:, :3])
        assert np.all(r["trace"][1:, 7] <= 2 * np.maximum(r["trace"][1:, 2], 1))      # exact preconditioner: <= 2 CG iterations per pressure solve
        if backend.endswith("_fdm"):
            assert 0 < r["trace"][1:, 6].max() <= 40                                  # block fast diagonalisation of the displacement system
    u = stitch([r["u"] for r in R], off_u, plane_u, P.desc.n_dofs_u)
    p = stitch([r["p"] for r in R], off_p, plane_p, P.desc.n_dofs_p)
    rhs = stitch([r["rhs_u"] for r in R], off_u, plane_u, P.desc.n_dofs_u)
    for a, b in zip(R[:-1], R[1:]):                                   # both copies of every shared plane agree
        assert np.abs(a["u"][-plane_u:] - b["u"][:plane_u]).max() <= 1e-14 * np.abs(u).max()
    # the rhs after the step depends on p, which both sides converge to the reference's 1e-8 tolerances only
    assert np.linalg.norm(rhs - O.get(pk.VEC_RHS_U)) <= 1e-9 * np.linalg.norm(rhs)
    assert np.linalg.norm(u - O.get(pk.VEC_U)) <= 1e-8 * np.linalg.norm(u)
    assert np.linalg.norm(p - O.get(pk.VEC_P)) <= 1e-10 * np.linalg.norm(p)
    Ax = stitch([r["Ax"] for r in R], off_u, plane_u, P.desc.n_dofs_u)
    y = O.apply(pk.MAT_A_U, np.sin(0.11 * np.arange(P.desc.n_dofs_u)))
    assert np.abs(Ax - y).max() <= 1e-12 * np.abs(y).max()
    O.close(); P.close()


def test_rccl_data_plane_single_rank(monkeypatch):
    """RCCL plumbing on one GPU: dlopen, ncclCommInitRank(1 rank) + the library's communicator self-test (all-reduce, grouped
    send/recv to itself), then a solve through the partitioned code path (explicit sums + ncclAllReduce on the compute stream)."""
    monkeypatch.setenv("PORO_FORCE_PARTITIONED_PATH", "1")
    P = box_problem(3, 4, 2)
    O = oracle_py.Oracle(P, hoisted=True)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    try:
        G.comm_rccl(pk.rccl_unique_id())
        p = 10e6 * (1 + 0.1 * np.sin(0.37 * np.arange(G.n_p)))
        O.set(pk.VEC_P, p); G.set(pk.VEC_P, p)
        O.disp_assemble_system(True); G.disp_assemble_system(True)
        assert O.disp_solve()[0] == 0 and G.disp_solve(max_iter=5000)[0] == 0
        assert np.linalg.norm(G.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
        assert abs(G.norm(pk.VEC_U)[0] - np.linalg.norm(O.get(pk.VEC_U))) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
        # block fast diagonalisation of the displacement system through the partitioned path (RCCL all-reduce of the face flags / slab sizes, self block of the all-to-all)
        assert G.supports_preconditioner(0, pk.PREC_FDM)
        G.fill(pk.VEC_U, 0.0)
        rc, info = G.disp_solve(max_iter=200, prec=pk.PREC_FDM)
        assert rc == 0 and info.iterations <= 40
        assert np.linalg.norm(G.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
        # Chebyshev-CG through the partitioned path (elementwise recurrence after the exchange, Lanczos estimate with all-reduced dots)
        G.fill(pk.VEC_U, 0.0)
        rc, info = G.disp_solve(max_iter=2000, prec=pk.PREC_CHEBYSHEV)
        assert rc == 0 and np.linalg.norm(G.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_U))
        # the distributed fast-diagonalisation solve with RCCL as the communicator (1 rank: all-reduce + the self block of the all-to-all)
        assert G.supports_preconditioner(1, pk.PREC_FDM)
        for S in (O, G):
            S.set(pk.VEC_P_OLD, 0.99 * p); S.fill(pk.VEC_EPSV, -2e-6); S.fill(pk.VEC_EPSV0, -2e-6)
            S.pres_assemble_residual(60.0); S.pres_assemble_jacobian(60.0)
        rc0, _ = O.pres_solve(rel_tol=1e-13); rc, info = G.pres_solve(prec=pk.PREC_FDM)
        assert rc0 == 0 and rc == 0 and info.iterations <= 2
        assert np.linalg.norm(G.get(pk.VEC_DP) - O.get(pk.VEC_DP)) <= 1e-9 * np.linalg.norm(O.get(pk.VEC_DP))
    finally:
        G.close(); O.close(); P.close()
