"""BASELINE config 4's real partition - 72^3 cells in 8 slabs of 9 cell layers - end to end on one GPU: 8 rank THREADS of one process (a one-GPU box admits only a
handful of GPU processes) drive 8 contexts through the partitioned code path with the callback communicator (tools/rank_threads.py).  Asserted: the single-rank
iteration counts, identical fixed-stress / pressure iteration rows, equal copies of every shared plane, the single-rank fields."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import rank_threads  # noqa: E402

pytestmark = pytest.mark.gpu


def check(rec, its_slack):
    single = rec["cg_iterations_u_single_rank"]
    for its in rec["cg_iterations_u_by_rank"]:
        assert its == rec["cg_iterations_u_by_rank"][0]                                    # replicated recurrences: every rank counts the same
        assert all(abs(a - b) <= its_slack for s, t in zip(its, single) for a, b in zip(s, t)), (its, single)
    assert rec["fss_rows_equal_single_rank"]
    assert rec["shared_plane_copies_max_abs_diff_u"] <= 1e-18 + 1e-13 * 1e-5              # |u| ~ 1e-5: the two copies of a shared plane agree to the last bits
    assert rec["rel_diff_u_vs_single_rank"] <= 1e-7 and rec["rel_diff_p_vs_single_rank"] <= 1e-9, rec


@pytest.mark.parametrize("world,n,prec", [(3, 8, "block_fdm"), (4, 8, "chebyshev"), (5, 10, "jacobi"), (2, 24, "block_fdm"), (3, 48, "block_fdm")])   # (24 / 48 cells: 2 / 4 tiles per half line = the both-parity z pass at its other sizes; 72 cells: 5 tiles)
def test_rank_threads_small(world, n, prec):
    check(rank_threads.rehearse(world, 3, [n, n, n], 2, prec, 2), its_slack=2)


def test_config4_partition_8_slabs_of_9_layers():
    rec = rank_threads.rehearse(8, 3, [72, 72, 72], 2, "block_fdm", 1)
    assert rec["layers_per_rank"] == [9] * 8
    check(rec, its_slack=1)
    assert max(max(s) for s in rec["cg_iterations_u_by_rank"][0]) <= 20
    it = rec["per_cg_iteration_u_on_an_interior_rank"]
    assert it["operator_applications"] <= 1.3 and it["alltoalls_all_systems"] <= 5.0, it   # per iteration: 1 operator application (+ the initial residual), 2 all-to-alls (+ the share of the pressure / projection solves: 16 per step)
    assert it["block_fdm_applications_in_slab_form"] >= 1.0, it                          # the quadrant-form kernels (kernels_fdmo.hip) did the preconditioning, not the nodal fallback


def test_nodal_fallback_of_the_distributed_q1_fast_diagonalisation(monkeypatch):
    """boxes whose Q1 lines are longer than 80 vertices keep the nodal form of the distributed pressure / projection fast diagonalisation (batched window copies around
    the two all-to-alls); forced here on a small box, together with the two-workgroup z stage of the displacement system"""
    monkeypatch.setenv("PORO_FDM_P_UNFUSED", "1")
    monkeypatch.setenv("PORO_FDMO_SLAB_TWO_PARITY_WORKGROUPS", "1")
    check(rank_threads.rehearse(3, 3, [10, 9, 24], 2, "block_fdm", 2), its_slack=2)
