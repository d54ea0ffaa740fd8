"""Cross-implementation goldens: tests/golden/independent_traces.json is produced by tools/independent_model.py - a numpy / scipy restatement of the
time step that shares no code with the oracle or the product (operators from Kronecker products of 1D matrices, sparse direct solves).  The oracle
(CPU suite) and the HIP path (GPU suite) must reproduce its fixed-stress / pressure iteration counts exactly, |p|_inf to the 1e-8 of the reference's own
solver tolerances, and the fields at the probe points.  (The reference itself holds no vectors: parity stays unpinned by it, but the oracle is no
longer checked against its own output only.)"""
import json
import os

import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import GOLDEN, REF, box_problem

with open(os.path.join(GOLDEN, "independent_traces.json")) as f:
    GOLD = json.load(f)


def check(gold, trace, u, p):
    rows = trace[1:]
    assert [int(r[2]) for r in rows] == [t["pressure_iterations"] for t in gold["trace"]]
    assert [int(r[1]) for r in rows] == [t["fss_iteration"] for t in gold["trace"]]
    for r, t in zip(rows, gold["trace"]):
        assert abs(r[4] - t["p_linf"]) <= 1e-10 * t["p_linf"]
    assert abs(np.linalg.norm(u) - gold["u_l2"]) <= 1e-8 * gold["u_l2"] and abs(np.linalg.norm(p) - gold["p_l2"]) <= 1e-11 * gold["p_l2"]
    up, pp = u[:: max(1, len(u) // 16)][:16], p[:: max(1, len(p) // 16)][:16]
    assert np.abs(up - np.array(gold["u_probe"])).max() <= 1e-8 * np.abs(u).max()
    assert np.abs(pp - np.array(gold["p_probe"])).max() <= 1e-11 * np.abs(p).max()


@pytest.mark.parametrize("key", sorted(GOLD))
def test_oracle_reproduces_the_independent_model(key):
    g = GOLD[key]
    P = box_problem(g["dim"], g["n"], g["degree"])
    O = oracle_py.Oracle(P, hoisted=True)
    try:
        t, _ = O.run(g["steps"], REF["p_init"], REF["dt"], max_it=20000)
        check(g, t, O.get(pk.VEC_U), O.get(pk.VEC_P))
    finally:
        O.close(); P.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["jacobi", "chebyshev", "block_fdm"])
@pytest.mark.parametrize("key", sorted(GOLD))
def test_device_reproduces_the_independent_model(key, prec):
    g = GOLD[key]
    P = box_problem(g["dim"], g["n"], g["degree"])
    try:
        t, G = pk.run_problem(P, g["steps"], REF["p_init"], REF["dt"], operator_mode=pk.OP_MATRIX_FREE, max_it=20000,
                              prec={"jacobi": pk.PREC_JACOBI, "chebyshev": pk.PREC_CHEBYSHEV, "block_fdm": pk.PREC_FDM}[prec])
        check(g, t, G.get(pk.VEC_U), G.get(pk.VEC_P))
        G.close()
    finally:
        P.close()
