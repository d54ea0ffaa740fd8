"""Terzaghi's one-dimensional consolidation problem (the "corrected physics" switches validated against a consolidation benchmark).

A column of height H, rollers on the sides and the bottom, a compressive traction sigma0 on the top face, which is also DRAINED (p = 0; an extension:
the reference has no pressure boundary conditions, PoroElasticPressureSolver.h:69-70).  Starting from the undrained response p0 = (alpha sigma0 / Kv) /
(alpha^2 / Kv + 1 / M), Kv = lambda + 2G, the pressure obeys the heat equation with c_v = (k / mu) / (alpha^2 / Kv + 1 / M):
    p(y, t) = (4 p0 / pi) sum_m 1 / (2m + 1) sin((2m + 1) pi d / (2H)) exp(-(2m + 1)^2 pi^2 c_v t / (4 H^2)),   d = depth below the drained top.
The reference's loop as committed (get_volumetric_strain() commented out at PoroelasticityFSS.h:399, storage term against eps_v0) cannot reproduce this;
with `coupled_fss` (the call restored) and `incremental_strain` (storage term against the previous step) the fixed-stress iteration converges to the
backward-Euler solution of Biot's equations, which is compared with the series here: on the oracle (CPU suite) and on the device (GPU suite)."""
import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import material

H, SIGMA0 = 10.0, 1.0e6


def column(dim, ny, deg, nx=2):
    n = [nx] * (dim - 1) + [ny]                                   # the column's axis is the last direction
    last = dim - 1
    bc = [(2 * d, d, 0.0) for d in range(dim - 1)] + [(2 * d + 1, d, 0.0) for d in range(dim - 1)] + [(2 * last, last, 0.0)]      # rollers on the sides and the bottom
    m = material(flow_rate=0.0)                                    # no well
    P = pk.Problem.box(dim, n, [10.0] * (dim - 1) + [H], deg, m, bc, [(2 * last + 1, last, -SIGMA0)])                            # traction = value * n on the top face
    P.set_pressure_bc([(2 * last + 1, 0.0)])                       # drained top
    return P, m


def analytic(m, depth, t):
    Kv = m.lame_lambda + 2 * m.shear_G
    s = m.biot_alpha ** 2 / Kv + 1.0 / m.biot_M
    p0 = (m.biot_alpha * SIGMA0 / Kv) / s; cv = m.k_over_mu / s
    out = np.zeros_like(depth)
    for k in range(400):
        a = (2 * k + 1) * np.pi / (2 * H)
        out += 4 * p0 / np.pi / (2 * k + 1) * np.sin(a * depth) * np.exp(-a * a * cv * t)
    return out, p0, cv


def profile(P, p, dim):
    """pressure along the column's axis (nodes of the first vertical line) as (depth below the top, p)"""
    X = np.ctypeslib.as_array(P.desc.vertex_coords, shape=(P.desc.n_vertices, dim))
    line = np.all(np.abs(X[:, :dim - 1] - X[0, :dim - 1]) < 1e-12, axis=1)
    return H / 2 - X[line, dim - 1], p[line]


KW = dict(fss_tol=1e-11, pressure_tol=1e-11, max_fss=200, max_it=50000)    # tolerances are absolute residual norms (PoroelasticityFSS.h:364-371)


def run_and_compare(backend, dim, ny, deg, dt, steps, **switches):
    P, m = column(dim, ny, deg)
    _, p0, cv = analytic(m, np.zeros(1), 0.0)
    try:
        if backend == "oracle":
            G = oracle_py.Oracle(P, hoisted=True)
            tr, _ = G.run(steps, p0, dt, **KW, **switches)
        else:
            tr, G = pk.run_problem(P, steps, p0, dt, operator_mode=pk.OP_MATRIX_FREE, prec=pk.PREC_CHEBYSHEV, **KW, **switches)
        depth, pn = profile(P, G.get(pk.VEC_P), dim)
        G.close()
        pa, _, _ = analytic(m, depth, steps * dt)
        return np.abs(pn - pa).max() / p0, tr
    finally:
        P.close()


def test_oracle_consolidation_matches_terzaghi():
    # t = 600 s = 0.27 H^2 / c_v: a well developed profile; backward Euler in time: the error shrinks with dt
    e1, tr = run_and_compare("oracle", 2, 20, 2, 60.0, 10, coupled_fss=True, incremental_strain=True)
    e2, _ = run_and_compare("oracle", 2, 20, 2, 30.0, 20, coupled_fss=True, incremental_strain=True)
    assert e1 < 0.015 and e2 < 0.6 * e1, (e1, e2)          # measured 1.16e-2 -> 5.6e-3 (-> 2.6e-3 at dt = 15 s): first order in dt
    assert len(tr) - 1 > 10                                        # a genuine fixed-stress iteration: more than one pass per step


def test_reference_loop_does_not_consolidate():
    """the loop as committed (quirks Q1 / Q2) stays far from the consolidation solution on the same problem - what the switches are for"""
    e, _ = run_and_compare("oracle", 2, 20, 2, 60.0, 10)
    assert e > 0.2, e                                             # measured 0.36


@pytest.mark.gpu
@pytest.mark.parametrize("dim,ny,deg", [(2, 40, 2), (3, 24, 2), (3, 30, 1)], ids=str)
def test_device_consolidation_matches_terzaghi(dim, ny, deg):
    e, tr = run_and_compare("hip", dim, ny, deg, 30.0, 20, coupled_fss=True, incremental_strain=True)
    assert e < 0.02 and len(tr) - 1 > 20, (e, len(tr))


@pytest.mark.gpu
def test_device_follows_the_oracle_with_prescribed_pressures():
    """parity of the extension itself: same trace and fields as the oracle on the drained column (reference loop and corrected loop)"""
    for kw in (dict(), dict(coupled_fss=True, incremental_strain=True)):
        P, m = column(2, 12, 2)
        O = oracle_py.Oracle(P, hoisted=True)
        try:
            t0, _ = O.run(3, 2e5, 60.0, prec=oracle_py.PREC_JACOBI, **KW, **kw)
            t1, G = pk.run_problem(P, 3, 2e5, 60.0, operator_mode=pk.OP_CSR, **KW, **kw)
            assert np.array_equal(t1[:, :3], t0[:, :3])
            assert np.abs(G.get(pk.VEC_P) - O.get(pk.VEC_P)).max() <= 1e-8 * np.abs(O.get(pk.VEC_P)).max()
            assert np.linalg.norm(G.get(pk.VEC_U) - O.get(pk.VEC_U)) <= 1e-8 * np.linalg.norm(O.get(pk.VEC_U))
            G.close()
        finally:
            O.close(); P.close()
