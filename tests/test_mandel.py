"""Mandel's problem (the corrected-physics switches "validated on Terzaghi / Mandel"): a plane-strain slab of width 2a and height 2b squeezed between two rigid,
frictionless plates by a force 2F per unit thickness, drained at its two free sides.  Unlike Terzaghi's column the coupling is two-dimensional: the pore pressure at the centre first
RISES above its undrained value (Mandel-Cryer effect), which an uncoupled diffusion solver cannot show.  One quadrant is modelled: rollers on x = 0 and y = 0, the plate on y = b as
a tie of the u_y dofs (rigid plate = one unknown value, an ordinary constraint-list entry x[dof] = x[master], Problem.tie_boundary) loaded by the uniform traction -F/a, the side
x = a traction free and drained (prescribed pressure 0, Problem.set_pressure_bc); both are extensions, the reference has neither.  Series solution (Mandel 1953; Abousleiman et al.
1996 for compressible constituents):
    p(x, t) = 2 F B (1 + nu_u) / (3 a) * sum_i sin(a_i) / (a_i - sin(a_i) cos(a_i)) * (cos(a_i x / a) - cos(a_i)) * exp(-a_i^2 c t / a^2),   tan(a_i) = (1 - nu) / (nu_u - nu) * a_i
with c the consolidation coefficient of test_terzaghi.py.  Oracle on the CPU, device on the GPU."""
import numpy as np
import pytest
from scipy.optimize import brentq

import poroelasticity_dealii_amd as pk
import oracle_py
from common import material

A, B_, F = 10.0, 2.0, 1.0e7      # half width, half height, half force per unit thickness


def slab(nx, ny, deg):
    m = material(flow_rate=0.0)
    bc = [(0, 0, 0.0), (2, 1, 0.0)]                                   # rollers: u_x = 0 on x = 0, u_y = 0 on y = 0
    P = pk.Problem.box(2, [nx, ny], [A, B_], deg, m, bc, [(3, 1, -F / A)])
    P.tie_boundary([(3, 1)])                                        # rigid plate on y = b
    P.set_pressure_bc([(1, 0.0)])                                   # drained side x = a
    return P, m


def constants(m):
    K, G, al, M = m.bulk_K, m.shear_G, m.biot_alpha, m.biot_M
    Ku = K + al * al * M
    nu, nuu = (3 * K - 2 * G) / (2 * (3 * K + G)), (3 * Ku - 2 * G) / (2 * (3 * Ku + G))
    Bs = al * M / Ku
    Kv = K + 4 * G / 3
    c = m.k_over_mu / (1 / M + al * al / Kv)
    c2 = 2 * m.k_over_mu * Bs ** 2 * G * (1 - nu) * (1 + nuu) ** 2 / (9 * (1 - nuu) * (nuu - nu))      # the textbook form of the same coefficient
    assert abs(c - c2) <= 1e-9 * c, (c, c2)
    return nu, nuu, Bs, c


def analytic(m, x, t, terms=200):
    """x measured from the centre (0) to the drained side (A)"""
    nu, nuu, Bs, c = constants(m)
    r = (1 - nu) / (nuu - nu)
    roots = [brentq(lambda a: np.tan(a) - r * a, i * np.pi + 1e-9, i * np.pi + np.pi / 2 - 1e-12) for i in range(terms)]
    out = np.zeros_like(x)
    for a in roots:
        out += np.sin(a) / (a - np.sin(a) * np.cos(a)) * (np.cos(a * x / A) - np.cos(a)) * np.exp(-a * a * c * t / A ** 2)
    return 2 * F * Bs * (1 + nuu) / (3 * A) * out, F * Bs * (1 + nuu) / (3 * A), c


KW = dict(fss_tol=1e-11, pressure_tol=1e-11, max_fss=400, max_it=50000, coupled_fss=True, incremental_strain=True)


def run(backend, nx, ny, deg, dt, steps):
    P, m = slab(nx, ny, deg)
    _, p0, c = analytic(m, np.zeros(1), 0.0)
    try:
        if backend == "oracle":
            G = oracle_py.Oracle(P, hoisted=True)
            tr, _ = G.run(steps, p0, dt, prec=oracle_py.PREC_JACOBI, **KW)
        else:
            tr, G = pk.run_problem(P, steps, p0, dt, operator_mode=pk.OP_MATRIX_FREE, prec=pk.PREC_CHEBYSHEV, **KW)     # constraint lists: general matrix-free operator condensed on the fly, Chebyshev-CG
        p = G.get(pk.VEC_P)
        G.close()
        X = np.ctypeslib.as_array(P.desc.vertex_coords, shape=(P.desc.n_vertices, 2)).copy()
        x = X[:, 0] + A / 2                                          # the box is centred at the origin: x in [-a/2, a/2] -> [0, a]
        pa, _, _ = analytic(m, x, steps * dt)
        centre = np.argmin(x + 1e3 * np.abs(X[:, 1] - X[:, 1].min()))
        return np.abs(p - pa).max() / p0, p[centre] / p0, pa.max() / p0, tr, c
    finally:
        P.close()


def test_oracle_reproduces_mandels_solution():
    # t = 0.05 a^2 / c: the centre pressure is still ABOVE its undrained value (Mandel-Cryer effect)
    _, _, c = analytic(material(flow_rate=0.0), np.zeros(1), 0.0)
    dt = 0.0025 * A * A / c
    e, pc, pa_max, tr, _ = run("oracle", 16, 4, 2, dt, 20)
    assert pa_max > 1.012 and pc > 1.008, (pc, pa_max)              # the rise above p0 (1.5 % for this material: nu = 0.30, nu_u = 0.35), in the series and in the computation
    assert e < 0.01, e                                              # measured 4.0e-3 of p0
    assert len(tr) - 1 > 20                                         # genuine fixed-stress iterations


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,deg", [(32, 8, 2), (48, 8, 1)], ids=str)
def test_device_reproduces_mandels_solution(nx, ny, deg):
    _, _, c = analytic(material(flow_rate=0.0), np.zeros(1), 0.0)
    dt = 0.00125 * A * A / c
    e, pc, pa_max, tr, _ = run("hip", nx, ny, deg, dt, 40)
    assert pc > 1.008 and e < 0.01, (e, pc)
