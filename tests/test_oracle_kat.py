"""Known-answer tests that pin the CPU oracle.  The reference ships no tests or golden vectors and
deal.II cannot be run here, so these analytical facts — none of which depends on deal.II — are what anchors the
restatement; the committed golden trace (tests/golden/) additionally freezes the oracle's own output."""
import json
import os

import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import BC_2D, BC_3D, DOMAIN_MSH, GOLDEN, INPUT_DATA, REF, box_problem, csr_to_scipy, material, node_coords_box


def test_k1_derived_constants():
    """R18: InputDataPoroel::compute_derived_parameters on input.data (values quoted in SURVEY 8a R18)."""
    g = oracle_py.derived_parameters(REF["E"], REF["nu"], REF["alpha"], REF["poro"], REF["f_comp"], REF["perm_mD"], REF["visc"])
    E, nu, al, ph, cf = REF["E"], REF["nu"], REF["alpha"], REF["poro"], REF["f_comp"]
    lam = E * nu / ((1 + nu) * (1 - 2 * nu)); G = E / (2 * (1 + nu)); K = lam + 2 * G / 3
    Ks = K / (1 - al); N = Ks / (al - ph); M = (N / cf) / (N * ph + 1 / cf)
    for k, v in dict(zip(["lambda", "G", "K", "Ks", "N", "M"], [lam, G, K, Ks, N, M])).items():
        assert abs(g[k] - v) <= 1e-15 * abs(v), k
    assert abs(g["lambda"] - 8.0769230769e9) < 1e0 and abs(g["G"] - 5.3846153846e9) < 1e0 and abs(g["K"] - 1.1666666667e10) < 1e1
    assert abs(g["Ks"] - 1.1666666667e11) < 1e2 and abs(g["N"] - 1.9444444444e11) < 1e2 and abs(g["M"] - 5.5821371611e9) < 1e0
    assert abs(g["k_over_mu"] - 9.869233e-12) < 1e-24
    assert abs(REF["alpha"] / g["K"] - 7.714286e-11) < 1e-16


@pytest.mark.parametrize("dim,n,deg", [(2, 16, 1), (2, 16, 2), (3, 3, 1), (3, 2, 2)])
def test_k2_patch_test(dim, n, deg):
    """the reference's initialisation step (PoroelasticityFSS.h:311-317): uniform p => int alpha p div(phi_i) vanishes on free dofs,
    u is the linear field of the Dirichlet data, projected eps_aa = -1e-6, eps_v0 = -dim*1e-6."""
    P = box_problem(dim, n, deg)
    O = oracle_py.Oracle(P)
    O.fill(pk.VEC_P, REF["p_init"]); O.disp_assemble_system(True)
    rc, info = O.disp_solve()                       # reference controls: abs 1e-12, 1000 its, SSOR(1.2)
    assert rc == 0 and info.iterations < 1000
    X = node_coords_box(dim, n, deg); u = O.get(pk.VEC_U)
    for c in range(dim):
        assert np.abs(u[c::dim] + 1e-5 * (X[:, c] + 5) / 10).max() <= 1e-17
    comps = [a * dim + a for a in range(dim)]
    O.proj_assemble_matrix(); O.proj_assemble_rhs(comps)
    for e in ([0, 2] if dim == 2 else [0, 3, 5]):
        assert O.proj_solve(e, rel_tol=1e-13)[0] == 0
        assert np.abs(O.get(pk.VEC_STRAIN0 + e) + 1e-6).max() <= 1e-16
    O.get_volumetric_strain()
    assert np.abs(O.get(pk.VEC_EPSV) + dim * 1e-6).max() <= 1e-15
    O.close(); P.close()


def test_k3_first_residual_is_minus_source():
    """eps_v = eps_v0, p = p_old = const => R = -q up to ||K.1|| round-off; sum q = s * (well area resolved by the quadrature)."""
    P = box_problem(2, 16, 2)
    O = oracle_py.Oracle(P)
    O.fill(pk.VEC_P, REF["p_init"]); O.fill(pk.VEC_P_OLD, REF["p_init"]); O.fill(pk.VEC_EPSV, -2e-6); O.fill(pk.VEC_EPSV0, -2e-6)
    l2 = O.pres_assemble_residual(REF["dt"])
    R, q = O.get(pk.VEC_RESIDUAL_P), O.get(pk.VEC_SOURCE_P)
    assert np.abs(R + q).max() <= 1e-9 * np.abs(q).max()      # K*p_const round-off scaled by k/mu*p ~ 1e-4*1e-16
    assert abs(l2 - np.linalg.norm(R)) <= 1e-14 * l2
    s = -REF["flow_rate"] / (3.1415926 * REF["r_well"] ** 2)
    assert abs(s + 3.1830989161e-6) < 1e-15
    # quadrature points of QGauss(2) inside the unit-radius well: h = 10/16, points at cell_origin + h*(1/2 -+ 1/(2 sqrt 3))
    h = 10 / 16; g = np.array([0.5 - 0.5 / np.sqrt(3), 0.5 + 0.5 / np.sqrt(3)])
    pts = (-5 + h * (np.arange(16)[:, None] + g[None, :])).ravel()
    inside = (pts[:, None] ** 2 + pts[None, :] ** 2) <= REF["r_well"] ** 2
    assert abs(q.sum() - s * inside.sum() * (h / 2) ** 2) <= 1e-18
    O.close(); P.close()


@pytest.mark.parametrize("dim,n,deg", [(2, 6, 1), (2, 5, 2), (3, 3, 1), (3, 2, 2)])
def test_k4_matrix_identities(dim, n, deg):
    """sum M = |Omega|, K.1 = 0, symmetry; the unconstrained stiffness annihilates the 3 (2D) / 6 (3D) rigid-body modes."""
    P = box_problem(dim, n, deg, bc=[])               # no Dirichlet conditions: pure stiffness
    O = oracle_py.Oracle(P)
    M = csr_to_scipy(*O.export_csr(pk.MAT_MASS_P)); K = csr_to_scipy(*O.export_csr(pk.MAT_LAPLACE_P))
    assert abs(M.sum() - 10.0 ** dim) <= 1e-11 * 10.0 ** dim
    assert np.abs(K @ np.ones(K.shape[0])).max() <= 1e-12 * abs(K).max()
    assert abs(M - M.T).max() == 0 or abs(M - M.T).max() <= 1e-15 * abs(M).max()
    assert abs(K - K.T).max() <= 1e-14 * abs(K).max()
    O.fill(pk.VEC_P, 0.0); O.disp_assemble_system(True)
    A = csr_to_scipy(*O.export_csr(pk.MAT_A_U))
    assert abs(A - A.T).max() <= 1e-13 * abs(A).max()
    X = node_coords_box(dim, n, deg); N = X.shape[0]
    modes = []
    for c in range(dim):
        t = np.zeros((N, dim)); t[:, c] = 1; modes.append(t.ravel())
    for a in range(dim):
        for b in range(a + 1, dim):
            r = np.zeros((N, dim)); r[:, a] = -X[:, b]; r[:, b] = X[:, a]; modes.append(r.ravel())
    assert len(modes) == (3 if dim == 2 else 6)
    for m in modes:
        assert np.abs(A @ m).max() <= 1e-12 * abs(A).max() * np.abs(m).max()
    # and nothing else: a pure stretch has energy
    s = np.zeros((N, dim)); s[:, 0] = X[:, 0]
    assert s.ravel() @ (A @ s.ravel()) > 1e-3 * abs(A).max()
    # translation invariance of the element matrix: all interior element blocks of the uniform mesh are equal
    O.close(); P.close()


def test_k5_single_element_q1_plane_strain_closed_form():
    """one Q1 square cell: the stiffness from the closed-form integrals (sympy) of lambda div.div + 2G eps:eps."""
    import sympy as sp
    lam, G = material().lame_lambda, material().shear_G
    x, y = sp.symbols("x y")
    Lh = 10
    N = [(1 - x / Lh) * (1 - y / Lh), (x / Lh) * (1 - y / Lh), (1 - x / Lh) * (y / Lh), (x / Lh) * (y / Lh)]   # lexicographic vertices on [0,L]^2
    def eps(s, c):
        g = [sp.diff(N[s], x), sp.diff(N[s], y)]
        E = sp.zeros(2, 2)
        for b in range(2):
            E[c, b] += g[b] / 2; E[b, c] += g[b] / 2
        return E
    Kref = np.zeros((8, 8))
    for i in range(8):
        Ei = eps(i // 2, i % 2)
        for j in range(8):
            Ej = eps(j // 2, j % 2)
            integrand = lam * Ei.trace() * Ej.trace() + 2 * G * sum(Ei[a, b] * Ej[a, b] for a in range(2) for b in range(2))
            Kref[i, j] = float(sp.integrate(sp.integrate(integrand, (x, 0, Lh)), (y, 0, Lh)))
    P = box_problem(2, 1, 1, bc=[])
    O = oracle_py.Oracle(P)
    O.fill(pk.VEC_P, 0.0); O.disp_assemble_system(True)
    A = csr_to_scipy(*O.export_csr(pk.MAT_A_U)).toarray()
    assert np.abs(A - Kref).max() <= 1e-13 * np.abs(Kref).max()
    O.close(); P.close()


@pytest.mark.parametrize("deg", [1, 2])
def test_k6_uniaxial_pressure_gradient(deg):
    """linear p(x) with u_x = 0 on the x faces and u_y = 0 on the y faces: (lambda+2G) u_x'' = alpha p'
    => u_x = alpha p' (x^2 - 25) / (2 (lambda+2G)), u_y = 0; quadratic, so Q2 is exact and Q1 is nodally exact."""
    n = 8
    bc = [(0, 0, 0.0), (1, 0, 0.0), (2, 1, 0.0), (3, 1, 0.0)]
    P = box_problem(2, n, deg, bc=bc)
    O = oracle_py.Oracle(P)
    m = material(); dpdx = 2.0e5
    Xp = node_coords_box(2, n, 1)
    O.set(pk.VEC_P, 1e6 + dpdx * Xp[:, 0]); O.disp_assemble_system(True)
    rc, _ = O.disp_solve(abs_tol=1e-14, max_iter=5000)
    assert rc == 0
    X = node_coords_box(2, n, deg); u = O.get(pk.VEC_U)
    ux = m.biot_alpha * dpdx * (X[:, 0] ** 2 - 25.0) / (2 * (m.lame_lambda + 2 * m.shear_G))
    assert np.abs(u[0::2] - ux).max() <= 1e-11 * np.abs(ux).max()
    assert np.abs(u[1::2]).max() <= 1e-11 * np.abs(ux).max()
    O.close(); P.close()


def test_naive_and_hoisted_assembly_agree():
    """the reference-faithful i x q x j loop (quirk Q6) and the hoisted variant used for larger parity runs give the same matrix"""
    P = box_problem(3, 2, 2)
    A, B = oracle_py.Oracle(P, hoisted=False), oracle_py.Oracle(P, hoisted=True)
    for O in (A, B):
        O.fill(pk.VEC_P, 3e6); O.disp_assemble_system(True)
    va, vb = A.export_csr(pk.MAT_A_U)[2], B.export_csr(pk.MAT_A_U)[2]
    assert np.abs(va - vb).max() <= 1e-15 * np.abs(va).max()
    assert np.abs(A.get(pk.VEC_RHS_U) - B.get(pk.VEC_RHS_U)).max() <= 1e-15 * np.abs(A.get(pk.VEC_RHS_U)).max()
    A.close(); B.close(); P.close()


def test_ssor_cg_equals_direct_solve():
    """SolverCG + PreconditionSSOR restatement against a sparse direct solve of the oracle's own system"""
    import scipy.sparse.linalg as spla
    P = box_problem(2, 8, 2)
    O = oracle_py.Oracle(P)
    p = 10e6 * (1 + 0.1 * np.sin(0.37 * np.arange(O.n_p)))
    O.set(pk.VEC_P, p); O.disp_assemble_system(True)
    A = csr_to_scipy(*O.export_csr(pk.MAT_A_U)).tocsc(); b = O.get(pk.VEC_RHS_U)
    rc, info = O.disp_solve()
    assert rc == 0
    u = O.get(pk.VEC_U)
    d = P.desc
    dd = np.ctypeslib.as_array(d.dirichlet_dof, shape=(d.n_dirichlet,)); dv = np.ctypeslib.as_array(d.dirichlet_value, shape=(d.n_dirichlet,))
    x = spla.spsolve(A, b); x[dd] = dv
    assert np.linalg.norm(u - x) <= 1e-10 * np.linalg.norm(x)
    O.close(); P.close()


def test_k7_config1_golden_trace():
    """BASELINE config 1: domain.msh + input.data, Q1/Q1 (BASELINE) and Q2/Q1 (reference behaviour), 1 timestep.
    The trace was generated by tests/golden/make_golden.py from this oracle; quirk Q1 => exactly one FSS iteration."""
    with open(os.path.join(GOLDEN, "config1_trace.json")) as f:
        gold = json.load(f)
    inp = pk.read_input(INPUT_DATA)
    for deg in (1, 2):
        P = pk.Problem.gmsh(DOMAIN_MSH, deg, inp.material, BC_2D)
        O = oracle_py.Oracle(P)
        tr, _ = O.run(1, inp.p_init, inp.time_step, inp.fss_tol, inp.pressure_tol, inp.max_fss_iterations, inp.max_pressure_iterations)
        g = gold[f"Q{deg}"]
        assert O.noconvergence_count() == 0
        assert tr.shape[0] == 2 and tr[1, 1] == 1                       # initialisation row + ONE fixed-stress iteration (quirk Q1)
        assert int(tr[1, 2]) == g["pressure_iterations"]
        assert tr[1, 3] < 1e-8 and tr[1, 5] < 1e-8
        assert abs(tr[1, 4] - g["p_linf"]) <= 1e-9 * g["p_linf"]
        assert abs(np.linalg.norm(O.get(pk.VEC_U)) - g["u_l2"]) <= 1e-8 * g["u_l2"]
        assert abs(np.linalg.norm(O.get(pk.VEC_P)) - g["p_l2"]) <= 1e-10 * g["p_l2"]
        assert abs(np.linalg.norm(O.get(pk.VEC_EPSV)) - g["epsv_l2"]) <= 1e-6 * g["epsv_l2"]
        O.close(); P.close()


def test_box_regression_goldens():
    """Committed oracle traces of three small box configurations (tests/golden/box_traces.json, made by make_golden.py from this oracle): the
    reference loop and the coupled / incremental variant.  Guards the oracle itself against silent changes."""
    from common import REF, box_problem, host_material
    with open(os.path.join(GOLDEN, "box_traces.json")) as f:
        gold = json.load(f)
    for name, g in gold.items():
        P = box_problem(g["dim"], g["n"], g["degree"], mat=host_material())
        O = oracle_py.Oracle(P)
        kw = {"coupled_fss": True, "incremental_strain": True} if g["variant"] == "coupled" else {}
        tr, _ = O.run(2, REF["p_init"], REF["dt"], max_it=2000, **kw)
        rows = np.array(g["rows"])
        assert tr.shape[0] == rows.shape[0] and np.array_equal(tr[:, :3], rows[:, :3]), name
        assert np.allclose(tr[:, 4], rows[:, 4], rtol=1e-10), name
        assert abs(np.linalg.norm(O.get(pk.VEC_U)) - g["u_l2"]) <= 1e-8 * g["u_l2"], name
        assert abs(np.linalg.norm(O.get(pk.VEC_P)) - g["p_l2"]) <= 1e-10 * g["p_l2"], name
        assert O.noconvergence_count() == g["noconvergence"]
        O.close(); P.close()
