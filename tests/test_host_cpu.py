"""CPU tests of the host layer (mesh / DoF / FE-table provider, parameter-file front end) and of the C-ABI library's
loadability.  No compute calls are made: there is no GPU here and the product has no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import poroelasticity_dealii_amd as pk
import oracle_py
from common import BC_2D, BC_3D, DOMAIN_MSH, INPUT_DATA, box_problem, material

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "poroel_hip.h")).read()
    declared = set(re.findall(r"\b(poro_[a-z_0-9]+)\s*\(", hdr)) - {"poro_allreduce_fn", "poro_sendrecv_fn"}
    assert declared == set(pk.HIP_SYMBOLS), declared ^ set(pk.HIP_SYMBOLS)
    L = pk.load_hip()
    for s in declared:
        assert hasattr(L, s), s
    L.poro_abi_version.restype = C.c_int
    assert L.poro_abi_version() == 4


def test_product_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    P = box_problem(2, 4, 2)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        pk.Context(P, 0, pk.OP_CSR)
    P.close()


def test_product_sources_never_touch_the_oracle():
    pkg = os.path.join(ROOT, "poroelasticity_dealii_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(d, f)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("deg", [1, 2])
def test_fe_tables_match_the_oracles_independent_tables(dim, deg):
    P = box_problem(dim, 2, deg)
    fe = P.desc.fe
    ns, nq, nqp, nsp = (deg + 1) ** dim, (deg + 1) ** dim, 2 ** dim, 2 ** dim
    assert (fe.nq_u, fe.nq_p, fe.ns_u, fe.ns_p, fe.nq_f) == (nq, nqp, ns, nsp, (deg + 1) ** (dim - 1))
    arr = lambda p, n: np.ctypeslib.as_array(p, shape=(n,))
    tol = 2e-15
    assert np.abs(arr(fe.w_qu, nq) - oracle_py.fe_table(dim, deg, deg + 1, 2)).max() < tol
    assert np.abs(arr(fe.w_qp, nqp) - oracle_py.fe_table(dim, 1, 2, 2)).max() < tol
    assert np.abs(arr(fe.u_qu, nq * ns) - oracle_py.fe_table(dim, deg, deg + 1, 0)).max() < tol
    assert np.abs(arr(fe.du_qu, nq * ns * dim) - oracle_py.fe_table(dim, deg, deg + 1, 1)).max() < 8 * tol
    assert np.abs(arr(fe.du_qp, nqp * ns * dim) - oracle_py.fe_table(dim, deg, 2, 1)).max() < 8 * tol
    assert np.abs(arr(fe.q1_qu, nq * nsp) - oracle_py.fe_table(dim, 1, deg + 1, 0)).max() < tol
    assert np.abs(arr(fe.dq1_qu, nq * nsp * dim) - oracle_py.fe_table(dim, 1, deg + 1, 1)).max() < 8 * tol
    assert np.abs(arr(fe.q1_qp, nqp * nsp) - oracle_py.fe_table(dim, 1, 2, 0)).max() < tol
    assert np.abs(arr(fe.dq1_qp, nqp * nsp * dim) - oracle_py.fe_table(dim, 1, 2, 1)).max() < 8 * tol
    assert abs(arr(fe.w_qu, nq).sum() - 1) < 1e-15 and abs(arr(fe.w_qf, fe.nq_f).sum() - 1) < 1e-15
    # partition of unity of the face tables, zero off the face
    uf = arr(fe.u_qf, 2 * dim * fe.nq_f * ns).reshape(2 * dim, fe.nq_f, ns)
    assert np.abs(uf.sum(axis=2) - 1).max() < 1e-14
    P.close()


def test_box_mesh_sizes_follow_survey_formulas():
    for dim, n, deg in [(2, 16, 2), (2, 10, 1), (3, 4, 2), (3, 5, 1)]:
        P = box_problem(dim, n, deg)
        d = P.desc
        assert d.n_cells == n ** dim and d.n_dofs_p == (n + 1) ** dim and d.n_dofs_u == dim * (deg * n + 1) ** dim
        assert d.n_bfaces == 2 * dim * n ** (dim - 1)
        ids = np.ctypeslib.as_array(d.bface_id, shape=(d.n_bfaces,))
        assert sorted(set(ids)) == list(range(2 * dim))                              # colorize: 2d / 2d+1 per direction
        assert d.n_dirichlet == 2 * dim * (deg * n + 1) ** (dim - 1)                  # one component per face, faces are disjoint per component
        X = np.ctypeslib.as_array(d.vertex_coords, shape=(d.n_vertices, dim))
        assert X.min() == -5.0 and X.max() == 5.0                                    # hyper_rectangle(+size/2, -size/2)
        assert d.box.enabled == 1 and list(d.box.n)[:dim] == [n] * dim
        P.close()
    P = box_problem(2, 16, 2)                                                        # reference default: refinement level 4 (input.data:5)
    assert (P.desc.n_dofs_u, P.desc.n_dofs_p) == (2178, 289)
    P.close()


def test_gmsh_reader_on_the_bundled_domain_msh():
    """domain.msh: 121 nodes, 40 boundary lines with physical tags 0..3 = bottom/right/top/left, 100 quads."""
    for deg, nu in ((1, 242), (2, 882)):
        P = pk.Problem.gmsh(DOMAIN_MSH, deg, material(), BC_2D)
        d = P.desc
        assert (d.n_vertices, d.n_cells, d.n_bfaces, d.n_dofs_p, d.n_dofs_u) == (121, 100, 40, 121, nu)
        assert d.box.enabled == 0
        X = np.ctypeslib.as_array(d.vertex_coords, shape=(121, 2))
        cv = np.ctypeslib.as_array(d.cell_vertices, shape=(100, 4))
        # lexicographic local ordering: positive Jacobian, v0->v1 and v2->v3 parallel
        e1, e2 = X[cv[:, 1]] - X[cv[:, 0]], X[cv[:, 2]] - X[cv[:, 0]]
        assert np.all(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0] > 0)
        area = 0.5 * np.abs((X[cv[:, 3]] - X[cv[:, 0]])[:, 0] * (X[cv[:, 2]] - X[cv[:, 1]])[:, 1] - (X[cv[:, 3]] - X[cv[:, 0]])[:, 1] * (X[cv[:, 2]] - X[cv[:, 1]])[:, 0])
        assert abs(area.sum() - 100.0) < 1e-9
        bc_, bl, bi = (np.ctypeslib.as_array(getattr(d, k), shape=(40,)) for k in ("bface_cell", "bface_local", "bface_id"))
        fv = np.array([[0, 2], [1, 3], [0, 1], [2, 3]])
        mid = 0.5 * (X[cv[bc_, fv[bl, 0]]] + X[cv[bc_, fv[bl, 1]]])
        assert np.allclose(mid[bi == 0][:, 1], -5) and np.allclose(mid[bi == 1][:, 0], 5) and np.allclose(mid[bi == 2][:, 1], 5) and np.allclose(mid[bi == 3][:, 0], -5)
        assert [int((bi == k).sum()) for k in range(4)] == [10, 10, 10, 10]
        # Dirichlet list: first condition wins on shared corners (interpolate_boundary_values skips constrained dofs)
        dd = np.ctypeslib.as_array(d.dirichlet_dof, shape=(d.n_dirichlet,)); dv = np.ctypeslib.as_array(d.dirichlet_value, shape=(d.n_dirichlet,))
        per_side = 10 * deg + 1
        assert d.n_dirichlet == 4 * per_side - 2                                     # u_x: bottom+right share one corner, u_y: top+left share one
        assert len(set(dd)) == d.n_dirichlet and set(np.unique(dv)) == {0.0, -1e-5}
        P.close()


def test_parameter_file_front_end():
    i = pk.read_input(INPUT_DATA)
    assert (i.dim, list(i.domain_size)[:2], i.initial_refinement_level, i.max_refinement_level) == (2, [10.0, 10.0], 4, 6)
    assert (i.youngs_modulus, i.poisson_ratio, i.biot_coef, i.bulk_density, i.f_comp, i.poro, i.visc, i.r_well, i.flow_rate) == (1.4e10, 0.3, 0.9, 2700, 5.8e-10, 0.3, 1e-3, 1, 1e-5)
    assert i.perm == 10 * 9.869233e-16                                             # mD -> m^2 (InputDataPoroel.h:162,168)
    assert (i.time_step, i.t_max, i.p_init) == (60, 1e3, 10e6)
    assert (i.fss_tol, i.pressure_tol, i.max_fss_iterations, i.max_pressure_iterations) == (1e-8, 1e-8, 50, 50)   # declared defaults :138-141
    assert i.n_dirichlet == 4 and list(i.dirichlet_labels)[:4] == [0, 1, 2, 3] and list(i.dirichlet_components)[:4] == [0, 0, 1, 1]
    assert list(i.dirichlet_values)[:4] == [0, -1e-5, 0, -1e-5] and i.n_neumann == 0
    g = oracle_py.derived_parameters(1.4e10, 0.3, 0.9, 0.3, 5.8e-10, 10, 1e-3)
    assert (i.lame_constant, i.shear_modulus, i.bulk_modulus, i.grain_bulk_modulus, i.n_modulus, i.m_modulus) == (g["lambda"], g["G"], g["K"], g["Ks"], g["N"], g["M"])
    assert i.material.k_over_mu == g["k_over_mu"]
    d = pk.read_input(None)                                                          # declare_parameters defaults (:89-147)
    assert (d.youngs_modulus, d.perm, d.r_well, d.flow_rate, d.t_max, d.initial_refinement_level) == (7e9, 9.869233e-16, 0.1, 1e-6, 60, 3)
    assert list(d.dirichlet_labels)[:4] == [0, 2, 3, 1] and list(d.dirichlet_values)[:4] == [0, 0, 0, -0.1]


def test_parameter_file_rejects_out_of_range(tmp_path):
    bad = tmp_path / "bad.data"
    bad.write_text("subsection Properties\n  set Poisson ratio = 0.7\nend\n")       # Patterns::Double(0, 0.5)
    with pytest.raises(RuntimeError, match="out of range"):
        pk.read_input(str(bad))
    bad.write_text("subsection Mesh\n  set Dimensions 2\nend\n")
    with pytest.raises(RuntimeError):
        pk.read_input(str(bad))
    with pytest.raises(RuntimeError, match="cannot open"):
        pk.read_input(str(tmp_path / "missing.data"))


@pytest.mark.parametrize("dim,n,deg,ranks", [(2, (6, 7), 2, 3), (3, (3, 3, 8), 1, 4), (3, (2, 2, 5), 2, 2)])
def test_slab_partition_tiles_the_global_box(dim, n, deg, ranks):
    """z-slabs (y-slabs in 2D) of whole cell layers; consecutive slabs share exactly one node plane."""
    Pg = box_problem(dim, n, deg)
    tot_cells = 0; own_u = 0; own_p = 0; z_hi = None
    for r in range(ranks):
        P = box_problem(dim, n, deg, rank=r, n_ranks=ranks)
        d = P.desc; part = d.part
        assert (part.rank, part.n_ranks, part.has_lower, part.has_upper) == (r, ranks, int(r > 0), int(r < ranks - 1))
        plane_u = dim * int(np.prod([deg * m + 1 for m in n[:-1]])); plane_p = int(np.prod([m + 1 for m in n[:-1]]))
        assert (part.plane_u, part.plane_p) == (plane_u, plane_p)
        tot_cells += d.n_cells
        own_u += d.n_dofs_u - (plane_u if part.has_upper else 0); own_p += d.n_dofs_p - (plane_p if part.has_upper else 0)
        X = np.ctypeslib.as_array(d.vertex_coords, shape=(d.n_vertices, dim))
        if z_hi is not None:
            assert abs(X[:, -1].min() - z_hi) < 1e-12                                # slabs abut
        z_hi = X[:, -1].max()
        ids = set(np.ctypeslib.as_array(d.bface_id, shape=(d.n_bfaces,)))
        assert (2 * (dim - 1) in ids) == (r == 0) and (2 * (dim - 1) + 1 in ids) == (r == ranks - 1)   # interface planes are not boundaries
        P.close()
    assert (tot_cells, own_u, own_p) == (Pg.desc.n_cells, Pg.desc.n_dofs_u, Pg.desc.n_dofs_p)
    assert abs(z_hi - 5.0) < 1e-12
    Pg.close()


def test_graded_box_carries_its_tensor_grid():
    """poro_desc.tensor of the graded-box provider: no box tag, the 1D vertex grids of the tensor product are exactly the distinct vertex coordinates per direction
    (the fast-diagonalisation preconditioners assemble their 1D matrices on them)"""
    import numpy as np
    from common import BC_3D, material
    n, grading = (5, 4, 6), (1.5, 0.0, -0.8)
    P = pk.Problem.graded_box(3, list(n), [10.0] * 3, 2, material(), BC_3D, list(grading))
    try:
        d = P.desc
        assert d.box.enabled == 0 and d.tensor.enabled == 1 and list(d.tensor.n) == list(n)
        X = np.ctypeslib.as_array(d.vertex_coords, shape=(d.n_vertices, 3))
        for a in range(3):
            g = np.ctypeslib.as_array(d.tensor.grid[a], shape=(n[a] + 1,))
            assert np.all(np.diff(g) > 0) and abs(g[0] + 5.0) <= 1e-13 and abs(g[-1] - 5.0) <= 1e-13
            assert np.abs(np.unique(np.round(X[:, a], 12)) - g).max() <= 1e-11
            ratio = np.diff(g).max() / np.diff(g).min()
            assert (ratio < 1 + 1e-9) == (grading[a] == 0.0)
    finally:
        P.close()
