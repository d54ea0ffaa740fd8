"""Shared problem definitions of the test-suite: the bundled input.data values (reference input.data:1-41),
committed here as numbers because /root/reference does not exist on the GPU box."""
import os

import numpy as np

import poroelasticity_dealii_amd as pk

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INPUT_DATA = os.path.join(GOLDEN, "input.data")       # byte copy of the reference's example parameter file (a data fixture)
DOMAIN_MSH = os.path.join(GOLDEN, "domain.msh")       # byte copy of the reference's example Gmsh mesh (a data fixture)

# derived constants of input.data
REF = dict(E=1.4e10, nu=0.3, alpha=0.9, poro=0.3, f_comp=5.8e-10, perm_mD=10.0, visc=1e-3, r_well=1.0, flow_rate=1e-5, p_init=10e6, dt=60.0)


def material(**over):
    import oracle_py
    d = dict(REF); d.update(over)
    g = oracle_py.derived_parameters(d["E"], d["nu"], d["alpha"], d["poro"], d["f_comp"], d["perm_mD"], d["visc"])
    return pk.Material(g["lambda"], g["G"], d["alpha"], g["K"], g["M"], g["k_over_mu"], d["r_well"], d["flow_rate"])


def host_material(path=None):
    """material through the product's own parameter front end"""
    return pk.read_input(path or INPUT_DATA).material


# input.data:14-16 (x faces fix u_x, y faces fix u_y); z faces per SURVEY Q9
BC_2D = [(0, 0, 0.0), (1, 0, -1e-5), (2, 1, 0.0), (3, 1, -1e-5)]
BC_3D = BC_2D + [(4, 2, 0.0), (5, 2, -1e-5)]


def box_problem(dim, n, degree, rank=0, n_ranks=1, mat=None, bc=None, neumann=()):
    n = [n] * dim if np.isscalar(n) else list(n)
    return pk.Problem.box(dim, n, [10.0] * dim, degree, mat or material(), bc if bc is not None else (BC_2D if dim == 2 else BC_3D), neumann, rank, n_ranks)


def node_coords_box(dim, n, degree, size=10.0):
    """coordinates of the lexicographic u nodes of the box"""
    n = [n] * dim if np.isscalar(n) else list(n)
    ax = [np.linspace(-size / 2, size / 2, degree * m + 1) for m in n]
    grids = np.meshgrid(*ax[::-1], indexing="ij")     # slowest first
    return np.stack([g.ravel() for g in grids[::-1]], axis=1)   # [node][x,y(,z)]


def csr_to_scipy(rp, col, val):
    import scipy.sparse as sp
    return sp.csr_matrix((val, col, rp), shape=(len(rp) - 1, len(rp) - 1))


def global_problem(mesh, deg):
    """global (unpartitioned) problem of the general-partition tests: mesh = "gmsh" (the bundled domain.msh) | "box:nx,ny[,nz]" | "refined:nx,ny[,nz]" (the box with its
    middle block [n/4, 3n/4) refined once: hanging nodes)"""
    if mesh == "gmsh":
        return pk.Problem.gmsh(DOMAIN_MSH, deg, material(), BC_2D)
    n = [int(v) for v in mesh.split(":")[1].split(",")]
    if mesh.startswith("refined:"):
        return pk.Problem.refined_box(len(n), n, [10.0] * len(n), deg, material(), BC_2D if len(n) == 2 else BC_3D, [m // 4 for m in n], [max(3 * m // 4, m // 4 + 1) for m in n])
    return box_problem(len(n), n, deg)
