"""Structured operator on a box whose displacement vector exceeds 2^31 bytes (the kernels address memory with 32-bit offsets relative to each z-chunk's first plane):
3D Q1, 500 x 500 x 400 cells = 100.6 M nodes, 301.8 M displacement dofs, 2.41 GB per vector.  The context's set-up self-check compares the structured kernel with the
generic element-matrix kernel (64-bit addressing) on the full vector; here additionally the rigid-body null space and symmetry.  Usage: python tools/big_box_check.py [nx ny nz [degree]]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import poroelasticity_dealii_amd as pk
from bench import material

n = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [500, 500, 400]
deg = int(sys.argv[4]) if len(sys.argv) > 4 else 1
t0 = time.time()
P = pk.Problem.box(3, n, [10.0, 10.0, 8.0], deg, material(), [])          # no Dirichlet conditions: the operator has the rigid-body modes in its null space
t1 = time.time(); print("host mesh", round(t1 - t0, 1), "s", flush=True)
G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
G.fill(pk.VEC_P, 0.0); G.disp_assemble_system(True)                      # runs the structured-vs-generic self-check on the full vector
t2 = time.time(); print("context + assembly + self-check", round(t2 - t1, 1), "s", flush=True)
nu = G.n_u; nn = [deg * m + 1 for m in n]
out = {"cells": n, "degree": deg, "N_u": int(nu), "bytes_per_vector": int(8 * nu)}
scale = float(G.get(pk.VEC_DIAG_U).max())
ax = [np.linspace(-5, 5, nn[0]), np.linspace(-5, 5, nn[1]), np.linspace(-4, 4, nn[2])]
z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
res = []
t = np.zeros((x.size, 3)); t[:, 2] = 1.0
res.append(float(np.abs(G.apply(pk.MAT_A_U, t.ravel())).max() / scale))
t[:] = 0; t[:, 0] = -y.ravel(); t[:, 1] = x.ravel()
res.append(float(np.abs(G.apply(pk.MAT_A_U, t.ravel())).max() / scale / 5.0))
out["rigid_body_residual_rel"] = res
a, b = np.sin(0.37 * np.arange(nu)), np.cos(0.11 * np.arange(nu)) + 0.2
Aa, Ab = G.apply(pk.MAT_A_U, a), G.apply(pk.MAT_A_U, b)
out["symmetry_rel"] = float(abs(b @ Aa - a @ Ab) / (abs(b @ Aa) + np.linalg.norm(Aa) * np.linalg.norm(b)))
out["spd"] = bool(a @ Aa > 0)
out["us_per_apply"] = 1e6 * G.bench_operator(pk.OP_MATRIX_FREE, 10)
out["GBs_index_free"] = 16.0 * nu / out["us_per_apply"] / 1e3
print(json.dumps(out), flush=True)
assert max(res) < 1e-12 and out["symmetry_rel"] < 1e-12 and out["spd"]
G.close(); P.close()
