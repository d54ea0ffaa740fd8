"""PORO_PREC_TWO_LEVEL on locally refined boxes (hanging nodes): displacement CG iterations and solve time per uniform refinement of the whole configuration, next to
Chebyshev-Jacobi and Jacobi on the same meshes (general matrix-free operator + operator-level condensation).  Usage: python tools/two_level.py [n ...] > out.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT]
import poroelasticity_dealii_amd as pk
import bench

sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32]
out = {"mesh": "n^3 box, cells [n/4, 3n/4)^3 refined once (hanging nodes on the block's faces), Q2/Q1", "rel_tol": 1e-8, "cases": []}
for n in sizes:
    P = pk.Problem.refined_box(3, [n] * 3, [10.0] * 3, 2, bench.material(), bench.BC_3D, [n // 4] * 3, [3 * n // 4] * 3)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    rec = {"coarse_cells": n, "n_cells": int(P.desc.n_cells), "n_dofs_u": int(P.desc.n_dofs_u), "hanging_dofs_u": int(P.desc.cons_u.n)}
    G.set(pk.VEC_P, bench.INPUT["p_init"] * (1 + 0.3 * np.sin(0.37 * np.arange(G.n_p)))); G.disp_assemble_system(True)
    for name, prec, cap in (("two_level", pk.PREC_TWO_LEVEL, 2000), ("chebyshev", pk.PREC_CHEBYSHEV, 20000), ("jacobi", pk.PREC_JACOBI, 100000)):
        if name == "jacobi" and n > 16:
            continue
        best = None
        for rep in range(2):
            G.fill(pk.VEC_U, 0.0); G.synchronize(); t0 = time.perf_counter()
            rc, info = G.disp_solve(abs_tol=1e-14, rel_tol=1e-8, max_iter=cap, prec=prec); G.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rec[name] = {"converged": rc == 0, "cg_iterations": int(info.iterations), "ms_per_solve": round(1e3 * best, 3)}
    out["cases"].append(rec); G.close(); P.close()
print(json.dumps(out, indent=1))
