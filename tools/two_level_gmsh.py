"""PORO_PREC_TWO_LEVEL on read_mesh()'s Gmsh grid and its uniform refinements (auxiliary uniform box as coarse space): displacement CG iterations and solve time next to
Jacobi and Chebyshev.  Usage: python tools/two_level_gmsh.py [max_refine] > out.json"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT]
import poroelasticity_dealii_amd as pk
import bench

MSH = os.path.join(ROOT, "tests", "golden", "domain.msh")
BC_2D = bench.BC_3D[:4]
out = {"mesh": "tests/golden/domain.msh (10 x 10 quadrilaterals, Gmsh numbering and boundary ids) after r uniform refinements, Q2/Q1", "rel_tol": 1e-10, "cases": []}
for r in range(int(sys.argv[1]) + 1 if len(sys.argv) > 1 else 6):
    P = pk.Problem.gmsh(MSH, 2, bench.material(), BC_2D, refine=r)
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    rec = {"refinements": r, "n_cells": int(P.desc.n_cells), "n_dofs_u": int(P.desc.n_dofs_u)}
    G.set(pk.VEC_P, bench.INPUT["p_init"] * (1 + 0.2 * np.sin(0.37 * np.arange(G.n_p)))); G.disp_assemble_system(True)
    for name, prec, cap in (("two_level", pk.PREC_TWO_LEVEL, 1000), ("chebyshev", pk.PREC_CHEBYSHEV, 20000), ("jacobi", pk.PREC_JACOBI, 200000)):
        if not G.supports_preconditioner(0, prec):       # (the box's lines pass 320 points at r = 4: with different conditions at their two ends only the nodal kernels of up to 320 points apply)
            rec[name] = "not supported at this size"; continue
        best = None
        for rep in range(2):
            G.fill(pk.VEC_U, 0.0); G.synchronize(); t0 = time.perf_counter()
            rc, info = G.disp_solve(abs_tol=1e-14, rel_tol=1e-10, max_iter=cap, prec=prec); G.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rec[name] = {"converged": rc == 0, "cg_iterations": int(info.iterations), "ms_per_solve": round(1e3 * best, 3)}
    out["cases"].append(rec); G.close(); P.close()
print(json.dumps(out, indent=1))
