"""Mesh independence of the fast-diagonalisation preconditioners on tensor-product grids WITHOUT the box tag (poro_desc.tensor; graded boxes): displacement CG iteration
counts and time of one fixed-stress time step per refinement, next to Chebyshev-Jacobi on the same meshes.  Usage: python tools/graded_fdm.py [n ...] > out.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT]
import poroelasticity_dealii_amd as pk
import bench

sizes = [int(a) for a in sys.argv[1:]] or [12, 24, 48]
grading = [1.5, 0.0, -1.0]
out = {"grading": grading, "degree": 2, "cases": []}
for n in sizes:
    P = pk.Problem.graded_box(3, [n] * 3, [10.0] * 3, 2, bench.material(), bench.BC_3D, grading)
    rec = {"cells": n, "n_dofs_u": int(P.desc.n_dofs_u), "cell_size_ratio_max_over_min": None}
    for name, prec in (("block_fdm", pk.PREC_FDM), ("chebyshev", pk.PREC_CHEBYSHEV)):
        R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=bench.INPUT["p_init"], dt=bench.INPUT["dt"], abs_u=1e-12, rel_u=1e-8, max_it=50000, prec=prec, reduction=True)
        R.initialize(); R.save_state(); ts = []
        for k in range(3):
            R.restore_state(); R.ctx.synchronize(); t0 = time.perf_counter(); tr, w = R.step(); R.ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        rec[name] = {"cg_iterations_u": int(tr[0][6]), "pressure_cg_iterations": int(tr[0][7]), "ms_per_step": round(min(ts), 3)}
        R.close()
    out["cases"].append(rec); P.close()
print(json.dumps(out, indent=1))
