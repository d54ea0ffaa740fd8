"""A whole time step (PoroelasticityFSS.h:328-407) on locally refined boxes with the host driver's automatic choice (PORO_PREC_TWO_LEVEL on the displacement and pressure systems, Jacobi on the projection),
next to the same step with the two-level form on the displacement system only (pressure / projection: Jacobi) - the CG counts per step and the step time.
Usage: python tools/refined_step.py [n ...] > out.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT]
import poroelasticity_dealii_amd as pk
import bench

sizes = [int(a) for a in sys.argv[1:]] or [8, 16, 32]
out = {"mesh": "n^3 box, cells [n/4, 3n/4)^3 refined once (hanging nodes on the block's faces), Q2/Q1, input.data tolerances", "steps_timed": 3, "cases": []}
for n in sizes:
    P = pk.Problem.refined_box(3, [n] * 3, [10.0] * 3, 2, bench.material(), bench.BC_3D, [n // 4] * 3, [3 * n // 4] * 3)
    rec = {"coarse_cells": n, "n_cells": int(P.desc.n_cells), "n_dofs_u": int(P.desc.n_dofs_u), "n_dofs_p": int(P.desc.n_dofs_p)}
    for name, jp in (("two_level_u_and_p", False), ("two_level_u_jacobi_p", True)):
        R = pk.Runner(P, 0, pk.OP_MATRIX_FREE, p_init=bench.INPUT["p_init"], dt=bench.INPUT["dt"], max_it=20000, prec=-1, jacobi_p=jp)
        R.initialize(); R.step(); R.ctx.synchronize()
        w0 = R.work(); t0 = time.perf_counter()
        for _ in range(3):
            tr, _w = R.step()
        R.ctx.synchronize(); dt = (time.perf_counter() - t0) / 3; w1 = R.work()
        rec[name] = {"ms_per_step": round(1e3 * dt, 3), "fss_iterations_last_step": len(tr), **{k: (w1[k] - w0[k]) / 3 for k in ("cg_u", "cg_p", "cg_proj")}}
        R.close()
    out["cases"].append(rec); P.close()
print(json.dumps(out, indent=1))
