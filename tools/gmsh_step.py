"""A time step on read_mesh()'s Gmsh grid refined r times, the host driver choosing the preconditioners by itself (auxiliary-box two-level form for the displacement
system and, from 4096 pressure dofs on, the pressure system).  Usage: python tools/gmsh_step.py [r ...] > out.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import poroelasticity_dealii_amd as pk
import bench
from common import BC_2D, DOMAIN_MSH

out = {"mesh": "tests/golden/domain.msh refined r times, Q2/Q1, input.data tolerances", "cases": []}
for r in [int(a) for a in sys.argv[1:]] or [3, 4, 5]:
    P = pk.Problem.gmsh(DOMAIN_MSH, 2, bench.material(), BC_2D, refine=r)
    rec = {"refinements": r, "n_cells": int(P.desc.n_cells), "n_dofs_u": int(P.desc.n_dofs_u), "n_dofs_p": int(P.desc.n_dofs_p)}
    for name, jp in (("automatic", False), ("jacobi_on_the_pressure_system", True)):
        R = pk.Runner(P, 0, pk.OP_MATRIX_FREE, p_init=bench.INPUT["p_init"], dt=bench.INPUT["dt"], max_it=50000, prec=-1, jacobi_p=jp)
        R.initialize(); R.step(); R.ctx.synchronize()
        w0 = R.work(); t0 = time.perf_counter()
        for _ in range(3):
            tr, _w = R.step()
        R.ctx.synchronize(); dt = (time.perf_counter() - t0) / 3; w1 = R.work()
        rec[name] = {"ms_per_step": round(1e3 * dt, 3), **{k: (w1[k] - w0[k]) / 3 for k in ("cg_u", "cg_p", "cg_proj")}}
        R.close()
    out["cases"].append(rec); P.close()
print(json.dumps(out, indent=1))
