"""one fixed-stress time step on boxes beyond BASELINE's size (half lines of more than 80 entries: the block fast diagonalisation falls back to the nodal transform kernels):
python tools/big_box_step.py [cells ...] > out.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT]
import poroelasticity_dealii_amd as pk
import bench

out = []
for n in [int(a) for a in sys.argv[1:]] or [96, 128]:
    P = pk.Problem.box(3, [n] * 3, [10.0] * 3, 2, bench.material(), bench.BC_3D)
    rec = {"cells": n, "n_dofs_u": int(P.desc.n_dofs_u), "n_dofs_p": int(P.desc.n_dofs_p)}
    for name, prec in (("block_fdm", pk.PREC_FDM), ("chebyshev", pk.PREC_CHEBYSHEV)):
        R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=bench.INPUT["p_init"], dt=bench.INPUT["dt"], abs_u=1e-12, rel_u=1e-8, max_it=50000, prec=prec, reduction=True)
        R.initialize(); R.save_state(); ts = []
        for k in range(3):
            R.ctx.synchronize(); t0 = time.perf_counter(); tr, w = R.step(restore=True); R.ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        rec[name] = {"cg_iterations_u": int(tr[0][6]), "ms_per_step": round(min(ts), 3)}
        R.close()
    out.append(rec); P.close()
    print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
