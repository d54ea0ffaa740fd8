"""Probe: can two RCCL ranks share GPU 0 on this box (would let the partitioned path run over RCCL without a second GPU)?  Each rank creates a slab context on device 0 and
initialises the communicator; prints the outcome.  Usage (rank 0 and rank 1 as two processes): python tools/rccl_two_ranks_one_gpu.py <rank> <idfile>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import poroelasticity_dealii_amd as pk
from bench import material, BC_3D
rank, idfile = int(sys.argv[1]), sys.argv[2]
if rank == 0:
    uid = pk.rccl_unique_id(); open(idfile + ".tmp", "wb").write(bytes(uid)); os.replace(idfile + ".tmp", idfile)
else:
    for _ in range(600):
        if os.path.exists(idfile): break
        time.sleep(0.1)
    uid = open(idfile, "rb").read()
P = pk.Problem.box(3, [8, 8, 8], [10.0] * 3, 2, material(), BC_3D, (), rank, 2)
R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=10e6, dt=60.0, abs_u=1e-12, rel_u=1e-8, max_it=5000, prec=pk.PREC_CHEBYSHEV, reduction=True)
try:
    R.ctx.comm_rccl(uid)
    R.initialize(); tr, w = R.step()
    print("rank", rank, "RCCL with two ranks on one GPU: ok, CG iterations", int(tr[0][6]), flush=True)
except Exception as e:
    print("rank", rank, "RCCL with two ranks on one GPU: refused:", str(e)[:300], flush=True)
