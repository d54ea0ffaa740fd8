"""The partitioned code path on ONE rank with RCCL as the communicator (PORO_FORCE_PARTITIONED_PATH=1): single-reduction PCG, elementwise Chebyshev recurrence after the operator,
explicit sums + ncclAllReduce on the compute stream - what every rank of a multi-GPU run executes, minus the neighbour exchange.  Prints the step time next to the single-rank path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 72
force = len(sys.argv) > 2 and sys.argv[2] == "force"
if force:
    os.environ["PORO_FORCE_PARTITIONED_PATH"] = "1"
import poroelasticity_dealii_amd as pk
from bench import material, BC_3D
P = pk.Problem.box(3, [n] * 3, [10.0] * 3, 2, material(), BC_3D)
only = sys.argv[3] if len(sys.argv) > 3 else None
for name, prec in (("chebyshev", pk.PREC_CHEBYSHEV), ("block_fdm", pk.PREC_FDM), ("jacobi", pk.PREC_JACOBI)):
    if only and name != only:
        continue
    R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=10e6, dt=60.0, abs_u=1e-12, rel_u=1e-8, max_it=50000, prec=prec, reduction=True)
    if force:
        R.ctx.comm_rccl(pk.rccl_unique_id())
    R.initialize(); R.save_state()
    ts = []
    for k in range(5):
        R.restore_state(); R.ctx.synchronize(); t0 = time.perf_counter(); tr, w = R.step(); R.ctx.synchronize(); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
    print(("partitioned path (1 RCCL rank)" if force else "single-rank path"), name, "ms per step:", ts, "CG iterations", int(tr[0][6]), flush=True)
    R.close()
