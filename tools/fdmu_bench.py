"""times one application of the block fast-diagonalisation preconditioner (and the operator) at BASELINE config sizes"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests")]
import numpy as np
import poroelasticity_dealii_amd as pk
from common import box_problem

for dim, n, deg in [(3, 72, 2), (3, 99, 1)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:4])]:
    P = box_problem(dim, n, deg); G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    G.fill(pk.VEC_P, 0.0); G.disp_assemble_system(True)
    g = np.sin(0.37 * np.arange(G.n_u))
    z, t = G.apply_preconditioner_u(pk.PREC_FDM, g, reps=int(os.environ.get("REPS", "20")))
    top = G.bench_operator(pk.OP_MATRIX_FREE, 20)
    flop = 2.0 * dim * 2 * sum((deg * n + 1) for _ in range(dim)) * (deg * n + 1) ** dim
    print(f"{dim}D Q{deg} {n}^{dim}: block-FDM apply {t * 1e6:.1f} us ({flop / t / 1e12:.1f} TFLOP/s useful), operator {top * 1e6:.1f} us", flush=True)
    G.close(); P.close()
