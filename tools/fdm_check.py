"""Quick check of the fast-diagonalisation preconditioner against Jacobi-PCG on the pressure Jacobian / projection mass matrix."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import poroelasticity_dealii_amd as pk
from common import box_problem, host_material
for dim, n in ((3, 8), (2, 40), (3, 72), (3, 99), (2, 336)):
    P = box_problem(dim, n, 1, mat=host_material())
    G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
    npn = G.n_p
    p = 10e6 * (1 + 0.05 * np.sin(0.37 * np.arange(npn)))
    G.set(pk.VEC_P, p); G.set(pk.VEC_P_OLD, p * 0.99); G.set(pk.VEC_EPSV, -2e-6 * (1 + 0.3 * np.sin(0.5 * np.arange(npn)))); G.set(pk.VEC_EPSV0, -2e-6 * np.ones(npn))
    G.pres_assemble_residual(60.0); G.pres_assemble_jacobian(60.0)
    res = {}
    for prec in (pk.PREC_FDM, pk.PREC_JACOBI):
        for rep in range(2):
            G.fill(pk.VEC_DP, 0.0)
            t0 = time.perf_counter(); rc, info = G.pres_solve(rel_tol=1e-10, max_iter=3000, prec=prec); t1 = time.perf_counter()
        res[prec] = (G.get(pk.VEC_DP), info.iterations, t1 - t0, rc, info.final_residual)
    a, b = res[pk.PREC_FDM], res[pk.PREC_JACOBI]
    print(dim, n, "fdm it", a[1], "wall %.2f ms" % (1e3 * a[2]), "| jacobi it", b[1], "wall %.2f ms" % (1e3 * b[2]),
          "| rel diff %.2e" % (np.linalg.norm(a[0] - b[0]) / np.linalg.norm(b[0])), a[3], b[3], a[4], b[4], flush=True)
    G.close(); P.close()
