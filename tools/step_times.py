"""Wall time of consecutive repetitions of time step 1 (72^3 Q2/Q1) per preconditioner: shows one-off costs inside the first steps."""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
import poroelasticity_dealii_amd as pk
from bench import material, BC_3D
P = pk.Problem.box(3, [72]*3, [10.0]*3, 2, material(), BC_3D)
for name, prec in (("fdm", pk.PREC_FDM), ("cheb", pk.PREC_CHEBYSHEV)):
    R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=10e6, dt=60.0, abs_u=1e-12, rel_u=1e-8, max_it=50000, prec=prec, reduction=True)
    R.initialize(); R.save_state()
    ts = []
    for k in range(7):
        R.restore_state(); t0 = time.perf_counter(); tr, w = R.step(); ts.append((round(1e3 * (time.perf_counter() - t0), 2), int(w['cg_u']), int(w['apply_u']), round(w['usec_solve_u'] / 1e3, 2)))
        if k == 0 and len(sys.argv) > 1: R.ctx.timers_reset(); R.ctx.timers_enable(int(sys.argv[1]))     # argument: event stride (as bench.py switches it on after the warm-up)
    print(name, '(ms, cg_u, apply_u, ms in the displacement solve):', ts)
    R.close()
