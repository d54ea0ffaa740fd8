"""Generates tests/golden/config1_trace.json from the CPU oracle (NOT from the reference: deal.II is unavailable, see
DESIGN.md).  BASELINE config 1: the reference's domain.msh + input.data, one time step, Q1/Q1 and Q2/Q1."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import poroelasticity_dealii_amd as pk  # noqa: E402
import oracle_py  # noqa: E402
from common import BC_2D, DOMAIN_MSH, INPUT_DATA  # noqa: E402

inp = pk.read_input(INPUT_DATA)
out = {}
for deg in (1, 2):
    P = pk.Problem.gmsh(DOMAIN_MSH, deg, inp.material, BC_2D)
    O = oracle_py.Oracle(P)
    tr, _ = O.run(1, inp.p_init, inp.time_step, inp.fss_tol, inp.pressure_tol, inp.max_fss_iterations, inp.max_pressure_iterations)
    out[f"Q{deg}"] = {"n_dofs_u": int(P.desc.n_dofs_u), "n_dofs_p": int(P.desc.n_dofs_p), "rows": tr.tolist(), "fss_iterations": int(tr[-1, 1]),
                      "pressure_iterations": int(tr[1, 2]), "p_linf": float(tr[1, 4]), "u_l2": float(np.linalg.norm(O.get(pk.VEC_U))),
                      "p_l2": float(np.linalg.norm(O.get(pk.VEC_P))), "epsv_l2": float(np.linalg.norm(O.get(pk.VEC_EPSV))),
                      "u_cg_iterations_ssor": [int(r[6]) for r in tr], "noconvergence": int(O.noconvergence_count())}
with open(os.path.join(HERE, "config1_trace.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
