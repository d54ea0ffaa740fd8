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

# Regression goldens for the box configurations (miniatures of BASELINE configs 2-5): 2 time steps each, reference loop and the
# coupled variant.  Same provenance: this repository's oracle, not the reference.
from common import REF, box_problem, host_material  # noqa: E402

boxes = {}
for name, (dim, n, deg) in {"2d_q2_16": (2, 16, 2), "3d_q1_4": (3, 4, 1), "3d_q2_4": (3, 4, 2)}.items():
    P = box_problem(dim, n, deg, mat=host_material())
    for variant, kw in (("reference", {}), ("coupled", {"coupled_fss": True, "incremental_strain": True})):
        O = oracle_py.Oracle(P)
        tr, _ = O.run(2, REF["p_init"], REF["dt"], max_it=2000, **kw)
        boxes[f"{name}_{variant}"] = {"dim": dim, "n": n, "degree": deg, "variant": variant, "rows": tr[:, :6].tolist(), "u_l2": float(np.linalg.norm(O.get(pk.VEC_U))),
                                      "p_l2": float(np.linalg.norm(O.get(pk.VEC_P))), "p_linf": float(np.abs(O.get(pk.VEC_P)).max()),
                                      "epsv_l2": float(np.linalg.norm(O.get(pk.VEC_EPSV))), "noconvergence": int(O.noconvergence_count())}
        O.close()
    P.close()
with open(os.path.join(HERE, "box_traces.json"), "w") as f:
    json.dump(boxes, f, indent=1)
