"""Create / run / destroy a dozen contexts (both operator modes) and watch the free device memory: no growth = no leak."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, numpy as np
import poroelasticity_dealii_amd as pk
from common import box_problem, host_material, REF
free0 = None
for it in range(12):
    P = box_problem(3, 16, 2, mat=host_material())
    R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE if it % 2 == 0 else pk.OP_CSR, p_init=REF["p_init"], dt=REF["dt"], max_it=20000)
    R.initialize(); R.step(); R.postprocess()
    R.close(); P.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 1: free0 = free
    print(it, "free MB", free // 2**20, flush=True)
assert free0 - free < 64 * 2**20, (free0, free)
print("no leak")
