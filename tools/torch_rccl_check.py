import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
os.environ["PORO_FORCE_PARTITIONED_PATH"] = "1"
import torch
torch.cuda.set_device(0); torch.zeros(4, device="cuda").sum().item()
import numpy as np
import poroelasticity_dealii_amd as pk
from common import box_problem
P = box_problem(3, 6, 2)
G = pk.Context(P, 0, pk.OP_MATRIX_FREE)
G.comm_rccl(pk.rccl_unique_id())
G.fill(pk.VEC_P, 10e6); G.disp_assemble_system(True)
rc, info = G.disp_solve(max_iter=5000)
print("torch", torch.__version__, "rc", rc, "its", info.iterations, "res", info.final_residual)
import ctypes
print([l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l or "amdhip" in l][:6])
