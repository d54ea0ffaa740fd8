"""Worker of the general-partition tests (last sentence: contiguous cell ranges in Morton order + indexed interface lists): one process per
rank, gloo.  Every rank builds the GLOBAL problem, takes its piece (Problem.partition) and runs one time step on it, with the oracle or with the HIP
library (all ranks share GPU 0; the interface sums and dots are host-staged through the callbacks, RCCL on a multi-GPU node).
Usage: python mr_general_worker.py rank world port mesh degree out.npz backend      mesh = gmsh | box:nx,ny[,nz]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import poroelasticity_dealii_amd as pk  # noqa: E402
import oracle_py  # noqa: E402
from common import REF, global_problem  # noqa: E402


def main():
    rank, world, port, mesh, deg, out, backend = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), sys.argv[6], sys.argv[7]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def allreduce(buf):
        t = torch.from_numpy(buf.copy()); dist.all_reduce(t); buf[:] = t.numpy()

    def sendrecv(send, recv, peer):
        ts, tr = torch.from_numpy(send.copy()), torch.empty(len(recv), dtype=torch.float64)
        for r in [dist.isend(ts, peer), dist.irecv(tr, peer)]:
            r.wait()
        recv[:] = tr.numpy()

    PG = global_problem(mesh, deg)
    P = PG.partition(rank, world)
    if backend == "oracle":
        O = oracle_py.Oracle(P, hoisted=True)
        O.comm_callbacks(allreduce, sendrecv)
        tr, _ = O.run(1, REF["p_init"], REF["dt"], max_it=20000, prec=oracle_py.PREC_JACOBI)
    else:
        R = pk.Runner(P, device=0, operator_mode=pk.OP_CSR if backend == "hip_csr" else pk.OP_MATRIX_FREE, p_init=REF["p_init"], dt=REF["dt"], max_it=20000,
                      prec=pk.PREC_CHEBYSHEV if backend == "hip_mf_cheb" else pk.PREC_JACOBI)
        O = R.ctx
        O.comm_callbacks(allreduce, sendrecv)
        R.initialize()
        t1, _ = R.step()
        tr = np.vstack([np.zeros((1, 8)), t1])
    xg = np.sin(0.11 * np.arange(PG.desc.n_dofs_u))
    res = {"trace": tr, "u": O.get(pk.VEC_U), "p": O.get(pk.VEC_P), "epsv": O.get(pk.VEC_EPSV), "rhs_u": O.get(pk.VEC_RHS_U), "Ax": O.apply(pk.MAT_A_U, xg[P.local_to_global_u]),
           "l2g_u": P.local_to_global_u, "l2g_p": P.local_to_global_p, "owned": np.array([P.desc.part.n_owned_u, P.desc.part.n_owned_p]),
           "neighbours": np.array([P.desc.part.neighbour_rank[k] for k in range(P.desc.part.n_neighbours)])}
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
