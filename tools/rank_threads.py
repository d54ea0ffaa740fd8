"""N ranks of a slab-partitioned run as N THREADS of one process, all on GPU 0, exchanging through the library's callback communicator
(poro_ctx_comm_init_callbacks) wired to in-process barriers / queues.  A one-GPU box admits only a handful of GPU processes, so this is how BASELINE
config 4's real partition - 72^3 cells cut into 8 slabs of 9 cell layers - is rehearsed end to end on the hardware there is: same partition
descriptors, same partitioned code path (single-reduction PCG, neighbour exchanges, all-to-alls of the distributed fast diagonalisation) as an 8-GPU
RCCL run, with host-staged transport instead of xGMI.

Usage: python tools/rank_threads.py [--ranks 8] [--cells 72] [--degree 2] [--prec block_fdm|chebyshev|jacobi] [--steps 1] [--json out.json]
Prints / writes: per-step iteration counts of every rank, the single-rank counts for comparison, exchange / all-reduce / all-to-all counts per CG
iteration, and the largest disagreement between the two copies of every shared plane and against the single-rank fields."""
import argparse
import json
import os
import queue
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import poroelasticity_dealii_amd as pk  # noqa: E402
import bench                           # noqa: E402  (material / boundary conditions of the bundled input.data, as the benchmark uses them)

REF = dict(p_init=bench.INPUT["p_init"], dt=bench.INPUT["dt"])


def box_problem(dim, n, deg, rank=0, n_ranks=1):
    return pk.Problem.box(dim, list(n), [10.0] * dim, deg, bench.material(), bench.BC_3D[:2 * dim], (), rank, n_ranks)

PREC = {"jacobi": pk.PREC_JACOBI, "chebyshev": pk.PREC_CHEBYSHEV, "block_fdm": pk.PREC_FDM}
FAMILIES = ("apply_u_matrix_free", "apply_u_chebyshev_fused", "precondition_u_fdm", "fdm_u_slab_z_stage", "halo_exchange", "allreduce", "alltoall")


class ThreadComm:
    """the two callbacks of the library for `n` rank threads: a sum over all ranks (in rank order, so every rank gets the same bits) and a pairwise exchange"""

    def __init__(self, n):
        self.n, self.barrier, self.slots = n, threading.Barrier(n), [None] * n
        self.mail = {(a, b): queue.Queue() for a in range(n) for b in range(n) if a != b}
        self.counts = [dict(allreduce=0, sendrecv=0) for _ in range(n)]

    def allreduce(self, rank, buf):
        self.counts[rank]["allreduce"] += 1
        self.slots[rank] = np.array(buf, copy=True)
        self.barrier.wait()
        total = self.slots[0].copy()
        for r in range(1, self.n):
            total += self.slots[r]
        self.barrier.wait()
        buf[:] = total

    def sendrecv(self, rank, send, recv, peer):
        self.counts[rank]["sendrecv"] += 1
        self.mail[(rank, peer)].put(np.array(send, copy=True))
        recv[:] = self.mail[(peer, rank)].get(timeout=600)


def run_ranks(world, dim, n, deg, prec, steps, reduction=True, rel_u=1e-8, max_it=50000):
    """returns (per-rank results, seconds of the timed steps); every rank: initialise, then `steps` consecutive time steps"""
    comm = ThreadComm(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            P = box_problem(dim, n, deg, rank=r, n_ranks=world)
            R = pk.Runner(P, device=0, operator_mode=pk.OP_MATRIX_FREE, p_init=REF["p_init"], dt=REF["dt"], abs_u=1e-12, rel_u=rel_u, max_it=max_it, prec=prec, reduction=reduction)
            G = R.ctx
            if world > 1:
                G.comm_callbacks(lambda b: comm.allreduce(r, b), lambda s, v, peer: comm.sendrecv(r, s, v, peer))
            R.initialize()
            G.timers_reset(); before = dict(comm.counts[r])
            comm.barrier.wait(); t0 = time.perf_counter()
            traces = [R.step()[0] for _ in range(steps)]
            G.synchronize(); comm.barrier.wait(); elapsed = time.perf_counter() - t0
            fam = {k: G.timer(k) for k in FAMILIES}
            out[r] = {"traces": traces, "u": G.get(pk.VEC_U), "p": G.get(pk.VEC_P), "n_u": P.desc.n_dofs_u, "n_p": P.desc.n_dofs_p, "plane_u": P.desc.part.plane_u, "plane_p": P.desc.part.plane_p,
                      "elapsed": elapsed, "families": fam, "callbacks": {k: comm.counts[r][k] - before[k] for k in before}}
            R.close(); P.close()
        except BaseException as exc:   # noqa: BLE001 - a dead rank must not leave the others waiting for ever
            errs.append((r, repr(exc)))
            comm.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise RuntimeError(f"rank threads failed: {errs}")
    return out


def stitch(parts, planes):
    """global vector from the slabs (the upper shared plane of rank r is the lower one of rank r + 1)"""
    pieces = [parts[0]] + [p[planes:] for p in parts[1:]]
    return np.concatenate(pieces)


def rehearse(world, dim, n, deg, prec_name, steps):
    prec = PREC[prec_name]
    multi = run_ranks(world, dim, n, deg, prec, steps)
    single = run_ranks(1, dim, n, deg, prec, steps)[0]
    cg = lambda res: [[int(r[6]) for r in t] for t in res["traces"]]   # noqa: E731
    rec = {"ranks": world, "cells": n, "degree": deg, "preconditioner": prec_name, "steps": steps,
           "layers_per_rank": [int(round((m["n_u"] // m["plane_u"] - 1) / deg)) for m in multi],
           "cg_iterations_u_by_rank": [cg(m) for m in multi], "cg_iterations_u_single_rank": cg(single),
           "fss_rows_equal_single_rank": all(np.array_equal(np.vstack(m["traces"])[:, :3], np.vstack(single["traces"])[:, :3]) for m in multi),
           "seconds_ranks_sharing_one_gpu": max(m["elapsed"] for m in multi), "seconds_single_rank": single["elapsed"]}
    its = sum(sum(x) for x in rec["cg_iterations_u_by_rank"][0])
    lf = multi[min(1, world - 1)]["families"]          # an interior rank (two neighbours)
    rec["per_cg_iteration_u_on_an_interior_rank"] = {"operator_applications": lf["apply_u_matrix_free"][1] / max(its, 1), "halo_exchanges_all_systems": lf["halo_exchange"][1] / max(its, 1),
                                                      "allreduces_all_systems": lf["allreduce"][1] / max(its, 1), "alltoalls_all_systems": lf["alltoall"][1] / max(its, 1),
                                                      "block_fdm_applications_in_slab_form": lf["fdm_u_slab_z_stage"][1] / max(its, 1),
                                                      "note": "exchange / reduction counts cover ALL solves of the steps (displacement, pressure Newton, 3 projections, residual norms) divided by the displacement CG iterations"}
    u = stitch([m["u"] for m in multi], multi[0]["plane_u"]); p = stitch([m["p"] for m in multi], multi[0]["plane_p"])
    rec["shared_plane_copies_max_abs_diff_u"] = max((float(np.abs(a["u"][-a["plane_u"]:] - b["u"][:a["plane_u"]]).max()) for a, b in zip(multi[:-1], multi[1:])), default=0.0)
    rec["rel_diff_u_vs_single_rank"] = float(np.linalg.norm(u - single["u"]) / np.linalg.norm(single["u"]))
    rec["rel_diff_p_vs_single_rank"] = float(np.abs(p - single["p"]).max() / np.abs(single["p"]).max())
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8); ap.add_argument("--cells", type=int, default=72); ap.add_argument("--dim", type=int, default=3); ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--prec", choices=sorted(PREC), default="block_fdm"); ap.add_argument("--steps", type=int, default=1); ap.add_argument("--json", default=None)
    a = ap.parse_args()
    rec = rehearse(a.ranks, a.dim, [a.cells] * a.dim, a.degree, a.prec, a.steps)
    print(json.dumps(rec, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rec, f, indent=1)
