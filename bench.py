#!/usr/bin/env python3
"""Benchmark of the fixed-stress Biot hot path on MI355X.

Metric (BASELINE.json): DoF-updates/sec in assemble + SpMV per fixed-stress iteration.
One "step" = one time step of the reference's loop (PoroelasticityFSS.h:328-407) = one fixed-stress iteration
(quirk Q1): pressure Newton loop, displacement RHS assembly + PCG solve with the matrix-free A_u, strain
projection (RHS assembly + dim CG solves), residual check.  A DoF-update is one entry of a vector
produced by an operator application y = A x or by an assembly pass; `value` = DoF-updates of the K timed steps
(all ranks) / wall time of those steps, INCLUDING the Krylov vector work, reductions and host control.
Only useful work is counted: (iterations + 1) operator applications per CG solve (launches the batched PCG loop
enqueues behind the finishing iteration are no-ops), the Jacobian only when dt changed.

Workload: 3D Q2/Q1 uniform box of 72^3 cells (N_u = 9 145 875: BASELINE config "3D Q2/Q1 ~10M DoF").  N > 1 cuts THAT mesh into N z-slabs
(strong scaling: 9 cell layers per GPU at N = 8); the weak-scaled variant (72 layers per GPU, 72 x 72 x 72N box, same h) is measured
as well and reported under "weak_scaling_line".  Every timed step must do a real displacement solve: by default each timed step is time
step 1 from the initial equilibrium (the device state is rolled back before every step, so all steps do the same work whatever --steps /
--warmup are; --transient times consecutive steps instead), the CG stops on the reduction of the step's own initial residual
(PORO_STOP_REDUCTION), the per-step iteration counts are printed and a step with 0 iterations fails the run.
Input is synthetic in the sense of SURVEY 8d: the bundled input.data material / BC values on a generated mesh.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import poroelasticity_dealii_amd as pk  # noqa: E402

INPUT = dict(E=1.4e10, nu=0.3, alpha=0.9, poro=0.3, f_comp=5.8e-10, perm_mD=10.0, visc=1e-3, r_well=1.0, flow_rate=1e-5, p_init=10e6, dt=60.0)   # input.data:13-40
BC_3D = [(0, 0, 0.0), (1, 0, -1e-5), (2, 1, 0.0), (3, 1, -1e-5), (4, 2, 0.0), (5, 2, -1e-5)]   # input.data:14-16 + z faces
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def material():
    # InputDataPoroel::compute_derived_parameters (InputDataPoroel.h:213-222) on the input.data values
    E, nu, alpha, poro, cf = INPUT["E"], INPUT["nu"], INPUT["alpha"], INPUT["poro"], INPUT["f_comp"]
    lam = E * nu / ((1. + nu) * (1. - 2. * nu)); G = 0.5 * E / (1 + nu); K = lam + 2. / 3. * G
    Ks = K / (1. - alpha); N = Ks / (alpha - poro); M = (N / cf) / (N * poro + 1. / cf)
    return pk.Material(lam, G, alpha, K, M, INPUT["perm_mD"] * 9.869233e-16 / INPUT["visc"], INPUT["r_well"], INPUT["flow_rate"])


def dof_updates(w, n_u, n_p, dim):
    """operator applications and assembly passes, each counted by the length of the vector it produces
    (the pressure residual holds two SpMVs, mass and Laplace)."""
    return (w["apply_u"] * n_u + w["asm_rhs_u"] * n_u + w["apply_p"] * n_p + w["residual_p"] * 2 * n_p + w["jacobian_p"] * n_p + w["proj_rhs"] * dim * n_p)


def bytes_per_apply(dim, degree, n_u, n_cells, operator):
    """algorithmic HBM bytes of one A_u application: matrix-free 16 N + 4 dpc n_cells; CSR 12 nnz + 24 N"""
    dpc = dim * (degree + 1) ** dim
    if operator == "matrix_free":
        return 16.0 * n_u + 4.0 * dpc * n_cells
    raise ValueError


def cpu_baseline(dim, degree, n, rel_tol, reduction):
    """the oracle (CPU restatement of the reference algorithm, 1 thread) on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    P = pk.Problem.box(dim, [n] * dim, [10.0] * dim, degree, material(), BC_3D[:2 * dim])
    O = oracle_py.Oracle(P)                      # naive i x q x j assembly (quirk Q6) + SSOR-CG = the reference's algorithm
    kw = dict(abs_u=1e-12, rel_u=rel_tol, max_it=100000, reduction=reduction)
    ta = time.perf_counter(); O.run(0, INPUT["p_init"], INPUT["dt"], **kw); tb = time.perf_counter()      # initialisation only
    w_init = O.work_counts(reset=True)
    tc = time.perf_counter(); O.run(1, INPUT["p_init"], INPUT["dt"], **kw); td = time.perf_counter()      # initialisation + 1 time step
    w_all = O.work_counts(reset=True)
    w = {k: w_all[k] - w_init[k] for k in w_all}
    dt_step = max((td - tc) - (tb - ta), 1e-9)
    upd = dof_updates(w, P.desc.n_dofs_u, P.desc.n_dofs_p, dim)
    out = {"value": upd / dt_step, "unit": "DoF-updates/s", "cores": 1, "kind": "port",
           "sample": f"oracle = CPU restatement of the reference (naive assembly + SSOR-CG, g++ -O2, 1 thread); {dim}D Q{degree}/Q1 {n}^{dim} cells "
                     f"(N_u={P.desc.n_dofs_u}), one time step = {dt_step:.2f} s, {w['apply_u']} A_u applications",
           "host_cpu": _cpu_name(), "host_cores_available": os.cpu_count()}
    O.close(); P.close()
    # context figure: what the assembled-CSR data structure gives on all host cores (not the reference's serial algorithm)
    try:
        threads = max(1, min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0))))   # the 1-GPU box's CPU share is 16 cores
        n2 = 12 if dim == 3 else 192
        P2 = pk.Problem.box(dim, [n2] * dim, [10.0] * dim, degree, material(), BC_3D[:2 * dim])
        O2 = oracle_py.Oracle(P2, hoisted=True)
        O2.fill(pk.VEC_P, INPUT["p_init"]); O2.disp_assemble_system(True)
        t_spmv, t_cg = O2.bench_spmv_threads(threads, reps=20)
        out["all_cores_csr"] = {"threads": threads, "sample": f"{dim}D Q{degree}/Q1 {n2}^{dim} cells, N_u={P2.desc.n_dofs_u}, assembled CSR A_u",
                                "spmv_DoF_updates_per_s": P2.desc.n_dofs_u / t_spmv, "jacobi_cg_iterations_per_s": 1.0 / t_cg,
                                "jacobi_cg_DoF_updates_per_s": P2.desc.n_dofs_u / t_cg}
        O2.close(); P2.close()
    except Exception as exc:                      # the headline cpu_baseline stays valid without the context figure
        out["all_cores_csr"] = {"error": str(exc)}
    return out


def _cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_stamp():
    """sha256 of the operator-kernel source: PMC traffic figures under profiles/ are only quoted for the build they were measured on"""
    import hashlib
    with open(os.path.join(ROOT, "poroelasticity_dealii_amd", "csrc", "kernels_kron.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def committed_traffic(dim, deg, n, fused_cheb=False):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
    tools/bench_ops.py on this workload, gfx950 x2 read correction; tools/pmc_summary.py).  The file carries the stamp of the kernel source it was
    measured on; a different source (or no file) gives null rather than a stale number."""
    try:
        key = {(2, 72): "poro::k_kron3_q2", (1, 99): "poro::k_kron3_q1"}[(deg, n)] + ("_cheb" if fused_cheb else "") if dim == 3 else None
        with open(os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")) as f:
            rec = json.load(f)
        if rec.get("kernel_source_sha16") != kernel_source_stamp():
            return None
        return rec[key]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def run_case(args, scaling, steps, warmup, rank, world, local_rank, torch, dist, rccl_ids, prec=None, label="jacobi"):
    """`warmup` untimed + `steps` timed time steps of one configuration; returns the measurements of this rank (elapsed = max over ranks)"""
    dim, deg = args.dim, args.degree
    n, size = [args.n] * dim, [10.0] * dim
    if world > 1 and scaling == "weak":
        n[dim - 1] = args.n * world; size[dim - 1] = 10.0 * world
    P = pk.Problem.box(dim, n, size, deg, material(), BC_3D[:2 * dim], (), rank, world)
    R = pk.Runner(P, device=local_rank, operator_mode=pk.OP_MATRIX_FREE, p_init=INPUT["p_init"], dt=INPUT["dt"], abs_u=1e-12, rel_u=args.rel_tol, max_it=args.max_iter,
                  prec=pk.PREC_JACOBI if prec is None else prec, reduction=args.stop == "reduction", cheb_degree=args.cheb_degree, cheb_ratio=args.cheb_ratio)
    G = R.ctx
    if world > 1 and args.share_gpu:
        import numpy as np

        def _allreduce(buf):
            t = torch.from_numpy(buf.copy()); dist.all_reduce(t); buf[:] = t.numpy()

        def _sendrecv(send, recv, peer):
            ts, tr = torch.from_numpy(np.array(send, copy=True)), torch.empty(len(recv), dtype=torch.float64)
            for r in [dist.isend(ts, peer), dist.irecv(tr, peer)]:
                r.wait()
            recv[:] = tr.numpy()
        G.comm_callbacks(_allreduce, _sendrecv)
    elif world > 1:
        G.comm_rccl(rccl_ids[label + scaling])

    part = P.desc.part                        # shared interface planes are counted once, by their upper owner
    own_u = P.desc.n_dofs_u - (part.plane_u if part.has_upper else 0)
    own_p = P.desc.n_dofs_p - (part.plane_p if part.has_upper else 0)
    n_cells = P.desc.n_cells
    if world > 1:
        t = torch.tensor([own_u, own_p, n_cells], dtype=torch.int64); dist.all_reduce(t)
        n_u_glob, n_p_glob = int(t[0]), int(t[1])
    else:
        n_u_glob, n_p_glob = own_u, own_p

    def sync():
        if torch is not None:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    R.initialize()                            # PoroelasticityFSS.h:308-317 (initial equilibrium; not part of a step)
    repeat = not args.transient
    if repeat:
        R.save_state()                        # every step below is "time step 1 from the initial equilibrium": a device-side rollback (~30 us of copies, inside the timed region)
    for _ in range(warmup):
        if repeat:
            R.restore_state()
        R.step()
    before = R.work()
    G.timers_reset()                          # HIP events around every kernel family on the launch stream
    # every launch with events costs the step ~7 % (2.5 us per launch, 650 launches per step): sample every `event_stride`-th launch of each kernel family
    # (the library scales the sampled time to all launches of the family)
    G.timers_enable(0 if getattr(args, "no_kernel_events", False) else args.event_stride)
    sync()
    t0 = time.perf_counter()
    traces, step_seconds = [], []
    for _ in range(steps):
        ts = time.perf_counter()
        if repeat:
            R.restore_state()
        traces.append(R.step()[0]); step_seconds.append(time.perf_counter() - ts)   # step() returns after a stream sync
    sync()
    elapsed = time.perf_counter() - t0
    after = R.work()
    G.timers_enable(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); elapsed = float(tt[0])
    work = {k: after[k] - before[k] for k in after}
    families = ("apply_u_matrix_free", "apply_u_chebyshev_fused", "apply_u_dirichlet_rows", "assemble_u_rhs", "projection_rhs", "pressure_residual", "pressure_jacobian", "apply_p_csr", "apply_p_stencil",
                "precondition_p_fdm", "precondition_u_fdm", "precondition_u_chebyshev", "halo_exchange", "allreduce", "alltoall")
    out = {"n": n, "n_u_glob": n_u_glob, "n_p_glob": n_p_glob, "n_u_local": P.desc.n_dofs_u, "n_cells_local": n_cells, "elapsed": elapsed, "work": work,
           "updates": dof_updates(work, n_u_glob, n_p_glob, dim), "traces": traces, "step_seconds": step_seconds,
           "kernel_time": {k: G.timer(k) for k in families}}
    R.close(); P.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--n", "--cells", dest="n", type=int, default=72, help="cells per direction (use --cells under torch.distributed.run, whose own parser treats --n as an abbreviation)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong = the fixed BASELINE mesh (72^3: 9 cell layers per GPU at N = 8, SURVEY 8e) cut into N z-slabs; weak = 72 layers per GPU (reported as an extra key by default)")
    ap.add_argument("--no-weak-line", action="store_true", help="N > 1: skip the extra weak-scaled measurement")
    ap.add_argument("--stop", choices=["reduction", "rhs"], default="reduction",
                    help="displacement CG stops at rel_tol x (reduction: the residual of the step's warm start | rhs: ||b||); with `rhs` a slow transient lets later steps accept the warm start")
    ap.add_argument("--rel-tol", type=float, default=1e-8, help="displacement CG: recursive residual <= max(1e-12, rel_tol * reference norm of --stop)")
    ap.add_argument("--transient", action="store_true",
                    help="time consecutive steps of the transient (BASELINE config 5) instead of repeating time step 1: the steps then differ (the input.data well rate is tiny, "
                         "the transient dies within ~10 steps) and ms_per_step depends on the window")
    ap.add_argument("--prec", choices=["chebyshev", "jacobi", "block_fdm"], default="chebyshev",
                    help="preconditioner of the displacement CG in the headline run: chebyshev = Chebyshev polynomial around Jacobi, on 3D boxes fused into the operator kernel "
                         "(the other one and the block fast diagonalisation are measured as well and reported under time_to_solution)")
    ap.add_argument("--cheb-degree", type=int, default=6)
    ap.add_argument("--cheb-ratio", type=int, default=0, help="lambda_max / lambda_min of the Chebyshev interval (0: the library default, a few times the mesh-dependent lambda_min)")
    ap.add_argument("--max-iter", type=int, default=50000)
    ap.add_argument("--cpu-n", type=int, default=9, help="cells per direction of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra time-to-solution measurement with the block fast-diagonalisation preconditioner")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: all ranks use device 0 and exchange through gloo (host-staged callbacks) instead of RCCL")
    ap.add_argument("--event-stride", type=int, default=8, help="attach HIP events to every n-th launch of each kernel family inside the timed region (1: every launch)")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: do not attach HIP events to the kernel dispatches inside the timed region (no roofline figures)")
    ap.add_argument("--trace-out", default=None, help="write the per-step record (SURVEY 8d config 5: FSS / pressure / Krylov iteration counts, wall-clock) to this JSON file")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch N>1 as: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch = dist = None
    try:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group(backend="gloo")   # control plane only (barrier, max, id broadcast); the data plane is RCCL inside the library
    except ImportError:
        torch = None

    dim, deg = args.dim, args.degree
    PREC = {"jacobi": pk.PREC_JACOBI, "chebyshev": pk.PREC_CHEBYSHEV, "block_fdm": pk.PREC_FDM}
    head = args.prec
    others = [] if args.no_variants else [k for k in ("jacobi", "chebyshev", "block_fdm") if k != head]
    weak_line = world > 1 and args.scaling == "strong" and not args.no_weak_line
    rccl_ids = {}
    if world > 1 and not args.share_gpu:      # one communicator per context that will be created, ids from rank 0
        labels = [head + args.scaling] + [k + args.scaling for k in others] + ([head + "weak"] if weak_line else [])
        ids = [{lb: pk.rccl_unique_id() for lb in labels} if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        rccl_ids = ids[0]

    M = run_case(args, args.scaling, args.steps, args.warmup, rank, world, local_rank, torch, dist, rccl_ids, prec=PREC[head], label=head)
    work, elapsed, updates, traces, step_seconds = M["work"], M["elapsed"], M["updates"], M["traces"], M["step_seconds"]
    n, n_u_glob, n_p_glob = M["n"], M["n_u_glob"], M["n_p_glob"]
    kernel_time = M["kernel_time"]

    # roofline of the dominant kernel (per launch, per GPU).  Chebyshev-CG on a 3D box: the structured operator with the polynomial recurrence fused into
    # its stores (k_kron3_*_cheb: reads z_j and g, writes z_{j+1} = 24 B / DoF, + the 8d element-index bytes the structured kernels do not read);
    # otherwise the plain operator y = A_u x (16 B / DoF + index bytes).  Launches enqueued behind the finishing iteration of a solve return at once
    # (no-ops, ~2 us): their time stays in the numerator, the average is taken over the useful applications only.
    t_plain, n_plain_launched = kernel_time["apply_u_matrix_free"]
    t_cheb, n_cheb_launched = kernel_time["apply_u_chebyshev_fused"]
    alg_plain = bytes_per_apply(dim, deg, M["n_u_local"], M["n_cells_local"], "matrix_free")
    cg_total = int(work["cg_u"]); solves = int(work["asm_rhs_u"])
    fused_cheb = n_cheb_launched > 0
    if fused_cheb:
        n_cheb_useful = args.cheb_degree * (cg_total + solves) if args.cheb_degree % 2 == 0 else (args.cheb_degree + 1) * (cg_total + solves)
        n_plain_useful = cg_total + solves
        n_apply, t_apply, n_launched = n_cheb_useful, t_cheb, n_cheb_launched
        alg_bytes = alg_plain + 8.0 * M["n_u_local"]
        kernel_label = ("k_kron3_q%d_cheb" % deg if dim == 3 else "k_kron2<%d, true>" % deg) + " (matrix-free A_u z_j with the Chebyshev root-form update z_{j+1} = z_j + omega_j D^-1 (g - A z_j) fused into the stores)"
    else:
        n_apply = int(work["apply_u"]) or n_plain_launched
        n_plain_useful = n_apply; t_apply, n_launched = t_plain, n_plain_launched
        alg_bytes = alg_plain
        kernel_label = ("k_kron3_q%d" % deg if dim == 3 else "k_kron2<%d>" % deg) + " (matrix-free y = A_u x, sum-factorised)"
    avg_apply = t_apply / max(n_apply, 1)
    achieved = alg_bytes / avg_apply / 1e9 if (n_apply and avg_apply > 0) else 0.0
    avg_plain = t_plain / max(n_plain_useful, 1)
    t_fix, n_fix = kernel_time["apply_u_dirichlet_rows"]
    traffic = committed_traffic(dim, deg, args.n, fused_cheb) if world == 1 else None

    # the same steps with the other preconditioners of the displacement CG (time to solution; the headline is the run above)
    def summary(V, prec_name):
        d = {"ms_per_step": 1e3 * V["elapsed"] / args.steps, "cg_iterations_u": [[int(r[6]) for r in t] for t in V["traces"]],
             "operator_applications_per_step": V["work"]["apply_u"] / args.steps, "DoF_updates_per_s": V["updates"] / V["elapsed"]}
        if prec_name == "block_fdm":
            d.update({"seconds_precondition_u": V["kernel_time"]["precondition_u_fdm"][0], "applications_precondition_u": V["kernel_time"]["precondition_u_fdm"][1],
                      "seconds_alltoall": V["kernel_time"]["alltoall"][0]})
        return d
    tts = {head: summary(M, head)}
    for k in others:
        try:
            tts[k] = summary(run_case(args, args.scaling, args.steps, args.warmup, rank, world, local_rank, torch, dist, rccl_ids, prec=PREC[k], label=k), k)
        except RuntimeError as exc:           # a context that cannot use it (e.g. Dirichlet data not face-separable) says so; the headline stays valid
            tts[k] = {"error": str(exc)}
    weak = None
    if weak_line:
        W = run_case(args, "weak", min(args.steps, 3), 1, rank, world, local_rank, torch, dist, rccl_ids, prec=PREC[head], label=head)
        weak = {"value": W["updates"] / W["elapsed"], "ms_per_step": 1e3 * W["elapsed"] / min(args.steps, 3), "cells": "x".join(map(str, W["n"])), "N_u": W["n_u_glob"],
                "cg_iterations_u": [[int(r[6]) for r in t] for t in W["traces"]]}

    cg_u = [[int(r[6]) for r in t] for t in traces]
    dead = [i for i, its in enumerate(cg_u) if not its or min(its) == 0]
    if rank == 0:
        stop_txt = (f"recursive residual <= max(1e-12, {args.rel_tol:g}*||g_0||), g_0 = residual of the step's warm start (ReductionControl), cap {args.max_iter}" if args.stop == "reduction"
                    else f"recursive residual <= max(1e-12, {args.rel_tol:g}*||b||), cap {args.max_iter}")
        out = {
            "metric": "DoF-updates/sec in assemble+SpMV per fixed-stress iter", "value": updates / elapsed, "unit": "DoF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{dim}D Q{deg}/Q1 uniform box, {'x'.join(map(str, n))} cells, N_u={n_u_glob}, N_p={n_p_glob}; {'consecutive time steps' if args.transient else 'time step 1 repeated (device-side rollback before each step)'}; one time step = one fixed-stress iteration "
                                   f"(pressure loop + matrix-free PCG displacement solve, preconditioner: {head}" + (f" degree {args.cheb_degree}" if head == "chebyshev" else "") + " + strain projection); input.data material/BCs, z-face BCs per SURVEY Q9",
                       "parallelism": f"z-slab x{world}" if world > 1 else "single GPU", "operator": "matrix_free",
                       "stopping_rule_u": stop_txt},
            "cg_iterations_u": cg_u,
            "roofline": {"bound": "hbm", "kernel": kernel_label, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": 1e6 * avg_apply, "launches_timed": -(-n_launched // max(args.event_stride, 1)), "launches_enqueued": n_launched, "useful_launches": n_apply, "events_on_every_nth_launch": args.event_stride,
                         "frac_without_index_bytes": (alg_bytes - 4.0 * dim * (deg + 1) ** dim * M["n_cells_local"]) / avg_apply / 1e9 / HBM_PEAK_GBS if (n_apply and avg_apply > 0) else 0.0,
                         "plain_operator": {"kernel": ("k_kron3_q%d" % deg if dim == 3 else "k_kron2<%d>" % deg), "avg_launch_us": 1e6 * avg_plain, "algorithmic_bytes_per_launch": alg_plain,
                                            "achieved": alg_plain / avg_plain / 1e9 if n_plain_useful and t_plain else 0.0, "frac": alg_plain / avg_plain / 1e9 / HBM_PEAK_GBS if n_plain_useful and t_plain else 0.0},
                         "note": ("kernel alone (HIP events attached to the dispatch, as rocprofv3 reports it; every events_on_every_nth_launch-th launch of the timed steps carries events, since events on every launch slow the step by 7 percent); inside PCG the Dirichlet dofs are inert (zero residual / direction), so no row fix-up runs "
                                  "(k_kron_fix_constrained only serves poro_apply_operator: %d launches in the timed region); "
                                  "algorithmic bytes follow SURVEY 8d (16 N + 4 dpc n_cells per operator application, + 8 N for the extra stream g of the fused Chebyshev update) although the "
                                  "structured kernels read no index arrays (frac_without_index_bytes leaves them out); "
                                  "traffic = PMC bytes per launch from profiles/r02_pmc_traffic.json, null unless that file was measured on this kernel source") % n_fix},
            "work_per_step": {k: work[k] / args.steps for k in work},
            "kernel_only": {"apply_u_DoF_updates_per_s": (M["n_u_local"] / (avg_plain + t_fix / max(n_fix, 1))) if n_plain_useful and t_plain else 0.0,
                            "seconds_by_family": {k: v[0] for k, v in kernel_time.items()}, "launches_by_family": {k: v[1] for k, v in kernel_time.items()}},
            "fss_iterations_per_step": [int(len(t)) for t in traces],
        }
        out["time_to_solution"] = tts
        if weak is not None:
            out["weak_scaling_line"] = weak
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dim, deg, args.cpu_n, args.rel_tol, args.stop == "reduction")
        print(json.dumps(out), flush=True)
        if args.trace_out:
            # trace rows: [step, fss iteration, pressure iterations, inner pressure error, |p|_inf, error after displacement, u CG its, p CG its]
            rec = [{"step": int(t[-1][0]), "fss_iterations": int(len(t)), "pressure_iterations": [int(r[2]) for r in t], "cg_iterations_u": [int(r[6]) for r in t],
                    "cg_iterations_p": [int(r[7]) for r in t], "p_inf": float(t[-1][4]), "fss_error": float(t[-1][5]), "seconds": sec} for t, sec in zip(traces, step_seconds)]
            with open(args.trace_out, "w") as f:
                json.dump({"workload": out["config"]["workload"], "n_gpus": world, "steps": rec, "seconds_total": elapsed}, f, indent=1)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    if dead:
        # a timed step whose displacement solve accepted its warm start did no Krylov work: the line above is not a measurement of the hot path
        print(f"bench.py: timed steps {dead} ran 0 displacement CG iterations", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
