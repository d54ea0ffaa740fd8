#!/usr/bin/env python3
"""Benchmark of the fixed-stress Biot hot path on MI355X.

Metric (BASELINE.json): DoF-updates/sec in assemble + SpMV per fixed-stress iteration.
One "step" = one time step of the reference's loop (PoroelasticityFSS.h:328-407) = one fixed-stress iteration
(quirk Q1): pressure Newton loop, displacement RHS assembly + PCG solve with the matrix-free A_u, strain
projection (RHS assembly + dim CG solves), residual check.  A DoF-update is one entry of a vector
produced by an operator application y = A x or by an assembly pass; `value` = DoF-updates of the K timed steps
(all ranks) / wall time of those steps, INCLUDING the Krylov vector work, the preconditioner, reductions and host control.
Only Krylov-level work is counted: (iterations + 1) operator applications per CG solve, the Jacobian only when dt changed.

The headline run is the FASTEST solver (time to solution): CG preconditioned by the block fast diagonalisation.  The metric's unit
rewards operator applications, not solved steps - the polynomial (Chebyshev) preconditioner does 18 x more applications per step and
therefore shows a 6 x higher `value` at 2.7 x the time per step - so the line also carries `best_ms_per_step` and, under
`time_to_solution`, the same step with the other preconditioners, and `config5`: 100 CONSECUTIVE steps of the transient (BASELINE config 5).

Workload: 3D Q2/Q1 uniform box of 72^3 cells (N_u = 9 145 875: BASELINE config "3D Q2/Q1 ~10M DoF").  N > 1 cuts THAT mesh into N z-slabs
(strong scaling: 9 cell layers per GPU at N = 8); the weak-scaled variant (72 layers per GPU, 72 x 72 x 72N box, same h) is measured
as well and reported under "weak_scaling_line".  `python bench.py --gpus N` starts its own N rank processes when it was not started by a launcher.
Every timed step must do a real displacement solve: by default each timed step is time
step 1 from the initial equilibrium (the device state is rolled back before every step - which also clears the solver's iteration-count history - so all
steps do the same work whatever --steps / --warmup are; --transient times consecutive steps instead), the CG stops on the reduction of the step's own initial residual
(PORO_STOP_REDUCTION), the per-step iteration counts are printed and a step with 0 iterations fails the run.
Input is synthetic in the sense of SURVEY 8d: the bundled input.data material / BC values on a generated mesh.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import poroelasticity_dealii_amd as pk  # noqa: E402

INPUT = dict(E=1.4e10, nu=0.3, alpha=0.9, poro=0.3, f_comp=5.8e-10, perm_mD=10.0, visc=1e-3, r_well=1.0, flow_rate=1e-5, p_init=10e6, dt=60.0)   # input.data:13-40
BC_3D = [(0, 0, 0.0), (1, 0, -1e-5), (2, 1, 0.0), (3, 1, -1e-5), (4, 2, 0.0), (5, 2, -1e-5)]   # input.data:14-16 + z faces
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6   # dense fp64 matrix peak of MI355X = half the f32 matrix rate of MI355X_MICROARCH.md (157.3); v_mfma_f64_16x16x4_f64 measured at 77.8 (tools/micro/mfma_rate.hip)


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


def cpu_baseline(dim, degree, n, rel_tol, reduction, csr_n):
    """the oracle (CPU restatement of the reference algorithm, 1 thread) on a bounded sample of the same workload: ONE run = initialisation (not timed as part of
    the step) + one time step, split by the oracle itself"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    P = pk.Problem.box(dim, [n] * dim, [10.0] * dim, degree, material(), BC_3D[:2 * dim])
    O = oracle_py.Oracle(P)                      # naive i x q x j assembly (quirk Q6) + SSOR-CG = the reference's algorithm
    O.work_counts(reset=True)
    O.run(1, INPUT["p_init"], INPUT["dt"], abs_u=1e-12, rel_u=rel_tol, max_it=100000, reduction=reduction)
    w_all = O.work_counts(reset=True); w_init, t_init, dt_step = O.last_run_split()
    w = {k: w_all[k] - w_init[k] for k in w_all}
    upd = dof_updates(w, P.desc.n_dofs_u, P.desc.n_dofs_p, dim)
    out = {"value": upd / max(dt_step, 1e-9), "unit": "DoF-updates/s", "cores": 1, "kind": "port",
           "sample": f"oracle = CPU restatement of the reference (naive assembly + SSOR-CG, g++ -O2, 1 thread); {dim}D Q{degree}/Q1 {n}^{dim} cells "
                     f"(N_u={P.desc.n_dofs_u}), one time step = {dt_step:.2f} s ({w['apply_u']} A_u applications), after {t_init:.1f} s of initialisation",
           "host_cpu": _cpu_name(), "host_cores_available": os.cpu_count()}
    O.close(); P.close()
    # context figure: what the assembled-CSR data structure gives on all host cores the lease exposes (not the reference's serial algorithm), out of cache
    try:
        threads = host_cores()
        n2 = csr_n if dim == 3 else 16 * csr_n
        P2 = pk.Problem.box(dim, [n2] * dim, [10.0] * dim, degree, material(), BC_3D[:2 * dim])
        O2 = oracle_py.Oracle(P2, hoisted=True)
        O2.fill_synthetic_matrix()               # real pattern, synthetic SPD values: assembling 32^3 with the serial cell loop takes minutes, and SpMV speed does not depend on the values
        t_spmv, t_cg = O2.bench_spmv_threads(threads, reps=10)
        out["all_cores_csr"] = {"threads": threads, "sample": f"{dim}D Q{degree}/Q1 {n2}^{dim} cells, N_u={P2.desc.n_dofs_u}, CSR pattern of A_u (~{12 * (8 * n2 + 1) ** 3 * 9 / 1e9:.1f} GB of values + columns) with synthetic SPD values" if dim == 3 else f"{dim}D {n2}^2 cells, CSR pattern of A_u, synthetic SPD values",
                                "spmv_DoF_updates_per_s": P2.desc.n_dofs_u / t_spmv, "jacobi_cg_iterations_per_s": 1.0 / t_cg,
                                "jacobi_cg_DoF_updates_per_s": P2.desc.n_dofs_u / t_cg}
        O2.close(); P2.close()
    except Exception as exc:                      # the headline cpu_baseline stays valid without the context figure
        out["all_cores_csr"] = {"error": str(exc)}
    return out


def host_cores():
    """cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (a one-GPU lease of this pool exposes 256 logical CPUs but a 16-core quota)"""
    n = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, int(os.environ.get("PORO_BENCH_MAX_THREADS", "64")))


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
        for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            with open(path) as f:
                rec = json.load(f)
            if rec.get("kernel_source_sha16") == kernel_source_stamp():
                return rec[key]["hbm_bytes_per_launch"]
        return None
    except (OSError, KeyError, ValueError):
        return None


def committed_fdm_traffic():
    """HBM bytes per launch of the block-FDM transform pass from the committed PMC passes (profiles/r03_fdmo_pmc_traffic.json, stamped with the kernel source)"""
    import hashlib
    try:
        with open(os.path.join(ROOT, "profiles", "r03_fdmo_pmc_traffic.json")) as f:
            rec = json.load(f)
        with open(os.path.join(ROOT, "poroelasticity_dealii_amd", "csrc", "kernels_fdmo.hip"), "rb") as f:
            if rec.get("kernel_source_sha16") != hashlib.sha256(f.read()).hexdigest()[:16]:
                return None
        return rec["hbm_bytes_per_launch_mean"]
    except (OSError, KeyError, ValueError):
        return None


def run_case(args, scaling, steps, warmup, rank, world, local_rank, torch, dist, rccl_ids, prec=None, label="jacobi", transient=None, events=True):
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
        # (RCCL prints a version banner on the C-level stdout when a communicator is created: keep stdout for the JSON line)
        sys.stdout.flush(); saved_fd = os.dup(1); os.dup2(2, 1)
        try:
            G.comm_rccl(rccl_ids[label + scaling])
        finally:
            os.dup2(saved_fd, 1); os.close(saved_fd)

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
    repeat = not (args.transient if transient is None else transient)
    if repeat:
        R.save_state()                        # every step below is "time step 1 from the initial equilibrium": a device-side rollback (~30 us of copies, inside the timed region)
    for _ in range(warmup):
        R.step(restore=repeat)
    before = R.work()
    G.timers_reset()                          # HIP events around every kernel family on the launch stream
    # every launch with events costs the step ~7 % (2.5 us per launch, 650 launches per step): sample every `event_stride`-th launch of each kernel family
    # (the library scales the sampled time to all launches of the family)
    G.timers_enable(0 if (getattr(args, "no_kernel_events", False) or not events) else args.event_stride)
    sync()
    t0 = time.perf_counter()
    traces, step_seconds = [], []
    for _ in range(steps):
        ts = time.perf_counter()
        traces.append(R.step(restore=repeat)[0]); step_seconds.append(time.perf_counter() - ts)   # (rollback + step in one call; the step's last entry point waits for the device)
    sync()
    elapsed = time.perf_counter() - t0
    after = R.work()
    G.timers_enable(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); elapsed = float(tt[0])
    work = {k: after[k] - before[k] for k in after}
    families = ("apply_u_matrix_free", "apply_u_chebyshev_fused", "apply_u_dirichlet_rows", "assemble_u_rhs", "projection_rhs", "pressure_residual", "pressure_jacobian", "apply_p_csr", "apply_p_stencil",
                "precondition_p_fdm", "precondition_u_fdm", "fdm_u_pass1", "fdm_u_pass2", "fdm_u_pass3", "precondition_u_chebyshev", "halo_exchange", "allreduce", "alltoall")
    out = {"n": n, "n_u_glob": n_u_glob, "n_p_glob": n_p_glob, "n_u_local": P.desc.n_dofs_u, "n_cells_local": n_cells, "elapsed": elapsed, "work": work,
           "updates": dof_updates(work, n_u_glob, n_p_glob, dim), "traces": traces, "step_seconds": step_seconds,
           "kernel_time": {k: G.timer(k) for k in families}}
    R.close(); P.close()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh children, one per GPU, before anything in this process touches the GPU),
    rank 0's JSON line goes to our stdout, the exit code is the worst of the children's"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
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
                    help="time consecutive steps of the transient in the headline run instead of repeating time step 1 (the `config5` block always runs consecutive steps)")
    ap.add_argument("--prec", choices=["chebyshev", "jacobi", "block_fdm"], default="block_fdm",
                    help="preconditioner of the displacement CG in the headline run; default = the fastest time to solution (block fast diagonalisation); the others are measured as well "
                         "(one GPU) and reported under time_to_solution")
    ap.add_argument("--cheb-degree", type=int, default=6)
    ap.add_argument("--cheb-ratio", type=int, default=0, help="lambda_max / lambda_min of the Chebyshev interval (0: the library default, a few times the mesh-dependent lambda_min)")
    ap.add_argument("--max-iter", type=int, default=50000)
    ap.add_argument("--cpu-n", type=int, default=16, help="cells per direction of the single-thread CPU-baseline sample (SURVEY 8d: 16^3-24^3; 16: ~20 s per time step + as much initialisation on the GPU box's EPYC 9575F)")
    ap.add_argument("--cpu-csr-n", type=int, default=32, help="cells per direction of the all-cores CSR SpMV / CG sample (32: 1.8 GB of matrix, out of cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra time-to-solution measurements with the other preconditioners")
    ap.add_argument("--config5-steps", type=int, default=100, help="consecutive time steps of the `config5` block (BASELINE config 5: 100); 0 = skip")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: all ranks use device 0 and exchange through gloo (host-staged callbacks) instead of RCCL")
    ap.add_argument("--event-stride", type=int, default=8, help="attach HIP events to every n-th launch of each kernel family inside the timed region (1: every launch)")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: do not attach HIP events to the kernel dispatches inside the timed region (no roofline figures)")
    ap.add_argument("--trace-out", default=None, help="write the per-step record of the config5 block (SURVEY 8d config 5: FSS / pressure / Krylov iteration counts, wall-clock) to this JSON file")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        self_launch(args)                      # does not return
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
            # control plane only (barrier, max, id broadcast); the data plane is RCCL inside the library.  gloo announces its connections on the C-level stdout, which must
            # carry nothing but the one JSON line: file descriptor 1 points at stderr while the group is set up and its first collective runs
            sys.stdout.flush(); saved_fd = os.dup(1); os.dup2(2, 1)
            try:
                dist.init_process_group(backend="gloo")
                dist.barrier()
            finally:
                sys.stdout.flush(); os.dup2(saved_fd, 1); os.close(saved_fd)
    except ImportError:
        torch = None

    dim, deg = args.dim, args.degree
    PREC = {"jacobi": pk.PREC_JACOBI, "chebyshev": pk.PREC_CHEBYSHEV, "block_fdm": pk.PREC_FDM}
    head = args.prec
    # one RCCL communicator per context: partitioned runs measure the headline solver (+ its weak-scaled and transient runs) only
    others = [] if (args.no_variants or world > 1) else [k for k in ("block_fdm", "chebyshev", "jacobi") if k != head]
    weak_line = world > 1 and args.scaling == "strong" and not args.no_weak_line
    cfg5 = args.config5_steps > 0
    rccl_ids = {}
    if world > 1 and not args.share_gpu:      # ids from rank 0
        labels = [head + args.scaling] + ([head + "weak"] if weak_line else []) + ([head + "config5" + args.scaling] if cfg5 else [])
        ids = [{lb: pk.rccl_unique_id() for lb in labels} if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        rccl_ids = ids[0]

    def attempt(*a, **kw):
        """run_case, with the ranks agreeing on a failure (a rank that raised alone would leave the others waiting in a collective of the next case)"""
        err, out = "", None
        try:
            out = run_case(*a, **kw)
        except RuntimeError as exc:
            err = str(exc) or "RuntimeError"
        if dist is not None:
            errs = [None] * world; dist.all_gather_object(errs, err)
            err = next((e for e in errs if e), "")
        if err:
            raise RuntimeError(err)
        return out

    M = attempt(args, args.scaling, args.steps, args.warmup, rank, world, local_rank, torch, dist, rccl_ids, prec=PREC[head], label=head)
    work, elapsed, updates, traces, step_seconds = M["work"], M["elapsed"], M["updates"], M["traces"], M["step_seconds"]
    n, n_u_glob, n_p_glob = M["n"], M["n_u_glob"], M["n_p_glob"]
    kernel_time = M["kernel_time"]

    # ---- rooflines (per launch, per GPU; kernel durations from HIP events attached to sampled dispatches inside the timed steps) ----
    # operator y = A_u x (matrix-free, the kernel north_star names): HBM-bound, 16 N + 4 dpc n_cells algorithmic bytes (the structured kernels read no index arrays:
    # `frac_without_index_bytes`); with the Chebyshev preconditioner the polynomial update is fused into its stores (+ 8 N for the extra stream g).
    t_plain, n_plain_launched = kernel_time["apply_u_matrix_free"]
    t_cheb, n_cheb_launched = kernel_time["apply_u_chebyshev_fused"]
    alg_plain = bytes_per_apply(dim, deg, M["n_u_local"], M["n_cells_local"], "matrix_free")
    index_bytes = 4.0 * dim * (deg + 1) ** dim * M["n_cells_local"]
    cg_total = int(work["cg_u"]); solves = int(work["asm_rhs_u"])
    fused_cheb = n_cheb_launched > 0
    n_plain_useful = (cg_total + solves) if (fused_cheb or head == "block_fdm") else (int(work["apply_u"]) or n_plain_launched)
    avg_plain = t_plain / max(n_plain_useful, 1)
    op_kernel = "k_kron3_q%d" % deg if dim == 3 else "k_kron2<%d>" % deg
    roof_operator = {"bound": "hbm", "kernel": op_kernel + " (matrix-free y = A_u x, sum-factorised)", "achieved": alg_plain / avg_plain / 1e9 if (n_plain_useful and t_plain) else 0.0,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": alg_plain, "avg_launch_us": 1e6 * avg_plain, "useful_launches": n_plain_useful,
                     "traffic": committed_traffic(dim, deg, args.n, False) if world == 1 else None}
    roof_operator["frac"] = roof_operator["achieved"] / HBM_PEAK_GBS
    roof_operator["frac_without_index_bytes"] = (alg_plain - index_bytes) / avg_plain / 1e9 / HBM_PEAK_GBS if (n_plain_useful and t_plain) else 0.0
    if fused_cheb:
        n_cheb_useful = (args.cheb_degree + (args.cheb_degree & 1)) * (cg_total + solves)
        avg_cheb = t_cheb / max(n_cheb_useful, 1); alg_cheb = alg_plain + 8.0 * M["n_u_local"]
        roofline = {"bound": "hbm", "kernel": ("k_kron3_q%d_cheb" % deg if dim == 3 else "k_kron2<%d, true>" % deg) + " (matrix-free A_u z_j with the Chebyshev root-form update fused into the stores)",
                    "achieved": alg_cheb / avg_cheb / 1e9 if avg_cheb > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": committed_traffic(dim, deg, args.n, True) if world == 1 else None,
                    "algorithmic_bytes_per_launch": alg_cheb, "avg_launch_us": 1e6 * avg_cheb, "useful_launches": n_cheb_useful,
                    "frac_without_index_bytes": (alg_cheb - index_bytes) / avg_cheb / 1e9 / HBM_PEAK_GBS if avg_cheb > 0 else 0.0}
        roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
    elif head == "block_fdm" and kernel_time["fdm_u_pass1"][1] > 0:
        # dominant kernel of the headline run: the transform pass of the block fast diagonalisation (3 launches per preconditioner application): GEMM-shaped, fp64 MFMA-bound.
        # Algorithmic flops of one pass = 2 GEMMs per (component, octant, block) of the UNPADDED half-size problem: pass 1 / 3: 2 h_y h_x (h_x + h_y) per plane, pass 2: 4 h_z^2 per column.
        nn = [deg * m + 1 for m in M["n"]]; h = [(v + 1) // 2 for v in nn]
        flops_pass = [24.0 * h[2] * 2.0 * h[1] * h[0] * (h[0] + h[1]), 24.0 * h[0] * h[1] * 4.0 * h[2] * h[2], 24.0 * h[2] * 2.0 * h[1] * h[0] * (h[0] + h[1])]
        t_pass = [kernel_time["fdm_u_pass%d" % (k + 1)][0] / max(kernel_time["fdm_u_pass%d" % (k + 1)][1], 1) for k in range(3)]
        n_app = kernel_time["fdm_u_pass1"][1]
        avg = sum(t_pass) / 3.0; fl = sum(flops_pass) / 3.0
        n_oct_bytes = 8.0 * 24 * h[2] * h[1] * (h[0] + (h[0] & 1))
        roofline = {"bound": "mfma", "kernel": "k_fdmo_pass<NT, MODE> (one sweep of the octant-form block fast diagonalisation: two chained half-size GEMMs per (component, octant, block) on v_mfma_f64_16x16x4_f64; 3 launches per preconditioner application)",
                    "achieved": fl / avg / 1e12 if avg > 0 else 0.0, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "algorithmic_flops_per_launch": fl, "avg_launch_us": 1e6 * avg,
                    "avg_launch_us_by_pass": [1e6 * t for t in t_pass], "launches_enqueued": 3 * n_app,
                    "traffic": committed_fdm_traffic() if world == 1 else None, "algorithmic_bytes_per_launch": 2.0 * n_oct_bytes,
                    "hbm_frac_of_the_same_kernel": 2.0 * n_oct_bytes / avg / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0}
        roofline["frac"] = roofline["achieved"] / MFMA_F64_PEAK_TFLOPS
    else:
        roofline = dict(roof_operator)
    roofline["events_on_every_nth_launch"] = args.event_stride
    roofline["note"] = ("kernel alone: HIP events attached to the dispatch itself (hipExtLaunchKernelGGL), as rocprofv3 reports it, on every events_on_every_nth_launch-th launch of the timed steps; "
                        "`roofline` = the dominant kernel of the headline run, `roofline_operator` = the matrix-free operator application (HBM-bound; algorithmic bytes per SURVEY 8d, "
                        "frac_without_index_bytes leaves out the element-index bytes the structured kernel does not read); traffic = PMC bytes per launch from profiles/, null unless measured on this kernel source")

    # ---- the same steps with the other preconditioners of the displacement CG ----
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
            tts[k] = summary(attempt(args, args.scaling, args.steps, args.warmup, rank, world, local_rank, torch, dist, rccl_ids, prec=PREC[k], label=k), k)
        except RuntimeError as exc:           # a context that cannot use it (e.g. Dirichlet data not face-separable) says so; the headline stays valid
            tts[k] = {"error": str(exc)}
    best = min((k for k in tts if "ms_per_step" in tts[k]), key=lambda k: tts[k]["ms_per_step"])
    weak = None
    if weak_line:
        W = attempt(args, "weak", min(args.steps, 3), 1, rank, world, local_rank, torch, dist, rccl_ids, prec=PREC[head], label=head)
        weak = {"value": W["updates"] / W["elapsed"], "ms_per_step": 1e3 * W["elapsed"] / min(args.steps, 3), "cells": "x".join(map(str, W["n"])), "N_u": W["n_u_glob"],
                "cg_iterations_u": [[int(r[6]) for r in t] for t in W["traces"]]}
    # ---- BASELINE config 5: consecutive steps of the transient from the initial equilibrium (no rollback, no kernel events) ----
    config5 = None
    if cfg5:
        try:
            T = attempt(args, args.scaling, args.config5_steps, 0, rank, world, local_rank, torch, dist, rccl_ids, prec=PREC[head], label=head + "config5", transient=True, events=False)
            config5 = {"steps": args.config5_steps, "seconds": T["elapsed"], "ms_per_step_mean": 1e3 * T["elapsed"] / args.config5_steps, "preconditioner": head,
                       "cg_iterations_u": [int(t[-1][6]) for t in T["traces"]], "fss_iterations": [int(len(t)) for t in T["traces"]], "pressure_iterations": [int(t[-1][2]) for t in T["traces"]],
                       "DoF_updates_per_s": T["updates"] / T["elapsed"], "note": "consecutive time steps dt = 60 s from the initial equilibrium, wall clock of the whole loop incl. host control"}
            if rank == 0 and args.trace_out:
                rec = [{"step": int(t[-1][0]), "fss_iterations": int(len(t)), "pressure_iterations": [int(r[2]) for r in t], "cg_iterations_u": [int(r[6]) for r in t],
                        "cg_iterations_p": [int(r[7]) for r in t], "p_inf": float(t[-1][4]), "fss_error": float(t[-1][5]), "seconds": sec} for t, sec in zip(T["traces"], T["step_seconds"])]
                with open(args.trace_out, "w") as f:
                    json.dump({"workload": f"{dim}D Q{deg}/Q1 {'x'.join(map(str, n))} cells, preconditioner {head}", "n_gpus": world, "steps": rec, "seconds_total": T["elapsed"]}, f, indent=1)
        except RuntimeError as exc:
            config5 = {"error": str(exc)}

    cg_u = [[int(r[6]) for r in t] for t in traces]
    dead = [i for i, its in enumerate(cg_u) if not its or min(its) == 0]
    if rank == 0:
        stop_txt = (f"recursive residual <= max(1e-12, {args.rel_tol:g}*||g_0||), g_0 = residual of the step's warm start (ReductionControl), cap {args.max_iter}" if args.stop == "reduction"
                    else f"recursive residual <= max(1e-12, {args.rel_tol:g}*||b||), cap {args.max_iter}")
        out = {
            "metric": "DoF-updates/sec in assemble+SpMV per fixed-stress iter", "value": updates / elapsed, "unit": "DoF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{dim}D Q{deg}/Q1 uniform box, {'x'.join(map(str, n))} cells, N_u={n_u_glob}, N_p={n_p_glob}; {'consecutive time steps' if args.transient else 'time step 1 repeated (device-side rollback before each step; the rollback also clears the iteration-count history of the solver)'}; one time step = one fixed-stress iteration "
                                   f"(pressure loop + matrix-free PCG displacement solve, preconditioner: {head}" + (f" degree {args.cheb_degree}" if head == "chebyshev" else "") + " + strain projection); input.data material/BCs, z-face BCs per SURVEY Q9",
                       "parallelism": f"z-slab x{world}" if world > 1 else "single GPU", "operator": "matrix_free",
                       "stopping_rule_u": stop_txt},
            "best_ms_per_step": tts[best]["ms_per_step"], "best_preconditioner": best,
            "cg_iterations_u": cg_u,
            "roofline": roofline, "roofline_operator": roof_operator,
            "work_per_step": {k: work[k] / args.steps for k in work},
            "kernel_only": {"seconds_by_family": {k: v[0] for k, v in kernel_time.items()}, "launches_by_family": {k: v[1] for k, v in kernel_time.items()}},
            "fss_iterations_per_step": [int(len(t)) for t in traces],
        }
        out["time_to_solution"] = tts
        if config5 is not None:
            out["config5"] = config5
        if weak is not None:
            out["weak_scaling_line"] = weak
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dim, deg, args.cpu_n, args.rel_tol, args.stop == "reduction", args.cpu_csr_n)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    if dead:
        # a timed step whose displacement solve accepted its warm start did no Krylov work: the line above is not a measurement of the hot path
        print(f"bench.py: timed steps {dead} ran 0 displacement CG iterations", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
