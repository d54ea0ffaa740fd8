"""bench.py contract (task statement: one JSON line with metric / value / roofline / cpu_baseline ...) on a small mesh, and a rehearsal of
its multi-rank path: 2 ranks launched by torch.distributed.run exactly as the driver does, sharing the one GPU of the test box and
exchanging through gloo callbacks instead of RCCL (--share-gpu)."""
import json
import os
import subprocess
import sys

import pytest

from test_multirank_cpu import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--cells", "8", "--cpu-n", "3", "--cpu-csr-n", "4", "--config5-steps", "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 0
    # the headline is the fastest solver (block fast diagonalisation); its dominant kernel is the GEMM-shaped transform pass (fp64 MFMA roof), the operator application
    # (the kernel north_star names, HBM roof) is reported beside it
    assert "block_fdm" in d["config"]["workload"] and d["best_preconditioner"] in d["time_to_solution"] and d["best_ms_per_step"] <= d["ms_per_step"] * 1.0001
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "mfma" and "k_fdmo_pass" in d["roofline"]["kernel"]
    assert d["roofline"]["unit"] == "TFLOP/s" and 0 < d["roofline"]["frac"] < 1
    ro = d["roofline_operator"]
    assert set(ro) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_without_index_bytes"} and ro["bound"] == "hbm" and ro["avg_launch_us"] > 0 and ro["frac_without_index_bytes"] < ro["frac"]
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port" and "error" not in d["cpu_baseline"]["all_cores_csr"]
    assert "workload" in d["config"]
    assert len(d["cg_iterations_u"]) == 1 and min(d["cg_iterations_u"][0]) > 0     # every timed step does a live displacement solve
    assert "reduction" in d["config"]["stopping_rule_u"].lower() or "g_0" in d["config"]["stopping_rule_u"]
    assert set(d["time_to_solution"]) == {"jacobi", "chebyshev", "block_fdm"}
    tts = d["time_to_solution"]["block_fdm"]
    assert "error" not in tts, tts
    assert max(tts["cg_iterations_u"][0]) <= 40 and tts["applications_precondition_u"] > 0
    assert d["time_to_solution"]["chebyshev"]["cg_iterations_u"][0][0] < d["time_to_solution"]["jacobi"]["cg_iterations_u"][0][0]
    # BASELINE config 5 in miniature: consecutive steps, every one of them solved
    c5 = d["config5"]
    assert c5["steps"] == 6 and len(c5["cg_iterations_u"]) == 6 and min(c5["cg_iterations_u"]) > 0 and c5["seconds"] > 0 and c5["fss_iterations"] == [1] * 6


def test_chebyshev_headline_reports_the_fused_kernel():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--cells", "8", "--prec", "chebyshev", "--no-variants", "--no-cpu-baseline", "--config5-steps", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["roofline"]["bound"] == "hbm" and "cheb" in d["roofline"]["kernel"] and d["roofline_operator"]["avg_launch_us"] > 0 and "config5" not in d


def test_gpus_flag_without_a_launcher_starts_its_own_ranks():
    """the driver's N = 1 command is plain `python bench.py --gpus 1`; the same form with N > 1 must start N rank processes itself"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "0", "--cells", "8", "--no-weak-line", "--config5-steps", "2"],
                       capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config5"]["steps"] == 2 and min(d["config5"]["cg_iterations_u"]) > 0


def test_live_steps_do_not_depend_on_the_window():
    """the warm-started transient keeps solving: later steps take about as many CG iterations as the first (the r1 bench measured empty steps)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "4", "--cells", "8", "--no-cpu-baseline", "--no-variants", "--config5-steps", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    its = [t[0] for t in _line(r.stdout)["cg_iterations_u"]]
    assert min(its) > 0 and max(its) == min(its), its             # time step 1 repeated: identical work in every timed step
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "4", "--cells", "8", "--no-cpu-baseline", "--no-variants", "--transient", "--config5-steps", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert min(t[0] for t in _line(r.stdout)["cg_iterations_u"]) > 0


def test_dead_steps_fail_the_run():
    """with the ||b||-relative rule the transient dies after a few steps on this tiny mesh: bench.py must refuse to report such a window"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "40", "--cells", "4", "--stop", "rhs", "--rel-tol", "1e-6", "--no-cpu-baseline", "--no-variants", "--transient", "--config5-steps", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stdout[-500:], r.stderr[-500:])


def test_two_rank_rehearsal():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "8", "--share-gpu", "--config5-steps", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:500]      # stdout carries the one JSON line and nothing else (gloo's connection messages go to stderr)
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "8x8x8" in d["config"]["workload"]                      # strong scaling: the BASELINE mesh itself is cut into slabs
    assert d["weak_scaling_line"]["cells"] == "8x8x16" and d["weak_scaling_line"]["value"] > 0     # the box grown along the partitioned direction
    assert d["work_per_step"]["cg_p"] <= 2 * d["work_per_step"]["residual_p"]     # distributed fast diagonalisation in use
    # pressure / projection systems: single-reduction PCG, one all-reduce per CG iteration (+ one per finished solve, per residual norm and per reported vector norm);
    # displacement system with the block fast diagonalisation in quadrant form: SolverCG's own recurrence on the device-side scalars, two all-reduces per iteration
    # (d.Ad, then g.g with g.z) + one set at the start; one grouped neighbour exchange per operator application / assembled vector
    w, fam = d["work_per_step"], d["kernel_only"]["launches_by_family"]
    its = 2 * w["cg_u"] + w["cg_p"] + w["cg_proj"]; solves = 1 + w["residual_p"] + 3
    assert fam["allreduce"] <= its + 2 * solves + w["residual_p"] + 8, (fam, w)
    assert fam["halo_exchange"] <= w["apply_u"] + w["apply_p"] + 2 * solves + 24, (fam, w)


def test_partitioned_chebyshev_takes_the_single_rank_iteration_count():
    """the polynomial's interval comes from a Lanczos estimate of lambda_max; on a partition its start vector must agree on the shared planes, otherwise the estimate
    drifts with the number of ranks (it once hit the loose element bound at 4 ranks: 46 instead of 38 iterations at 72^3).  Same mesh on 1 and on 3 ranks: same count."""
    base = [os.path.join(ROOT, "bench.py"), "--cells", "24", "--steps", "1", "--warmup", "1", "--prec", "chebyshev", "--no-variants", "--no-weak-line", "--no-cpu-baseline", "--config5-steps", "0"]
    r1 = subprocess.run([sys.executable] + base, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, (r1.stdout + r1.stderr)[-2000:]
    r3 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + base + ["--gpus", "3", "--share-gpu"],
                        capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, (r3.stdout + r3.stderr)[-2000:]
    i1, i3 = _line(r1.stdout)["cg_iterations_u"][0][0], _line(r3.stdout)["cg_iterations_u"][0][0]
    assert abs(i1 - i3) <= 1, (i1, i3)
