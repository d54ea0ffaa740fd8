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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--cells", "8", "--cpu-n", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 0
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"]
    assert len(d["cg_iterations_u"]) == 1 and min(d["cg_iterations_u"][0]) > 0     # every timed step does a live displacement solve
    assert "reduction" in d["config"]["stopping_rule_u"].lower() or "g_0" in d["config"]["stopping_rule_u"]
    assert set(d["time_to_solution"]) == {"jacobi", "chebyshev", "block_fdm"}
    tts = d["time_to_solution"]["block_fdm"]
    assert "error" not in tts, tts
    assert max(tts["cg_iterations_u"][0]) <= 40 and tts["applications_precondition_u"] > 0
    assert d["time_to_solution"]["chebyshev"]["cg_iterations_u"][0][0] < d["time_to_solution"]["jacobi"]["cg_iterations_u"][0][0]
    assert "cheb" in d["roofline"]["kernel"] and d["roofline"]["plain_operator"]["avg_launch_us"] > 0


def test_live_steps_do_not_depend_on_the_window():
    """the warm-started transient keeps solving: later steps take about as many CG iterations as the first (the r1 bench measured empty steps)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "4", "--cells", "8", "--no-cpu-baseline", "--no-variants"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    its = [t[0] for t in _line(r.stdout)["cg_iterations_u"]]
    assert min(its) > 0 and max(its) == min(its), its             # time step 1 repeated: identical work in every timed step
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "4", "--cells", "8", "--no-cpu-baseline", "--no-variants", "--transient"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert min(t[0] for t in _line(r.stdout)["cg_iterations_u"]) > 0


def test_dead_steps_fail_the_run():
    """with the ||b||-relative rule the transient dies after a few steps on this tiny mesh: bench.py must refuse to report such a window"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "40", "--cells", "4", "--stop", "rhs", "--rel-tol", "1e-6", "--no-cpu-baseline", "--no-variants", "--transient"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 3, (r.returncode, r.stdout[-500:], r.stderr[-500:])


def test_two_rank_rehearsal():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "8", "--share-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "8x8x8" in d["config"]["workload"]                      # strong scaling: the BASELINE mesh itself is cut into slabs
    assert d["weak_scaling_line"]["cells"] == "8x8x16" and d["weak_scaling_line"]["value"] > 0     # the box grown along the partitioned direction
    assert d["work_per_step"]["cg_p"] <= 2 * d["work_per_step"]["residual_p"]     # distributed fast diagonalisation in use
    # single-reduction PCG: one all-reduce per CG iteration of any system (+ one per finished solve, per residual norm and per reported vector norm), and one grouped
    # neighbour exchange per operator application / assembled vector (the round-1 recurrence needed two all-reduces per iteration)
    w, fam = d["work_per_step"], d["kernel_only"]["launches_by_family"]
    its = w["cg_u"] + w["cg_p"] + w["cg_proj"]; solves = 1 + w["residual_p"] + 3
    assert fam["allreduce"] <= its + 2 * solves + w["residual_p"] + 8, (fam, w)
    assert fam["halo_exchange"] <= w["apply_u"] + w["apply_p"] + 2 * solves + 24, (fam, w)


def test_partitioned_chebyshev_takes_the_single_rank_iteration_count():
    """the polynomial's interval comes from a Lanczos estimate of lambda_max; on a partition its start vector must agree on the shared planes, otherwise the estimate
    drifts with the number of ranks (it once hit the loose element bound at 4 ranks: 46 instead of 38 iterations at 72^3).  Same mesh on 1 and on 3 ranks: same count."""
    base = [os.path.join(ROOT, "bench.py"), "--cells", "24", "--steps", "1", "--warmup", "1", "--no-variants", "--no-weak-line", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable] + base, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, (r1.stdout + r1.stderr)[-2000:]
    r3 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(free_port())] + base + ["--gpus", "3", "--share-gpu"],
                        capture_output=True, text=True, timeout=600)
    assert r3.returncode == 0, (r3.stdout + r3.stderr)[-2000:]
    i1, i3 = _line(r1.stdout)["cg_iterations_u"][0][0], _line(r3.stdout)["cg_iterations_u"][0][0]
    assert abs(i1 - i3) <= 1, (i1, i3)
