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


def test_two_rank_rehearsal():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "8", "--share-gpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "8x8x16" in d["config"]["workload"]                     # weak scaling: the box grows along the partitioned direction
    assert d["work_per_step"]["cg_p"] <= 2 * d["work_per_step"]["residual_p"]     # distributed fast diagonalisation in use
