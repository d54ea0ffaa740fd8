#!/bin/bash
# BASELINE config 5 (100 consecutive steps) under the kernel trace: idle gaps and the kernel totals of the steady part
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/config5_probe; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-variants --no-cpu-baseline --config5-steps 100 --no-kernel-events "$@" > $OUT/line.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
T=$(ls $OUT/ktrace/*/*kernel_trace.csv | head -1)
python3 $ROOT/tools/gap_analysis.py $T 15 > $OUT/gaps.txt
python3 - $T > $OUT/kernels_steady.txt <<PY
import collections, csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); rows = rows[len(rows) // 3:]
tot = collections.defaultdict(lambda: [0, 0]); 
for s, e, n in rows: tot[n][0] += 1; tot[n][1] += e - s
span = rows[-1][1] - rows[0][0]; busy = sum(v[1] for v in tot.values())
print(f"span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]: print(f"{t/1e6:8.3f} ms {100*t/span:5.1f}% {c:6d} x {t/c/1e3:7.1f} us  {n[:90]}")
PY
rm -rf $OUT/ktrace
python3 -c "
import json; d = json.load(open('$OUT/line.json')); print('config5 (profiled):', d['config5']['seconds'])"
head -30 $OUT/gaps.txt; head -45 $OUT/kernels_steady.txt
