#!/bin/bash
# probe on the GPU box: FDM parity tests, then kernel statistics of the block-FDM bench step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/step_probe; mkdir -p $OUT
cd $ROOT && timeout -k 10 600 python -m pytest tests/test_fdm_u_gpu.py -x -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/bstats; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bstats -- python3 $ROOT/bench.py --prec block_fdm --no-variants --no-cpu-baseline --steps 6 --warmup 1 --no-kernel-events > $OUT/bench_profiled.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob, json
d = json.load(open("$OUT/bench_profiled.json")); print("profiled run: ms/step", round(d["ms_per_step"], 3), "cg", d["cg_iterations_u"][0])
f = glob.glob("$OUT/bstats/*/*kernel_stats.csv")[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < ${TOPN:-14}: print("%-64s calls %5s avg %8.1f us total %8.2f ms" % (r["Name"].replace("poro::(anonymous namespace)::","").replace("void ","")[:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
cd $ROOT && python3 bench.py --prec block_fdm --no-variants --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench.json 2>> $OUT/bench.err && python3 -c "
import json; d = json.load(open('$OUT/bench.json')); print('unprofiled: ms/step', round(d['ms_per_step'], 3))"
