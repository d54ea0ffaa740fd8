#!/bin/bash
# quick probe on the GPU box: FDM parity tests, then per-kernel durations of one preconditioner application at config 4 (for a list of stagger values)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/fdmo_probe; mkdir -p $OUT
cd $ROOT && timeout -k 10 600 python -m pytest tests/test_fdm_u_gpu.py -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for STG in ${STAGGERS:-700}; do
  rm -rf $OUT/stats; PORO_FDMO_STAGGER=$STG timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/fdmu_bench.py 3 72 2 > $OUT/bench.log 2>&1
  echo "== stagger $STG: $(grep block-FDM $OUT/bench.log)"
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "fdmo_pass" in r["Name"]: print("   %-50s calls %4s avg %8.1f us min %8.1f max %8.1f" % (r["Name"].replace("poro::(anonymous namespace)::","")[:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
PORO_FDMO_STAMPS=/tmp/st.txt REPS=4 python3 $ROOT/tools/fdmu_bench.py 3 72 2 > /dev/null 2>&1 && python3 $ROOT/tools/fdmo_stamps.py /tmp/st.txt > $OUT/stamps.txt; cat $OUT/stamps.txt
