#!/bin/bash
# kernel statistics of the partitioned code path on one RCCL rank (tools/partitioned_path_1rank.py <n> force): which launches a slab rank adds to the single-rank step
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/slab_probe; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/partitioned_path_1rank.py ${1:-72} force block_fdm > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:28]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("poro::", "").split("(")[0]
    print("%-52s calls %6s avg %8.1f us total %8.2f ms" % (n[:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
grep "ms per step" $OUT/run.log
rm -rf $OUT/stats
