#!/bin/bash
# A/B of the structured operator kernels on the GPU box: time (unprofiled) and PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) of tools/bench_ops.py,
# with the environment given as arguments (e.g. PORO_KRON_COLMAJOR=1).  Output: gpurun_out/ab_<tag>/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/ab_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/bench_ops.py 3,72,2,mf > $OUT/time.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $ROOT/tools/bench_ops.py 3,72,2,mf > $OUT/f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $ROOT/tools/bench_ops.py 3,72,2,mf > $OUT/w.log 2>&1 || exit 1
FD=$(dirname $(ls $OUT/f/*/*counter_collection.csv | head -1)); WD=$(dirname $(ls $OUT/w/*/*counter_collection.csv | head -1))
python3 $ROOT/tools/pmc_summary.py $FD $WD $OUT/traffic.json > $OUT/traffic.txt || exit 1
rm -rf $OUT/f $OUT/w
cat $OUT/time.log; cat $OUT/traffic.txt
