#!/bin/bash
# GPU idle gaps of the default bench step (kernel trace of 8 timed steps): tools/gap_analysis.py on the rocprofv3 kernel trace
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/gap_probe; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-variants --no-cpu-baseline --config5-steps 0 --no-kernel-events "$@" > $OUT/line.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 $ROOT/tools/gap_analysis.py $(ls $OUT/ktrace/*/*kernel_trace.csv | head -1) 20 > $OUT/gaps.txt
cp $(ls $OUT/ktrace/*/*kernel_trace.csv | head -1) $OUT/kernel_trace.csv
rm -rf $OUT/ktrace
head -40 $OUT/gaps.txt
