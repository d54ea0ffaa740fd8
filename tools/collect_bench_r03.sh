#!/bin/bash
# the bench-line part of tools/collect_profiles_r03.sh alone (steps 3 and 4): default line, the same command under the kernel trace, idle gaps, configs 2 and 3
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles_r03; mkdir -p $OUT
TAG=r03
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --trace-out $OUT/${TAG}_config5_100steps_block_fdm.json > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_stderr.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-variants --no-cpu-baseline --config5-steps 0 --steps 20 --warmup 1 > $OUT/${TAG}_bench_line_profiled.json 2>> $OUT/${TAG}_bench_stderr.log || exit 1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-variants --no-cpu-baseline --config5-steps 0 --no-kernel-events > /dev/null 2>&1
python3 $ROOT/tools/gap_analysis.py $(ls $OUT/ktrace/*/*kernel_trace.csv | head -1) 20 > $OUT/${TAG}_gpu_idle_gaps.txt
python3 $ROOT/bench.py --dim 3 --degree 1 --cells 99 --no-cpu-baseline --config5-steps 20 > $OUT/${TAG}_bench_line_c3.json 2>/dev/null
python3 $ROOT/bench.py --dim 2 --cells 336 --no-cpu-baseline --config5-steps 20 > $OUT/${TAG}_bench_line_c2.json 2>/dev/null
python3 $ROOT/tools/partitioned_path_1rank.py 72 noforce > $OUT/${TAG}_partitioned_path_1rank.txt 2>&1
python3 $ROOT/tools/partitioned_path_1rank.py 72 force >> $OUT/${TAG}_partitioned_path_1rank.txt 2>&1
rm -rf $OUT/stats $OUT/ktrace
python3 $ROOT/tools/show_line.py $OUT/${TAG}_bench_line.json; head -6 $OUT/${TAG}_gpu_idle_gaps.txt; grep block_fdm $OUT/${TAG}_partitioned_path_1rank.txt
