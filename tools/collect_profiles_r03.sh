#!/bin/bash
# Round-3 evidence, collected on the GPU box (gpurun): everything lands in gpurun_out/profiles_r03/ and is copied into profiles/ by hand.
#   PMC traffic of the operator kernels and of the block-FDM transform passes (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, never with a trace domain),
#   SQ counters of both, the default bench line, rocprofv3 kernel statistics + GPU idle gaps of the same command, the config-5 trace, the general
#   matrix-free kernel at 72^3 without the box tag, the partitioned code path on one RCCL rank.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles_r03; mkdir -p $OUT
TAG=r03
cd /tmp && export TMPDIR=/tmp
# 1. operator kernels: traffic (stamped with the kernel source) before the bench, which quotes it
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/bench_ops.py 3,72,2,mf 3,99,1,mf > $OUT/${TAG}_ops_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/bench_ops.py 3,72,2,mf 3,99,1,mf > $OUT/${TAG}_ops_write.log 2>&1 || exit 1
FD=$(dirname $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1)); WD=$(dirname $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1))
python3 $ROOT/tools/pmc_summary.py $FD $WD $OUT/${TAG}_pmc_traffic_raw.json > $OUT/${TAG}_pmc_traffic.txt || exit 1
python3 - <<PY
import json, hashlib
d = json.load(open("$OUT/${TAG}_pmc_traffic_raw.json"))
d["kernel_source_sha16"] = hashlib.sha256(open("$ROOT/poroelasticity_dealii_amd/csrc/kernels_kron.hip", "rb").read()).hexdigest()[:16]
d["how"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/bench_ops.py 3,72,2,mf 3,99,1,mf; read bytes = 2 x FETCH_SIZE x 1024 (gfx950), tools/pmc_summary.py"
json.dump(d, open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
PY
cp $OUT/${TAG}_pmc_traffic.json $ROOT/profiles/${TAG}_pmc_traffic.json
echo "[1] operator traffic done"
# 2. block-FDM transform passes: kernel stats, traffic, SQ counters
bash $ROOT/tools/kernel_counters.sh fdmo_final k_fdmo_pass tools/fdmu_bench.py 3 72 2 > /dev/null 2>&1
cp $ROOT/gpurun_out/counters_fdmo_final/summary.txt $OUT/${TAG}_fdmo_counters.txt
python3 - <<PY
import re, json, hashlib
txt = open("$OUT/${TAG}_fdmo_counters.txt").read()
rd = [float(x) for x in re.findall(r"HBM-side traffic: read ([0-9.]+) MB", txt)]; wr = [float(x) for x in re.findall(r"write ([0-9.]+) MB", txt)]
rec = {"kernels": "k_fdmo_pass<5, 0 | 1 | 2> at 72^3 Q2 (tools/fdmu_bench.py 3 72 2)", "read_MB_by_pass": rd, "write_MB_by_pass": wr,
       "hbm_bytes_per_launch_mean": 1e6 * (sum(rd) + sum(wr)) / max(len(rd), 1),
       "kernel_source_sha16": hashlib.sha256(open("$ROOT/poroelasticity_dealii_amd/csrc/kernels_fdmo.hip", "rb").read()).hexdigest()[:16],
       "how": "tools/kernel_counters.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; read bytes = 2 x FETCH_SIZE x 1024 (gfx950)"}
json.dump(rec, open("$OUT/${TAG}_fdmo_pmc_traffic.json", "w"), indent=1)
PY
cp $OUT/${TAG}_fdmo_pmc_traffic.json $ROOT/profiles/${TAG}_fdmo_pmc_traffic.json
echo "[2] FDM pass counters done"
# 3. the bench line as the driver runs it, then the same command under the kernel trace (statistics + idle gaps)
python3 $ROOT/bench.py --trace-out $OUT/${TAG}_config5_100steps_block_fdm.json > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_stderr.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-variants --no-cpu-baseline --config5-steps 0 --steps 20 --warmup 1 > $OUT/${TAG}_bench_line_profiled.json 2>> $OUT/${TAG}_bench_stderr.log || exit 1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-variants --no-cpu-baseline --config5-steps 0 --no-kernel-events > /dev/null 2>&1
python3 $ROOT/tools/gap_analysis.py $(ls $OUT/ktrace/*/*kernel_trace.csv | head -1) 20 > $OUT/${TAG}_gpu_idle_gaps.txt
echo "[3] bench lines done"
# 4. other BASELINE configs
python3 $ROOT/bench.py --dim 3 --degree 1 --cells 99 --no-cpu-baseline --config5-steps 20 > $OUT/${TAG}_bench_line_c3.json 2>/dev/null
python3 $ROOT/bench.py --dim 2 --cells 336 --no-cpu-baseline --config5-steps 20 > $OUT/${TAG}_bench_line_c2.json 2>/dev/null
echo "[4] configs 2, 3 done"
# 5. general matrix-free kernel without the box tag: 1.17 M and 9.1 M dofs (the kernel source did not change after these were taken: only if asked for)
if [ -n "$WITH_MFG" ]; then
python3 $ROOT/tools/mfg_bench.py 36 2 > $OUT/${TAG}_mfg_bench_36.json 2>/dev/null
python3 $ROOT/tools/mfg_bench.py 72 2 > $OUT/${TAG}_mfg_bench_72.json 2>/dev/null
REPS=10 bash $ROOT/tools/kernel_counters.sh mfg_72 k_mfg3_sf tools/mfg_bench.py 72 2 > /dev/null 2>&1
cp $ROOT/gpurun_out/counters_mfg_72/summary.txt $OUT/${TAG}_mfg_counters_72.txt
fi
echo "[5] general kernel done"
# 6. the partitioned code path on one RCCL rank next to the single-rank path (+ its kernel statistics), the rank-thread rehearsals of config 4's partitions
python3 $ROOT/tools/partitioned_path_1rank.py 72 noforce > $OUT/${TAG}_partitioned_path_1rank.txt 2>&1
python3 $ROOT/tools/partitioned_path_1rank.py 72 force >> $OUT/${TAG}_partitioned_path_1rank.txt 2>&1
bash $ROOT/tools/slab_probe.sh 72 > $OUT/${TAG}_partitioned_path_1rank_kernels.txt 2>&1
python3 $ROOT/tools/rank_threads.py --ranks 8 --cells 72 --steps 3 --json $OUT/${TAG}_rank_threads_8x9_block_fdm.json > /dev/null 2>&1
python3 $ROOT/tools/rank_threads.py --ranks 2 --cells 72 --steps 3 --json $OUT/${TAG}_rank_threads_2x36_block_fdm.json > /dev/null 2>&1
echo "[6] partitioned path done"
# 7. SQ counters of the structured operator
bash $ROOT/tools/sq_counters.sh $TAG > /dev/null 2>&1 && cp $ROOT/gpurun_out/sq_$TAG/summary.txt $OUT/${TAG}_sq_counters_kron3.txt
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/ktrace
ls -la $OUT
