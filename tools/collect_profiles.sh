#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel stats of the default bench run, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, as the
# guide prescribes) of the structured operator kernels and of the bench run, and the CSR / assembly micro-benchmarks.  Output: gpurun_out/profiles_r02/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/profiles_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02}
# PMC traffic first: bench.py quotes roofline.traffic from profiles/r02_pmc_traffic.json only while its stamp matches the kernel source, so the fresh file is installed before the bench runs
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $ROOT/tools/bench_ops.py 3,72,2,mf 3,99,1,mf > $OUT/${TAG}_ops_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $ROOT/tools/bench_ops.py 3,72,2,mf 3,99,1,mf > $OUT/${TAG}_ops_write.log 2>&1 || exit 1
FD=$(dirname $(ls $OUT/pmc_fetch_$TAG/*/*counter_collection.csv | head -1)); WD=$(dirname $(ls $OUT/pmc_write_$TAG/*/*counter_collection.csv | head -1))
python3 $ROOT/tools/pmc_summary.py $FD $WD $OUT/${TAG}_pmc_traffic_raw.json > $OUT/${TAG}_pmc_traffic.txt || exit 1
python3 - <<PY
import json, hashlib
d = json.load(open("$OUT/${TAG}_pmc_traffic_raw.json"))
d["kernel_source_sha16"] = hashlib.sha256(open("$ROOT/poroelasticity_dealii_amd/csrc/kernels_kron.hip", "rb").read()).hexdigest()[:16]
d["how"] = "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/bench_ops.py 3,72,2,mf 3,99,1,mf; read bytes = 2 x FETCH_SIZE x 1024 (gfx950), tools/pmc_summary.py"
json.dump(d, open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
PY
cp $OUT/${TAG}_pmc_traffic.json $ROOT/profiles/${TAG}_pmc_traffic.json
# the bench line as the driver runs it (no profiler attached: HIP-event timings are inflated by ~5 % under rocprofv3)
python3 $ROOT/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_stderr.log || exit 1
# kernel statistics of the headline run alone, many steady steps so that the few early-exit launches of the first solves do not weigh on the averages
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -- python3 $ROOT/bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 1 > $OUT/${TAG}_bench_line_profiled.json 2>> $OUT/${TAG}_bench_stderr.log || exit 1
cp $(ls $OUT/stats_$TAG/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
python3 - <<PY
import csv, glob, json, statistics
t = sorted(glob.glob("$OUT/stats_$TAG/*/*kernel_trace.csv"))[-1]
out = {}
for key in ("k_kron3_q2_cheb", "k_kron3_q2(", "k_pcg_update_g_fused", "k_pcg_update_d_fused"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t)) if key in r["Kernel_Name"]]
    real = [x for x in d if x > 20.0]
    if d:
        out[key.rstrip("(")] = {"launches": len(d), "early_exit_launches": len(d) - len(real), "mean_us_all": statistics.mean(d), "mean_us_without_early_exits": statistics.mean(real), "median_us": statistics.median(d)}
json.dump(out, open("$OUT/${TAG}_bench_kernel_durations.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
PORO_DIAG_SKIP_SELFCHECK=1 python3 $ROOT/tools/asm_bench.py 3,32,2 3,48,2 3,72,2 2,336,2 > $OUT/${TAG}_csr_asm_spmv.jsonl 2> $OUT/${TAG}_csr_asm_stderr.log
python3 $ROOT/tools/fdmu_bench.py > $OUT/${TAG}_fdmu_apply.txt 2>&1
python3 $ROOT/bench.py --dim 3 --degree 1 --cells 99 --no-cpu-baseline > $OUT/${TAG}_bench_line_c3.json 2>/dev/null
python3 $ROOT/bench.py --dim 2 --cells 336 --no-cpu-baseline > $OUT/${TAG}_bench_line_c2.json 2>/dev/null
# where the GPU idles inside a step (host round trips): kernel trace of the headline run without per-dispatch events
rocprofv3 --kernel-trace --output-format csv -d $OUT/ktrace_$TAG -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-variants --no-weak-line --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
python3 $ROOT/tools/gap_analysis.py $(ls $OUT/ktrace_$TAG/*/*kernel_trace.csv | head -1) 4 > $OUT/${TAG}_gpu_idle_gaps.txt
python3 $ROOT/tools/step_times.py 8 > $OUT/${TAG}_step_times.txt 2>/dev/null
bash $ROOT/tools/sq_counters.sh $TAG > /dev/null 2>&1 && cp $ROOT/gpurun_out/sq_$TAG/summary.txt $OUT/${TAG}_sq_counters_kron3.txt
rm -rf $OUT/stats_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/ktrace_$TAG
ls -la $OUT
