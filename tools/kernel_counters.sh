#!/bin/bash
# Per-kernel evidence for one kernel family (runs on the GPU box): rocprofv3 kernel-trace statistics, HBM traffic (FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes, as MI355X_MICROARCH.md prescribes; read bytes = 2 x FETCH_SIZE x 1024 on gfx950) and SQ counters (one --pmc pass per group, never together
# with a trace domain).  Usage: tools/kernel_counters.sh <tag> <kernel-name-substring> <workload.py> [args...]     Output: gpurun_out/counters_<tag>/summary.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; PAT=$2; WL=$3; shift 3
case "$WL" in /*) ;; *) WL=$ROOT/$WL ;; esac
set -- "$WL" "$@"
OUT=$ROOT/gpurun_out/counters_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 "$@" > $OUT/workload.log 2>&1 || { tail -5 $OUT/workload.log; exit 1; }
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i ($grp) failed" >> $OUT/failed.txt; tail -3 $OUT/p$i.log >> $OUT/failed.txt; }
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
pat = "$PAT"
def short(k): return k.replace("(anonymous namespace)::", "").replace("void ", "").replace("poro::", "").split("(")[0]
print("== kernel-trace statistics (rocprofv3 --kernel-trace --stats) ==")
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if pat in r["Name"]: print("%-44s calls %5s  avg %9.1f us  min %9.1f  max %9.1f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("== counters, mean per launch ==")
for k in sorted(acc):
    print(k)
    a = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    for c in sorted(a): print("  %-32s %16.1f (n=%d)" % (c, a[c], cnt[k][c]))
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        print("  -> HBM-side traffic: read %.1f MB (2 x FETCH_SIZE KiB), write %.1f MB" % (2 * 1024 * a["FETCH_SIZE"] / 1e6, 1024 * a["WRITE_SIZE"] / 1e6))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CU_CYCLES" in a:
        print("  -> MFMA pipe busy / CU busy cycles: %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_BUSY_CU_CYCLES"], 1.0)))
    if "TCC_HIT_sum" in a: print("  -> L2 hit rate %.3f" % (a["TCC_HIT_sum"] / max(a["TCC_HIT_sum"] + a.get("TCC_MISS_sum", 0.0), 1.0)))
PY
rm -rf $OUT/p? $OUT/stats
cat $OUT/summary.txt; [ -f $OUT/failed.txt ] && cat $OUT/failed.txt; true
