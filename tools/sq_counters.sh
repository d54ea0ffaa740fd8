#!/bin/bash
# SQ counters of the structured operator kernels (one rocprofv3 --pmc pass per group, tools/bench_ops.py as the workload).  Output: gpurun_out/sq_<tag>/summary.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$ROOT/gpurun_out/sq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_FMA_F64" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/bench_ops.py 3,72,2,mf > $OUT/p$i.log 2>&1 || exit 1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "kron3" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]): print("  %-28s %16.1f per launch (n=%d)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
PY
rm -rf $OUT/p?
cat $OUT/summary.txt
