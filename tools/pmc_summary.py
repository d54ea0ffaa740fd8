"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-launch HBM traffic per kernel.
gfx950 corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; FETCH_SIZE reports 1/2 of coalesced streaming reads
(calibrated here on k_axpy: reads 2 vectors, writes 1) -> read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
Usage: python tools/pmc_summary.py <dir_FETCH> <dir_WRITE> <out.json>"""
import collections, csv, glob, json, sys

def load(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fe:
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    f, w = fe[k], wr.get(k, [0.0])
    out[short] = {"launches": len(f), "FETCH_SIZE_KiB_mean": sum(f) / len(f), "WRITE_SIZE_KiB_mean": sum(w) / len(w),
                  "read_bytes_corrected": 2 * 1024 * sum(f) / len(f), "write_bytes": 1024 * sum(w) / len(w),
                  "hbm_bytes_per_launch": 2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:40s} n={v['launches']:4d} read={v['read_bytes_corrected']/1e6:9.2f} MB write={v['write_bytes']/1e6:9.2f} MB")
