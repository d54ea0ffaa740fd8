"""average duration per kernel of a rocprofv3 kernel trace (csv): python tools/kernel_avgs.py trace.csv [n]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1]))); n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("poro::", "").split("(")[0]
    a = acc[k]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:n]:
    print("%-40s %6d calls avg %7.1f us" % (k[:40], a[0], a[1] / a[0]))
