"""Idle time on the GPU inside the bench's steps: reads a rocprofv3 kernel trace (csv), orders the dispatches by start time and lists where the stream sat idle
(end of one kernel -> start of the next), grouped by the kernel that FOLLOWS the gap.  Usage: python tools/gap_analysis.py <kernel_trace.csv> [min_gap_us]"""
import collections, csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 4e3
# the steady part: from the last third of the trace
rows = rows[len(rows) // 3:]
busy = sum(e - s for s, e, _ in rows); span = rows[-1][1] - rows[0][0]
gaps = collections.defaultdict(lambda: [0, 0.0]); prev_end = rows[0][1]; prev_name = rows[0][2]
small = 0.0
for s, e, name in rows[1:]:
    g = s - prev_end
    if g > min_gap:
        k = prev_name + "  ->  " + name; gaps[k][0] += 1; gaps[k][1] += g
    elif g > 0:
        small += g
    prev_end = max(prev_end, e); prev_name = name
print(f"span {span/1e6:.2f} ms, kernels busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), gaps < {min_gap/1e3:.0f} us: {small/1e6:.2f} ms in total")
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:8.3f} ms  {n:5d} x {t/n/1e3:7.1f} us   {k}")
