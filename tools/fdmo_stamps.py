"""Reads the per-block time stamps of the octant fast-diagonalisation passes (PORO_FDMO_STAMPS=<file>, kernels_fdmo.hip) and prints, per pass: the phase durations of a block
(load -> LDS, GEMM 1, intermediate -> LDS, GEMM 2, store), how many blocks are resident per CU over time and the launch span.  100 MHz clock."""
import sys, collections
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
for p in range(3):
    R = [r for r in rows if r[0] == p]
    if not R: continue
    t0 = min(r[2] for r in R); t1 = max(r[7] for r in R)
    ph = [[(r[3 + k] - r[2 + k]) / 100.0 for r in R] for k in range(5)]
    names = ["load->LDS", "GEMM 1", "mid->LDS", "GEMM 2", "store"]
    print(f"pass {p + 1}: {len(R)} blocks, span {(t1 - t0) / 100.0:.1f} us; block life mean {sum((r[7] - r[2]) for r in R) / len(R) / 100.0:.1f} us")
    for n, v in zip(names, ph): v.sort(); print(f"   {n:10s} mean {sum(v) / len(v):6.2f} us  median {v[len(v) // 2]:6.2f}  p90 {v[int(0.9 * len(v))]:6.2f}")
    # residency per CU: key = xcc, se, sh, cu  (HW_ID: cu 11:8, sh 12, se 15:13)
    ev = collections.defaultdict(list)
    for r in R:
        key = (r[9] & 0xf, (r[8] >> 8) & 0xff)
        ev[key].append((r[2], 1)); ev[key].append((r[7], -1))
    tot = 0.0; mx = 0
    for k, v in ev.items():
        v.sort(); cur = 0; last = t0; area = 0
        for t, d in v: area += cur * (t - last); last = t; cur += d; mx = max(mx, cur)
        tot += area / max(t1 - t0, 1)
    print(f"   {len(ev)} CUs seen; mean resident blocks per CU over the span {tot / len(ev):.2f}, max {mx}")
    starts = sorted(r[2] - t0 for r in R)
    print("   block starts (us) at 10/25/50/75/90 %:", [round(starts[int(q * len(starts))] / 100.0, 1) for q in (0.1, 0.25, 0.5, 0.75, 0.9)])
