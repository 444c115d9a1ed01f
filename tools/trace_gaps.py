"""From a rocprofv3 --kernel-trace CSV of back-to-back N = 4096 updates: per update, the duration of the resident chain kernel and the
largest idle gap on the main stream between two consecutive kernels of that update.  usage: trace_gaps.py KERNEL_TRACE.csv"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id")) for r in rows)
main_stream = next(st for s, e, n, st in ks if "potrf_follow_kernel" in n)
upd = []                                                     # (chain duration, [main-stream kernels inside])
chains = [(s, e) for s, e, n, st in ks if "potrf_chain_kernel" in n]
main = [(s, e, n) for s, e, n, st in ks if st == main_stream]
gaps_all, durs = [], []
for cs, ce in chains:
    inside = [(s, e, n) for s, e, n in main if s >= cs - 200000 and e <= ce + 100000 and ("potrf_" in n or "gram" in n or "scale_points" in n)]
    inside.sort()
    g = [(inside[i + 1][0] - inside[i][1]) / 1e3 for i in range(len(inside) - 1)]
    gaps_all.append(max(g) if g else 0.0)
    durs.append((ce - cs) / 1e3)
durs, gaps_all = np.array(durs), np.array(gaps_all)
print(f"updates {len(durs)}: chain kernel duration p50 {np.median(durs):.1f} us, p99 {np.percentile(durs, 99):.1f}, max {durs.max():.1f}; "
      f"largest main-stream gap inside an update: p50 {np.median(gaps_all):.1f} us, max {gaps_all.max():.1f} us; updates with a gap > 50 us: {(gaps_all > 50).sum()}")
worst = np.argsort(-durs)[:5]
print("slowest updates (index, chain us, largest gap us):", [(int(i), round(float(durs[i]), 1), round(float(gaps_all[i]), 1)) for i in worst])
