#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV of a pipelined bench run: per stream (queue) busy fractions, per-call critical path, and the
average per-call sequence of kernels with start offsets relative to the call's first kernel.  usage: trace_timeline.py kernel_trace.csv [skip_calls]"""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].replace("void ", "").replace("msm::", ""), r["Queue_Id"], r["Stream_Id"],
      int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
K.sort()
# the timed region: from the (skip)-th k_tile_hist to the last one
hists = [k for k in K if k[2] == "k_tile_hist"]
t_lo, t_hi = hists[skip][0], hists[-1][0]
ncalls = len(hists) - 1 - skip
print(f"{ncalls} calls in {(t_hi - t_lo) / 1e6:.2f} ms: {(t_hi - t_lo) / 1e3 / ncalls:.1f} us per call")
W = [k for k in K if t_lo <= k[0] < t_hi]
by_q = collections.defaultdict(list)
for k in W:
    by_q[(k[3], k[4])].append(k)
for q, ks in sorted(by_q.items()):
    busy = sum(e - s for s, e, *_ in ks)
    names = collections.Counter(k[2] for k in ks)
    print(f"queue/stream {q}: busy {busy / (t_hi - t_lo) * 100:.1f} % of the region; " + ", ".join(f"{n} x{c}" for n, c in names.most_common(6)))
# union busy (any kernel running)
ev = sorted([(s, 1) for s, e, *_ in W] + [(e, -1) for s, e, *_ in W])
depth, last, idle, hist = 0, t_lo, 0, collections.Counter()
for t, d in ev:
    if depth == 0:
        idle += t - last
    hist[depth] += t - last
    last = t
    depth += d
print(f"GPU idle (no kernel at all) {idle / (t_hi - t_lo) * 100:.1f} %; time by number of kernels running: " + ", ".join(f"{k}: {v / (t_hi - t_lo) * 100:.0f} %" for k, v in sorted(hist.items())))
# per kernel: average duration, average grid (workgroups)
agg = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n, q, st, wg in W:
    a = agg[n]
    a[0] += 1
    a[1] += e - s
    a[2] += wg
for n, (c, d, wg) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"  {n:28s} x{c:5d}  avg {d / c / 1e3:8.1f} us  workgroups {wg // c:7d}  share of region x streams {d / (t_hi - t_lo) * 100:6.1f} %")
