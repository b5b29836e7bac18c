#!/usr/bin/env python3
"""Fill and drain of the driver's form (5 warmup calls, synchronise, 20 timed calls, two in flight) from a rocprofv3 --kernel-trace CSV of
`bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline`: every kernel of the timed region longer than 15 us with its start and end
relative to the region's first kernel, the stream it ran on, and the gaps in which no kernel ran.
usage: fill_drain_trace.py kernel_trace.csv [warmup_calls] [timed_calls]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("<")[0].split("(")[0].replace("void ", "").replace("msm::", ""), r["Stream_Id"]) for r in rows)
hists = [i for i, k in enumerate(K) if k[2] == "k_tile_hist"]
# bench.py: one probe call, `warm` warmup calls, `steps` timed calls, then calls one at a time: the timed region starts at histogram 1 + warm
first = hists[1 + warm]
last_hist = hists[1 + warm + steps - 1]
nxt = hists[1 + warm + steps] if len(hists) > 1 + warm + steps else len(K)
R = K[first:nxt]
t0 = R[0][0]
end = max(e for s, e, *_ in R)
print(f"timed region on the GPU: {(end - t0) / 1e6:.3f} ms for {steps} calls = {(end - t0) / 1e6 / steps:.3f} ms per call; idle before it {(t0 - max(e for s, e, *_ in K[:first])) / 1e3:.0f} us")
call = 0
busy_to = t0
for s, e, n, st in R:
    if s > busy_to + 5000:
        print(f"      -- nothing running for {(s - busy_to) / 1e3:7.1f} us")
    busy_to = max(busy_to, e)
    if n == "k_tile_hist":
        call += 1
    if e - s >= 15000 and (call <= 4 or call >= steps - 2):
        print(f"{(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  stream {st:>3s}  {n}" + (f"   <- call {call}" if n == "k_tile_hist" else ""))
acc = [(s, e) for s, e, n, st in R if n == "k_accumulate"]
print("k_accumulate durations (ms):", " ".join(f"{(e - s) / 1e6:.2f}" for s, e in acc))
print("gaps between consecutive k_accumulate launches (us):", " ".join(f"{(b[0] - a[1]) / 1e3:.0f}" for a, b in zip(acc, acc[1:])))
