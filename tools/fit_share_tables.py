#!/usr/bin/env python3
"""kShareMs tables of csrc/dist.hip from the per-share measurements of tools/shard_table.py (its .jsonl output: the last record holds
every (points per rank, window groups) share with 2 / 3 / 4 calls in flight).  A cell = the best of the three depths; cells the sweep
did not visit are the BLS12-381 G1 table's cell scaled by the ratio of the nearest measured cell of the same row (else of the nearest
row).  usage: fit_share_tables.py NAME file.jsonl  -> a C initialiser for rows 2^13 .. 2^26, columns gw = 1, 2, 4, 8"""
import json
import math
import re
import sys

G1 = {}
src = open(__file__.replace("tools/fit_share_tables.py", "zprize-wasm-msm_amd/csrc/dist.hip")).read()
m = re.search(r"kShareMsG1\[14\]\[4\] = \{(.*?)\};", src, re.S) or re.search(r"kShareMs\[14\]\[4\] = \{(.*?)\};", src, re.S)
rows = re.findall(r"\{([^{}]*)\}", m.group(1))
for i, r in enumerate(rows):
    G1[13 + i] = [float(x.strip().rstrip("f")) for x in r.split(",") if x.strip()]

name, path = sys.argv[1], sys.argv[2]
shares = None
for ln in open(path):
    d = json.loads(ln)
    if "shares" in d:
        shares = d["shares"]
meas = {}
for s in shares:
    lp = math.log2(s["points"])
    if abs(lp - round(lp)) > 1e-9 or s["gw"] not in (1, 2, 4, 8):
        continue
    v = min(x for x in (s["ms_depth2"], s["ms_depth3"], s["ms_depth4"]) if x)
    meas[(int(round(lp)), s["gw"])] = v
print(f"static const float {name}[14][4] = {{")
for lp in range(13, 27):
    cells = []
    for j, gw in enumerate((1, 2, 4, 8)):
        if (lp, gw) in meas:
            cells.append(meas[(lp, gw)])
            continue
        # nearest measured cell: same row first, then rows outward
        ratio = None
        for dl in range(0, 14):
            for l2 in (lp - dl, lp + dl):
                cand = [(abs(math.log2(g2) - math.log2(gw)), g2) for (l3, g2) in meas if l3 == l2]
                if cand:
                    g2 = min(cand)[1]
                    ratio = meas[(l2, g2)] / G1[l2][(1, 2, 4, 8).index(g2)]
                    break
            if ratio is not None:
                break
        cells.append(G1[lp][j] * ratio)
    print("    {" + ", ".join(f"{c:.3f}f" for c in cells) + "},   // 2^" + str(lp) + ("" if all((lp, g) in meas for g in (1, 2, 4, 8)) else "  (partly scaled from the G1 table)"))
print("};")
