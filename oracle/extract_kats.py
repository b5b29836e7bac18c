#!/usr/bin/env python3
"""DEV-TIME: transcribe the reference's own known-answer DATA (inputs + expected outputs) from
wasmcurves/test/batchAffine.js into tests/golden/kats_batchaffine.json.  Only numbers are taken
(points, scalars, expected values); no reference code is copied."""
import json
import os
import re

REF = "/root/reference/wasmcurves/test/batchAffine.js"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "kats_batchaffine.json")

lines = open(REF).read().split("\n")


def hexints(s):
    return [int(h, 16) for h in re.findall(r"0x([0-9a-fA-F]+)n?", s)]


# multiExp end-to-end KAT: batchAffine.js:1177-1255 (points :1179-1188, scalars :1190-1193, expected :1195)
pts = hexints("\n".join(lines[1178:1189]))
scal = hexints("\n".join(lines[1189:1193]))
exp = hexints(lines[1194])
assert len(pts) == 20 and len(scal) == 10 and len(exp) == 2, (len(pts), len(scal), len(exp))

# getChunk KAT: batchAffine.js:116-155
gc_scalars = hexints("\n".join(lines[116:118]))
gc_chunk = [int(x) for x in re.findall(r"\d+", lines[118].split("=")[1])]
gc_ssize = [int(x) for x in re.findall(r"\d+", lines[119].split("=")[1])]
gc_nch = [int(x) for x in re.findall(r"\d+", lines[120].split("=")[1])]
gc_exp = [[int(x) for x in re.findall(r"\d+", l)] for l in lines[122:134]]
assert len(gc_scalars) == 12 and len(gc_exp) == 12

out = {
    "multiexp_case1": {
        "source": "wasmcurves/test/batchAffine.js:1177-1255",
        "curve": "bls12381", "group": 1, "scalar_size": 32,
        "points_affine_plain": [hex(p) for p in pts],      # x0,y0,x1,y1,... NOT Montgomery
        "scalars": [hex(s) for s in scal],
        "expected_affine_plain": [hex(e) for e in exp],
    },
    "get_chunk": {
        "source": "wasmcurves/test/batchAffine.js:116-155",
        "scalars": [hex(s) for s in gc_scalars], "chunk_size": gc_chunk, "scalar_size": gc_ssize,
        "num_chunks": gc_nch, "expected": gc_exp,
    },
}
json.dump(out, open(OUT, "w"), indent=1)
print("wrote", OUT)
