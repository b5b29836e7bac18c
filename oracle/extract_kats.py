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



# Stage-level KATs of the opt path re-expressed as MSMs: batchAffine.js:884-977 (multiExpSingleChunk), :979-1056 and
# :1058-1175 (multiExpChunks case 1 / 2).  Their inputs are 10 affine points and per-chunk "point schedules"
# (entry = pointIdx << 32 | bucket digit, 0xFFFF...F = digit 0); the value they compute is
# sum_i (sum_chunks digit[chunk][i] * 2^(chunkSize*chunk)) * P_i, so points + derived scalars + their expected
# affine output are one more known answer for multiexpAffine.
def block(title):
    start = next(i for i, l in enumerate(lines) if title in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("    it(") or lines[i].startswith("    // it("))
    return start, end


def array_after(start, end, name):
    i = next(k for k in range(start, end) if ("let " + name + " =") in lines[k])
    depth, buf = 0, []
    for k in range(i, end):
        buf.append(lines[k])
        depth += lines[k].count("[") - lines[k].count("]")
        if depth == 0 and "[" in "".join(buf):
            break
    return "\n".join(buf)


def sched_kat(title, src):
    a, b = block(title)
    pts_ = hexints(array_after(a, b, "inputPoints"))
    sched_txt = array_after(a, b, "pointSchedules")
    rows = re.findall(r"\[([^\[\]]*)\]", sched_txt[sched_txt.index("=") + 1:])
    if len(rows) > 1 and not rows[0].strip():
        rows = rows[1:]
    rows = [hexints(r) for r in rows if r.strip()] or [hexints(sched_txt)]
    cs = int(re.search(r"let chunkSize = (\d+);", "\n".join(lines[a:b])).group(1))
    exp_ = hexints(array_after(a, b, "expectedOutput"))
    assert len(pts_) == 20 and len(exp_) == 2 and all(len(r) == 10 for r in rows), (len(pts_), len(exp_), [len(r) for r in rows])
    scalars = [0] * 10
    for chunk, row in enumerate(rows):
        for e in row:
            if e == 0xFFFFFFFFFFFFFFFF:
                continue
            scalars[e >> 32] += (e & 0xFFFFFFFF) << (cs * chunk)
    return {"source": src, "curve": "bls12381", "group": 1, "scalar_size": 32, "chunk_size": cs, "num_chunks": len(rows),
            "points_affine_plain": [hex(p) for p in pts_], "scalars": [hex(x) for x in scalars], "expected_affine_plain": [hex(e) for e in exp_]}


sched = {
    "single_chunk": sched_kat('it("multiExpSingleChunk is correct."', "wasmcurves/test/batchAffine.js:884-977"),
    "chunks_case1": sched_kat('it("multiExpChunks is correct (case 1)."', "wasmcurves/test/batchAffine.js:979-1056"),
    "chunks_case2": sched_kat('it("multiExpChunks is correct (case 2)."', "wasmcurves/test/batchAffine.js:1058-1175"),
}

out = {
    "scheduled": sched,
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
