#!/usr/bin/env python3
"""Instruction mix of the hot loop of a kernel, from its disassembly (round 4: where BN254's issue slots go).
usage: loop_mix.py file.s <substring of the kernel's mangled name>
Finds the kernel, its largest backward-branch loop, and inside it the straight-line path that is taken when no rare branch is (every
forward branch whose target lies beyond a `long` distance is taken to guard a rare block -- the doubling case, infinity -- and the block
it skips is NOT counted when the branch is conditional on exec/vcc being zero).  Prints the mnemonic histogram of the loop, grouped."""
import collections
import re
import sys

text = open(sys.argv[1]).read().splitlines()
want = sys.argv[2]
start = next(i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <.*" + re.escape(want), l))
end = next((i for i in range(start + 1, len(text)) if re.match(r"^[0-9a-f]+ <_Z", text[i])), len(text))
ins = []   # (addr, mnemonic, branch target or None)
for l in text[start + 1:end]:
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", l)
    if m:
        tm = re.search(r"\+0x([0-9a-f]+)>\s*$", m.group(4))
        ins.append((int(m.group(3), 16), m.group(1), int(tm.group(1), 16) if tm else None))
base = ins[0][0]
addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
loops = []
for i, (a, mn, tgt) in enumerate(ins):
    if (mn.startswith("s_cbranch") or mn == "s_branch") and tgt is not None:
        t = base + tgt
        if t in addr_index and addr_index[t] < i:
            loops.append((i - addr_index[t], addr_index[t], i))
loops.sort(reverse=True)
if not loops:
    print("no loop found")
    sys.exit(1)
n, lo, hi = loops[0]
body = ins[lo:hi + 1]
hist = collections.Counter(mn for _, mn, _ in body)


def grp(mn):
    if mn.startswith("v_mad_u64_u32"):
        return "v_mad_u64_u32 (multiply-add)"
    if mn.startswith("v_mul_lo_u32") or mn.startswith("v_mul_hi_u32") or mn.startswith("v_mul_u32"):
        return "v_mul (the m of the reduction)"
    if mn.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_alignbit", "v_lshrrev_b32", "v_lshlrev_b32", "v_lshl_", "v_ashr")):
        return "shifts"
    if mn.startswith(("v_and_b32", "v_bfe_", "v_and_or", "v_or_b32", "v_bfi", "v_xor", "v_or3", "v_xad")):
        return "masks / bit ops"
    if mn.startswith(("v_add", "v_sub", "v_addc", "v_subb", "v_add3", "v_lshl_add", "v_add_lshl")):
        return "adds / carries"
    if mn.startswith(("v_cmp", "v_cndmask")):
        return "compares / selects"
    if mn.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_perm", "v_swap")):
        return "moves"
    if mn.startswith("v_"):
        return "other VALU"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vector memory"
    if mn.startswith("ds_"):
        return "LDS"
    if mn.startswith("s_nop"):
        return "s_nop"
    if mn.startswith("s_waitcnt"):
        return "s_waitcnt"
    return "scalar / branch"


g = collections.Counter()
for mn, c in hist.items():
    g[grp(mn)] += c
valu = sum(c for k, c in g.items() if k not in ("vector memory", "LDS", "s_nop", "s_waitcnt", "scalar / branch"))
print(f"kernel {want}: {len(ins)} instructions; largest loop: {len(body)} instructions ({valu} VALU)")
for k, c in g.most_common():
    print(f"  {k:34s} {c:6d}  {c / valu * 100:5.1f} % of the loop's VALU" if k not in ("vector memory", "LDS", "s_nop", "s_waitcnt", "scalar / branch") else f"  {k:34s} {c:6d}")
print("  top mnemonics:", ", ".join(f"{m} x{c}" for m, c in hist.most_common(14)))
