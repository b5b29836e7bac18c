#!/usr/bin/env python3
"""Post-link guard on the gfx950 code objects inside libmsm_amd.so (called by __graft_entry__.build_engine).

Why: twice in round 1 a build whose NON-KERNEL device functions had grown past the reach of a direct branch (s_branch / s_cbranch
take a signed 16-bit dword offset: +-128 KiB) hung the GPU on its first BLS12-381 G2 call.  Such a function is
the only place where the compiler's long-branch expansion (s_getpc_b64 / s_add_u32 / s_addc_u32 / s_setpc_b64 on a scavenged SGPR
pair) appears inside a callee.  The build therefore fails when
  * a non-kernel FUNC symbol of any embedded code object is larger than LIMIT bytes (default 100 KiB: the largest shipped function is 89 KB, 123 KB is known to run and 136 KB to hang), or
  * a non-kernel function contains an s_setpc_b64 through anything but the return address s[30:31] (a relaxed long branch),
so that the next inlining change cannot silently bring the hang back.  Kernels are exempt (500-KiB kernels run fine).

Round 4, a third rule: the PRIVATE SEGMENT of the MSM path's kernels.  The HSA runtime gives a hardware queue 64 x 8192 x (bytes of
private segment per lane) of scratch when a kernel that needs it is dispatched, and aborts the process when that allocation fails on
a nearly full device.  msm_amd_create reserves MSM_SCRATCH_RESERVE (csrc/engine_ctx.hpp) bytes per lane on every stream while failure
can still be reported; every kernel that an MSM call may launch must stay within that -- the build fails when one of them does not.
Kernels of the other entry points (group FFTs, batch conversions, batchApplyKey, the table build, point generation) are exempt BY NAME:
they call out-of-line point arithmetic, and their launch sites ask msm_scratch_guard first.

Usage: check_code_objects.py path/to/lib.so [--limit BYTES] [--list] [--sources csrc_dir]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """Every amdgcn ELF image of every offload bundle in the file (one bundle per translation unit)."""
    data = open(path, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        (nb,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(nb):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, data[pos + off:pos + off + size]))
        pos = data.find(MAGIC, pos + 1)
    return out


def inspect(elf_bytes):
    """[(name, size, is_kernel, long_jumps)] for the FUNC symbols of one code object"""
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(elf_bytes)
        tmp = f.name
    try:
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", tmp], capture_output=True, text=True, check=True).stdout
        funcs, kds = {}, set()
        for ln in syms.splitlines():
            t = ln.split()
            if len(t) >= 8 and t[3] == "FUNC":
                funcs[t[7]] = max(funcs.get(t[7], 0), int(t[2]))
            elif len(t) >= 8 and t[3] == "OBJECT" and t[7].endswith(".kd"):
                kds.add(t[7][:-3])
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", tmp], capture_output=True, text=True, check=True).stdout
        jumps, cur = {}, None
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
            if m:
                cur = m.group(1)
            elif cur and "s_setpc_b64" in ln and "s[30:31]" not in ln:
                jumps[cur] = jumps.get(cur, 0) + 1
        return [(n, s, n in kds, jumps.get(n, 0)) for n, s in funcs.items()]
    finally:
        os.unlink(tmp)


GUARDED_AT_RUNTIME = ("k_fft_", "k_g_fft_", "k_fr_fft", "k_batch_", "k_fr_batch", "k_gen_points", "k_table_next", "k_glv_preprocess")


def private_segments(elf_bytes):
    """[(kernel name, private segment bytes per lane)] from the code object's metadata notes"""
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(elf_bytes)
        tmp = f.name
    try:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", tmp], capture_output=True, text=True, check=True).stdout
    finally:
        os.unlink(tmp)
    out, name, priv = [], None, None
    for ln in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "name" and name is None and v.startswith("_Z") or (k == "name" and name is None and v.startswith("k_")):
            name = v
        elif k == "private_segment_fixed_size":
            priv = int(v)
        elif k == "wavefront_size":      # (the last key of a kernel's record)
            if name is not None and priv is not None:
                out.append((name, priv))
            name, priv = None, None
    return out


def scratch_reserve(src_dir):
    text = open(os.path.join(src_dir, "engine_ctx.hpp")).read()
    m = re.search(r"MSM_SCRATCH_RESERVE\s*=\s*(\d+)", text)
    return int(m.group(1)) if m else None


def byvalue_points(src_dir):
    """Source scan: an out-of-line (__noinline__) device function must not take a point (XYZZ / XYZZU / Jacobian / Affine struct: up to
    448 bytes) BY VALUE -- two 384-byte by-value operands arrived corrupted in the callee.  Field-element operands
    are bounded by static_asserts against MSM_BYVALUE_MAX next to the functions themselves."""
    bad = []
    for fn in sorted(os.listdir(src_dir)):
        if not fn.endswith((".hpp", ".hip")):
            continue
        text = open(os.path.join(src_dir, fn)).read()
        for m in re.finditer(r"__noinline__[^;{]*?\(([^;{]*?)\)\s*[{;]", text, re.S):
            for prm in m.group(1).split(","):
                prm = prm.strip()
                if re.match(r"^(const\s+)?(XYZZU|XYZZ|Jacobian|Affine)\b(<[^>]*>)?\s+\w+$", prm):
                    line = text.count("\n", 0, m.start()) + 1
                    bad.append(f"{fn}:{line}: out-of-line device function takes a point by value: '{prm}'")
    return bad


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    limit = 100 * 1024
    if "--limit" in argv:
        limit = int(argv[argv.index("--limit") + 1])
    objs = code_objects(argv[1])
    if not objs:
        print(f"check_code_objects: no amdgcn code object found in {argv[1]}", file=sys.stderr)
        return 1
    bad, rows = [], []
    for triple, blob in objs:
        for name, size, is_kernel, lj in inspect(blob):
            rows.append((size, is_kernel, lj, name))
            if not is_kernel and size > limit:
                bad.append(f"device function of {size} bytes (> {limit}): {name}")
            if not is_kernel and lj:
                bad.append(f"device function with {lj} relaxed long branch(es) (s_setpc_b64 as a jump): {name}")
    if "--list" in argv:
        for size, is_kernel, lj, name in sorted(rows, reverse=True)[:40]:
            print(f"{size:9d}  {'kernel  ' if is_kernel else 'function'}  long-jumps={lj}  {name[:140]}")
    nk = [r for r in rows if not r[1]]
    print(f"check_code_objects: {len(objs)} code objects, {len(rows) - len(nk)} kernels (largest {max([r[0] for r in rows if r[1]] or [0])} B), "
          f"{len(nk)} device functions (largest {max([r[0] for r in nk] or [0])} B, limit {limit} B)")
    if "--sources" in argv:
        src = argv[argv.index("--sources") + 1]
        bad += byvalue_points(src)
        reserve = scratch_reserve(src)
        if reserve is None:
            bad.append("MSM_SCRATCH_RESERVE not found in engine_ctx.hpp")
        else:
            worst, reserve_kernel = (0, ""), None
            for triple, blob in objs:
                for name, priv in private_segments(blob):
                    demangled = name
                    if "k_scratch_reserve" in name:      # the kernel msm_amd_create reserves the scratch WITH: at least the reserve
                        reserve_kernel = priv
                        continue
                    if any(("3msm%d%s" % (len(pfx) + k, pfx)) in name or name.startswith(pfx) for pfx in GUARDED_AT_RUNTIME for k in range(0, 24)):
                        continue
                    worst = max(worst, (priv, demangled))
                    if priv > reserve:
                        bad.append(f"kernel of the MSM path with {priv} bytes of private segment per lane (> MSM_SCRATCH_RESERVE = {reserve}): {name}")
            if reserve_kernel is None or reserve_kernel < reserve:
                bad.append(f"k_scratch_reserve carries {reserve_kernel} bytes of private segment, less than MSM_SCRATCH_RESERVE = {reserve}")
            print(f"check_code_objects: largest private segment among the MSM path's kernels: {worst[0]} B per lane (reserved at create: {reserve_kernel} B >= {reserve})")
    for b in bad:
        print("check_code_objects: FAIL:", b, file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
