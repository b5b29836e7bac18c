#!/usr/bin/env python3
"""Generates zprize-wasm-msm_amd/csrc/fp_asm_gen.hpp: gfx950 inline-asm carry chains for the
modular add / sub / conditional-negate / final Montgomery reduction of each supported modulus.

Why generated asm: hipcc turns the portable `(uint64_t)a + b + c` limb loops into 64-bit add / shift
emulation (~70-110 VALU instructions per 12-limb modular add).  A v_add_co/v_addc_co chain on VCC is
12 instructions, the conditional correction is done under an EXEC mask (s_and_saveexec) instead of
with a select per limb, and the modulus limbs are 32-bit literals in the instruction stream.
Semantics are those of wasmcurves/src/build_f1m.js:71-117 (results fully reduced to [0, q)).
"""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
params = open(os.path.join(HERE, "..", "zprize-wasm-msm_amd", "csrc", "params_gen.hpp")).read()


def limbs_of(struct, name):
    m = re.search(r"struct %s \{.*?static constexpr uint32_t %s\[(\d+)\] = \{(.*?)\};" % (struct, name), params, re.S)
    return [int(x.strip().rstrip("u"), 16) for x in m.group(2).split(",")]


def lit(v):
    return "0x%08x" % v


def gen(struct):
    Q = limbs_of(struct, "Q")
    N = len(Q)
    H = N // 2
    o = []
    A = o.append
    A(f"template <> struct FpAsm<{struct}> {{")
    A(f"    static constexpr int N = {N};")

    # ---- reduce: x in [0, 2q) -> [0, q), in place -------------------------------------------------
    ops = ", ".join(f'"+v"(x[{i}])' for i in range(N))
    lines = []
    for i in range(N):
        if i == 0:
            lines.append(f"v_subrev_co_u32 %{i}, vcc, {lit(Q[i])}, %{i}")
        else:
            lines.append(f"v_subbrev_co_u32 %{i}, vcc, {lit(Q[i])}, %{i}, vcc")
    lines.append("s_nop 0")
    lines.append(f"s_and_saveexec_b64 %{N}, vcc")           # lanes that borrowed (x < q) get q added back
    for i in range(N):
        if i == 0:
            lines.append(f"v_add_co_u32 %{i}, vcc, {lit(Q[i])}, %{i}")
        else:
            lines.append(f"v_addc_co_u32 %{i}, vcc, {lit(Q[i])}, %{i}, vcc")
    lines.append(f"s_mov_b64 exec, %{N}")
    body = "\\n\\t".join(lines)
    A("    // x < 2q  ->  x mod q, in place (subtract q everywhere, add it back on the lanes that borrowed)")
    A(f"    static __device__ __forceinline__ void reduce(uint32_t (&x)[{N}]) {{")
    A("        uint64_t saved;")
    A(f'        asm("{body}"')
    A(f'            : {ops}, "=&s"(saved)')
    A('            :')
    A('            : "vcc");')
    A("    }")

    # ---- add_raw / sub_raw: two half chains, carry handed over in an SGPR pair ---------------------
    for name, first, rest in (("add_raw", "v_add_co_u32", "v_addc_co_u32"), ("sub_raw", "v_sub_co_u32", "v_subb_co_u32")):
        A(f"    // r = a {'+' if name == 'add_raw' else '-'} b on {N} limbs; returns the final carry/borrow as a lane mask")
        A(f"    static __device__ __forceinline__ uint64_t {name}(uint32_t (&r)[{N}], const uint32_t (&a)[{N}], const uint32_t (&b)[{N}]) {{")
        A("        uint64_t c0, c1;")
        # first half
        lines = []
        for i in range(H):
            if i == 0:
                lines.append(f"{first} %{i}, vcc, %{H + 1 + i}, %{2 * H + 1 + i}")
            else:
                lines.append(f"{rest} %{i}, vcc, %{H + 1 + i}, %{2 * H + 1 + i}, vcc")
        lines.append(f"s_mov_b64 %{H}, vcc")
        body = "\\n\\t".join(lines)
        outs = ", ".join(f'"=&v"(r[{i}])' for i in range(H)) + ', "=&s"(c0)'
        ins = ", ".join(f'"v"(a[{i}])' for i in range(H)) + ", " + ", ".join(f'"v"(b[{i}])' for i in range(H))
        A(f'        asm("{body}"')
        A(f"            : {outs}")
        A(f"            : {ins}")
        A('            : "vcc");')
        # second half
        lines = [f"s_mov_b64 vcc, %{3 * H + 1}", "s_nop 1"]
        for i in range(H):
            lines.append(f"{rest} %{i}, vcc, %{H + 1 + i}, %{2 * H + 1 + i}, vcc")
        lines.append(f"s_mov_b64 %{H}, vcc")
        body = "\\n\\t".join(lines)
        outs = ", ".join(f'"=&v"(r[{H + i}])' for i in range(H)) + ', "=&s"(c1)'
        ins = ", ".join(f'"v"(a[{H + i}])' for i in range(H)) + ", " + ", ".join(f'"v"(b[{H + i}])' for i in range(H)) + ', "s"(c0)'
        A(f'        asm("{body}"')
        A(f"            : {outs}")
        A(f"            : {ins}")
        A('            : "vcc");')
        A("        return c1;")
        A("    }")

    # ---- add q on the lanes of a mask (borrow fix-up of sub) / q - x on the lanes of a mask (negate) -
    for name, first, rest, doc in (("add_q_masked", "v_add_co_u32", "v_addc_co_u32", "x += q"),
                                   ("rsub_q_masked", "v_sub_co_u32", "v_subb_co_u32", "x = q - x")):
        lines = [f"s_and_saveexec_b64 %{N}, %{N + 1}"]
        for i in range(N):
            if i == 0:
                lines.append(f"{first} %{i}, vcc, {lit(Q[i])}, %{i}")
            else:
                lines.append(f"{rest} %{i}, vcc, {lit(Q[i])}, %{i}, vcc")
        lines.append(f"s_mov_b64 exec, %{N}")
        body = "\\n\\t".join(lines)
        ops = ", ".join(f'"+v"(x[{i}])' for i in range(N))
        A(f"    // {doc} on the lanes whose bit is set in `mask`; other lanes unchanged")
        A(f"    static __device__ __forceinline__ void {name}(uint32_t (&x)[{N}], uint64_t mask) {{")
        A("        uint64_t saved;")
        A(f'        asm("{body}"')
        A(f'            : {ops}, "=&s"(saved)')
        A('            : "s"(mask)')
        A('            : "vcc");')
        A("    }")
    A("};")
    return "\n".join(o)


out = ["// GENERATED by tools/gen_fp_asm.py -- do not edit.", "#pragma once", "#include <cstdint>", '#include "params_gen.hpp"',
       "namespace msm {", "template <class P> struct FpAsm;", "#if defined(__HIP_DEVICE_COMPILE__)"]
for s in ("Bls12381Fp", "Bn254Fp"):
    out.append(gen(s))
out += ["#endif", "}  // namespace msm", ""]
path = os.path.join(HERE, "..", "zprize-wasm-msm_amd", "csrc", "fp_asm_gen.hpp")
open(path, "w").write("\n".join(out))
print("wrote", os.path.normpath(path))
