#!/usr/bin/env python3
"""Generates zprize-wasm-msm_amd/csrc/fp_asm_gen.hpp: gfx950 inline-asm carry chains for the DEVICE
field arithmetic of each supported modulus.

Device values are kept "relaxed", in [0, 2q): with R = 2^(32N) > 4q a Montgomery product of two
relaxed values is again < 2q without any final subtraction (t < ab/R + q < 4q^2/R + q < 2q), so the
multiplier drops its conditional subtract, and add/sub correct by 2q instead of q.  Values are made
canonical ([0, q), the reference's representation, wasmcurves/src/build_f1m.js:71-117,754-765)
only when they leave the engine.

Why asm: hipcc lowers the portable `(uint64_t)a + b + c` limb loops to 64-bit add/shift emulation
(70-110 VALU instructions per 12-limb modular op).  Here: v_add_co/v_addc_co chains on VCC.
Encoding constraint that shapes the code: a VOP2 carry-in instruction already reads VCC on the constant
bus, so it cannot also take a 32-bit literal (or an SGPR) -- modulus limbs are first moved/ANDed into
a VGPR temp (v_and_b32 t, literal, mask doubles as the conditional select).
Inline-asm statements take at most 30 operands, so 12-limb chains are split in two halves and the
carry travels between the halves in an SGPR pair.
"""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
params = open(os.path.join(HERE, "..", "zprize-wasm-msm_amd", "csrc", "params_gen.hpp")).read()


def limbs_of(struct, name):
    m = re.search(r"struct %s \{.*?static constexpr uint32_t %s\[(\d+)\] = \{(.*?)\};" % (struct, name), params, re.S)
    return [int(x.strip().rstrip("u"), 16) for x in m.group(2).split(",")]


def to_int(l):
    return sum(v << (32 * i) for i, v in enumerate(l))


def to_limbs(v, n):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def lit(v):
    return "0x%08x" % v


class Asm:
    """Collects one asm statement: numbered operands + instruction lines."""

    def __init__(self):
        self.outs, self.ins, self.lines = [], [], []

    def out(self, constraint, expr):
        self.outs.append((constraint, expr))
        return len(self.outs) - 1

    def inp(self, constraint, expr):
        self.ins.append((constraint, expr))
        return ("in", len(self.ins) - 1)

    def ref(self, h):
        if isinstance(h, tuple):
            return "%%%d" % (len(self.outs) + h[1])
        return "%%%d" % h

    def emit(self, fmt, *hs):
        self.lines.append(fmt.format(*[self.ref(h) for h in hs]))

    def render(self, indent="        "):
        assert len(self.outs) + len(self.ins) <= 30, (len(self.outs), len(self.ins))
        body = "\\n\\t".join(self.lines)
        o = ", ".join(f'"{c}"({e})' for c, e in self.outs)
        i = ", ".join(f'"{c}"({e})' for c, e in self.ins)
        return f'{indent}asm("{body}"\n{indent}    : {o}\n{indent}    : {i}\n{indent}    : "vcc");'


def gen(struct):
    Q = limbs_of(struct, "Q")
    N = len(Q)
    H = N // 2
    q = to_int(Q)
    R = 1 << (32 * N)
    assert 4 * q < R, "relaxed arithmetic needs R > 4q"
    M = to_limbs(2 * q, N)              # 2q
    C = to_limbs(R - 2 * q, N)          # 2^(32N) - 2q (adding it subtracts 2q; carry-out <=> value >= 2q)
    L = []
    A = L.append
    A(f"template <> struct FpAsm<{struct}> {{")
    A(f"    static constexpr int N = {N};")

    def chain2(name, first, rest, doc):
        """r = a (op) b in two half chains; returns the final carry/borrow lane mask"""
        A(f"    // {doc}")
        A(f"    static __device__ __forceinline__ uint64_t {name}(uint32_t (&r)[{N}], const uint32_t (&a)[{N}], const uint32_t (&b)[{N}]) {{")
        A("        uint64_t c0, c1;")
        for half in range(2):
            s = Asm()
            rr = [s.out("=&v", f"r[{half * H + i}]") for i in range(H)]
            c = s.out("=&s", "c0" if half == 0 else "c1")
            aa = [s.inp("v", f"a[{half * H + i}]") for i in range(H)]
            bb = [s.inp("v", f"b[{half * H + i}]") for i in range(H)]
            if half == 1:
                cin = s.inp("s", "c0")
                s.emit("s_mov_b64 vcc, {}", cin)
                s.emit("s_nop 1")
            for i in range(H):
                if half == 0 and i == 0:
                    s.emit(first + " {}, vcc, {}, {}", rr[i], aa[i], bb[i])
                else:
                    s.emit(rest + " {}, vcc, {}, {}, vcc", rr[i], aa[i], bb[i])
            s.emit("s_mov_b64 {}, vcc", c)
            A(s.render())
        A("        return c1;")
        A("    }")

    chain2("add_raw", "v_add_co_u32", "v_addc_co_u32", f"r = a + b on {N} limbs; returns the carry out of limb {N - 1} as a lane mask")
    chain2("sub_raw", "v_sub_co_u32", "v_subb_co_u32", f"r = a - b on {N} limbs; returns the final borrow as a lane mask")

    def add_const_masked(name, K, doc, masked):
        """x += K (& lane mask if masked); returns carry-out lane mask.  K limbs go through VGPR temps."""
        A(f"    // {doc}")
        sig = f"uint32_t (&x)[{N}]" + (", uint64_t mask" if masked else "")
        A(f"    static __device__ __forceinline__ uint64_t {name}({sig}) {{")
        A("        uint64_t c0, c1;")
        A("        uint32_t t0, t1;")
        if masked:
            A("        uint32_t m;")
            A('        asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m) : "s"(mask));')
        for half in range(2):
            s = Asm()
            xx = [s.out("+v", f"x[{half * H + i}]") for i in range(H)]
            t = [s.out("=&v", "t0"), s.out("=&v", "t1")]
            c = s.out("=&s", "c0" if half == 0 else "c1")
            # "+v" operands occupy an input slot each as well
            nslots = len(s.outs) + H
            mm = s.inp("v", "m") if masked else None
            cin = s.inp("s", "c0") if half == 1 else None
            assert nslots + len(s.ins) <= 30

            def ref_in(h):
                # a "+v" operand is numbered once (as an output); inputs follow the outputs
                return "%%%d" % (len(s.outs) + h[1])
            if half == 1:
                s.lines.append(f"s_mov_b64 vcc, {ref_in(cin)}")
                s.lines.append("s_nop 1")
            for i in range(H):
                k = K[half * H + i]
                tt = "%%%d" % t[i & 1]
                xi = "%%%d" % xx[i]
                if masked:
                    s.lines.append(f"v_and_b32 {tt}, {lit(k)}, {ref_in(mm)}")
                else:
                    s.lines.append(f"v_mov_b32 {tt}, {lit(k)}")
                if half == 0 and i == 0:
                    s.lines.append(f"v_add_co_u32 {xi}, vcc, {xi}, {tt}")
                else:
                    s.lines.append(f"v_addc_co_u32 {xi}, vcc, {xi}, {tt}, vcc")
            s.lines.append("s_mov_b64 %%%d, vcc" % c)
            body = "\\n\\t".join(s.lines)
            o = ", ".join(f'"{cc}"({e})' for cc, e in s.outs)
            i_ = ", ".join(f'"{cc}"({e})' for cc, e in s.ins)
            A(f'        asm("{body}"')
            A(f"            : {o}")
            A(f"            : {i_}")
            A('            : "vcc");')
        A("        return c1;")
        A("    }")

    add_const_masked("add_2q_masked", M, "x += 2q on the lanes of `mask` (borrow fix-up of a relaxed subtraction)", True)
    add_const_masked("add_minus_2q", C, "x += 2^(32N) - 2q, i.e. x -= 2q mod 2^(32N); the returned carry mask is set where the old x >= 2q", False)
    add_const_masked("add_minus_q", to_limbs(R - q, N), "x += 2^(32N) - q, i.e. x -= q mod 2^(32N); carry mask set where the old x >= q", False)
    add_const_masked("add_q_masked", Q, "x += q on the lanes of `mask`", True)

    # ---- x = 2q - x on the lanes of mask (EXEC-masked; one statement because EXEC must be restored) -------
    s_lines = []
    xs = [f"%{i}" for i in range(N)]
    saved, t0, t1 = f"%{N}", f"%{N + 1}", f"%{N + 2}"
    maskop = f"%{N + 3}"         # outputs: N x, saved, t0, t1 ; then the mask input
    s_lines.append(f"s_and_saveexec_b64 {saved}, {maskop}")
    for i in range(N):
        tt = t0 if (i & 1) == 0 else t1
        s_lines.append(f"v_mov_b32 {tt}, {lit(M[i])}")
        if i == 0:
            s_lines.append(f"v_sub_co_u32 {xs[i]}, vcc, {tt}, {xs[i]}")
        else:
            s_lines.append(f"v_subb_co_u32 {xs[i]}, vcc, {tt}, {xs[i]}, vcc")
    s_lines.append(f"s_mov_b64 exec, {saved}")
    assert N + 3 + N + 1 <= 30
    body = "\\n\\t".join(s_lines)
    ops = ", ".join(f'"+v"(x[{i}])' for i in range(N)) + ', "=&s"(saved), "=&v"(t0), "=&v"(t1)'
    A("    // x = 2q - x on the lanes of `mask` (callers exclude x == 0); other lanes unchanged")
    A(f"    static __device__ __forceinline__ void rsub_2q_masked(uint32_t (&x)[{N}], uint64_t mask) {{")
    A("        uint64_t saved;")
    A("        uint32_t t0, t1;")
    A(f'        asm("{body}"')
    A(f"            : {ops}")
    A('            : "s"(mask)')
    A('            : "vcc");')
    A("    }")
    A("};")
    return "\n".join(L)


out = ["// GENERATED by tools/gen_fp_asm.py -- do not edit.", "#pragma once", "#include <cstdint>", '#include "params_gen.hpp"',
       "namespace msm {", "template <class P> struct FpAsm;", "#if defined(__HIP_DEVICE_COMPILE__)"]
for s in ("Bls12381Fp", "Bn254Fp"):
    out.append(gen(s))
out += ["#endif", "}  // namespace msm", ""]
path = os.path.join(HERE, "..", "zprize-wasm-msm_amd", "csrc", "fp_asm_gen.hpp")
open(path, "w").write("\n".join(out))
print("wrote", os.path.normpath(path))
