#!/usr/bin/env node
/*
 * gen_golden_fft.js -- DEV-TIME generator of tests/golden/golden_fft.json: the reference's frm_fft / frm_ifft (field NTT over Fr)
 * and g1m_/g2m_ fft / ifft (the same transform over group elements, butterflies through g?m_timesFr) --
 * wasmcurves/src/build_fft.js:178-245 (fft, ifft), :247-394 (rawfft), :396-516 (fftFinal) -- run on the prebuilt
 * wasmcurves/build/{bls12381,bn128}.wasm.  Group outputs are recorded raw and normalised.
 */
"use strict";
const fs = require("fs");
const path = require("path");
const crypto = require("crypto");
const REF = "/root/reference/wasmcurves/build";
const OUT = path.join(__dirname, "..", "tests", "golden");
const CURVES = { bls12381: { wasm: "bls12381.wasm", n8: 48, r: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001n },
                 bn128: { wasm: "bn128.wasm", n8: 32, r: 21888242871839275222246405745257275088548364400416034343698204186575808495617n } };
const hex = (b) => Buffer.from(b).toString("hex");
const prng = (label, n) => { let out = Buffer.alloc(0), c = 0; while (out.length < n) out = Buffer.concat([out, crypto.createHash("sha256").update(label + ":" + (c++)).digest()]); return out.slice(0, n); };
const le2big = (b) => { let v = 0n; for (let i = b.length - 1; i >= 0; i--) v = (v << 8n) | BigInt(b[i]); return v; };

(async () => {
    const result = { generator: "oracle/gen_golden_fft.js", cases: [] };
    for (const curveName of Object.keys(CURVES)) {
        const C = CURVES[curveName];
        const memory = new WebAssembly.Memory({ initial: 400 });
        const ex = (await WebAssembly.instantiate(fs.readFileSync(path.join(REF, C.wasm)), { env: { memory } })).instance.exports;
        const u8 = () => new Uint8Array(memory.buffer);
        const u32 = () => new Uint32Array(memory.buffer);
        const alloc = (n) => { const p = u32()[0]; u32()[0] = p + ((n + 7) & ~7); return p; };
        const setBig = (p, v, len) => { for (let i = 0; i < len; i++) { u8()[p + i] = Number(v & 0xffn); v >>= 8n; } };
        for (const N of [1, 2, 4, 8, 32, 256, 1024]) {
            const p = alloc(N * 32);
            for (let i = 0; i < N; i++) { setBig(p + 32 * i, le2big(prng(`${curveName}:fft:${N}:${i}`, 40)) % C.r, 32); ex.frm_toMontgomery(p + 32 * i, p + 32 * i); }
            if (N >= 4) u8().fill(0, p + 32, p + 64);
            const inp = Buffer.from(u8().slice(p, p + N * 32));
            ex.frm_fft(p, N);
            const f = Buffer.from(u8().slice(p, p + N * 32));
            u8().set(inp, p);
            ex.frm_ifft(p, N);
            const fi = Buffer.from(u8().slice(p, p + N * 32));
            result.cases.push({ curve: curveName, group: 0, n: N, in: hex(inp), fft: hex(f), ifft: hex(fi) });
        }
        for (const g of [1, 2]) {
            const pf = g === 1 ? "g1m" : "g2m";
            const fsz = C.n8 * g, asz = 2 * fsz, jsz = 3 * fsz;
            const pool = fs.readFileSync(path.join(OUT, `pool_${curveName}_g${g}.bin`));
            for (const N of (g === 1 ? [1, 2, 4, 8, 16] : [2, 4, 8])) {
                const pJ = alloc(N * jsz), pT = alloc(jsz), pW = alloc(N * jsz);
                for (let i = 0; i < N; i++) {
                    u8().set(pool.slice(i * asz, (i + 1) * asz), pT);
                    ex[pf + "_toJacobian"](pT, pJ + i * jsz);
                    if (i % 3 === 1) ex[pf + "_double"](pJ + i * jsz, pJ + i * jsz);
                }
                if (N >= 4) ex[pf + "_zero"](pJ + 2 * jsz);
                const inp = Buffer.from(u8().slice(pJ, pJ + N * jsz));
                const run = (fn) => {
                    u8().set(inp, pW);
                    ex[pf + fn](pW, N);
                    const raw = Buffer.from(u8().slice(pW, pW + N * jsz));
                    for (let i = 0; i < N; i++) ex[pf + "_normalize"](pW + i * jsz, pW + i * jsz);
                    return [raw, Buffer.from(u8().slice(pW, pW + N * jsz))];
                };
                const [fr, fn_] = run("_fft"), [ir, in_] = run("_ifft");
                result.cases.push({ curve: curveName, group: g, n: N, in: hex(inp), fft_raw: hex(fr), fft: hex(fn_), ifft_raw: hex(ir), ifft: hex(in_) });
            }
        }
    }
    fs.writeFileSync(path.join(OUT, "golden_fft.json"), JSON.stringify(result));
    console.log("wrote golden_fft.json", result.cases.length, "cases");
})().catch((e) => { console.error(e); process.exit(1); });
