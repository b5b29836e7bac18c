#!/usr/bin/env node
/*
 * gen_golden_fft2.js -- DEV-TIME generator of tests/golden/golden_fft2.json: the reference's blocked-FFT building blocks
 *   <p>_fftJoin / _fftJoinExt / _fftJoinExtInv / _fftMix / _fftFinal / _prepareLagrangeEvaluation   (p = frm, g1m, g2m)
 *   frm_batchToMontgomery / frm_batchFromMontgomery
 * wasmcurves/src/build_fft.js:657-1353, build_batchconvertion.js -- run on the prebuilt wasmcurves/build/{bls12381,bn128}.wasm
 * (Node's built-in WebAssembly only).  Group outputs are recorded normalised.
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
    const result = { generator: "oracle/gen_golden_fft2.js", cases: [] };
    for (const curveName of Object.keys(CURVES)) {
        const C = CURVES[curveName];
        const memory = new WebAssembly.Memory({ initial: 400 });
        const ex = (await WebAssembly.instantiate(fs.readFileSync(path.join(REF, C.wasm)), { env: { memory } })).instance.exports;
        const u8 = () => new Uint8Array(memory.buffer);
        const u32 = () => new Uint32Array(memory.buffer);
        const alloc = (n) => { const p = u32()[0]; u32()[0] = p + ((n + 7) & ~7); return p; };
        const setBig = (p, v, len) => { for (let i = 0; i < len; i++) { u8()[p + i] = Number(v & 0xffn); v >>= 8n; } };
        const frElem = (label) => { const p = alloc(32); setBig(p, le2big(prng(label, 40)) % C.r, 32); ex.frm_toMontgomery(p, p); return p; };
        for (const g of [0, 1, 2]) {
            const pf = g === 0 ? "frm" : `g${g}m`;
            const esz = g === 0 ? 32 : 3 * C.n8 * g, asz = 2 * C.n8 * g;
            const pool = g ? fs.readFileSync(path.join(OUT, `pool_${curveName}_g${g}.bin`)) : null;
            const fill = (p, N, tag) => {
                for (let i = 0; i < N; i++) {
                    if (g === 0) { setBig(p + 32 * i, le2big(prng(`${curveName}:${tag}:${N}:${i}`, 40)) % C.r, 32); ex.frm_toMontgomery(p + 32 * i, p + 32 * i); }
                    else {
                        const pT = alloc(asz), k = (i * 7 + tag.length) % 200;
                        u8().set(pool.slice(k * asz, (k + 1) * asz), pT);
                        ex[pf + "_toJacobian"](pT, p + i * esz);
                        if (i % 3 === 1) ex[pf + "_double"](p + i * esz, p + i * esz);
                    }
                }
                if (N >= 4) { if (g === 0) u8().fill(0, p + 32, p + 64); else ex[pf + "_zero"](p + 2 * esz); }
            };
            const snap = (p, N) => Buffer.from(u8().slice(p, p + N * esz));
            const norm = (p, N) => { if (g) for (let i = 0; i < N; i++) ex[pf + "_normalize"](p + i * esz, p + i * esz); return snap(p, N); };
            const sizes = g === 0 ? [1, 4, 16, 64] : g === 1 ? [1, 4, 8] : [2, 4];
            for (const N of sizes) {
                const first = frElem(`${curveName}:first:${g}:${N}`), inc = frElem(`${curveName}:inc:${g}:${N}`), factor = frElem(`${curveName}:factor:${g}:${N}`);
                const keyHex = { first: hex(u8().slice(first, first + 32)), inc: hex(u8().slice(inc, inc + 32)), factor: hex(u8().slice(factor, factor + 32)) };
                const b1 = alloc(N * esz), b2 = alloc(N * esz), w1 = alloc(N * esz), w2 = alloc(N * esz);
                fill(b1, N, "b1"); fill(b2, N, "b2x");
                const in1 = snap(b1, N), in2 = snap(b2, N);
                const rec = { curve: curveName, group: g, n: N, in1: hex(in1), in2: hex(in2), ...keyHex, ops: {} };
                const pair = (name, ...extra) => {
                    u8().set(in1, w1); u8().set(in2, w2);
                    ex[`${pf}_${name}`](w1, w2, N, first, inc, ...extra);
                    return { out1: hex(norm(w1, N)), out2: hex(norm(w2, N)) };
                };
                rec.ops.fftJoin = pair("fftJoin");
                for (const tb of [3, 7]) {
                    rec.ops[`fftJoinExt:${tb}`] = pair("fftJoinExt", tb);
                    rec.ops[`fftJoinExtInv:${tb}`] = pair("fftJoinExtInv", tb);
                    rec.ops[`prepareLagrangeEvaluation:${tb}`] = pair("prepareLagrangeEvaluation", tb);
                }
                for (let exp = 1; (1 << exp) <= N; exp++) {
                    u8().set(in1, w1);
                    ex[`${pf}_fftMix`](w1, N, exp);
                    rec.ops[`fftMix:${exp}`] = { out1: hex(norm(w1, N)) };
                }
                u8().set(in1, w1);
                ex[`${pf}_fftFinal`](w1, N, factor);
                rec.ops.fftFinal = { out1: hex(norm(w1, N)) };
                if (N === 4 && g === 0) {   // an odd length exercises fftFinal's middle element
                    u8().set(in1, w1);
                    ex[`${pf}_fftFinal`](w1, 3, factor);
                    rec.ops["fftFinal:odd3"] = { out1: hex(snap(w1, 3)) };
                }
                result.cases.push(rec);
            }
        }
        // frm_batchToMontgomery / frm_batchFromMontgomery (build_batchconvertion.js over frm_toMontgomery / frm_fromMontgomery)
        const N = 33, p = alloc(N * 32), q = alloc(N * 32);
        for (let i = 0; i < N; i++) setBig(p + 32 * i, i === 5 ? 0n : i === 6 ? C.r - 1n : le2big(prng(`${curveName}:mont:${i}`, 40)) % C.r, 32);
        const plain = Buffer.from(u8().slice(p, p + N * 32));
        ex.frm_batchToMontgomery(p, N, q);
        const mont = Buffer.from(u8().slice(q, q + N * 32));
        ex.frm_batchFromMontgomery(q, N, p);
        result.cases.push({ curve: curveName, group: -1, n: N, plain: hex(plain), mont: hex(mont), back: hex(u8().slice(p, p + N * 32)) });
    }
    fs.writeFileSync(path.join(OUT, "golden_fft2.json"), JSON.stringify(result));
    console.log("wrote golden_fft2.json", result.cases.length, "cases");
})().catch((e) => { console.error(e); process.exit(1); });
