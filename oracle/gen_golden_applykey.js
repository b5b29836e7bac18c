#!/usr/bin/env node
/*
 * gen_golden_applykey.js -- DEV-TIME generator of tests/golden/golden_applykey.json: the reference's batchApplyKey exports
 * (wasmcurves/src/build_applykey.js:20-76 wired at bls12381/build_bls12381.js:88-92, bn128/build_bn128.js:84-88) run on the
 * prebuilt wasmcurves/build/{bls12381,bn128}.wasm:  out[i] = (first * inc^i) * in[i]  with first, inc in Fr (Montgomery).
 *   g?m_batchApplyKey       Jacobian in  -> Jacobian out          g?m_batchApplyKeyMixed   affine in -> Jacobian out
 *   frm_batchApplyKey       Fr in (Montgomery) -> Fr out
 * Group outputs are recorded raw and normalised (g?m_normalize): the reference returns some Jacobian representative.
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
    const result = { generator: "oracle/gen_golden_applykey.js", cases: [] };
    for (const curveName of Object.keys(CURVES)) {
        const C = CURVES[curveName];
        const memory = new WebAssembly.Memory({ initial: 200 });
        const ex = (await WebAssembly.instantiate(fs.readFileSync(path.join(REF, C.wasm)), { env: { memory } })).instance.exports;
        const u8 = () => new Uint8Array(memory.buffer);
        const u32 = () => new Uint32Array(memory.buffer);
        const alloc = (n) => { const p = u32()[0]; u32()[0] = p + ((n + 7) & ~7); return p; };
        const setBig = (p, v, len) => { for (let i = 0; i < len; i++) { u8()[p + i] = Number(v & 0xffn); v >>= 8n; } };
        const pFirst = alloc(32), pInc = alloc(32);
        const mkFr = (label, p) => { setBig(p, le2big(prng(label, 40)) % C.r, 32); ex.frm_toMontgomery(p, p); return Buffer.from(u8().slice(p, p + 32)); };
        for (const variant of ["random", "first_one", "inc_zero"]) {
            let first = mkFr(`${curveName}:${variant}:first`, pFirst), inc = mkFr(`${curveName}:${variant}:inc`, pInc);
            if (variant === "first_one") { ex.frm_one(pFirst); first = Buffer.from(u8().slice(pFirst, pFirst + 32)); }
            if (variant === "inc_zero") { u8().fill(0, pInc, pInc + 32); inc = Buffer.alloc(32); }
            // Fr batch
            const NF = 20, pF = alloc(NF * 32), pFo = alloc(NF * 32);
            for (let i = 0; i < NF; i++) mkFr(`${curveName}:${variant}:fr:${i}`, pF + 32 * i);
            u8().fill(0, pF + 5 * 32, pF + 6 * 32);
            const frIn = Buffer.from(u8().slice(pF, pF + NF * 32));
            ex.frm_batchApplyKey(pF, NF, pFirst, pInc, pFo);
            const frOut = Buffer.from(u8().slice(pFo, pFo + NF * 32));
            const rec = { curve: curveName, variant, first: hex(first), inc: hex(inc), fr_n: NF, fr_in: hex(frIn), fr_out: hex(frOut), groups: {} };
            for (const g of [1, 2]) {
                const pf = g === 1 ? "g1m" : "g2m";
                const fsz = C.n8 * g, asz = 2 * fsz, jsz = 3 * fsz;
                const pool = fs.readFileSync(path.join(OUT, `pool_${curveName}_g${g}.bin`));
                const N = 12;
                const pAff = alloc(N * asz), pJac = alloc(N * jsz), pOut = alloc(N * jsz), pT = alloc(jsz);
                u8().set(pool.slice(0, N * asz), pAff);
                u8().fill(0, pAff + 4 * asz, pAff + 5 * asz);   // an affine infinity
                for (let i = 0; i < N; i++) {
                    ex[pf + "_toJacobian"](pAff + i * asz, pJac + i * jsz);
                    if (i % 3 === 1) ex[pf + "_double"](pJac + i * jsz, pJac + i * jsz);                      // Z != 1
                    if (i % 3 === 2) { ex[pf + "_double"](pJac + i * jsz, pT); ex[pf + "_add"](pT, pJac + i * jsz, pJac + i * jsz); }
                }
                const affIn = Buffer.from(u8().slice(pAff, pAff + N * asz)), jacIn = Buffer.from(u8().slice(pJac, pJac + N * jsz));
                const grab = () => { const raw = Buffer.from(u8().slice(pOut, pOut + N * jsz)); for (let i = 0; i < N; i++) ex[pf + "_normalize"](pOut + i * jsz, pOut + i * jsz); return [raw, Buffer.from(u8().slice(pOut, pOut + N * jsz))]; };
                ex[pf + "_batchApplyKeyMixed"](pAff, N, pFirst, pInc, pOut);
                const [mRaw, mNorm] = grab();
                ex[pf + "_batchApplyKey"](pJac, N, pFirst, pInc, pOut);
                const [jRaw, jNorm] = grab();
                rec.groups["g" + g] = { n: N, affine_in: hex(affIn), jacobian_in: hex(jacIn), mixed_out_raw: hex(mRaw), mixed_out_norm: hex(mNorm), jac_out_raw: hex(jRaw), jac_out_norm: hex(jNorm) };
            }
            result.cases.push(rec);
        }
    }
    fs.writeFileSync(path.join(OUT, "golden_applykey.json"), JSON.stringify(result));
    console.log("wrote golden_applykey.json", result.cases.length, "cases");
})().catch((e) => { console.error(e); process.exit(1); });
