#!/usr/bin/env node
/*
 * gen_golden_ingroup.js -- DEV-TIME generator of tests/golden/golden_ingroup.json: the reference's subgroup check
 * g1m_inGroupAffine / g2m_inGroupAffine (wasmcurves/src/bls12381/build_bls12381.js:1494-1576, 1370-1480) run on the
 * prebuilt wasmcurves/build/bls12381.wasm, on
 *   - points of the committed pools (in the group), affine infinity (the reference answers 1),
 *   - the points of the reference's own tests (test/bls12381.js:737-802): x = 4 resp. (0, 4) with y := x^3 + b (not on the
 *     curve), with y := sqrt(x^3 + b) (on the curve, not in the group), and that point times the cofactor (in the group),
 *   - further curve points built the same way from small x (almost never in the group), some of them cofactor-cleared.
 */
"use strict";
const fs = require("fs");
const path = require("path");
const REF = "/root/reference/wasmcurves/build";
const OUT = path.join(__dirname, "..", "tests", "golden");
const hex = (b) => Buffer.from(b).toString("hex");
const H1 = 0x396c8c005555e1568c00aaab0000aaabn;   // test/bls12381.js:760
const H2 = 0x05d543a95414e7f1091d50792876a202cd91de4547085abaa68a205b2e5a7ddfa628f1cb4d9e82ef21537e293a6691ae1616ec6e786f0c70cf1c38e31c7238e5n;   // :795

(async () => {
    const n8 = 48;
    const memory = new WebAssembly.Memory({ initial: 200 });
    const ex = (await WebAssembly.instantiate(fs.readFileSync(path.join(REF, "bls12381.wasm")), { env: { memory } })).instance.exports;
    const consts = require("/root/reference/wasmcurves/build/bls12381_wasm.js");
    const u8 = () => new Uint8Array(memory.buffer);
    const u32 = () => new Uint32Array(memory.buffer);
    const alloc = (n) => { const p = u32()[0]; u32()[0] = p + ((n + 7) & ~7); return p; };
    const setInt = (p, v, len) => { for (let i = 0; i < len; i++) { u8()[p + i] = Number(v & 0xffn); v >>= 8n; } };
    const result = { generator: "oracle/gen_golden_ingroup.js", cases: [] };
    for (const g of [1, 2]) {
        const pf = g === 1 ? "g1m" : "g2m", ff = g === 1 ? "f1m" : "f2m";
        const fsz = n8 * g, asz = 2 * fsz, jsz = 3 * fsz;
        const pB = g === 1 ? consts.pG1b : consts.pG2b;
        const pool = fs.readFileSync(path.join(OUT, `pool_bls12381_g${g}.bin`));
        const pts = [];
        const push = (p) => pts.push(Buffer.from(u8().slice(p, p + asz)));
        const p1 = alloc(jsz), pT = alloc(jsz), pS = alloc(64), pChk = alloc(fsz);
        for (let i = 0; i < 6; i++) pts.push(Buffer.from(pool.slice(i * asz, (i + 1) * asz)));
        pts.push(Buffer.alloc(asz));   // affine infinity
        for (let xv = 4n; pts.length < 40 && xv < 200n; xv++) {
            // x = xv (G1) or (xv - 4, 4) (G2: the reference's (0, 4) first)
            u8().fill(0, p1, p1 + jsz);
            if (g === 1) setInt(p1, xv, n8);
            else { setInt(p1, xv - 4n, n8); setInt(p1 + n8, 4n, n8); }
            ex[ff + "_toMontgomery"](p1, p1);
            ex[ff + "_square"](p1, p1 + fsz);
            ex[ff + "_mul"](p1, p1 + fsz, p1 + fsz);
            ex[ff + "_add"](p1 + fsz, pB, p1 + fsz);          // y := x^3 + b  (not on the curve in general)
            if (xv < 7n) push(p1);
            if (!ex[ff + "_isSquare"](p1 + fsz)) continue;    // x^3 + b is not a square (f?m_sqrt does not terminate on those)
            u8().copyWithin(pChk, p1 + fsz, p1 + asz);
            ex[ff + "_sqrt"](p1 + fsz, p1 + fsz);
            ex[ff + "_square"](p1 + fsz, pT);
            if (!ex[ff + "_eq"](pT, pChk)) throw new Error("sqrt check failed");
            push(p1);                                          // on the curve, (almost surely) not in the group
            if (xv % 3n === 1n) {                              // cofactor-cleared: in the group
                if (g === 1) { setInt(pS, H1, 16); ex.g1m_timesScalarAffine(p1, pS, 16, pT); }
                else { setInt(pS, H2, 64); ex.g2m_timesScalarAffine(p1, pS, 64, pT); }
                ex[pf + "_toAffine"](pT, pT);
                push(pT);
            }
        }
        const pIn = alloc(asz);
        const inGroup = [], inCurve = [];
        for (const b of pts) {
            u8().set(b, pIn);
            inGroup.push(ex[pf + "_inGroupAffine"](pIn));
            inCurve.push(ex[pf + "_inCurveAffine"](pIn));
        }
        result.cases.push({ curve: "bls12381", group: g, n: pts.length, affine_in: hex(Buffer.concat(pts)), inGroupAffine: inGroup, inCurveAffine: inCurve });
        console.log(`g${g}: ${pts.length} points, in group ${inGroup.reduce((a, b) => a + b, 0)}, on curve ${inCurve.reduce((a, b) => a + b, 0)}`);
    }
    fs.writeFileSync(path.join(OUT, "golden_ingroup.json"), JSON.stringify(result));
    console.log("wrote golden_ingroup.json");
})().catch((e) => { console.error(e); process.exit(1); });
