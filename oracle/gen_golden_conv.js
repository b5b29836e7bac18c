#!/usr/bin/env node
/*
 * gen_golden_conv.js -- DEV-TIME generator of tests/golden/golden_conv.json: the reference's batch conversion
 * exports either side of the MSM (g?m_batchToAffine, _batchToJacobian, _batchLEMtoU, _batchUtoLEM) run on the
 * prebuilt wasmcurves/build/{bls12381,bn128}.wasm.  Inputs come from the committed point pools
 * (tests/golden/pool_*.bin) plus Jacobian points derived with the reference's own g?m_double / g?m_add.
 */
"use strict";
const fs = require("fs");
const path = require("path");
const REF = "/root/reference/wasmcurves/build";
const OUT = path.join(__dirname, "..", "tests", "golden");
const CURVES = { bls12381: { wasm: "bls12381.wasm", n8: 48 }, bn128: { wasm: "bn128.wasm", n8: 32 } };
const hex = (b) => Buffer.from(b).toString("hex");

(async () => {
    const result = { generator: "oracle/gen_golden_conv.js", cases: [] };
    for (const curveName of Object.keys(CURVES)) {
        const C = CURVES[curveName];
        const memory = new WebAssembly.Memory({ initial: 200 });
        const ex = (await WebAssembly.instantiate(fs.readFileSync(path.join(REF, C.wasm)), { env: { memory } })).instance.exports;
        const u8 = () => new Uint8Array(memory.buffer);
        const u32 = () => new Uint32Array(memory.buffer);
        const alloc = (n) => { const p = u32()[0]; u32()[0] = p + ((n + 7) & ~7); return p; };
        for (const g of [1, 2]) {
            const pf = g === 1 ? "g1m" : "g2m";
            const fsz = C.n8 * g, asz = 2 * fsz, jsz = 3 * fsz;
            const pool = fs.readFileSync(path.join(OUT, `pool_${curveName}_g${g}.bin`));
            const N = 24;
            const pAff = alloc(N * asz), pJac = alloc(N * jsz), pOutA = alloc(N * asz), pOutJ = alloc(N * jsz), pU = alloc(N * asz), pBack = alloc(N * asz), pT = alloc(jsz);
            // affine inputs: pool points, with infinities at 3 and 17
            u8().set(pool.slice(0, N * asz), pAff);
            u8().fill(0, pAff + 3 * asz, pAff + 4 * asz);
            u8().fill(0, pAff + 17 * asz, pAff + 18 * asz);
            // Jacobian inputs: i%4==0 -> Z=1 ; 1 -> 2P ; 2 -> 3P ; 3 -> infinity (0,R,0) ; one (X,Y,0) with nonzero X,Y
            for (let i = 0; i < N; i++) {
                ex[pf + "_toJacobian"](pAff + i * asz, pJac + i * jsz);
                if (i % 4 === 1) ex[pf + "_double"](pJac + i * jsz, pJac + i * jsz);
                if (i % 4 === 2) { ex[pf + "_double"](pJac + i * jsz, pT); ex[pf + "_add"](pT, pJac + i * jsz, pJac + i * jsz); }
                if (i % 4 === 3) ex[pf + "_zero"](pJac + i * jsz);
            }
            u8().fill(0, pJac + 5 * jsz + 2 * fsz, pJac + 6 * jsz);   // Z := 0 on a point with X, Y != 0
            const jacIn = Buffer.from(u8().slice(pJac, pJac + N * jsz));
            const affIn = Buffer.from(u8().slice(pAff, pAff + N * asz));
            ex[pf + "_batchToAffine"](pJac, N, pOutA);
            const toAffine = Buffer.from(u8().slice(pOutA, pOutA + N * asz));
            ex[pf + "_batchToJacobian"](pAff, N, pOutJ);
            const toJac = Buffer.from(u8().slice(pOutJ, pOutJ + N * jsz));
            ex[pf + "_batchLEMtoU"](pAff, N, pU);
            const lemToU = Buffer.from(u8().slice(pU, pU + N * asz));
            ex[pf + "_batchUtoLEM"](pU, N, pBack);
            const uToLem = Buffer.from(u8().slice(pBack, pBack + N * asz));
            // inCurveAffine on: the affine inputs (incl. the two infinities) and the same points with x and y swapped
            const inCurve = [], swapped = Buffer.alloc(N * asz);
            for (let i = 0; i < N; i++) {
                inCurve.push(ex[pf + "_inCurveAffine"](pAff + i * asz));
                affIn.copy(swapped, i * asz, i * asz + fsz, (i + 1) * asz);
                affIn.copy(swapped, i * asz + fsz, i * asz, i * asz + fsz);
            }
            u8().set(swapped, pBack);
            const inCurveSwapped = [];
            for (let i = 0; i < N; i++) inCurveSwapped.push(ex[pf + "_inCurveAffine"](pBack + i * asz));
            result.cases.push({ inCurveAffine: inCurve, swapped_in: hex(swapped), inCurveAffine_swapped: inCurveSwapped, curve: curveName, group: g, n: N, jacobian_in: hex(jacIn), affine_in: hex(affIn), batchToAffine: hex(toAffine),
                batchToJacobian: hex(toJac), batchLEMtoU: hex(lemToU), batchUtoLEM_of_U: hex(uToLem) });
        }
    }
    fs.writeFileSync(path.join(OUT, "golden_conv.json"), JSON.stringify(result));
    console.log("wrote golden_conv.json");
})().catch((e) => { console.error(e); process.exit(1); });
