#!/usr/bin/env node
/*
 * gen_golden_compressed.js -- DEV-TIME generator of tests/golden/golden_compressed.json: the reference's compressed point form,
 * g?m_LEMtoC / g?m_CtoLEM (wasmcurves/src/build_curve_jacobian_a0.js:1168-1205, 1265-1328; sign convention build_f1m.js:135-156,
 * build_f2m.js:411-430) on the prebuilt wasmcurves/build/{bls12381,bn128}.wasm.
 * LEMtoC decides "infinity" with the JACOBIAN _isZero, i.e. it looks at the field element that FOLLOWS the affine input; here every
 * finite point is followed by a non-zero element and the affine infinity by zeros, which is the intended reading
 * (infinity <=> x = y = 0 -> 0x40 00 .. 00).
 */
"use strict";
const fs = require("fs");
const path = require("path");
const REF = "/root/reference/wasmcurves/build";
const OUT = path.join(__dirname, "..", "tests", "golden");
const CURVES = { bls12381: { wasm: "bls12381.wasm", n8: 48 }, bn128: { wasm: "bn128.wasm", n8: 32 } };
const hex = (b) => Buffer.from(b).toString("hex");

(async () => {
    const result = { generator: "oracle/gen_golden_compressed.js", cases: [] };
    for (const curveName of Object.keys(CURVES)) {
        const C = CURVES[curveName];
        const memory = new WebAssembly.Memory({ initial: 100 });
        const ex = (await WebAssembly.instantiate(fs.readFileSync(path.join(REF, C.wasm)), { env: { memory } })).instance.exports;
        const u8 = () => new Uint8Array(memory.buffer);
        const u32 = () => new Uint32Array(memory.buffer);
        const alloc = (n) => { const p = u32()[0]; u32()[0] = p + ((n + 7) & ~7); return p; };
        for (const g of [1, 2]) {
            const pf = g === 1 ? "g1m" : "g2m";
            const fsz = C.n8 * g, asz = 2 * fsz;
            const pool = fs.readFileSync(path.join(OUT, `pool_${curveName}_g${g}.bin`));
            const N = 24;
            const pIn = alloc(3 * fsz), pC = alloc(fsz), pBack = alloc(asz);
            const affs = [], comps = [], backs = [];
            for (let i = 0; i < N; i++) {
                u8().fill(0, pIn, pIn + 3 * fsz);
                if (i !== 5) {
                    u8().set(pool.slice(i * asz, (i + 1) * asz), pIn);
                    if (i % 2 === 1) ex[(g === 1 ? "f1m" : "f2m") + "_neg"](pIn + fsz, pIn + fsz);   // both signs of y
                    u8()[pIn + 2 * fsz] = 1;                                                            // "Z" != 0
                }
                affs.push(Buffer.from(u8().slice(pIn, pIn + asz)));
                ex[pf + "_LEMtoC"](pIn, pC);
                comps.push(Buffer.from(u8().slice(pC, pC + fsz)));
                ex[pf + "_CtoLEM"](pC, pBack);
                backs.push(Buffer.from(u8().slice(pBack, pBack + asz)));
            }
            result.cases.push({ curve: curveName, group: g, n: N, affine_in: hex(Buffer.concat(affs)), compressed: hex(Buffer.concat(comps)), decompressed: hex(Buffer.concat(backs)) });
            const same = affs.every((a, i) => a.equals(backs[i]));
            console.log(`${curveName} g${g}: CtoLEM(LEMtoC(p)) == p for all: ${same}; flags ${comps.map((c) => (c[0] >> 6)).join("")}`);
        }
    }
    fs.writeFileSync(path.join(OUT, "golden_compressed.json"), JSON.stringify(result));
    console.log("wrote golden_compressed.json");
})().catch((e) => { console.error(e); process.exit(1); });
