#!/usr/bin/env node
/*
 * bench.js -- the MSM through the Node surface a wasmcurves/ffjavascript user would call (index.js curve().G1.multiExpAffine):
 * host Uint8Arrays in, Promise<Uint8Array> out.  Reports ms per call for sequential awaits, for several promises in flight
 * (the addon's libuv workers share one context and overlap on the GPU), and with the bases uploaded once
 * (uploadBases / multiExpResident).  Inputs: the committed pool of BLS12-381 G1 points tiled to n, random 32-byte scalars < r.
 *     node bench.js [log2 n = 20] [calls = 8]
 */
"use strict";
const fs = require("fs");
const path = require("path");
const crypto = require("crypto");
const msm = require("./index.js");
const logn = parseInt(process.argv[2] || "20", 10), calls = parseInt(process.argv[3] || "8", 10), n = 1 << logn;
const R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001n;

(async () => {
    const pool = fs.readFileSync(path.join(__dirname, "..", "..", "tests", "golden", "pool_bls12381_g1.bin"));
    const bases = Buffer.alloc(n * 96);
    for (let off = 0; off < bases.length; off += pool.length) pool.copy(bases, off, 0, Math.min(pool.length, bases.length - off));
    const scalars = crypto.randomBytes(n * 32);
    for (let i = 0; i < n; i++) scalars[32 * i + 31] &= 0x3f;   // < 2^254 < r
    const G = msm.curve("bls12381").G1;
    const ref = Buffer.from(await G.multiExpAffine(bases, scalars)).toString("hex");   // warm-up + reference result
    let t = process.hrtime.bigint();
    for (let k = 0; k < calls; k++) await G.multiExpAffine(bases, scalars);
    const seq = Number(process.hrtime.bigint() - t) / 1e6 / calls;
    t = process.hrtime.bigint();
    const outs = await Promise.all(Array.from({ length: calls }, () => G.multiExpAffine(bases, scalars)));
    const par = Number(process.hrtime.bigint() - t) / 1e6 / calls;
    if (outs.some((o) => Buffer.from(o).toString("hex") !== ref)) throw new Error("results differ between calls");
    const h = G.uploadBases(bases);
    await G.multiExpResident(h, scalars);
    t = process.hrtime.bigint();
    for (let k = 0; k < calls; k++) {
        const o = await G.multiExpResident(h, scalars);
        if (k === 0 && Buffer.from(o).toString("hex") !== ref) throw new Error("resident-bases result differs");
    }
    const res = Number(process.hrtime.bigint() - t) / 1e6 / calls;
    console.log(JSON.stringify({ workload: `bls12381 G1 n=2^${logn} through node/index.js (host buffers)`, calls,
        ms_per_call_sequential: seq, ms_per_call_promises_in_flight: par, ms_per_call_resident_bases: res,
        pairs_per_s_promises_in_flight: n / (par / 1e3) }));
})().catch((e) => { console.error(e); process.exit(1); });
