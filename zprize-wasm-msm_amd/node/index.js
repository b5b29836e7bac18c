/*
 * index.js -- Node.js host side of the MI355X MSM engine, shaped like the reference.
 *
 * Two surfaces:
 *
 * 1. instantiate(curveName, {memory})  ->  an object with the reference's WASM export names operating
 *    on the caller's linear memory, so code written against the wasmcurves module keeps working:
 *        const m = msm.instantiate("bls12381", { memory });
 *        m.g1m_multiexpAffine(pBases, pScalars, scalarSize, n, pr);        // build_multiexp.js:251-257
 *        m.g1m_multiexpAffine_chunk(pBases, pScalars, scalarSize, n, startBit, chunkSize, pr);   // :96-104
 *    (also g1m_multiexp, g1m_multiexp_chunk, g2m_*; the opt-path alias g1m_multiexp_multiExp(p,s,n,pr)
 *    of build_multiexp_opt.js:1987-1996).  Results are written through `pr` as a NORMALISED Jacobian point.
 *
 * 2. curve(curveName).G1 / .G2 with multiExp(buffBases, buffScalars) / multiExpAffine(...) returning a
 *    Promise<Uint8Array> -- the call shape ffjavascript@0.2.56 gives its users (SURVEY.md section 8b; that
 *    package is not part of the reference tree, so this surface is modelled, not pinned).
 *
 * No fallback: if the addon or libmsm_amd.so or a GPU is missing, loading / instantiate throws.
 */
"use strict";
const path = require("path");

const native = require(path.join(__dirname, "msm_amd_napi.node"));

const CURVES = { bls12381: { id: 0, n8: 48 }, bn128: { id: 1, n8: 32 } };
const FLAG_JACOBIAN = 1;

let sharedCtx = null;
function context(device) {
    if (device === undefined) {
        if (!sharedCtx) sharedCtx = native.create(0);
        return sharedCtx;
    }
    return native.create(device);
}

function lookup(curveName) {
    const c = CURVES[curveName];
    if (!c) throw new RangeError(`unknown curve ${curveName} (bls12381 | bn128)`);
    return c;
}

/** WASM-export-shaped view over a caller-owned WebAssembly.Memory (or any {buffer}) */
function instantiate(curveName, imports, opts) {
    const C = lookup(curveName);
    const memory = imports && (imports.memory || (imports.env && imports.env.memory));
    if (!memory) throw new TypeError("instantiate(curve, {memory}) needs the linear memory the offsets refer to");
    const ctx = context(opts && opts.device);
    const u8 = () => new Uint8Array(memory.buffer);
    const ex = { memory };
    // in-place transforms, (p, n) as the reference's exports (build_fft.js:178-245); group results are normalised Jacobian points
    const fftFn = (grp, esz, inverse) => (p, n) => u8().set(native.fft(ctx, C.id, grp, inverse, u8().subarray(p, p + n * esz)), p);
    ex.frm_fft = fftFn(0, 32, 0);
    ex.frm_ifft = fftFn(0, 32, 1);
    ex.frm_batchApplyKey = (pIn, n, pFirst, pInc, pOut) => u8().set(native.batchApplyKey(ctx, C.id, 0, 0, u8().subarray(pIn, pIn + n * 32),
        u8().subarray(pFirst, pFirst + 32), u8().subarray(pInc, pInc + 32)), pOut);
    for (const g of [1, 2]) {
        const pf = `g${g}m`;
        const fsz = C.n8 * g;
        const mk = (flags, chunk) => function (pBases, pScalars, scalarSize, n, a, b, c) {
            const stride = (flags & FLAG_JACOBIAN ? 3 : 2) * fsz;
            const mem = u8();
            const bases = mem.subarray(pBases, pBases + n * stride);
            const scalars = mem.subarray(pScalars, pScalars + n * scalarSize);
            let out, pr;
            if (chunk) { out = native.multiexpChunk(ctx, C.id, g, flags, bases, scalars, scalarSize, n, a, b); pr = c; }
            else { out = native.multiexp(ctx, C.id, g, flags, bases, scalars, scalarSize, n); pr = a; }
            u8().set(out, pr);
        };
        ex[`${pf}_multiexp`] = mk(FLAG_JACOBIAN, false);
        ex[`${pf}_multiexpAffine`] = mk(0, false);
        ex[`${pf}_multiexp_chunk`] = mk(FLAG_JACOBIAN, true);
        ex[`${pf}_multiexpAffine_chunk`] = mk(0, true);
        // batch conversions either side of the MSM (same (pIn, n, pOut) signature as the reference's exports)
        const conv = (which, inSz, outSz) => (pIn, n, pOut) => u8().set(native.batchConvert(ctx, which, C.id, g, u8().subarray(pIn, pIn + n * inSz)), pOut);
        // validation: the reference's single-point exports (return 0/1) and batched forms (n points -> Uint8Array of flags)
        ex[`${pf}_inCurveAffine`] = (p) => native.batchCheck(ctx, 0, C.id, g, u8().subarray(p, p + 2 * fsz))[0];
        ex[`${pf}_batchInCurveAffine`] = (pIn, n) => native.batchCheck(ctx, 0, C.id, g, u8().subarray(pIn, pIn + n * 2 * fsz));
        if (curveName === "bls12381") {
            ex[`${pf}_inGroupAffine`] = (p) => native.batchCheck(ctx, 1, C.id, g, u8().subarray(p, p + 2 * fsz))[0];
            ex[`${pf}_batchInGroupAffine`] = (pIn, n) => native.batchCheck(ctx, 1, C.id, g, u8().subarray(pIn, pIn + n * 2 * fsz));
        }
        // out[i] = (first * inc^i) * in[i]; (pIn, n, pFirst, pInc, pOut) as the reference's exports; output normalised Jacobian
        const applyKey = (jacIn) => (pIn, n, pFirst, pInc, pOut) => u8().set(native.batchApplyKey(ctx, C.id, g, jacIn, u8().subarray(pIn, pIn + n * (jacIn ? 3 : 2) * fsz),
            u8().subarray(pFirst, pFirst + 32), u8().subarray(pInc, pInc + 32)), pOut);
        ex[`${pf}_fft`] = fftFn(g, 3 * fsz, 0);
        ex[`${pf}_ifft`] = fftFn(g, 3 * fsz, 1);
        ex[`${pf}_batchApplyKey`] = applyKey(1);
        ex[`${pf}_batchApplyKeyMixed`] = applyKey(0);
        ex[`${pf}_batchToAffine`] = conv(0, 3 * fsz, 2 * fsz);
        ex[`${pf}_batchToJacobian`] = conv(1, 2 * fsz, 3 * fsz);
        ex[`${pf}_batchLEMtoU`] = conv(2, 2 * fsz, 2 * fsz);
        ex[`${pf}_batchUtoLEM`] = conv(3, 2 * fsz, 2 * fsz);
        ex[`${pf}_batchLEMtoC`] = conv(4, 2 * fsz, fsz);
        ex[`${pf}_batchCtoLEM`] = conv(5, fsz, 2 * fsz);
        // comparison helpers the reference's tests use around the MSM
        ex[`${pf}_normalize`] = (p, pr) => u8().set(native.normalize(C.id, g, u8().subarray(p, p + 3 * fsz)), pr);
        ex[`${pf}_toAffine`] = (p, pr) => u8().set(native.toAffine(C.id, g, u8().subarray(p, p + 3 * fsz)), pr);
        ex[`${pf}_eq`] = (p1, p2) => (native.eq(C.id, g, u8().subarray(p1, p1 + 3 * fsz), u8().subarray(p2, p2 + 3 * fsz)) ? 1 : 0);
    }
    ex.g1m_multiexp_multiExp = (pPoints, pScalars, n, pr) => ex.g1m_multiexpAffine(pPoints, pScalars, 32, n, pr);
    if (curveName === "bls12381") {
        // Manta's GLV exports (wasmcurves/src/build_glv.js:269-272), same pointer conventions
        ex.g1m_glv_decomposeScalar = (pScalar, pScalarRes) => {
            const r = native.glvDecomposeScalar(u8().subarray(pScalar, pScalar + 32));
            u8().set(r.scalars, pScalarRes);
            return r.sign;
        };
        ex.g1m_glv_endomorphism = (pPoint, sign, pPointRes) => u8().set(native.glvEndomorphism(u8().subarray(pPoint, pPoint + 96), sign ? 1 : 0), pPointRes);
        ex.g1m_glv_preprocessEndomorphism = (pPoints, pScalars, n, pPointsRes, pScalarsRes) => {
            const r = native.glvPreprocess(ctx, u8().subarray(pPoints, pPoints + 96 * n), u8().subarray(pScalars, pScalars + 32 * n), n);
            u8().set(r.points, pPointsRes);
            u8().set(r.scalars, pScalarsRes);
        };
    }
    ex.f1m_toMontgomery = (p, pr) => u8().set(native.toMontgomery(C.id, u8().subarray(p, p + C.n8)), pr);
    ex.f1m_fromMontgomery = (p, pr) => u8().set(native.fromMontgomery(C.id, u8().subarray(p, p + C.n8)), pr);
    return ex;
}

/** ffjavascript-shaped async surface */
function curve(curveName, opts) {
    const C = lookup(curveName);
    const ctx = context(opts && opts.device);
    const maxSlice = (opts && opts.maxSlice) || (1 << 22);
    // opts.options = {name: value}: engine options (msm_amd_set_option), e.g. {slots: 4} -- how many calls may be in flight at once
    // (unresolved promises beyond it wait inside the engine): two suit n >= 2^19, three or four n <= 2^18 (DESIGN.md section 5)
    for (const [k, v] of Object.entries((opts && opts.options) || {})) native.setOption(ctx, k, v);
    const dist = opts && opts.dist;
    if (dist) native.distInit(ctx, dist.rank, dist.world, dist.id);
    const distQueue = [];
    let distInFlight = 0;
    const distPump = () => {
        while (distInFlight < 2 && distQueue.length) {
            const q = distQueue.shift();
            distInFlight++;
            let p;
            try { p = native.multiexpDistAsync(...q.args); } catch (e) { p = Promise.reject(e); }
            p.then(q.resolve, q.reject).then(() => { distInFlight--; distPump(); });
        }
    };
    const distCall = (...args) => new Promise((resolve, reject) => { distQueue.push({ args, resolve, reject }); distPump(); });
    const group = (g) => {
        const fsz = C.n8 * g;
        // ffjavascript's G.multiExp[Affine] (package ffjavascript@0.2.56, NOT in the reference tree: behaviour modelled from upstream
        // knowledge, parity unpinned at this level -- SURVEY.md 8b): n = bases / point size, scalarSize = scalars / n, the points are cut
        // into slices of at most 2^22, every slice is one multi-exponentiation (there: one `_chunk` task per window on a worker pool,
        // Horner on the main thread; here: one engine call, windows and Horner inside it) and the slice results are added.
        // opts.dist = {rank, world, id}: this process is one rank of a multi-GPU group (msm_amd_dist_*; every rank makes the same calls).
        // A call of a group is submitted to the engine inside native.multiexpDistAsync, on this (the JS) thread, so the order of the
        // group's exchanges is the order of the JS calls -- the same on every rank.  `distCall` keeps at most two of them submitted
        // (the engine's slots) and the rest in a FIFO: concurrent promises -- slices of one multiExp, a G1 and a G2 call of a
        // prover -- are queued, never reordered and never refused for want of a slot.
        const call = dist ? distCall : native.multiexpAsync;
        const run = (flags, stride) => async function (buffBases, buffScalars) {
            const n = Math.floor(buffBases.byteLength / stride);
            if (n * stride !== buffBases.byteLength) throw new RangeError("bases length is not a whole number of points");
            if (n === 0) return native.combine(C.id, g, new Uint8Array(0));
            const scalarSize = Math.floor(buffScalars.byteLength / n);
            if (scalarSize * n !== buffScalars.byteLength) throw new RangeError("scalars length is not a multiple of the number of points");
            if (n <= maxSlice) return call(ctx, C.id, g, flags, buffBases, buffScalars, scalarSize, n);
            const parts = [];
            for (let i = 0; i < n; i += maxSlice) {
                const m = Math.min(maxSlice, n - i);
                parts.push(call(ctx, C.id, g, flags, buffBases.subarray(i * stride, (i + m) * stride),
                                buffScalars.subarray(i * scalarSize, (i + m) * scalarSize), scalarSize, m));
                if (parts.length >= 2) await parts[parts.length - 2];      // at most two slices in flight (the engine's two slots)
            }
            const res = await Promise.all(parts);
            const jsz = 3 * fsz, all = new Uint8Array(res.length * jsz);
            res.forEach((r, k) => all.set(r, k * jsz));
            return native.combine(C.id, g, all);
        };
        return {
            F: { n8: fsz },
            // a prover's fixed bases (SRS): upload once, then only the scalars cross PCIe on every call
            uploadBases: (buffBases) => ({ handle: native.uploadBases(ctx, buffBases), n: Math.floor(buffBases.byteLength / (2 * fsz)), jacobian: false }),
            multiExpResident: async (bases, buffScalars) => {
                const scalarSize = Math.floor(buffScalars.byteLength / bases.n);
                return native.multiexpResident(ctx, C.id, g, 0, bases.handle, buffScalars, scalarSize, bases.n);
            },
            // the same SRS PREPARED (msm_amd_prepare_bases): its window table is built on the GPU once, every multiExp on it sorts all
            // windows into one bucket set -- 13 instead of 16 bucket additions per pair at n = 2^20, a quarter less time at 2^16..2^18.
            // scalarBits: the table serves scalars below 2^scalarBits (default: the curve order's bit length; a wider scalar rejects)
            prepareBases: (buffBases, opts2) => {
                const jac = !!(opts2 && opts2.jacobian);
                const n = Math.floor(buffBases.byteLength / ((jac ? 3 : 2) * fsz));
                const handle = native.prepareBases(ctx, C.id, g, jac ? FLAG_JACOBIAN : 0, buffBases, n, (opts2 && opts2.scalarBits) || 0);
                return { handle, n, info: () => native.basesInfo(handle) };
            },
            // what a table for n points would take, before building it: { c, windows, signedWindows, buckets, tableBytes, .. }
            planPrepared: (n, opts2) => native.planPrepared(C.id, g, n, (opts2 && opts2.scalarBits) || 0, (opts2 && opts2.windowBits) || 0),
            multiExpPrepared: (prepared, buffScalars) => {
                const scalarSize = Math.floor(buffScalars.byteLength / prepared.n);
                return native.multiexpPreparedAsync(ctx, prepared.handle, buffScalars, scalarSize, prepared.n);
            },
            multiExp: run(FLAG_JACOBIAN, 3 * fsz),
            multiExpAffine: run(0, 2 * fsz),
            // buffer-in / buffer-out forms of the batch conversions (ffjavascript exposes them under these names)
            inCurveAffine: async (buff) => native.batchCheck(ctx, 0, C.id, g, buff),
            inGroupAffine: async (buff) => native.batchCheck(ctx, 1, C.id, g, buff),
            batchToAffine: async (buff) => native.batchConvert(ctx, 0, C.id, g, buff),
            batchToJacobian: async (buff) => native.batchConvert(ctx, 1, C.id, g, buff),
            batchLEMtoU: async (buff) => native.batchConvert(ctx, 2, C.id, g, buff),
            batchUtoLEM: async (buff) => native.batchConvert(ctx, 3, C.id, g, buff),
            batchLEMtoC: async (buff) => native.batchConvert(ctx, 4, C.id, g, buff),
            batchCtoLEM: async (buff) => native.batchConvert(ctx, 5, C.id, g, buff),
            // Jacobian in, normalised Jacobian out (ffjavascript: G.fft / G.ifft / G.batchApplyKey)
            fft: async (buff) => native.fft(ctx, C.id, g, 0, buff),
            ifft: async (buff) => native.fft(ctx, C.id, g, 1, buff),
            batchApplyKey: async (buff, first, inc, inType = "jacobian") => native.batchApplyKey(ctx, C.id, g, inType === "affine" ? 0 : 1, buff, first, inc),
            normalize: (p) => native.normalize(C.id, g, p),
            toAffine: (p) => native.toAffine(C.id, g, p),
            eq: (a, b) => native.eq(C.id, g, a, b),
        };
    };
    // scalar field: Montgomery elements of 32 bytes
    const Fr = {
        n8: 32,
        fft: async (buff) => native.fft(ctx, C.id, 0, 0, buff),
        ifft: async (buff) => native.fft(ctx, C.id, 0, 1, buff),
        batchApplyKey: async (buff, first, inc) => native.batchApplyKey(ctx, C.id, 0, 0, buff, first, inc),
    };
    return { name: curveName, n8q: C.n8, n8r: 32, Fr, G1: group(1), G2: group(2), setWindowBits: (c) => native.setWindowBits(ctx, c),
        // {ranks, dead} of this process's multi-GPU group (msm_amd_dist_comm_count): the ranks the communicator itself counts, and whether an
        // exchange has timed out or failed -- a dead group rejects every call at once until it is destroyed and initialised anew
        distInfo: () => native.distCommCount(ctx) };
}

/** 128-byte id of a multi-GPU group: rank 0 creates it and hands it to the other processes (msm_amd_dist_unique_id) */
const distUniqueId = () => native.distUniqueId();

module.exports = { instantiate, curve, native, CURVES, distUniqueId, FLAG_JACOBIAN, FLAG_BASES_IN_SUBGROUP: 16 };
