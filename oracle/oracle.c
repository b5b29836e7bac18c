/*
 * oracle.c -- CPU restatement of the reference MSM path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the MI355X engine.  It restates, in plain C, the
 * algorithm that the reference's WASM generators emit for
 *     g1m_multiexp / g1m_multiexpAffine / *_chunk   and   g2m_* equivalents
 * on BLS12-381 and BN254 ("bn128").  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / CPU baseline.  The product library (libmsm_amd.so) never links or calls it.
 *
 * Pinning: oracle outputs are compared (tests/test_oracle_golden.py) with
 *   - the reference's own known-answer tests (wasmcurves/test/batchAffine.js:1177-1255,
 *     wasmcurves/test/bn128.js:462-497), and
 *   - golden vectors produced by running the reference's prebuilt
 *     wasmcurves/build/{bls12381,bn128}.wasm under Node in the build container
 *     (oracle/gen_golden.js, outputs committed under tests/golden/).
 *
 * Every function names the reference file:line it follows (paths relative to
 * /root/reference/wasmcurves/src/).  The reference computes on 32-bit limbs held in 64-bit
 * WASM locals (build_f1m.js:466-777); here limbs are 64-bit with a 128-bit accumulator.
 * Both use R = 2^(64*n64) and always return the fully reduced representative in [0,q)
 * (build_f1m.js:754-765), so every field value -- and therefore every byte of every
 * Jacobian coordinate -- is identical.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned __int128 u128;

#define MAXW 6          /* 64-bit words per base-field element (BLS12-381: 6, BN254: 4) */
#define MAXE (2 * MAXW) /* words per extension-field element (Fp2)                       */

/* ------------------------------------------------------------------------------------ */
/* Curve parameters: bls12381/build_bls12381.js:22-23,99-173 ; bn128/build_bn128.js:20-21 */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    int n64;         /* build_f1m.js:31: n64 = floor((bitLength(q-1)-1)/64)+1 */
    u64 q[MAXW];
    u64 one[MAXW];   /* R mod q   (build_f1m.js:43 pOne)  */
    u64 r2[MAXW];    /* R^2 mod q (build_f1m.js:42 pR2)   */
    u64 np;          /* -q^-1 mod 2^64 (build_f1m.js:504 uses the 32-bit version) */
    u64 qm2[MAXW];   /* q-2, exponent for inversion       */
} Fp;

static Fp FPS[2];
static Fp FRS[2];   /* scalar fields Fr (frm_* of the reference), same Montgomery machinery with modulus r */
static int fps_ready = 0;

/* little-endian 64-bit words */
static const u64 Q_BLS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                             0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 Q_BN[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                            0x30644e72e131a029ULL};

/* ---- plain multiword integers: build_int.js:186-228 (_add), :232-277 (_sub), :148-181 (_gte) */
static u64 int_add(int n, const u64 *a, const u64 *b, u64 *r) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static u64 int_sub(int n, const u64 *a, const u64 *b, u64 *r) {
    u64 borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d; borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static int int_gte(int n, const u64 *a, const u64 *b) {
    for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static int int_is_zero(int n, const u64 *a) { u64 o = 0; for (int i = 0; i < n; i++) o |= a[i]; return o == 0; }
static int int_eq(int n, const u64 *a, const u64 *b) { u64 o = 0; for (int i = 0; i < n; i++) o |= a[i] ^ b[i]; return o == 0; }

/* ---- Fp in Montgomery form: build_f1m.js ------------------------------------------------ */

/* build_f1m.js:71-89: carry-out or r>=q  =>  r -= q */
static void fp_add(const Fp *F, const u64 *x, const u64 *y, u64 *r) {
    int n = F->n64;
    if (int_add(n, x, y, r)) int_sub(n, r, F->q, r);
    else if (int_gte(n, r, F->q)) int_sub(n, r, F->q, r);
}
/* build_f1m.js:91-105: borrow => r += q */
static void fp_sub(const Fp *F, const u64 *x, const u64 *y, u64 *r) {
    int n = F->n64;
    if (int_sub(n, x, y, r)) int_add(n, r, F->q, r);
}
/* build_f1m.js:107-117: neg = 0 - x  (so neg(0) = 0) */
static void fp_neg(const Fp *F, const u64 *x, u64 *r) {
    u64 z[MAXW] = {0};
    fp_sub(F, z, x, r);
}
/* build_f1m.js:466-777 (f1m_mul): CIOS Montgomery product, output fully reduced. */
static void fp_mul(const Fp *F, const u64 *x, const u64 *y, u64 *r) {
    int n = F->n64;
    u64 t[MAXW + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) { c += (u128)x[j] * y[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[n]; t[n] = (u64)c; t[n + 1] = (u64)(c >> 64);
        u64 m = t[0] * F->np;
        c = (u128)m * F->q[0] + t[0]; c >>= 64;
        for (int j = 1; j < n; j++) { c += (u128)m * F->q[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[n]; t[n - 1] = (u64)c; t[n] = t[n + 1] + (u64)(c >> 64);
    }
    if (t[n] || int_gte(n, t, F->q)) int_sub(n, t, F->q, t); /* build_f1m.js:754-765 */
    memcpy(r, t, n * sizeof(u64));
}
/* build_f1m.js:779-1076 (f1m_square): same value as mul(x,x). */
static void fp_square(const Fp *F, const u64 *x, u64 *r) { fp_mul(F, x, x, r); }
/* build_f1m.js:1089-1096 */
static void fp_to_mont(const Fp *F, const u64 *x, u64 *r) { fp_mul(F, x, F->r2, r); }
/* build_f1m.js:1098-1110: Montgomery-reduce (x || 0) == mul by 1 */
static void fp_from_mont(const Fp *F, const u64 *x, u64 *r) {
    u64 o[MAXW] = {1};
    fp_mul(F, x, o, r);
}
/* build_f1m.js:1112-1122 (fromMontgomery; int_inverseMod; toMontgomery).  The modular
 * inverse is unique, so Fermat's x^(q-2) gives the same bytes.  inverse(0) = 0 as in
 * the reference's inverseMod of 0 (not reached on the MSM path: normalize tests Z first). */
static void fp_inverse(const Fp *F, const u64 *x, u64 *r) {
    int n = F->n64;
    u64 acc[MAXW], base[MAXW];
    memcpy(acc, F->one, sizeof acc); memcpy(base, x, n * sizeof(u64));
    for (int i = 0; i < n * 64; i++) {
        if ((F->qm2[i >> 6] >> (i & 63)) & 1) fp_mul(F, acc, base, acc);
        fp_mul(F, base, base, base);
    }
    memcpy(r, acc, n * sizeof(u64));
}

static void fp_setup(Fp *F, int n64, const u64 *q) {
    memset(F, 0, sizeof *F);
    F->n64 = n64;
    memcpy(F->q, q, n64 * sizeof(u64));
    /* np = -q^-1 mod 2^64 by Newton iteration */
    u64 inv = 1;
    for (int i = 0; i < 7; i++) inv *= 2 - q[0] * inv;
    F->np = (u64)0 - inv;
    /* R mod q and R^2 mod q by repeated modular doubling (build_f1m.js:42-43) */
    u64 v[MAXW] = {1};
    for (int i = 0; i < 2 * n64 * 64; i++) {
        if (int_add(n64, v, v, v) || int_gte(n64, v, q)) int_sub(n64, v, q, v);
        if (i == n64 * 64 - 1) memcpy(F->one, v, sizeof v);
    }
    memcpy(F->r2, v, sizeof v);
    u64 two[MAXW] = {2};
    int_sub(n64, q, two, F->qm2);
}
static void ensure_init(void) {
    if (fps_ready) return;
    fp_setup(&FPS[0], 6, Q_BLS);
    fp_setup(&FPS[1], 4, Q_BN);
    {
        /* group orders: bls12381/build_bls12381.js:23, bn128/build_bn128.js:21 */
        static const u64 R_BLS[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
        static const u64 R_BN[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
        fp_setup(&FRS[0], 4, R_BLS);
        fp_setup(&FRS[1], 4, R_BN);
    }
    fps_ready = 1;
}

/* ---- generic field (Fp or Fp2 = Fp[u]/(u^2+1)): build_f2m.js, mulNonResidueFn="f1m_neg"
 *      (bls12381/build_bls12381.js:48, bn128/build_bn128.js:43) ---------------------------- */
typedef struct { const Fp *fp; int deg; int nw; /* words per element = deg*n64 */ } Fld;

static void f_zero(const Fld *K, u64 *r) { memset(r, 0, K->nw * sizeof(u64)); }
static void f_copy(const Fld *K, const u64 *x, u64 *r) { memmove(r, x, K->nw * sizeof(u64)); }
static void f_one(const Fld *K, u64 *r) { f_zero(K, r); memcpy(r, K->fp->one, K->fp->n64 * sizeof(u64)); }
static int f_is_zero(const Fld *K, const u64 *x) { return int_is_zero(K->nw, x); }
static int f_eq(const Fld *K, const u64 *x, const u64 *y) { return int_eq(K->nw, x, y); }
/* build_f1m.js:1318-1327 ; build_f2m.js:432-449 (c0 == R and c1 == 0) */
static int f_is_one(const Fld *K, const u64 *x) {
    int n = K->fp->n64;
    if (!int_eq(n, x, K->fp->one)) return 0;
    return K->deg == 1 ? 1 : int_is_zero(n, x + n);
}
/* build_f2m.js:34-52 */
static void f_add(const Fld *K, const u64 *x, const u64 *y, u64 *r) {
    int n = K->fp->n64;
    for (int d = 0; d < K->deg; d++) fp_add(K->fp, x + d * n, y + d * n, r + d * n);
}
/* build_f2m.js:74-92 */
static void f_sub(const Fld *K, const u64 *x, const u64 *y, u64 *r) {
    int n = K->fp->n64;
    for (int d = 0; d < K->deg; d++) fp_sub(K->fp, x + d * n, y + d * n, r + d * n);
}
/* build_f2m.js:94-112 */
static void f_neg(const Fld *K, const u64 *x, u64 *r) {
    int n = K->fp->n64;
    for (int d = 0; d < K->deg; d++) fp_neg(K->fp, x + d * n, r + d * n);
}
/* build_f2m.js:152-188: A=x0y0, B=x1y1, C=(x0+x1)(y0+y1); r0=A+nr*B, r1=C-(A+B), nr*B = -B */
static void f_mul(const Fld *K, const u64 *x, const u64 *y, u64 *r) {
    const Fp *F = K->fp;
    if (K->deg == 1) { fp_mul(F, x, y, r); return; }
    int n = F->n64;
    u64 A[MAXW], B[MAXW], C[MAXW], D[MAXW], r0[MAXW], r1[MAXW];
    fp_mul(F, x, y, A);
    fp_mul(F, x + n, y + n, B);
    fp_add(F, x, x + n, C);
    fp_add(F, y, y + n, D);
    fp_mul(F, C, D, C);
    fp_neg(F, B, r0);
    fp_add(F, A, r0, r0);
    fp_add(F, A, B, r1);
    fp_sub(F, C, r1, r1);
    memcpy(r, r0, n * sizeof(u64)); memcpy(r + n, r1, n * sizeof(u64));
}
/* build_f2m.js:211-252: AB=x0x1; r0=(x0+x1)(x0+nr*x1)-(AB+nr*AB); r1=2AB */
static void f_square(const Fld *K, const u64 *x, u64 *r) {
    const Fp *F = K->fp;
    if (K->deg == 1) { fp_square(F, x, r); return; }
    int n = F->n64;
    u64 AB[MAXW], APB[MAXW], APNB[MAXW], ABPNAB[MAXW], r0[MAXW], r1[MAXW];
    fp_mul(F, x, x + n, AB);
    fp_add(F, x, x + n, APB);
    fp_neg(F, x + n, APNB);
    fp_add(F, x, APNB, APNB);
    fp_neg(F, AB, ABPNAB);
    fp_add(F, ABPNAB, AB, ABPNAB);
    fp_mul(F, APB, APNB, r0);
    fp_sub(F, r0, ABPNAB, r0);
    fp_add(F, AB, AB, r1);
    memcpy(r, r0, n * sizeof(u64)); memcpy(r + n, r1, n * sizeof(u64));
}
/* build_f2m.js:378-408: t = x0^2 - nr*x1^2; r0 = x0/t; r1 = -x1/t */
static void f_inverse(const Fld *K, const u64 *x, u64 *r) {
    const Fp *F = K->fp;
    if (K->deg == 1) { fp_inverse(F, x, r); return; }
    int n = F->n64;
    u64 t0[MAXW], t1[MAXW], t2[MAXW], t3[MAXW], r0[MAXW], r1[MAXW];
    fp_square(F, x, t0);
    fp_square(F, x + n, t1);
    fp_neg(F, t1, t2);
    fp_sub(F, t0, t2, t2);
    fp_inverse(F, t2, t3);
    fp_mul(F, x, t3, r0);
    fp_mul(F, x + n, t3, r1);
    fp_neg(F, r1, r1);
    memcpy(r, r0, n * sizeof(u64)); memcpy(r + n, r1, n * sizeof(u64));
}

/* ---- group law for y^2 = x^3 + b, Jacobian coordinates: build_curve_jacobian_a0.js ------ */
/* A Jacobian point is X||Y||Z (3*nw words); affine is x||y (2*nw words).                    */
#define PX(p) (p)
#define PY(p) ((p) + K->nw)
#define PZ(p) ((p) + 2 * K->nw)

/* build_curve_jacobian_a0.js:40-54 */
static int g_is_zero(const Fld *K, const u64 *p) { return f_is_zero(K, PZ(p)); }
/* :55-77 */
static int g_is_zero_affine(const Fld *K, const u64 *p) { return f_is_zero(K, PX(p)) && f_is_zero(K, PY(p)); }
/* :79-98 */
static void g_copy(const Fld *K, const u64 *s, u64 *d) { memmove(d, s, 3 * K->nw * sizeof(u64)); }
/* :124-150: (0, R, 0) */
static void g_zero(const Fld *K, u64 *r) { f_zero(K, PX(r)); f_one(K, PY(r)); f_zero(K, PZ(r)); }
/* :1014-1038 */
static void g_to_jacobian(const Fld *K, const u64 *p, u64 *r) {
    if (g_is_zero_affine(K, p)) { g_zero(K, r); return; }
    u64 t[2 * MAXE];
    memcpy(t, p, 2 * K->nw * sizeof(u64));
    f_copy(K, t, PX(r)); f_copy(K, t + K->nw, PY(r)); f_one(K, PZ(r));
}

/* :362-428 (doubleAffine; dbl-2007-bl with Z1=1) */
static void g_double_affine(const Fld *K, const u64 *p, u64 *r) {
    if (g_is_zero_affine(K, p)) { g_to_jacobian(K, p, r); return; }
    u64 XX[MAXE], YY[MAXE], YYYY[MAXE], S[MAXE], M[MAXE], e8[MAXE], x3[MAXE], y3[MAXE], z3[MAXE];
    f_square(K, PX(p), XX);
    f_square(K, PY(p), YY);
    f_square(K, YY, YYYY);
    f_add(K, PX(p), YY, S); f_square(K, S, S); f_sub(K, S, XX, S); f_sub(K, S, YYYY, S); f_add(K, S, S, S);
    f_add(K, XX, XX, M); f_add(K, M, XX, M);
    f_add(K, PY(p), PY(p), z3);
    f_square(K, M, x3); f_sub(K, x3, S, x3); f_sub(K, x3, S, x3);
    f_add(K, YYYY, YYYY, e8); f_add(K, e8, e8, e8); f_add(K, e8, e8, e8);
    f_sub(K, S, x3, y3); f_mul(K, y3, M, y3); f_sub(K, y3, e8, y3);
    f_copy(K, x3, PX(r)); f_copy(K, y3, PY(r)); f_copy(K, z3, PZ(r));
}

/* :291-359 (double; dbl-2009-l) with the isZero and Z==R branches :315-328 */
static void g_double(const Fld *K, const u64 *p, u64 *r) {
    if (g_is_zero(K, p)) { g_copy(K, p, r); return; }
    if (f_is_one(K, PZ(p))) { g_double_affine(K, p, r); return; }
    u64 A[MAXE], B[MAXE], C[MAXE], D[MAXE], E[MAXE], F[MAXE], G[MAXE], e8[MAXE], x3[MAXE], y3[MAXE], z3[MAXE];
    f_square(K, PX(p), A);
    f_square(K, PY(p), B);
    f_square(K, B, C);
    f_add(K, PX(p), B, D); f_square(K, D, D); f_sub(K, D, A, D); f_sub(K, D, C, D); f_add(K, D, D, D);
    f_add(K, A, A, E); f_add(K, E, A, E);
    f_square(K, E, F);
    f_mul(K, PY(p), PZ(p), G);
    f_add(K, D, D, x3); f_sub(K, F, x3, x3);
    f_add(K, C, C, e8); f_add(K, e8, e8, e8); f_add(K, e8, e8, e8);
    f_sub(K, D, x3, y3); f_mul(K, y3, E, y3); f_sub(K, y3, e8, y3);
    f_add(K, G, G, z3);
    f_copy(K, x3, PX(r)); f_copy(K, y3, PY(r)); f_copy(K, z3, PZ(r));
}

/* :764-847 (addAffine; both operands affine, result Jacobian) */
static void g_add_affine(const Fld *K, const u64 *p1, const u64 *p2, u64 *r) {
    if (g_is_zero_affine(K, p1)) { u64 t[2 * MAXE]; memcpy(t, p2, 2 * K->nw * 8); memcpy(r, t, 2 * K->nw * 8); f_one(K, PZ(r)); return; }
    if (g_is_zero_affine(K, p2)) { u64 t[2 * MAXE]; memcpy(t, p1, 2 * K->nw * 8); memcpy(r, t, 2 * K->nw * 8); f_one(K, PZ(r)); return; }
    if (f_eq(K, PX(p1), PX(p2)) && f_eq(K, PY(p1), PY(p2))) { g_double_affine(K, p2, r); return; }
    u64 H[MAXE], HH[MAXE], ym[MAXE], I[MAXE], J[MAXE], rr[MAXE], r2[MAXE], V[MAXE], V2[MAXE], yJ[MAXE], x3[MAXE], y3[MAXE], z3[MAXE];
    f_sub(K, PX(p2), PX(p1), H);
    f_sub(K, PY(p2), PY(p1), ym);
    f_square(K, H, HH);
    f_add(K, HH, HH, I); f_add(K, I, I, I);
    f_mul(K, H, I, J);
    f_add(K, ym, ym, rr);
    f_mul(K, PX(p1), I, V);
    f_square(K, rr, r2);
    f_add(K, V, V, V2);
    f_sub(K, r2, J, x3); f_sub(K, x3, V2, x3);
    f_mul(K, PY(p1), J, yJ); f_add(K, yJ, yJ, yJ);
    f_sub(K, V, x3, y3); f_mul(K, y3, rr, y3); f_sub(K, y3, yJ, y3);
    f_add(K, H, H, z3);
    f_copy(K, x3, PX(r)); f_copy(K, y3, PY(r)); f_copy(K, z3, PZ(r));
}

/* :661-761 (addMixed; p1 Jacobian, p2 affine; madd-2007-bl) incl. every branch :697-733 */
static void g_add_mixed(const Fld *K, const u64 *p1, const u64 *p2, u64 *r) {
    if (g_is_zero(K, p1)) { u64 t[2 * MAXE]; memcpy(t, p2, 2 * K->nw * 8); memcpy(r, t, 2 * K->nw * 8); f_one(K, PZ(r)); return; }
    if (g_is_zero_affine(K, p2)) { g_copy(K, p1, r); return; }
    if (f_is_one(K, PZ(p1))) { g_add_affine(K, p1, p2, r); return; }
    u64 Z1Z1[MAXE], U2[MAXE], Z1c[MAXE], S2[MAXE], H[MAXE], HH[MAXE], sm[MAXE], I[MAXE], J[MAXE], rr[MAXE], r2[MAXE], V[MAXE], V2[MAXE], yJ[MAXE], x3[MAXE], y3[MAXE], z3[MAXE];
    f_square(K, PZ(p1), Z1Z1);
    f_mul(K, PX(p2), Z1Z1, U2);
    f_mul(K, PZ(p1), Z1Z1, Z1c);
    f_mul(K, PY(p2), Z1c, S2);
    if (f_eq(K, PX(p1), U2) && f_eq(K, PY(p1), S2)) { g_double_affine(K, p2, r); return; }
    f_sub(K, U2, PX(p1), H);
    f_sub(K, S2, PY(p1), sm);
    f_square(K, H, HH);
    f_add(K, HH, HH, I); f_add(K, I, I, I);
    f_mul(K, H, I, J);
    f_add(K, sm, sm, rr);
    f_mul(K, PX(p1), I, V);
    f_square(K, rr, r2);
    f_add(K, V, V, V2);
    f_sub(K, r2, J, x3); f_sub(K, x3, V2, x3);
    f_mul(K, PY(p1), J, yJ); f_add(K, yJ, yJ, yJ);
    f_sub(K, V, x3, y3); f_mul(K, y3, rr, y3); f_sub(K, y3, yJ, y3);
    f_add(K, PZ(p1), H, z3); f_square(K, z3, z3); f_sub(K, z3, Z1Z1, z3); f_sub(K, z3, HH, z3);
    f_copy(K, x3, PX(r)); f_copy(K, y3, PY(r)); f_copy(K, z3, PZ(r));
}

/* :541-658 (add; add-2007-bl) incl. branches :583-629 */
static void g_add(const Fld *K, const u64 *p1, const u64 *p2, u64 *r) {
    if (g_is_zero(K, p1)) { g_copy(K, p2, r); return; }
    if (g_is_zero(K, p2)) { g_copy(K, p1, r); return; }
    if (f_is_one(K, PZ(p1))) { g_add_mixed(K, p2, p1, r); return; }
    if (f_is_one(K, PZ(p2))) { g_add_mixed(K, p1, p2, r); return; }
    u64 Z1Z1[MAXE], Z2Z2[MAXE], U1[MAXE], U2[MAXE], Z1c[MAXE], Z2c[MAXE], S1[MAXE], S2[MAXE], H[MAXE], sm[MAXE], I[MAXE], J[MAXE], rr[MAXE], r2[MAXE], V[MAXE], V2[MAXE], SJ[MAXE], x3[MAXE], y3[MAXE], z3[MAXE];
    f_square(K, PZ(p1), Z1Z1);
    f_square(K, PZ(p2), Z2Z2);
    f_mul(K, PX(p1), Z2Z2, U1);
    f_mul(K, PX(p2), Z1Z1, U2);
    f_mul(K, PZ(p1), Z1Z1, Z1c);
    f_mul(K, PZ(p2), Z2Z2, Z2c);
    f_mul(K, PY(p1), Z2c, S1);
    f_mul(K, PY(p2), Z1c, S2);
    if (f_eq(K, U1, U2) && f_eq(K, S1, S2)) { g_double(K, p1, r); return; }
    f_sub(K, U2, U1, H);
    f_sub(K, S2, S1, sm);
    f_add(K, H, H, I); f_square(K, I, I);
    f_mul(K, H, I, J);
    f_add(K, sm, sm, rr);
    f_mul(K, U1, I, V);
    f_square(K, rr, r2);
    f_add(K, V, V, V2);
    f_sub(K, r2, J, x3); f_sub(K, x3, V2, x3);
    f_mul(K, S1, J, SJ); f_add(K, SJ, SJ, SJ);
    f_sub(K, V, x3, y3); f_mul(K, y3, rr, y3); f_sub(K, y3, SJ, y3);
    f_add(K, PZ(p1), PZ(p2), z3); f_square(K, z3, z3); f_sub(K, z3, Z1Z1, z3); f_sub(K, z3, Z2Z2, z3); f_mul(K, z3, H, z3);
    f_copy(K, x3, PX(r)); f_copy(K, y3, PY(r)); f_copy(K, z3, PZ(r));
}

/* :849-868 */
static void g_neg(const Fld *K, const u64 *p, u64 *r) {
    u64 y[MAXE]; f_neg(K, PY(p), y);
    f_copy(K, PX(p), PX(r)); f_copy(K, y, PY(r)); f_copy(K, PZ(p), PZ(r));
}
/* :940-973 (normalize: Z := R; infinity := (0,R,0)) */
static void g_normalize(const Fld *K, const u64 *p, u64 *r) {
    if (g_is_zero(K, p)) { g_zero(K, r); return; }
    u64 zi[MAXE], z2[MAXE], z3[MAXE], x3[MAXE], y3[MAXE];
    f_inverse(K, PZ(p), zi);
    f_square(K, zi, z2);
    f_mul(K, zi, z2, z3);
    f_mul(K, PX(p), z2, x3);
    f_mul(K, PY(p), z3, y3);
    f_copy(K, x3, PX(r)); f_copy(K, y3, PY(r)); f_one(K, PZ(r));
}
/* :977-1011 (toAffine; infinity := (0,0)) */
static void g_to_affine(const Fld *K, const u64 *p, u64 *r) {
    if (g_is_zero(K, p)) { f_zero(K, PX(r)); f_zero(K, PY(r)); return; }
    u64 t[3 * MAXE];
    g_normalize(K, p, t);
    memcpy(r, t, 2 * K->nw * sizeof(u64));
}
/* :173-238 (eq) -- cross-multiplied comparison (the isOne shortcuts give the same truth value) */
static int g_eq(const Fld *K, const u64 *p1, const u64 *p2) {
    if (g_is_zero(K, p1)) return g_is_zero(K, p2);
    if (g_is_zero(K, p2)) return 0;
    u64 Z1Z1[MAXE], Z2Z2[MAXE], U1[MAXE], U2[MAXE], Z1c[MAXE], Z2c[MAXE], S1[MAXE], S2[MAXE];
    f_square(K, PZ(p1), Z1Z1); f_square(K, PZ(p2), Z2Z2);
    f_mul(K, PX(p1), Z2Z2, U1); f_mul(K, PX(p2), Z1Z1, U2);
    f_mul(K, PZ(p1), Z1Z1, Z1c); f_mul(K, PZ(p2), Z2Z2, Z2c);
    f_mul(K, PY(p1), Z2c, S1); f_mul(K, PY(p2), Z1c, S2);
    return f_eq(K, U1, U2) && f_eq(K, S1, S2);
}
/* build_timesscalar.js:20-87 semantics (double-and-add, MSB first); used only to build
 * test inputs and expected values, as the reference's tests do (test/bn128.js:484-491). */
static void g_times_scalar(const Fld *K, const u64 *p, const u8 *scalar, u32 scalar_len, u64 *r) {
    u64 acc[3 * MAXE], base[3 * MAXE];
    g_zero(K, acc); g_copy(K, p, base);
    for (int i = (int)scalar_len * 8 - 1; i >= 0; i--) {
        g_double(K, acc, acc);
        if ((scalar[i >> 3] >> (i & 7)) & 1) g_add(K, acc, base, acc);
    }
    g_copy(K, acc, r);
}

/* ---- the MSM: build_multiexp.js ---------------------------------------------------------- */

/* build_multiexp.js:25-94 (getChunk).  The reference does an unaligned 32-bit load that may
 * run up to 3 bytes past the scalar (masked off); here out-of-range bytes read as 0. */
static u32 get_chunk(const u8 *scalars, size_t total_bytes, size_t scalar_off, u32 scalar_size, u32 start_bit, u32 chunk_size) {
    int bits_to_end = (int)scalar_size * 8 - (int)start_bit;
    u32 mask = ((int)chunk_size > bits_to_end) ? ((1u << bits_to_end) - 1) : ((1u << chunk_size) - 1);
    size_t off = scalar_off + (start_bit >> 3);
    u32 w = 0;
    for (int b = 0; b < 4; b++) if (off + b < total_bytes) w |= (u32)scalars[off + b] << (8 * b);
    return (w >> (start_bit & 7)) & mask;
}

/* build_multiexp.js:373-461 (reduceTable): recursive halving */
static void reduce_table(const Fld *K, u64 *table, int p) {
    if (p == 1) return;
    size_t sz = 3 * K->nw;
    size_t half = (size_t)1 << (p - 1);
    u64 *acc = table + (half - 1) * sz;
    for (size_t i = 0; i + 1 < half; i++) {
        u64 *it1 = table + i * sz, *it2 = table + (half + i) * sz;
        g_add(K, it1, it2, it1);
        g_add(K, acc, it2, acc);
    }
    reduce_table(K, table, p - 1);
    for (int k = p - 1; k > 0; k--) g_double(K, acc, acc);
    g_add(K, table, acc, table);
}

/* build_multiexp.js:96-249 (fnName_chunk) */
static int multiexp_chunk(const Fld *K, int affine, const u8 *bases, const u8 *scalars, u32 scalar_size, u32 n,
                          u32 start_bit, u32 chunk_size, u64 *out) {
    if (n == 0) { g_zero(K, out); return 0; }
    size_t sz = 3 * K->nw;
    size_t ntable = (size_t)1 << chunk_size;
    u64 *table = (u64 *)malloc(ntable * sz * sizeof(u64));
    if (!table) return -1;
    for (size_t j = 0; j < ntable; j++) g_zero(K, table + j * sz);
    size_t base_stride = (affine ? 2 : 3) * K->nw; /* n8b: curve:1426-1430 */
    size_t total = (size_t)n * scalar_size;
    u64 pt[3 * MAXE];
    for (u32 i = 0; i < n; i++) {
        u32 idx = get_chunk(scalars, total, (size_t)i * scalar_size, scalar_size, start_bit, chunk_size);
        if (idx) {
            u64 *slot = table + (size_t)(idx - 1) * sz;
            memcpy(pt, bases + (size_t)i * base_stride * 8, base_stride * 8);
            if (affine) g_add_mixed(K, slot, pt, slot); else g_add(K, slot, pt, slot);
        }
    }
    reduce_table(K, table, (int)chunk_size);
    g_copy(K, table, out);
    free(table);
    return 0;
}

/* build_multiexp.js:275-280: window size by clz(n) */
static const u8 TSIZES[32] = {17, 17, 17, 17, 17, 17, 17, 17, 17, 17, 16, 16, 15, 14, 13, 13,
                              12, 11, 10, 9,  8,  7,  7,  6,  5,  4,  3,  2,  1,  1,  1,  1};

/* build_multiexp.js:251-371 (fnName) */
static int multiexp(const Fld *K, int affine, const u8 *bases, const u8 *scalars, u32 scalar_size, u32 n, u64 *out) {
    u64 acc[3 * MAXE], aux[3 * MAXE];
    g_zero(K, acc);
    if (n == 0) { g_copy(K, acc, out); return 0; }
    u32 chunk_size = TSIZES[__builtin_clz(n)];
    u32 nchunks = (scalar_size * 8 - 1) / chunk_size + 1;
    for (int it_bit = (int)((nchunks - 1) * chunk_size); it_bit >= 0; it_bit -= (int)chunk_size) {
        if (!g_is_zero(K, acc)) for (u32 j = 0; j < chunk_size; j++) g_double(K, acc, acc);
        int rc = multiexp_chunk(K, affine, bases, scalars, scalar_size, n, (u32)it_bit, chunk_size, aux);
        if (rc) return rc;
        g_add(K, acc, aux, acc);
    }
    g_copy(K, acc, out);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* exported C entry points (loaded with ctypes by oracle/oracle.py)                      */
/* curve: 0 = bls12381, 1 = bn128 ; group: 1 = G1 (Fp), 2 = G2 (Fp2)                     */
/* All byte buffers use the reference's layout: little-endian, Montgomery form.          */
/* ------------------------------------------------------------------------------------ */
static int mk(Fld *K, int curve, int group) {
    ensure_init();
    if (curve < 0 || curve > 1 || group < 1 || group > 2) return -1;
    K->fp = &FPS[curve]; K->deg = group; K->nw = group * K->fp->n64;
    return 0;
}
#define EXPORT __attribute__((visibility("default")))

EXPORT int oracle_n8(int curve) { ensure_init(); return (curve == 0 || curve == 1) ? FPS[curve].n64 * 8 : -1; }

EXPORT int oracle_multiexp(int curve, int group, int affine, const u8 *bases, const u8 *scalars, u32 scalar_size, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 r[3 * MAXE]; int rc = multiexp(&K, affine, bases, scalars, scalar_size, n, r);
    if (!rc) memcpy(out, r, 3 * K.nw * 8);
    return rc;
}
EXPORT int oracle_multiexp_chunk(int curve, int group, int affine, const u8 *bases, const u8 *scalars, u32 scalar_size, u32 n,
                                 u32 start_bit, u32 chunk_size, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    if (chunk_size == 0 || chunk_size > 25) return -2;
    u64 r[3 * MAXE]; int rc = multiexp_chunk(&K, affine, bases, scalars, scalar_size, n, start_bit, chunk_size, r);
    if (!rc) memcpy(out, r, 3 * K.nw * 8);
    return rc;
}
EXPORT u32 oracle_get_chunk(const u8 *scalar, u32 scalar_size, u32 start_bit, u32 chunk_size) {
    return get_chunk(scalar, scalar_size, 0, scalar_size, start_bit, chunk_size);
}
#define LOAD(dst, src, words) u64 dst[3 * MAXE]; memcpy(dst, src, (words) * 8)
EXPORT int oracle_g_add(int curve, int group, const u8 *p1, const u8 *p2, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); LOAD(b, p2, 3 * K.nw); u64 r[3 * MAXE];
    g_add(&K, a, b, r); memcpy(out, r, 3 * K.nw * 8); return 0;
}
EXPORT int oracle_g_add_mixed(int curve, int group, const u8 *p1, const u8 *p2, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); LOAD(b, p2, 2 * K.nw); u64 r[3 * MAXE];
    g_add_mixed(&K, a, b, r); memcpy(out, r, 3 * K.nw * 8); return 0;
}
EXPORT int oracle_g_add_affine(int curve, int group, const u8 *p1, const u8 *p2, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 2 * K.nw); LOAD(b, p2, 2 * K.nw); u64 r[3 * MAXE];
    g_add_affine(&K, a, b, r); memcpy(out, r, 3 * K.nw * 8); return 0;
}
EXPORT int oracle_g_double(int curve, int group, const u8 *p1, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); u64 r[3 * MAXE];
    g_double(&K, a, r); memcpy(out, r, 3 * K.nw * 8); return 0;
}
EXPORT int oracle_g_neg(int curve, int group, const u8 *p1, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); u64 r[3 * MAXE];
    g_neg(&K, a, r); memcpy(out, r, 3 * K.nw * 8); return 0;
}
EXPORT int oracle_g_normalize(int curve, int group, const u8 *p1, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); u64 r[3 * MAXE];
    g_normalize(&K, a, r); memcpy(out, r, 3 * K.nw * 8); return 0;
}
EXPORT int oracle_g_to_affine(int curve, int group, const u8 *p1, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); u64 r[3 * MAXE];
    g_to_affine(&K, a, r); memcpy(out, r, 2 * K.nw * 8); return 0;
}
EXPORT int oracle_g_eq(int curve, int group, const u8 *p1, const u8 *p2) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); LOAD(b, p2, 3 * K.nw);
    return g_eq(&K, a, b);
}
EXPORT int oracle_g_times_scalar(int curve, int group, const u8 *p1, const u8 *scalar, u32 scalar_len, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw); u64 r[3 * MAXE];
    g_times_scalar(&K, a, scalar, scalar_len, r); memcpy(out, r, 3 * K.nw * 8); return 0;
}
/* out_affine[i] = toAffine(k_i * P) for n scalars: builds test inputs (test/batchAffine.js:1268-1276 pattern) */
EXPORT int oracle_g_batch_times_scalar_affine(int curve, int group, const u8 *p1, const u8 *scalars, u32 scalar_len, u32 n, u8 *out_affine) {
    Fld K; if (mk(&K, curve, group)) return -1;
    LOAD(a, p1, 3 * K.nw);
    for (u32 i = 0; i < n; i++) {
        u64 r[3 * MAXE], af[2 * MAXE];
        g_times_scalar(&K, a, scalars + (size_t)i * scalar_len, scalar_len, r);
        g_to_affine(&K, r, af);
        memcpy(out_affine + (size_t)i * 2 * K.nw * 8, af, 2 * K.nw * 8);
    }
    return 0;
}
EXPORT int oracle_f_mul(int curve, int group, const u8 *x, const u8 *y, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 a[MAXE], b[MAXE], r[MAXE]; memcpy(a, x, K.nw * 8); memcpy(b, y, K.nw * 8);
    f_mul(&K, a, b, r); memcpy(out, r, K.nw * 8); return 0;
}
EXPORT int oracle_f_square(int curve, int group, const u8 *x, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 a[MAXE], r[MAXE]; memcpy(a, x, K.nw * 8);
    f_square(&K, a, r); memcpy(out, r, K.nw * 8); return 0;
}
EXPORT int oracle_f_add(int curve, int group, const u8 *x, const u8 *y, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 a[MAXE], b[MAXE], r[MAXE]; memcpy(a, x, K.nw * 8); memcpy(b, y, K.nw * 8);
    f_add(&K, a, b, r); memcpy(out, r, K.nw * 8); return 0;
}
EXPORT int oracle_f_sub(int curve, int group, const u8 *x, const u8 *y, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 a[MAXE], b[MAXE], r[MAXE]; memcpy(a, x, K.nw * 8); memcpy(b, y, K.nw * 8);
    f_sub(&K, a, b, r); memcpy(out, r, K.nw * 8); return 0;
}
EXPORT int oracle_f_inverse(int curve, int group, const u8 *x, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 a[MAXE], r[MAXE]; memcpy(a, x, K.nw * 8);
    f_inverse(&K, a, r); memcpy(out, r, K.nw * 8); return 0;
}
/* ---- batch conversions either side of the MSM ------------------------------ */
/* build_curve_jacobian_a0.js:1040-1125 (batchToAffine): (X/Z^2, Y/Z^3); Z == 0 -> (0, 0) */
EXPORT int oracle_g_batch_to_affine(int curve, int group, const u8 *in, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    for (u32 i = 0; i < n; i++) {
        u64 a[3 * MAXE], r[2 * MAXE];
        memcpy(a, in + (size_t)i * 3 * K.nw * 8, 3 * K.nw * 8);
        g_to_affine(&K, a, r);
        memcpy(out + (size_t)i * 2 * K.nw * 8, r, 2 * K.nw * 8);
    }
    return 0;
}
/* build_curve_jacobian_a0.js:1014-1038,1418 (batchToJacobian) */
EXPORT int oracle_g_batch_to_jacobian(int curve, int group, const u8 *in, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    for (u32 i = 0; i < n; i++) {
        u64 a[2 * MAXE], r[3 * MAXE];
        memcpy(a, in + (size_t)i * 2 * K.nw * 8, 2 * K.nw * 8);
        g_to_jacobian(&K, a, r);
        memcpy(out + (size_t)i * 3 * K.nw * 8, r, 3 * K.nw * 8);
    }
    return 0;
}
/* build_curve_jacobian_a0.js:1330-1353 (inCurveAffine): y^2 == x^3 + b ; b per curve/group from
 * bls12381/build_bls12381.js:40-53 (4 ; (4,4)) and bn128/build_bn128.js:35-48 (3 ; the twist constant) */
static void curve_b(const Fld *K, int curve, u64 *b) {
    static const char *BN_B2[2] = {"19485874751759354771024239261021720505790618469301721065564631296452457478373",
                                   "266929791119991161246907387137283842545076965332900288569378510910307636690"};
    memset(b, 0, K->nw * 8);
    for (int c = 0; c < K->deg; c++) {
        u64 plain[MAXW] = {0};
        if (curve == 0) plain[0] = 4;
        else if (K->deg == 1) plain[0] = 3;
        else { /* decimal -> limbs */
            for (const char *d = BN_B2[c]; *d; d++) {
                u128 carry = (u128)(*d - '0');
                for (int i = 0; i < K->fp->n64; i++) { u128 t = (u128)plain[i] * 10 + carry; plain[i] = (u64)t; carry = t >> 64; }
            }
        }
        fp_to_mont(K->fp, plain, b + c * K->fp->n64);
    }
}
EXPORT int oracle_g_batch_in_curve_affine(int curve, int group, const u8 *in, u32 n, u8 *flags) {
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 b[MAXE]; curve_b(&K, curve, b);
    for (u32 i = 0; i < n; i++) {
        u64 a[2 * MAXE], y2[MAXE], x3[MAXE];
        memcpy(a, in + (size_t)i * 2 * K.nw * 8, 2 * K.nw * 8);
        f_square(&K, a + K.nw, y2);
        f_square(&K, a, x3); f_mul(&K, a, x3, x3); f_add(&K, x3, b, x3);
        flags[i] = (u8)f_eq(&K, y2, x3);
    }
    return 0;
}
/* bls12381/build_bls12381.js:1494-1576 (g1m_inGroupAffine) and :1370-1480 (g2m_inGroupAffine): 1 for affine infinity, 0 off the
 * curve, else membership in the order-r subgroup.  The reference decides membership with endomorphism identities (Scott 2021);
 * this restatement uses the definition, r*P == infinity, which is the same predicate on curve points.  BLS12-381 only (the
 * reference builds no such export for BN254). */
EXPORT int oracle_g_batch_in_group_affine(int curve, int group, const u8 *in, u32 n, u8 *flags) {
    static const u8 R_BLS[32] = {0x01, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0x02, 0xa4, 0xbd, 0x53,
                                 0x05, 0xd8, 0xa1, 0x09, 0x08, 0xd8, 0x39, 0x33, 0x48, 0x7d, 0x9d, 0x29, 0x53, 0xa7, 0xed, 0x73};
    if (curve != 0) return -1;
    Fld K; if (mk(&K, curve, group)) return -1;
    u64 b[MAXE]; curve_b(&K, curve, b);
    for (u32 i = 0; i < n; i++) {
        u64 a[2 * MAXE], y2[MAXE], x3[MAXE], j[3 * MAXE], rp[3 * MAXE];
        memcpy(a, in + (size_t)i * 2 * K.nw * 8, 2 * K.nw * 8);
        if (g_is_zero_affine(&K, a)) { flags[i] = 1; continue; }
        f_square(&K, a + K.nw, y2);
        f_square(&K, a, x3); f_mul(&K, a, x3, x3); f_add(&K, x3, b, x3);
        if (!f_eq(&K, y2, x3)) { flags[i] = 0; continue; }
        g_to_jacobian(&K, a, j);
        g_times_scalar(&K, j, R_BLS, 32, rp);
        flags[i] = (u8)g_is_zero(&K, rp);
    }
    return 0;
}
/* build_applykey.js:20-76 with opGtimesF = g?m_timesFr / g?m_timesFrAffine (bls12381/build_bls12381.js:79-92, :buildGTimesFr;
 * bn128/build_bn128.js:84-88):  t = first; for i: out[i] = timesScalar(in[i], fromMontgomery(t)); t = t * inc  (Fr, Montgomery).
 * jac_in = 1: Jacobian inputs (batchApplyKey), 0: affine inputs (batchApplyKeyMixed).  Output: Jacobian, some representative. */
EXPORT int oracle_g_batch_apply_key(int curve, int group, int jac_in, const u8 *in, u32 n, const u8 *first, const u8 *inc, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    const Fp *R = &FRS[curve];
    u64 t[4], incw[4], plain[4];
    memcpy(t, first, 32); memcpy(incw, inc, 32);
    size_t fsz = (size_t)K.nw * 8, isz = (jac_in ? 3 : 2) * fsz;
    for (u32 i = 0; i < n; i++) {
        u64 p[3 * MAXE], r[3 * MAXE];
        if (jac_in) memcpy(p, in + (size_t)i * isz, isz);
        else { u64 a[2 * MAXE]; memcpy(a, in + (size_t)i * isz, isz); g_to_jacobian(&K, a, p); }
        fp_from_mont(R, t, plain);
        g_times_scalar(&K, p, (const u8 *)plain, 32, r);
        memcpy(out + (size_t)i * 3 * fsz, r, 3 * fsz);
        fp_mul(R, t, incw, t);
    }
    return 0;
}
/* frm_batchApplyKey (build_applykey.js with opGtimesF = frm_mul): out[i] = in[i] * first * inc^i in Fr (Montgomery) */
EXPORT int oracle_fr_batch_apply_key(int curve, const u8 *in, u32 n, const u8 *first, const u8 *inc, u8 *out) {
    ensure_init();
    if (curve != 0 && curve != 1) return -1;
    const Fp *R = &FRS[curve];
    u64 t[4], incw[4];
    memcpy(t, first, 32); memcpy(incw, inc, 32);
    for (u32 i = 0; i < n; i++) {
        u64 x[4], r[4];
        memcpy(x, in + (size_t)i * 32, 32);
        fp_mul(R, x, t, r);
        memcpy(out + (size_t)i * 32, r, 32);
        fp_mul(R, t, incw, t);
    }
    return 0;
}
/* ---- compressed point form: build_curve_jacobian_a0.js:1168-1205 (LEMtoC), :1265-1328 (CtoLEM) ------------------------------ */
/* build_f1m.js:135-156: 0 for zero, -1 if the plain value is >= (q+1)/2, else 1 */
static int fp_sign(const Fp *F, const u64 *x) {
    if (int_is_zero(F->n64, x)) return 0;
    u64 pl[MAXW], half[MAXW];
    fp_from_mont(F, x, pl);
    /* (q+1)/2 */
    { u128 c = 1; for (int i = 0; i < F->n64; i++) { c += F->q[i]; half[i] = (u64)c; c >>= 64; }
      for (int i = 0; i < F->n64; i++) half[i] = (half[i] >> 1) | (i < F->n64 - 1 ? half[i + 1] << 63 : 0); }
    return int_gte(F->n64, pl, half) ? -1 : 1;
}
/* build_f2m.js:411-430: sign of c1 unless it is zero, then sign of c0 */
static int f_sign(const Fld *K, const u64 *x) {
    if (K->deg == 1) return fp_sign(K->fp, x);
    int s = fp_sign(K->fp, x + K->fp->n64);
    return s ? s : fp_sign(K->fp, x);
}
static void fp_pow_words(const Fp *F, const u64 *base, const u64 *e, u64 *r) {
    u64 acc[MAXW], b[MAXW];
    memcpy(acc, F->one, (size_t)F->n64 * 8); memcpy(b, base, (size_t)F->n64 * 8);
    for (int i = 0; i < F->n64 * 64; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(F, acc, b, acc);
        fp_mul(F, b, b, b);
    }
    memcpy(r, acc, (size_t)F->n64 * 8);
}
/* a square root in Fp, q = 3 mod 4 (both base fields): a^((q+1)/4); returns 0 if a is not a square.  The reference's f1m_sqrt is
 * Tonelli-Shanks (build_f1m.js); whichever root it returns, CtoLEM then fixes the sign, so only "a root" matters. */
static int fp_sqrt(const Fp *F, const u64 *a, u64 *r) {
    u64 e[MAXW], t[MAXW];
    { u128 c = 1; for (int i = 0; i < F->n64; i++) { c += F->q[i]; e[i] = (u64)c; c >>= 64; }
      for (int k = 0; k < 2; k++) for (int i = 0; i < F->n64; i++) e[i] = (e[i] >> 1) | (i < F->n64 - 1 ? e[i + 1] << 63 : 0); }
    fp_pow_words(F, a, e, r);
    fp_mul(F, r, r, t);
    return int_eq(F->n64, t, a);
}
/* a square root in Fp2 = Fp[u]/(u^2+1) (complex method); returns 0 if a is not a square */
static int f_sqrt(const Fld *K, const u64 *a, u64 *r) {
    const Fp *F = K->fp; int n = F->n64;
    if (K->deg == 1) return fp_sqrt(F, a, r);
    const u64 *a0 = a, *a1 = a + n;
    u64 t0[MAXW], t1[MAXW], alpha[MAXW], delta[MAXW], x0[MAXW], x1[MAXW], two[MAXW], inv[MAXW], chk[MAXE];
    if (int_is_zero(n, a1)) {
        if (fp_sqrt(F, a0, r)) { memset(r + n, 0, (size_t)n * 8); return 1; }
        fp_neg(F, a0, t0);
        if (!fp_sqrt(F, t0, r + n)) return 0;
        memset(r, 0, (size_t)n * 8); return 1;
    }
    fp_mul(F, a0, a0, t0); fp_mul(F, a1, a1, t1); fp_add(F, t0, t1, t0);
    if (!fp_sqrt(F, t0, alpha)) return 0;
    fp_add(F, F->one, F->one, two); fp_inverse(F, two, inv);
    fp_add(F, a0, alpha, delta); fp_mul(F, delta, inv, delta);
    if (!fp_sqrt(F, delta, x0)) {
        fp_sub(F, a0, alpha, delta); fp_mul(F, delta, inv, delta);
        if (!fp_sqrt(F, delta, x0)) return 0;
    }
    fp_add(F, x0, x0, t0); fp_inverse(F, t0, t1); fp_mul(F, a1, t1, x1);
    memcpy(r, x0, (size_t)n * 8); memcpy(r + n, x1, (size_t)n * 8);
    f_mul(K, r, r, chk);
    return f_eq(K, chk, a);
}
static void reverse_bytes(const u8 *in, size_t n, u8 *out);
/* LEMtoC with the intended reading of infinity (affine x = y = 0): x big-endian plain, 0x80 on the first byte if sign(y) == -1,
 * infinity = 0x40 00 .. 00.  (The reference tests infinity with the Jacobian _isZero, which looks past the affine input.) */
EXPORT int oracle_g_batch_lem_to_c(int curve, int group, const u8 *in, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    size_t fsz = (size_t)K.nw * 8, n8 = (size_t)K.fp->n64 * 8;
    for (u32 i = 0; i < n; i++) {
        u64 a[2 * MAXE]; u8 pl[MAXE * 8];
        memcpy(a, in + (size_t)i * 2 * fsz, 2 * fsz);
        u8 *o = out + (size_t)i * fsz;
        if (g_is_zero_affine(&K, a)) { memset(o, 0, fsz); o[0] = 0x40; continue; }
        for (int c = 0; c < K.deg; c++) { u64 t[MAXW]; fp_from_mont(K.fp, a + c * K.fp->n64, t); memcpy(pl + c * n8, t, n8); }
        reverse_bytes(pl, fsz, o);
        if (f_sign(&K, a + K.nw) == -1) o[0] |= 0x80;
    }
    return 0;
}
/* CtoLEM: flags from the first byte; y = sqrt(x^3 + b) with the sign the 0x80 flag asks for.  Returns 1 + index of the first x
 * for which x^3 + b is not a square (the reference's sqrt does not terminate there), 0 on success. */
EXPORT int oracle_g_batch_c_to_lem(int curve, int group, const u8 *in, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    size_t fsz = (size_t)K.nw * 8, n8 = (size_t)K.fp->n64 * 8;
    u64 b[MAXE]; curve_b(&K, curve, b);
    for (u32 i = 0; i < n; i++) {
        const u8 *p = in + (size_t)i * fsz; u8 *o = out + (size_t)i * 2 * fsz;
        if (p[0] & 0x40) { memset(o, 0, 2 * fsz); continue; }
        int greatest = p[0] & 0x80;
        u8 tmp[MAXE * 8], pl[MAXE * 8];
        memcpy(tmp, p, fsz); tmp[0] &= 0x3f;
        reverse_bytes(tmp, fsz, pl);
        u64 x[MAXE], y[MAXE], t[MAXE];
        for (int c = 0; c < K.deg; c++) { u64 w[MAXW]; memcpy(w, pl + c * n8, n8); fp_to_mont(K.fp, w, x + c * K.fp->n64); }
        f_square(&K, x, t); f_mul(&K, x, t, t); f_add(&K, t, b, t);
        if (!f_sqrt(&K, t, y)) return (int)i + 1;
        int s = f_sign(&K, y);
        if ((s == -1) != (greatest != 0)) f_neg(&K, y, y);
        memcpy(o, x, fsz); memcpy(o + fsz, y, fsz);
    }
    return 0;
}
/* build_fft.js:23-68: roots of unity of Fr.  maxBits = 2-adicity of r-1, nr = smallest quadratic non-residue >= 2,
 * w[maxBits] = nr^((r-1)/2^maxBits), w[s] = w[s+1]^2; INV2[b] = 2^-b.  All kept in Montgomery form. */
static void fr_pow(const Fp *R, const u64 *base, const u64 *e, int ewords, u64 *r) {
    u64 acc[4], b[4];
    memcpy(acc, R->one, 32); memcpy(b, base, 32);
    for (int i = 0; i < ewords * 64; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(R, acc, b, acc);
        fp_mul(R, b, b, b);
    }
    memcpy(r, acc, 32);
}
static int fr_roots(int curve, u64 roots[33][4], u64 *inv2pow /* [33][4] */) {
    const Fp *R = &FRS[curve];
    u64 rem[4], one_plain[4] = {1, 0, 0, 0}, half[4];
    memcpy(rem, R->q, 32); rem[0] -= 1;   /* r - 1 */
    int maxbits = 0;
    while (!(rem[0] & 1)) { for (int i = 0; i < 4; i++) rem[i] = (rem[i] >> 1) | (i < 3 ? rem[i + 1] << 63 : 0); maxbits++; }
    /* (r-1)/2 for the Euler criterion */
    memcpy(half, R->q, 32); half[0] -= 1;
    for (int i = 0; i < 4; i++) half[i] = (half[i] >> 1) | (i < 3 ? half[i + 1] << 63 : 0);
    u64 nr_plain[4] = {2, 0, 0, 0}, nr[4], t[4];
    for (;;) {
        fp_to_mont(R, nr_plain, nr);
        fr_pow(R, nr, half, 4, t);
        if (!int_eq(4, t, R->one)) break;
        nr_plain[0]++;
    }
    (void)one_plain;
    fr_pow(R, nr, rem, 4, roots[maxbits]);
    for (int s = maxbits - 1; s >= 0; s--) fp_mul(R, roots[s + 1], roots[s + 1], roots[s]);
    /* 2^-b: (r+1)/2 is 1/2 */
    u64 h_plain[4], h[4];
    memcpy(h_plain, R->q, 32);
    { u128 c = 1; for (int i = 0; i < 4; i++) { c += h_plain[i]; h_plain[i] = (u64)c; c >>= 64; } }
    for (int i = 0; i < 4; i++) h_plain[i] = (h_plain[i] >> 1) | (i < 3 ? h_plain[i + 1] << 63 : 0);
    fp_to_mont(R, h_plain, h);
    memcpy(inv2pow, R->one, 32);
    for (int b = 1; b <= maxbits; b++) fp_mul(R, inv2pow + 4 * (b - 1), h, inv2pow + 4 * b);
    return maxbits;
}
static u32 bitrev(u32 x, int bits) { u32 r = 0; for (int i = 0; i < bits; i++) if (x & (1u << i)) r |= 1u << (bits - 1 - i); return r; }
/* frm_fft / frm_ifft (build_fft.js:178-245 -> rawfft :247-394 -> fftFinal :396-516): bit-reversal permutation, log2 n stages of
 * butterflies (u, t = W*x2) -> (u + t, u - t) with W running over the powers of ROOTs[s]; the inverse transform is the forward one
 * followed by out[i] = x[(n - i) mod n] * 2^-bits.  Elements: Fr in Montgomery form; in place semantics, here in -> out. */
EXPORT int oracle_fr_fft(int curve, int inverse, const u8 *in, u32 n, u8 *out) {
    ensure_init();
    if ((curve != 0 && curve != 1) || n == 0 || (n & (n - 1))) return -1;
    const Fp *R = &FRS[curve];
    static u64 roots[2][33][4], inv2[2][33 * 4]; static int mb[2] = {0, 0};
    if (!mb[curve]) mb[curve] = fr_roots(curve, roots[curve], inv2[curve]);
    int bits = 0; while ((1u << bits) < n) bits++;
    if (bits > mb[curve]) return -1;
    u64 *x = (u64 *)malloc((size_t)n * 32);
    for (u32 i = 0; i < n; i++) memcpy(x + 4 * (size_t)bitrev(i, bits), in + (size_t)i * 32, 32);
    for (int s = 1; s <= bits; s++) {
        u32 m = 1u << s, h = m >> 1;
        for (u32 k = 0; k < n; k += m) {
            u64 w[4]; memcpy(w, R->one, 32);
            for (u32 j = 0; j < h; j++) {
                u64 *a = x + 4 * (size_t)(k + j), *b = x + 4 * (size_t)(k + j + h), t[4], u[4];
                fp_mul(R, b, w, t); memcpy(u, a, 32);
                fp_add(R, u, t, a); fp_sub(R, u, t, b);
                fp_mul(R, w, roots[curve][s], w);
            }
        }
    }
    for (u32 i = 0; i < n; i++) {
        const u64 *src = inverse ? x + 4 * (size_t)((n - i) & (n - 1)) : x + 4 * (size_t)i;
        u64 r[4];
        if (inverse) fp_mul(R, src, inv2[curve] + 4 * bits, r); else memcpy(r, src, 32);
        memcpy(out + (size_t)i * 32, r, 32);
    }
    free(x);
    return 0;
}
/* g?m_fft / g?m_ifft: the same transform over Jacobian points, W*x2 through g?m_timesFr (fromMontgomery, then timesScalar) */
EXPORT int oracle_g_fft(int curve, int group, int inverse, const u8 *in, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    if (n == 0 || (n & (n - 1))) return -1;
    const Fp *R = &FRS[curve];
    static u64 roots[2][33][4], inv2[2][33 * 4]; static int mb[2] = {0, 0};
    if (!mb[curve]) mb[curve] = fr_roots(curve, roots[curve], inv2[curve]);
    int bits = 0; while ((1u << bits) < n) bits++;
    if (bits > mb[curve]) return -1;
    size_t jw = 3 * (size_t)K.nw, jsz = jw * 8;
    u64 *x = (u64 *)malloc((size_t)n * jsz);
    for (u32 i = 0; i < n; i++) memcpy(x + jw * bitrev(i, bits), in + (size_t)i * jsz, jsz);
    for (int s = 1; s <= bits; s++) {
        u32 m = 1u << s, h = m >> 1;
        for (u32 k = 0; k < n; k += m) {
            u64 w[4]; memcpy(w, R->one, 32);
            for (u32 j = 0; j < h; j++) {
                u64 *a = x + jw * (k + j), *b = x + jw * (k + j + h), t[3 * MAXE], u[3 * MAXE], nt[3 * MAXE], plain[4];
                fp_from_mont(R, w, plain);
                g_times_scalar(&K, b, (const u8 *)plain, 32, t);
                g_copy(&K, a, u);
                g_add(&K, u, t, a);
                g_neg(&K, t, nt); g_add(&K, u, nt, b);
                fp_mul(R, w, roots[curve][s], w);
            }
        }
    }
    for (u32 i = 0; i < n; i++) {
        const u64 *src = inverse ? x + jw * ((n - i) & (n - 1)) : x + jw * i;
        u64 r[3 * MAXE];
        if (inverse) { u64 plain[4]; fp_from_mont(R, inv2[curve] + 4 * bits, plain); g_times_scalar(&K, src, (const u8 *)plain, 32, r); }
        else g_copy(&K, src, r);
        memcpy(out + (size_t)i * jsz, r, jsz);
    }
    free(x);
    return 0;
}
static void reverse_bytes(const u8 *in, size_t n, u8 *out) { for (size_t i = 0; i < n; i++) out[i] = in[n - 1 - i]; }
/* build_curve_jacobian_a0.js:1206-1231,1413 (LEMtoU / batchLEMtoU): affine little-endian Montgomery ->
 * x || y big-endian plain; infinity -> zeros with first byte 0x40 */
EXPORT int oracle_g_batch_lem_to_u(int curve, int group, const u8 *in, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    size_t fsz = (size_t)K.nw * 8, n8 = (size_t)K.fp->n64 * 8;
    for (u32 i = 0; i < n; i++) {
        u64 a[2 * MAXE]; u8 pl[2 * MAXE * 8];
        memcpy(a, in + (size_t)i * 2 * fsz, 2 * fsz);
        u8 *o = out + (size_t)i * 2 * fsz;
        if (g_is_zero_affine(&K, a)) { memset(o, 0, 2 * fsz); o[0] = 0x40; continue; }
        for (int c = 0; c < 2 * K.deg; c++) {
            u64 t[MAXW]; fp_from_mont(K.fp, a + c * K.fp->n64, t); memcpy(pl + c * n8, t, n8);
        }
        reverse_bytes(pl, fsz, o);
        reverse_bytes(pl + fsz, fsz, o + fsz);
    }
    return 0;
}
/* build_curve_jacobian_a0.js:1233-1258,1415 (UtoLEM / batchUtoLEM) */
EXPORT int oracle_g_batch_u_to_lem(int curve, int group, const u8 *in, u32 n, u8 *out) {
    Fld K; if (mk(&K, curve, group)) return -1;
    size_t fsz = (size_t)K.nw * 8, n8 = (size_t)K.fp->n64 * 8;
    for (u32 i = 0; i < n; i++) {
        const u8 *p = in + (size_t)i * 2 * fsz; u8 *o = out + (size_t)i * 2 * fsz; u8 pl[2 * MAXE * 8];
        if (p[0] & 0x40) { memset(o, 0, 2 * fsz); continue; }
        reverse_bytes(p, fsz, pl);
        reverse_bytes(p + fsz, fsz, pl + fsz);
        for (int c = 0; c < 2 * K.deg; c++) {
            u64 t[MAXW], r[MAXW]; memcpy(t, pl + c * n8, n8); fp_to_mont(K.fp, t, r); memcpy(o + c * n8, r, n8);
        }
    }
    return 0;
}

/* base-field conversions (f1m_toMontgomery / f1m_fromMontgomery) applied per Fp component */
EXPORT int oracle_f_to_montgomery(int curve, const u8 *x, u32 count, u8 *out) {
    ensure_init(); if (curve < 0 || curve > 1) return -1;
    const Fp *F = &FPS[curve];
    for (u32 i = 0; i < count; i++) {
        u64 a[MAXW], r[MAXW]; memcpy(a, x + (size_t)i * F->n64 * 8, F->n64 * 8);
        fp_to_mont(F, a, r); memcpy(out + (size_t)i * F->n64 * 8, r, F->n64 * 8);
    }
    return 0;
}
EXPORT int oracle_f_from_montgomery(int curve, const u8 *x, u32 count, u8 *out) {
    ensure_init(); if (curve < 0 || curve > 1) return -1;
    const Fp *F = &FPS[curve];
    for (u32 i = 0; i < count; i++) {
        u64 a[MAXW], r[MAXW]; memcpy(a, x + (size_t)i * F->n64 * 8, F->n64 * 8);
        fp_from_mont(F, a, r); memcpy(out + (size_t)i * F->n64 * 8, r, F->n64 * 8);
    }
    return 0;
}
