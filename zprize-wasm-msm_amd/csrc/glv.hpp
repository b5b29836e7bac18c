// glv.hpp -- GLV pre-pass for BLS12-381 G1, host and device.
//
// Restates wasmcurves/src/build_glv.js:
//   _decomposeScalar (:53-146)  k -> (|k1|, |k2|, signs) with k = k1 + k2*lambda (mod r), |k_i| < 2^128,
//                               q1 = floor(k / r), q2 = floor(k * negV1 / r),
//                               k1 = k - q1 - q2*u0, k2 = q1*negV1 - q2      (constants :15-22)
//   _endomorphism (:150-163)    (x, y) -> (beta*x, sign ? y : -y)
//   _preprocessEndomorphism (:178-267)  n (point, scalar) pairs -> 2n pairs
// The reference works in 512-bit two's complement and divides with a generic int512_div; here the
// quotients come from one comparison ladder (q1 <= 2 because k < 2^256 < 3r) and a fixed-point
// reciprocal with a one-step correction (q2), which give the same integers.
#pragma once
#include <cstdint>

#include "field.hpp"

namespace msm {
namespace glv {

// little-endian 32-bit limbs
struct Consts {
    // r (group order), negV1 = -v1 = u0 - 1, u0 ; G = floor(negV1 * 2^384 / r) (257 bits -> 9 limbs)
    static constexpr uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    static constexpr uint32_t NEGV1[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};
    static constexpr uint32_t U0[4] = {0x00000000u, 0x00000001u, 0x0001a402u, 0xac45a401u};
    static constexpr uint32_t G[9] = {0x818be406u, 0xa1a872d6u, 0x27adc027u, 0x034eb4b9u, 0xf6cfee2eu, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u};
    // beta in Montgomery form (12 limbs)
    static constexpr uint32_t BETA[12] = {0x798a64e8u, 0x30f1361bu, 0x7ece5a2au, 0xf3b8ddabu, 0xc61577f7u, 0x16a8ca3au, 0x74fd029bu, 0xc26a2ff8u, 0x60701c6eu, 0x3636b766u, 0x241b6160u, 0x051ba4abu};
};

// r = a - b on n limbs, returns borrow
template <int N>
MSM_HD uint32_t sub_n(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t br = 0;
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)a[i] - b[i] - br;
        r[i] = (uint32_t)t;
        br = (t >> 32) & 1;
    }
    return (uint32_t)br;
}
template <int N>
MSM_HD bool gte_n(const uint32_t* a, const uint32_t* b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
// r[NA+NB] = a[NA] * b[NB]
template <int NA, int NB>
MSM_HD void mul_n(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    for (int i = 0; i < NA + NB; ++i) r[i] = 0;
    for (int i = 0; i < NA; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < NB; ++j) {
            uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + c;
            r[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        r[i + NB] = (uint32_t)c;
    }
}

// k: 8 limbs (any 256-bit value).  out: |k1| and |k2| as 4 limbs each (low 128 bits, as the reference stores them).
// Returns the sign word of build_glv.js: bit 0 set <=> k1 >= 0, bit 1 set <=> k2 >= 0.
MSM_HD uint32_t decompose(const uint32_t* k, uint32_t* k1abs, uint32_t* k2abs) {
    // q1 = floor(k / r) in {0, 1, 2}
    uint32_t r2[9];
    {
        uint32_t c = 0;
        for (int i = 0; i < 8; ++i) {
            uint32_t v = Consts::R[i];
            r2[i] = (v << 1) | c;
            c = v >> 31;
        }
        r2[8] = c;
    }
    uint32_t k9[9];
    for (int i = 0; i < 8; ++i) k9[i] = k[i];
    k9[8] = 0;
    uint32_t q1 = (gte_n<8>(k, Consts::R) ? 1u : 0u) + (gte_n<9>(k9, r2) ? 1u : 0u);
    // q2 = floor(k * negV1 / r): estimate floor(k * G / 2^384) (never too large, at most 1 too small), then correct
    uint32_t kg[17];
    mul_n<8, 9>(kg, k, Consts::G);
    uint32_t q2[6];
    for (int i = 0; i < 5; ++i) q2[i] = kg[12 + i];
    q2[5] = 0;
    uint32_t T[12];
    mul_n<8, 4>(T, k, Consts::NEGV1);            // k * negV1 < 2^384
    uint32_t qr[14];
    mul_n<6, 8>(qr, q2, Consts::R);              // q2 * r <= T
    uint32_t rem[12];
    (void)sub_n<12>(rem, T, qr);
    uint32_t r12[12];
    for (int i = 0; i < 12; ++i) r12[i] = i < 8 ? Consts::R[i] : 0u;
    if (gte_n<12>(rem, r12)) {                    // one correction step suffices
        uint64_t c = 1;
        for (int i = 0; i < 6; ++i) {
            uint64_t t = (uint64_t)q2[i] + c;
            q2[i] = (uint32_t)t;
            c = t >> 32;
        }
    }
    // k1 = k - (q1 + q2*u0)
    uint32_t B[10];
    mul_n<6, 4>(B, q2, Consts::U0);
    {
        uint64_t c = q1;
        for (int i = 0; i < 10 && c; ++i) {
            uint64_t t = (uint64_t)B[i] + c;
            B[i] = (uint32_t)t;
            c = t >> 32;
        }
    }
    uint32_t A[10];
    for (int i = 0; i < 10; ++i) A[i] = i < 8 ? k[i] : 0u;
    uint32_t d[10];
    uint32_t sign = 0;
    if (gte_n<10>(A, B)) {
        (void)sub_n<10>(d, A, B);
        sign |= 1u;
    } else {
        (void)sub_n<10>(d, B, A);
    }
    for (int i = 0; i < 4; ++i) k1abs[i] = d[i];
    // k2 = q1*negV1 - q2
    uint32_t C[10], q1l[1] = {q1}, D[10];
    {
        uint32_t t5[5];
        mul_n<1, 4>(t5, q1l, Consts::NEGV1);
        for (int i = 0; i < 10; ++i) C[i] = i < 5 ? t5[i] : 0u;
        for (int i = 0; i < 10; ++i) D[i] = i < 6 ? q2[i] : 0u;
    }
    if (gte_n<10>(C, D)) {
        (void)sub_n<10>(d, C, D);
        sign |= 2u;
    } else {
        (void)sub_n<10>(d, D, C);
    }
    for (int i = 0; i < 4; ++i) k2abs[i] = d[i];
    return sign;
}

// ------------------------------------------------------------------------------------------
// The engine's own split, used INSIDE the MSM (not a reference export): k == s * (k1 + k2*lambda) (mod r) with
// |k1| < 2^127 and 0 <= k2 < 2^127, lambda = -u0 (mod r), the eigenvalue of phi(x, y) = (beta*x, y) for the
// reference's beta (build_glv.js:15-22), so that k*P = s*(k1*P + k2*phi(P)); u0 = x^2 is the 128-bit integer the division uses.  Compared with build_glv.js's floor-based quotients the
// residue is first centred (k > (r-1)/2 -> r - k, remembered in s) and k2 is ROUNDED, so both halves stay one bit
// below 2^128: their signed radix-2^16 recoding then never carries out of the eighth window, and the MSM over
// the 2n half-length scalars needs exactly 8 windows.
//   kr = k mod r;  flip = kr > (r-1)/2;  kr = flip ? r - kr : kr
//   t = kr + floor(u0/2);  b = floor(t / u0);  k1 = (t mod u0) - floor(u0/2);  k2 = -b          (kr = k1 + b*u0 = k1 + k2*lambda mod r)
// Stores |k1| and b = |k2|.  Returns bit 0: the P term is negative (flip xor k1 < 0), bit 1: the phi(P) term is negative (not flip).
struct SplitConsts {
    static constexpr uint32_t RHALF[8] = {0x80000000u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};   // (r-1)/2
    static constexpr uint32_t LHALF[4] = {0x80000000u, 0x00000000u, 0x8000d201u, 0x5622d200u};   // floor(u0/2)
    static constexpr uint32_t M[9] = {0x818be409u, 0xa1a872d6u, 0x27adc027u, 0x034eb4b9u, 0xf6cfee2eu, 0x63f6e522u, 0xe01faaddu, 0x7c6becf1u, 0x00000001u};   // floor(2^384/u0)
};
MSM_HD uint32_t split_centered(const uint32_t* k, uint32_t* k1abs, uint32_t* k2abs) {
    // kr = k mod r  (k < 2^256 < 3r)
    uint32_t kr[8], tmp[8];
    for (int i = 0; i < 8; ++i) kr[i] = k[i];
    for (int rep = 0; rep < 2; ++rep) {
        if (gte_n<8>(kr, Consts::R)) {
            (void)sub_n<8>(tmp, kr, Consts::R);
            for (int i = 0; i < 8; ++i) kr[i] = tmp[i];
        }
    }
    uint32_t flip = 0;
    if (!gte_n<8>(SplitConsts::RHALF, kr)) {   // kr > (r-1)/2
        (void)sub_n<8>(tmp, Consts::R, kr);
        for (int i = 0; i < 8; ++i) kr[i] = tmp[i];
        flip = 1;
    }
    // t = kr + floor(u0/2)   (< 2^255)
    uint32_t t[8];
    {
        uint64_t c = 0;
        for (int i = 0; i < 8; ++i) {
            uint64_t v = (uint64_t)kr[i] + (i < 4 ? SplitConsts::LHALF[i] : 0u) + c;
            t[i] = (uint32_t)v;
            c = v >> 32;
        }
    }
    // q = floor(t / u0): Barrett estimate floor(t*M / 2^384) (never too large, at most 2 too small), then correct
    uint32_t tm[17];
    mul_n<8, 9>(tm, t, SplitConsts::M);
    uint32_t q[4] = {tm[12], tm[13], tm[14], tm[15]};
    uint32_t ql[8];
    mul_n<4, 4>(ql, q, Consts::U0);
    uint32_t rem[8];
    (void)sub_n<8>(rem, t, ql);   // < 3*u0 < 2^130
    uint32_t lam5[5] = {Consts::U0[0], Consts::U0[1], Consts::U0[2], Consts::U0[3], 0u};
    for (int rep = 0; rep < 2; ++rep) {
        if (gte_n<5>(rem, lam5)) {
            uint32_t r5[5];
            (void)sub_n<5>(r5, rem, lam5);
            for (int i = 0; i < 5; ++i) rem[i] = r5[i];
            uint64_t c = 1;
            for (int i = 0; i < 4; ++i) {
                uint64_t v = (uint64_t)q[i] + c;
                q[i] = (uint32_t)v;
                c = v >> 32;
            }
        }
    }
    for (int i = 0; i < 4; ++i) k2abs[i] = q[i];
    // k1 = rem - floor(u0/2), |k1| <= u0/2
    uint32_t neg1 = 0;
    if (gte_n<4>(rem, SplitConsts::LHALF)) {
        (void)sub_n<4>(k1abs, rem, SplitConsts::LHALF);
    } else {
        (void)sub_n<4>(k1abs, SplitConsts::LHALF, rem);
        neg1 = 1;
    }
    return (flip ^ neg1) | ((flip ^ 1u) << 1);
}

}  // namespace glv
}  // namespace msm
