// field_host.hpp -- the same Montgomery field as field.hpp, for the HOST side of the engine
// (window Horner combine, normalisation, the comparison helpers of the C ABI): 64-bit limbs with a
// 128-bit accumulator.  Same R = 2^(32*N), same fully reduced values, same byte image as Fp<P>,
// so device results are memcpy'd into these types.  Semantics: wasmcurves/src/build_f1m.js
// (add :71-89, sub :91-105, mul :466-777, toMontgomery :1089-1096, fromMontgomery :1098-1110).
#pragma once
#include <cstdint>
#include <cstring>

#include "field.hpp"

namespace msm {

template <class P>
struct FpH {
    static constexpr int N = P::N64;
    static constexpr int BYTES = 8 * P::N64;
    using Params = P;
    using u128 = unsigned __int128;
    uint64_t v[N];

    static uint64_t q(int i) { return (uint64_t)P::Q[2 * i] | ((uint64_t)P::Q[2 * i + 1] << 32); }
    static FpH zero() {
        FpH r;
        for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    static FpH one() {
        FpH r;
        for (int i = 0; i < N; ++i) r.v[i] = (uint64_t)P::ONE[2 * i] | ((uint64_t)P::ONE[2 * i + 1] << 32);
        return r;
    }
    bool is_zero() const {
        uint64_t o = 0;
        for (int i = 0; i < N; ++i) o |= v[i];
        return o == 0;
    }
    bool is_one() const { return eq(*this, one()); }
    static bool eq(const FpH& a, const FpH& b) {
        uint64_t o = 0;
        for (int i = 0; i < N; ++i) o |= a.v[i] ^ b.v[i];
        return o == 0;
    }
    static bool gte_q(const uint64_t* t) {
        for (int i = N - 1; i >= 0; --i) {
            if (t[i] > q(i)) return true;
            if (t[i] < q(i)) return false;
        }
        return true;
    }
    static void sub_q(uint64_t* t) {
        uint64_t br = 0;
        for (int i = 0; i < N; ++i) {
            u128 d = (u128)t[i] - q(i) - br;
            t[i] = (uint64_t)d;
            br = (uint64_t)(d >> 64) & 1;
        }
    }
    static FpH add(const FpH& a, const FpH& b) {
        FpH r;
        u128 c = 0;
        for (int i = 0; i < N; ++i) {
            c += (u128)a.v[i] + b.v[i];
            r.v[i] = (uint64_t)c;
            c >>= 64;
        }
        if (c || gte_q(r.v)) sub_q(r.v);
        return r;
    }
    static FpH dbl(const FpH& a) { return add(a, a); }
    static FpH sub(const FpH& a, const FpH& b) {
        FpH r;
        uint64_t br = 0;
        for (int i = 0; i < N; ++i) {
            u128 d = (u128)a.v[i] - b.v[i] - br;
            r.v[i] = (uint64_t)d;
            br = (uint64_t)(d >> 64) & 1;
        }
        if (br) {
            u128 c = 0;
            for (int i = 0; i < N; ++i) {
                c += (u128)r.v[i] + q(i);
                r.v[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    static FpH neg(const FpH& a) { return sub(zero(), a); }
    static FpH canon(const FpH& a) { return a; }
    static FpH cneg(const FpH& a, bool f) { return f ? neg(a) : a; }
    static FpH mul(const FpH& a, const FpH& b) {
        uint64_t t[N + 2];
        for (int i = 0; i < N + 2; ++i) t[i] = 0;
        for (int i = 0; i < N; ++i) {
            u128 c = 0;
            for (int j = 0; j < N; ++j) {
                c += (u128)a.v[j] * b.v[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N] = (uint64_t)c;
            t[N + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * P::NP64;
            c = (u128)m * q(0) + t[0];
            c >>= 64;
            for (int j = 1; j < N; ++j) {
                c += (u128)m * q(j) + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[N];
            t[N - 1] = (uint64_t)c;
            t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        if (t[N] || gte_q(t)) sub_q(t);
        FpH r;
        for (int i = 0; i < N; ++i) r.v[i] = t[i];
        return r;
    }
    static FpH sqr(const FpH& a) { return mul(a, a); }
    static FpH inv(const FpH& x) {
        FpH acc = one(), base = x;
        for (int i = 0; i < 64 * N; ++i) {
            if ((P::QM2[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    static FpH to_mont(const FpH& x) {
        FpH r2;
        for (int i = 0; i < N; ++i) r2.v[i] = (uint64_t)P::R2[2 * i] | ((uint64_t)P::R2[2 * i + 1] << 32);
        return mul(x, r2);
    }
    static FpH from_mont(const FpH& x) {
        FpH o = zero();
        o.v[0] = 1;
        return mul(x, o);
    }
};

// device field type -> host twin with the same byte image
template <class F>
struct HostOf;
template <class P, bool I>
struct HostOf<Fp<P, I>> {
    using type = FpH<P>;
};
template <class P, bool I>
struct HostOf<Fp2T<Fp<P, I>>> {
    using type = Fp2T<FpH<P>>;
};

}  // namespace msm
