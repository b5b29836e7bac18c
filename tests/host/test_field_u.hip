// Host-side check of the unsaturated field / curve code (csrc/field_u.hpp, csrc/curve_u.hpp) against the
// packed 64-bit host implementation (csrc/field_host.hpp + curve.hpp), which the GPU parity tests pin to the
// oracle.  The same templates run on the GPU; here they are compiled for the host only
// (hipcc --cuda-host-only) so the arithmetic can be exercised without a GPU.  Built and run by
// tests/test_field_u_host.py.
#include <cstdio>
#include <cstring>
#include <random>

// On the GPU G2 distributes an Fp2 element over a lane pair (Fp2d, exchanged with DPP moves): that form cannot run on the host, so
// the host check exercises the one-lane Fp2u kind (same formulas, same bound machinery); the lane-paired kernels are covered by
// the GPU parity tests against the oracle.
#define MSM_G2_PAIRED 0
#include "curve_u.hpp"
#include "field_host.hpp"

using namespace msm;

static std::mt19937_64 rng(12345);

template <class P>
static void rand_packed(uint32_t* w, int mode) {
    // value < q; mode 0 random, 1 zero, 2 one-ish small, 3 q-1
    const int N = P::N;
    if (mode == 1) { for (int i = 0; i < N; ++i) w[i] = 0; return; }
    if (mode == 2) { for (int i = 0; i < N; ++i) w[i] = 0; w[0] = 1 + (uint32_t)(rng() % 5); return; }
    if (mode == 3) { for (int i = 0; i < N; ++i) w[i] = P::Q[i]; w[0] -= 1; return; }
    for (;;) {
        for (int i = 0; i < N; ++i) w[i] = (uint32_t)rng();
        w[N - 1] &= (1u << ((P::MODBITS - 1) % 32 + 1)) - 1;
        // compare with q
        int lt = 0;
        for (int i = N - 1; i >= 0; --i) {
            if (w[i] < P::Q[i]) { lt = 1; break; }
            if (w[i] > P::Q[i]) { lt = 0; break; }
        }
        if (lt) return;
    }
}

template <class F, class HX>
static int check_point(const char* what, const XYZZU<typename UKind<F>::type>& u, const HX& h) {
    using K = typename UKind<F>::type;
    uint32_t got[4 * K::WORDS];
    u.to_packed(got);
    if (h.is_inf()) {
        if (!u.is_inf() || !words_all_zero(got + 2 * K::WORDS, K::WORDS)) { printf("FAIL %s: expected infinity\n", what); return 1; }
        return 0;
    }
    if (u.is_inf()) { printf("FAIL %s: unexpected infinity\n", what); return 1; }
    if (memcmp(got, &h, sizeof(h)) != 0) {
        printf("FAIL %s: coordinates differ\n", what);
        const uint32_t* e = reinterpret_cast<const uint32_t*>(&h);
        for (int i = 0; i < 4 * K::WORDS; ++i) if (got[i] != e[i]) { printf("  word %d: got %08x want %08x\n", i, got[i], e[i]); break; }
        return 1;
    }
    return 0;
}

template <class F>
static int run(const char* name) {
    using K = typename UKind<F>::type;
    using P = typename K::Params;
    using HF = typename HostOf<F>::type;
    using HX = XYZZ<HF>;
    using UX = XYZZU<K>;
    static_assert(sizeof(HX) == 4 * K::WORDS * 4, "packed size");
    int fails = 0;
    constexpr int CW = K::WORDS;
    const int NPTS = 64;
    static uint32_t pts[NPTS][2 * CW];
    for (int i = 0; i < NPTS; ++i) {
        int mode = i < 40 ? 0 : (i % 4);
        for (int c = 0; c < 2 * CW / P::N; ++c) rand_packed<P>(pts[i] + c * P::N, i < 40 ? 0 : ((i + c) % 4));
        (void)mode;
    }
    // chains of mixed additions with signs, including repeats (doubling) and opposite points
    for (int trial = 0; trial < 200; ++trial) {
        UX ua = UX::inf();
        HX ha = HX::inf();
        int len = 1 + (int)(rng() % 24);
        int prev = -1;
        for (int k = 0; k < len; ++k) {
            int idx = (int)(rng() % NPTS);
            bool neg = rng() & 1;
            int r = (int)(rng() % 16);
            if (r == 0 && prev >= 0) { idx = prev; }                    // same point again
            const uint32_t* w = pts[idx];
            if (words_all_zero(w, 2 * CW)) continue;
            // host
            typename HostOf<F>::type hx, hy;
            memcpy(&hx, w, CW * 4);
            memcpy(&hy, w + CW, CW * 4);
            bool hinf = hx.is_zero() && hy.is_zero();
            hy = HF::cneg(hy, neg && !hinf);
            HX::madd(ha, hx, hy);
            // unsaturated
            auto x2 = K::load(w);
            auto y2 = fcneg(K::load(w + CW), neg);
            UX::madd(ua, x2, y2);
            prev = idx;
            if (k == 0 && trial % 5 == 0) {   // add the same point (first addition: acc == P) -> doubling; then the opposite -> P
                HX::madd(ha, hx, hy);
                UX::madd(ua, x2, y2);
                fails += check_point<F>("madd doubling", ua, ha);
                auto y2n = fcneg(K::load(w + CW), !neg);
                HX::madd(ha, hx, HF::neg(hy));
                UX::madd(ua, x2, y2n);
                fails += check_point<F>("madd after doubling", ua, ha);
                // acc == P now; adding -P gives infinity
                HX t = ha;
                UX tu = ua;
                HX::madd(t, hx, HF::neg(hy));
                UX::madd(tu, x2, y2n);
                fails += check_point<F>("madd opposite", tu, t);
            }
        }
        char buf[64];
        snprintf(buf, sizeof buf, "madd chain %d", trial);
        fails += check_point<F>(buf, ua, ha);
        // full additions and doublings on the results
        static UX prev_u;
        static HX prev_h;
        if (trial > 0) {
            fails += check_point<F>("add", UX::add(ua, prev_u), HX::add(ha, prev_h));
            fails += check_point<F>("add_ni", UX::add_ni(prev_u, ua), HX::add(prev_h, ha));
            fails += check_point<F>("add self", UX::add(ua, ua), HX::add(ha, ha));
            fails += check_point<F>("dbl", UX::dbl(ua), HX::dbl(ha));
            fails += check_point<F>("add inf", UX::add(ua, UX::inf()), ha);
            fails += check_point<F>("inf add", UX::add(UX::inf(), ua), ha);
            // a + (-a): negate y
            UX na = ua;
            na.y = fstore<typename K::Y>(fmul(fneg(ua.y), K::one()));
            HX hn = HX::neg(ha);
            fails += check_point<F>("neg", na, hn);
            fails += check_point<F>("add opposite", UX::add(ua, na), HX::add(ha, hn));
            // long dependent chain: stresses the loop invariants of the stored types
            UX cu = ua;
            HX ch = ha;
            for (int it = 0; it < 50; ++it) {
                cu = UX::add(cu, prev_u);
                ch = HX::add(ch, prev_h);
                cu = UX::dbl(cu);
                ch = HX::dbl(ch);
            }
            fails += check_point<F>("add/dbl chain", cu, ch);
        }
        prev_u = ua;
        prev_h = ha;
    }
    printf("%s: %s\n", name, fails ? "FAILED" : "ok");
    return fails;
}

int main() {
    int fails = 0;
    fails += run<Fp<Bls12381Fp, true>>("bls12381 G1");
    fails += run<Fp<Bn254Fp, true>>("bn254 G1");
    fails += run<Fp2<Bls12381Fp>>("bls12381 G2");
    fails += run<Fp2<Bn254Fp>>("bn254 G2");
    return fails ? 1 : 0;
}
