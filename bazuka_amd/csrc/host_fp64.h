// Host-side BLS12-381 base field on 6 x 64-bit limbs (Montgomery, R = 2^384 - the byte layout of `Fp`'s 12 x 32-bit limbs on a
// little-endian host), for the few point operations an MSM leaves to the host: the Horner combine of the window sums (256 doublings +
// <= 32 additions) and the final to-affine inversion.  Round 3: these were 0.19 ms of a 4.0 ms MSM through the generic 32-bit-limb
// field code (62 ns per product incl. its additions: every add / sub walked 12 limbs with 64-bit temporaries); here a product is a
// fully unrolled CIOS on unsigned __int128 and additions are 6-limb carry chains.  Same values, so the packed affine result is
// byte-identical.  HFpOps / HFp2Ops satisfy the interface the generic XYZZ formulas (bzk_curve.cuh) expect.
#pragma once
#include <stdint.h>
#include <string.h>

#include "bzk_curve.cuh"

namespace bzk {

struct HFp {
    uint64_t l[6];
};

namespace hfp {
typedef unsigned __int128 u128;
struct Consts {
    uint64_t p[6], one[6], inv;
};
inline const Consts& consts() {
    static const Consts c = [] {
        Consts k;
        for (int i = 0; i < 6; ++i) {
            k.p[i] = (uint64_t)FpParams::MOD[2 * i] | ((uint64_t)FpParams::MOD[2 * i + 1] << 32);
            k.one[i] = (uint64_t)FpParams::ONE[2 * i] | ((uint64_t)FpParams::ONE[2 * i + 1] << 32);
        }
        const uint64_t x32 = (uint64_t)(uint32_t)(0u - FpParams::INV);  // p^-1 mod 2^32
        const uint64_t x64 = x32 * (2 - k.p[0] * x32);                  // one Newton step: p^-1 mod 2^64
        k.inv = (uint64_t)0 - x64;
        return k;
    }();
    return c;
}
inline bool geq_p(const uint64_t t[6], const uint64_t p[6]) {
    for (int i = 5; i >= 0; --i) {
        if (t[i] != p[i]) return t[i] > p[i];
    }
    return true;
}
inline void sub_p(uint64_t t[6], const uint64_t p[6]) {
    u128 b = 0;
    for (int i = 0; i < 6; ++i) {
        const u128 d = (u128)t[i] - p[i] - (uint64_t)b;
        t[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
}  // namespace hfp

struct HFpOps {
    typedef HFp T;
    static T zero() { return HFp{{0, 0, 0, 0, 0, 0}}; }
    static T one() {
        HFp r;
        memcpy(r.l, hfp::consts().one, 48);
        return r;
    }
    static bool is_zero(const T& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
    static bool eq(const T& a, const T& b) { return memcmp(a.l, b.l, 48) == 0; }
    static T add(const T& a, const T& b) {
        using namespace hfp;
        const Consts& k = consts();
        T r;
        u128 c = 0;
        for (int i = 0; i < 6; ++i) {
            c += (u128)a.l[i] + b.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq_p(r.l, k.p)) sub_p(r.l, k.p);  // p < 2^381: no carry out of six limbs
        return r;
    }
    static T sub(const T& a, const T& b) {
        using namespace hfp;
        const Consts& k = consts();
        T r;
        u128 bw = 0;
        for (int i = 0; i < 6; ++i) {
            const u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)bw;
            r.l[i] = (uint64_t)d;
            bw = (d >> 64) & 1;
        }
        if (bw) {
            u128 c = 0;
            for (int i = 0; i < 6; ++i) {
                c += (u128)r.l[i] + k.p[i];
                r.l[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return r;
    }
    static T neg(const T& a) { return is_zero(a) ? a : sub(zero(), a); }
    static T dbl(const T& a) { return add(a, a); }
    // Montgomery product, CIOS with the multiplication and reduction rows interleaved ("no-carry" form: the modulus' top bit is clear)
    static T mul(const T& a, const T& b) {
        using namespace hfp;
        const Consts& k = consts();
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6;
#define BZK_HFP_ROW(bi)                                                                       \
    {                                                                                          \
        u128 s = (u128)a.l[0] * (bi) + t0;                                                     \
        const uint64_t lo0 = (uint64_t)s;                                                      \
        uint64_t c = (uint64_t)(s >> 64);                                                      \
        s = (u128)a.l[1] * (bi) + t1 + c; t1 = (uint64_t)s; c = (uint64_t)(s >> 64);          \
        s = (u128)a.l[2] * (bi) + t2 + c; t2 = (uint64_t)s; c = (uint64_t)(s >> 64);          \
        s = (u128)a.l[3] * (bi) + t3 + c; t3 = (uint64_t)s; c = (uint64_t)(s >> 64);          \
        s = (u128)a.l[4] * (bi) + t4 + c; t4 = (uint64_t)s; c = (uint64_t)(s >> 64);          \
        s = (u128)a.l[5] * (bi) + t5 + c; t5 = (uint64_t)s; t6 = (uint64_t)(s >> 64);          \
        const uint64_t m = lo0 * k.inv;                                                        \
        s = (u128)m * k.p[0] + lo0; c = (uint64_t)(s >> 64);                                   \
        s = (u128)m * k.p[1] + t1 + c; t0 = (uint64_t)s; c = (uint64_t)(s >> 64);             \
        s = (u128)m * k.p[2] + t2 + c; t1 = (uint64_t)s; c = (uint64_t)(s >> 64);             \
        s = (u128)m * k.p[3] + t3 + c; t2 = (uint64_t)s; c = (uint64_t)(s >> 64);             \
        s = (u128)m * k.p[4] + t4 + c; t3 = (uint64_t)s; c = (uint64_t)(s >> 64);             \
        s = (u128)m * k.p[5] + t5 + c; t4 = (uint64_t)s; c = (uint64_t)(s >> 64);             \
        t5 = t6 + c;                                                                           \
    }
        BZK_HFP_ROW(b.l[0]) BZK_HFP_ROW(b.l[1]) BZK_HFP_ROW(b.l[2]) BZK_HFP_ROW(b.l[3]) BZK_HFP_ROW(b.l[4]) BZK_HFP_ROW(b.l[5])
#undef BZK_HFP_ROW
        T r = {{t0, t1, t2, t3, t4, t5}};
        if (geq_p(r.l, k.p)) sub_p(r.l, k.p);
        return r;
    }
    static T sqr(const T& a) { return mul(a, a); }
    static T inv(const T& a) {  // a^(p-2), fixed 4-bit windows: 380 squarings + <= 95 + 14 products (inv(0) = 0)
        using namespace hfp;
        uint64_t e[6];
        memcpy(e, consts().p, 48);
        e[0] -= 2;  // p is odd and its low limb is >= 2
        T pw[16];
        pw[0] = one();
        pw[1] = a;
        for (int i = 2; i < 16; ++i) pw[i] = mul(pw[i - 1], a);
        T r = one();
        for (int i = 95; i >= 0; --i) {  // 384 bits = 96 nibbles
            if (i != 95) { r = sqr(r); r = sqr(r); r = sqr(r); r = sqr(r); }
            const unsigned nib = (unsigned)(e[i >> 4] >> ((i & 15) * 4)) & 15u;
            if (nib) r = mul(r, pw[nib]);
        }
        return r;
    }
};

struct HFp2 {
    HFp c0, c1;
};
struct HFp2Ops {
    typedef HFp2 T;
    static T zero() { return {HFpOps::zero(), HFpOps::zero()}; }
    static T one() { return {HFpOps::one(), HFpOps::zero()}; }
    static bool is_zero(const T& a) { return HFpOps::is_zero(a.c0) && HFpOps::is_zero(a.c1); }
    static bool eq(const T& a, const T& b) { return HFpOps::eq(a.c0, b.c0) && HFpOps::eq(a.c1, b.c1); }
    static T add(const T& a, const T& b) { return {HFpOps::add(a.c0, b.c0), HFpOps::add(a.c1, b.c1)}; }
    static T sub(const T& a, const T& b) { return {HFpOps::sub(a.c0, b.c0), HFpOps::sub(a.c1, b.c1)}; }
    static T neg(const T& a) { return {HFpOps::neg(a.c0), HFpOps::neg(a.c1)}; }
    static T dbl(const T& a) { return {HFpOps::dbl(a.c0), HFpOps::dbl(a.c1)}; }
    static T mul(const T& a, const T& b) {  // Karatsuba over u^2 = -1
        const HFp aa = HFpOps::mul(a.c0, b.c0), bb = HFpOps::mul(a.c1, b.c1);
        const HFp s = HFpOps::mul(HFpOps::add(a.c0, a.c1), HFpOps::add(b.c0, b.c1));
        return {HFpOps::sub(aa, bb), HFpOps::sub(HFpOps::sub(s, aa), bb)};
    }
    static T sqr(const T& a) {
        const HFp s = HFpOps::add(a.c0, a.c1), d = HFpOps::sub(a.c0, a.c1), m = HFpOps::mul(a.c0, a.c1);
        return {HFpOps::mul(s, d), HFpOps::dbl(m)};
    }
    static T inv(const T& a) {
        const HFp d = HFpOps::inv(HFpOps::add(HFpOps::sqr(a.c0), HFpOps::sqr(a.c1)));
        return {HFpOps::mul(a.c0, d), HFpOps::mul(HFpOps::neg(a.c1), d)};
    }
};

// the fast host counterpart of a device-side field policy, and the (layout-preserving) conversions of XYZZ points
template <class F>
struct HostFast;
template <>
struct HostFast<FpOps> {
    typedef HFpOps Ops;
};
template <>
struct HostFast<Fp2Ops> {
    typedef HFp2Ops Ops;
};
template <class F>
inline XyzzT<typename HostFast<F>::Ops> to_host_fast(const XyzzT<F>& p) {
    static_assert(sizeof(XyzzT<F>) == sizeof(XyzzT<typename HostFast<F>::Ops>), "same limbs, same bytes");
    XyzzT<typename HostFast<F>::Ops> r;
    memcpy(&r, &p, sizeof r);
    return r;
}
template <class F>
inline XyzzT<F> from_host_fast(const XyzzT<typename HostFast<F>::Ops>& p) {
    XyzzT<F> r;
    memcpy(&r, &p, sizeof r);
    return r;
}

}  // namespace bzk
