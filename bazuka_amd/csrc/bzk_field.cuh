// Montgomery prime-field arithmetic for gfx950 (and the host side of libbzk), 32-bit limbs.
//
// Fr = BLS12-381 scalar field  (= reference `ZkScalar`, /root/reference/src/zk/mod.rs:196-206)
// Fp = BLS12-381 base field    (= bls12_381::Fp behind /root/reference/src/zk/groth16/mod.rs:19-20)
// Memory form = little-endian Montgomery limbs; 8 x u32 == the reference's [u64;4], 12 x u32 ==
// [u64;6] byte for byte, so buffers cross the C ABI without conversion.
//
// Why 32-bit limbs: CDNA4's widest integer multiply is v_mad_u64_u32 (32x32+64 -> 64).  The
// multiplier below is an operand-scanning CIOS with the "no-carry" shortcut (both moduli leave the
// top bit of the top limb clear), written so that every step is one mad with a 64-bit addend.
// MFMA is deliberately unused: a 381-bit modular product is a carry chain, not a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define BZK_HD __host__ __device__ __forceinline__
#else
#define BZK_HD __host__ __device__ inline
#endif

namespace bzk {

struct FrParams {
    static constexpr int N = 8;
    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static constexpr uint32_t MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                        0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
    // R = 2^256 mod r
    static constexpr uint32_t ONE[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                        0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
    // R^2 mod r
    static constexpr uint32_t R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                       0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
};

struct FpParams {
    static constexpr int N = 12;
    // p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    static constexpr uint32_t MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                         0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    static constexpr uint32_t INV = 0xfffcfffdu;  // -p^-1 mod 2^32
    // R = 2^384 mod p
    static constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                         0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
    // R^2 mod p
    static constexpr uint32_t R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                        0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
};

template <class P>
struct alignas(16) Fe {
    static constexpr int N = P::N;
    uint32_t l[N];

    BZK_HD static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = 0;
        return r;
    }
    BZK_HD static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = P::ONE[i];
        return r;
    }
    BZK_HD static Fe r2() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = P::R2[i];
        return r;
    }
    BZK_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= l[i];
        return o == 0;
    }
    BZK_HD bool equals(const Fe& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= l[i] ^ b.l[i];
        return o == 0;
    }
};

// r = a - MOD if a >= MOD (a < 2*MOD)
template <class P>
BZK_HD void fe_reduce_once(Fe<P>& a) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t d = (uint64_t)a.l[i] - P::MOD[i] - borrow;
        t[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    if (!borrow) {
#pragma unroll
        for (int i = 0; i < N; ++i) a.l[i] = t[i];
    }
}

template <class P>
BZK_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    Fe<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    // both moduli have a clear top bit, so a+b < 2^(32N): no carry out
    fe_reduce_once<P>(r);
    return r;
}

template <class P>
BZK_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    Fe<P> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c += (uint64_t)r.l[i] + (P::MOD[i] & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class P>
BZK_HD Fe<P> fe_neg(const Fe<P>& a) {
    return fe_sub<P>(Fe<P>::zero(), a);
}

template <class P>
BZK_HD Fe<P> fe_dbl(const Fe<P>& a) {
    return fe_add<P>(a, a);
}

// Montgomery product a*b*R^-1 mod p.  CIOS, multiplication and reduction rows interleaved,
// no-carry variant (top bit of the modulus is clear, so the running value stays below 2^(32N+1)
// and the two row carries can simply be summed into the top limb).
#if !defined(__HIP_DEVICE_COMPILE__)
// Host build of the same product on 64-bit limbs (the byte layout is identical): used by the few
// host-side point operations (window Horner, proof assembly), ~5x faster than the 32-bit form.
template <class P>
inline Fe<P> fe_mul_host64(const Fe<P>& a, const Fe<P>& b) {
    constexpr int M = P::N / 2;
    typedef unsigned __int128 u128;
    uint64_t A[M], B[M], Q[M], t[M + 2];
    for (int i = 0; i < M; ++i) {
        A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        Q[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
    }
    // -p^-1 mod 2^64 from the 32-bit constant by one Newton step
    const uint64_t x32 = (uint64_t)(uint32_t)(0u - P::INV);  // p^-1 mod 2^32
    const uint64_t x64 = x32 * (2 - Q[0] * x32);
    const uint64_t inv = (uint64_t)0 - x64;
    for (int i = 0; i < M + 2; ++i) t[i] = 0;
    for (int i = 0; i < M; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < M; ++j) {
            u128 s = (u128)A[j] * B[i] + t[j] + c;
            t[j] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        u128 s = (u128)t[M] + c;
        t[M] = (uint64_t)s;
        t[M + 1] = (uint64_t)(s >> 64);
        const uint64_t m = t[0] * inv;
        s = (u128)m * Q[0] + t[0];
        c = (uint64_t)(s >> 64);
        for (int j = 1; j < M; ++j) {
            s = (u128)m * Q[j] + t[j] + c;
            t[j - 1] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        s = (u128)t[M] + c;
        t[M - 1] = (uint64_t)s;
        t[M] = t[M + 1] + (uint64_t)(s >> 64);
    }
    Fe<P> r;
    for (int i = 0; i < M; ++i) {
        r.l[2 * i] = (uint32_t)t[i];
        r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    fe_reduce_once<P>(r);
    return r;
}
#endif

template <class P>
BZK_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_host64<P>(a, b);
#else
    constexpr int N = P::N;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t bi = b.l[i];
        uint64_t s = (uint64_t)a.l[0] * bi + t[0];
        uint32_t c1 = (uint32_t)(s >> 32);
        const uint32_t m = (uint32_t)s * P::INV;
        uint64_t q = (uint64_t)m * P::MOD[0] + (uint32_t)s;
        uint32_t c2 = (uint32_t)(q >> 32);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            s = (uint64_t)a.l[j] * bi + t[j] + c1;
            c1 = (uint32_t)(s >> 32);
            q = (uint64_t)m * P::MOD[j] + (uint32_t)s + c2;
            c2 = (uint32_t)(q >> 32);
            t[j - 1] = (uint32_t)q;
        }
        t[N - 1] = c1 + c2;
    }
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = t[i];
    fe_reduce_once<P>(r);
    return r;
#endif
}

template <class P>
BZK_HD Fe<P> fe_sqr(const Fe<P>& a) {
    return fe_mul<P>(a, a);
}

template <class P>
BZK_HD Fe<P> fe_from_mont(const Fe<P>& a) {  // -> canonical limbs
    Fe<P> o = Fe<P>::zero();
    o.l[0] = 1;
    return fe_mul<P>(a, o);
}

template <class P>
BZK_HD Fe<P> fe_to_mont(const Fe<P>& a) {
    return fe_mul<P>(a, Fe<P>::r2());
}

// a^(p-2); inv(0) = 0.  Not unrolled (code size).
template <class P>
__host__ __device__ inline Fe<P> fe_inv(const Fe<P>& a) {
    constexpr int N = P::N;
    uint32_t e[N];
    {
        uint64_t borrow = 2;
        for (int i = 0; i < N; ++i) {
            uint64_t d = (uint64_t)P::MOD[i] - borrow;
            e[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
    }
    Fe<P> r = Fe<P>::one();
    for (int i = 32 * N - 1; i >= 0; --i) {
        r = fe_sqr<P>(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = fe_mul<P>(r, a);
    }
    return r;
}

typedef Fe<FrParams> Fr;
typedef Fe<FpParams> Fp;

// ---- Fp2 = Fp[u]/(u^2+1)
struct Fp2 {
    Fp c0, c1;
};

// Uniform static interface so curve code can be generic over Fp / Fp2.
struct FpOps {
    typedef Fp T;
    static constexpr int LIMBS = 12;
    BZK_HD static T zero() { return Fp::zero(); }
    BZK_HD static T one() { return Fp::one(); }
    BZK_HD static bool is_zero(const T& a) { return a.is_zero(); }
    BZK_HD static bool eq(const T& a, const T& b) { return a.equals(b); }
    BZK_HD static T add(const T& a, const T& b) { return fe_add<FpParams>(a, b); }
    BZK_HD static T sub(const T& a, const T& b) { return fe_sub<FpParams>(a, b); }
    BZK_HD static T neg(const T& a) { return fe_neg<FpParams>(a); }
    BZK_HD static T dbl(const T& a) { return fe_dbl<FpParams>(a); }
#if defined(BZK_FP_NOINLINE) && defined(__HIP_DEVICE_COMPILE__)
    // by-value signature: the AMDGPU calling convention passes the 24 limbs in VGPRs (no scratch)
    __device__ __noinline__ static T mul_call(T a, T b) { return fe_mul<FpParams>(a, b); }
    BZK_HD static T mul(const T& a, const T& b) { return mul_call(a, b); }
    BZK_HD static T sqr(const T& a) { return mul_call(a, a); }
#else
    BZK_HD static T mul(const T& a, const T& b) { return fe_mul<FpParams>(a, b); }
    BZK_HD static T sqr(const T& a) { return fe_sqr<FpParams>(a); }
#endif
    __host__ __device__ static T inv(const T& a) {  // a^(p-2) through mul/sqr above (inv(0) = 0)
        uint32_t e[12];
        uint64_t borrow = 2;
        for (int i = 0; i < 12; ++i) {
            uint64_t d = (uint64_t)FpParams::MOD[i] - borrow;
            e[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
        T r = one();
        for (int i = 383; i >= 0; --i) {
            r = sqr(r);
            if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, a);
        }
        return r;
    }
};

struct Fp2Ops {
    typedef Fp2 T;
    static constexpr int LIMBS = 24;
    BZK_HD static T zero() { return {Fp::zero(), Fp::zero()}; }
    BZK_HD static T one() { return {Fp::one(), Fp::zero()}; }
    BZK_HD static bool is_zero(const T& a) { return a.c0.is_zero() && a.c1.is_zero(); }
    BZK_HD static bool eq(const T& a, const T& b) { return a.c0.equals(b.c0) && a.c1.equals(b.c1); }
    BZK_HD static T add(const T& a, const T& b) { return {FpOps::add(a.c0, b.c0), FpOps::add(a.c1, b.c1)}; }
    BZK_HD static T sub(const T& a, const T& b) { return {FpOps::sub(a.c0, b.c0), FpOps::sub(a.c1, b.c1)}; }
    BZK_HD static T neg(const T& a) { return {FpOps::neg(a.c0), FpOps::neg(a.c1)}; }
    BZK_HD static T dbl(const T& a) { return {FpOps::dbl(a.c0), FpOps::dbl(a.c1)}; }
    BZK_HD static T mul(const T& a, const T& b) {  // Karatsuba, 3 Fp products
        Fp aa = FpOps::mul(a.c0, b.c0), bb = FpOps::mul(a.c1, b.c1);
        Fp s = FpOps::mul(FpOps::add(a.c0, a.c1), FpOps::add(b.c0, b.c1));
        return {FpOps::sub(aa, bb), FpOps::sub(FpOps::sub(s, aa), bb)};
    }
    BZK_HD static T sqr(const T& a) {  // (c0+c1)(c0-c1), 2 c0 c1
        Fp s = FpOps::add(a.c0, a.c1), d = FpOps::sub(a.c0, a.c1), m = FpOps::mul(a.c0, a.c1);
        return {FpOps::mul(s, d), FpOps::dbl(m)};
    }
    __host__ __device__ static T inv(const T& a) {
        Fp d = FpOps::inv(FpOps::add(FpOps::sqr(a.c0), FpOps::sqr(a.c1)));
        return {FpOps::mul(a.c0, d), FpOps::mul(FpOps::neg(a.c1), d)};
    }
};

}  // namespace bzk
