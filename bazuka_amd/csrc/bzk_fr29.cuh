// Fr (BLS12-381 scalar field = ZkScalar) in reduced radix: 9 limbs of 29 bits, Montgomery factor 2^261.
// Same idea as bzk_fp28.cuh: a 64-bit column accumulator takes every partial product of a Montgomery
// product as a bare v_mad_u64_u32 (destination == addend), no carries, no register shuffling:
// 162 mads per product instead of the 8 x 32-bit CIOS's 128 mads + ~400 moves/adds.
//
// Bounds (k: value < k*r ; L: limbs < 2^L).  r / 2^261 = 2^-6.14, so products need ka * kb <= 70:
//   mul(a, b)        needs 9 * 2^(La+Lb) + 9 * 2^58 + carries < 2^64 (La + Lb <= 60) ; gives k 2, L 29
//   add(a, b)        limb-wise, k adds
//   norm(a)          carry propagation, L 29
//   wide_mac/reduce  sum of up to 6 products a_i * b_i with L 29 operands in 18 columns, ONE reduction:
//                    7 * 9 * 2^58 < 2^64 ; result k 2 when sum of ka*kb <= 70
// Memory / ABI form stays 8 x 32-bit Montgomery-256 (to29 / from29 convert with one product each).
#pragma once
#include "bzk_field.cuh"
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#endif

namespace bzk {

struct Fr29 {
    uint32_t l[9];
};

namespace fr29 {

static constexpr int N = 9;
static constexpr int W = 29;
static constexpr uint32_t MASK = (1u << W) - 1;
static constexpr uint32_t RINV = 0x1fffffffu;  // -r^-1 mod 2^29

struct Consts {
    uint32_t v[9];
};
static constexpr Consts R = {{0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u}};
static constexpr Consts C_IN = {{0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu}};   // 2^266 mod r
static constexpr Consts C_OUT = {{0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u}};  // 2^256 mod r
static constexpr Consts ONE = {{0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu}};    // 2^261 mod r

BZK_HD Fr29 zero() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = 0;
    return r;
}
BZK_HD Fr29 from_consts(const Consts& c) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = c.v[i];
    return r;
}

struct Wide {
    uint64_t c[2 * N];
};
BZK_HD void wide_zero(Wide& w) {
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) w.c[k] = 0;
}
// w += a * b (schoolbook into the 64-bit columns): 81 mads
BZK_HD void wide_mac(Wide& w, const Fr29& a, const Fr29& b) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
            assert(w.c[i + j] + (uint64_t)a.l[i] * b.l[j] >= w.c[i + j]);  // host harness: the column must not wrap
#endif
            w.c[i + j] += (uint64_t)a.l[i] * b.l[j];
        }
    }
}
// Montgomery reduction of the columns: value * 2^-261 mod r, normalised, k 2 (see header for the input bound)
BZK_HD Fr29 wide_reduce(Wide& w) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = ((uint32_t)w.c[i] * RINV) & MASK;
#pragma unroll
        for (int j = 0; j < N; ++j) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
            assert(w.c[i + j] + (uint64_t)m * R.v[j] >= w.c[i + j]);
#endif
            w.c[i + j] += (uint64_t)m * R.v[j];
        }
        w.c[i + 1] += w.c[i] >> W;
    }
    Fr29 r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        w.c[k + 1] += w.c[k] >> W;
        r.l[k - N] = (uint32_t)w.c[k] & MASK;
    }
    r.l[N - 1] = (uint32_t)w.c[2 * N - 1];
    return r;
}

BZK_HD Fr29 mul_body(const Fr29& a, const Fr29& b) {
    Wide w;
    wide_zero(w);
    wide_mac(w, a, b);
    return wide_reduce(w);
}
// a^2: the 36 off-diagonal partial products once against the doubled limbs + 9 diagonal ones = 45 mads instead of 81
// (126 instead of 162 with the reduction); same column bound as a * a, limb-identical result (the reduction only sees
// the value).  Limbs must stay below 2^31 for the doubling (they are <= 2^29.x wherever this is called).
BZK_HD Fr29 sqr_body(const Fr29& a) {
    Wide w;
    wide_zero(w);
#pragma unroll
    for (int i = 0; i < N; ++i) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
        assert(a.l[i] < (1u << 31));
#endif
        w.c[2 * i] += (uint64_t)a.l[i] * a.l[i];
        const uint32_t a2 = a.l[i] << 1;
#pragma unroll
        for (int j = i + 1; j < N; ++j) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
            assert(w.c[i + j] + (uint64_t)a2 * a.l[j] >= w.c[i + j]);
#endif
            w.c[i + j] += (uint64_t)a2 * a.l[j];
        }
    }
    return wide_reduce(w);
}
#if defined(__HIP_DEVICE_COMPILE__)
// vector (non-aggregate) arguments: two 9-dword structs exceed the ABI's 16 aggregate argument registers and the
// second one would travel through scratch memory (see bzk_fp28.cuh)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __noinline__ static Fr29 mul_call(u32x4 a0, u32x4 a1, uint32_t a2, u32x4 b0, u32x4 b1, uint32_t b2) {
    Fr29 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a.l[i] = a0[i]; a.l[4 + i] = a1[i];
        b.l[i] = b0[i]; b.l[4 + i] = b1[i];
    }
    a.l[8] = a2;
    b.l[8] = b2;
    return mul_body(a, b);
}
#define BZK_FR29_VEC(x) u32x4{x.l[0], x.l[1], x.l[2], x.l[3]}, u32x4{x.l[4], x.l[5], x.l[6], x.l[7]}, x.l[8]
BZK_HD Fr29 mul(const Fr29& a, const Fr29& b) { return mul_call(BZK_FR29_VEC(a), BZK_FR29_VEC(b)); }
__device__ __noinline__ static Fr29 sqr_call(u32x4 a0, u32x4 a1, uint32_t a2) {
    Fr29 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a.l[i] = a0[i]; a.l[4 + i] = a1[i];
    }
    a.l[8] = a2;
    return sqr_body(a);
}
BZK_HD Fr29 sqr(const Fr29& a) { return sqr_call(BZK_FR29_VEC(a)); }
#else
BZK_HD Fr29 mul(const Fr29& a, const Fr29& b) { return mul_body(a, b); }
BZK_HD Fr29 sqr(const Fr29& a) { return sqr_body(a); }
#endif

BZK_HD Fr29 add(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
BZK_HD Fr29 norm(const Fr29& a) {
    Fr29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        const uint32_t t = a.l[i] + c;
        r.l[i] = t & MASK;
        c = t >> W;
    }
    r.l[N - 1] = a.l[N - 1] + c;
    return r;
}

// a - b + 3r for a normalised b with k <= 2 (a product output); limbs below the top are raised so that none can
// underflow (same construction as fp28::sub<K>); result k = ka + 3, normalised
static constexpr Consts D3 = {{0x20000003u, 0x3fffffe7u, 0x3ec4ff3eu, 0x31d811feu, 0x3880ffb2u, 0x240c0e71u, 0x3f6266b5u, 0x3f2f9b0du, 0x015bc8f4u}};
BZK_HD Fr29 sub3(const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + D3.v[i] - b.l[i];
    return norm(r);
}
// x^5 for a normalised x with k <= 7  (49, 4, 14 <= 70)
BZK_HD Fr29 sbox5(const Fr29& x) {
    Fr29 x2 = sqr(x);
    Fr29 x4 = sqr(x2);
    return mul(x4, x);
}

// ---- conversions with the 8 x 32-bit Montgomery-256 memory form
BZK_HD Fr29 repack_from32(const Fr& a) {
    Fr29 r;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int bit = W * k, w = bit >> 5, sh = bit & 31;
        uint32_t v = a.l[w] >> sh;
        if (sh + W > 32 && w + 1 < 8) v |= a.l[w + 1] << (32 - sh);
        r.l[k] = v & MASK;
    }
    return r;
}
BZK_HD Fr repack_to32(const Fr29& a) {  // normalised, value < 2^256
    Fr r;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int lo = (32 * w) / W, sh = 32 * w - W * lo;
        uint64_t v = (uint64_t)a.l[lo] >> sh;
        if (lo + 1 < N) v |= (uint64_t)a.l[lo + 1] << (W - sh);
        if (2 * W - sh < 32 && lo + 2 < N) v |= (uint64_t)a.l[lo + 2] << (2 * W - sh);
        r.l[w] = (uint32_t)v;
    }
    return r;
}
BZK_HD Fr29 to29(const Fr& a) { return mul(repack_from32(a), from_consts(C_IN)); }  // k 2
BZK_HD Fr from29(const Fr29& a) {  // any k <= 35 -> canonical 8 x 32-bit Montgomery-256 limbs
    Fr29 t = mul(a, from_consts(C_OUT));  // [0, 2r)
    Fr29 s;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t d = t.l[i] - R.v[i] - borrow;
        borrow = d >> 31;
        s.l[i] = d & MASK;
    }
    if (!borrow) t = s;
    return repack_to32(t);
}

// out = canonical 8 x 32-bit limbs of (a * s * 2^-261 mod r): with s = x * 2^256 this is Montgomery-256(a' * x) for
// a = a' * 2^261; from29() is the case s = C_OUT.  Any ka <= 35, s < r.
BZK_HD Fr from29_scaled(const Fr29& a, const Fr29& s_) {
    Fr29 t = mul(a, s_);  // [0, 2r)
    Fr29 s;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t d = t.l[i] - R.v[i] - borrow;
        borrow = d >> 31;
        s.l[i] = d & MASK;
    }
    if (!borrow) t = s;
    return repack_to32(t);
}

}  // namespace fr29
}  // namespace bzk
