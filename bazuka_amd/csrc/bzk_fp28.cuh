// Fp in reduced radix for the gfx950 integer pipe: 14 limbs of 28 bits, Montgomery factor R' = 2^392.
//
// Why: CDNA4's multiplier is v_mad_u64_u32 (32x32 + 64 -> 64, measured 31.4 Top/s).  With saturated
// 32-bit limbs every partial product needs a carry (v_addc + VCC hazards) or, as hipcc lowers the
// textbook CIOS, ~2.6 v_mov per mad to build even-aligned 64-bit addend pairs (1419 instructions per
// product, 41.6 G products/s measured).  With 28-bit limbs a column accumulator is a plain u64 that
// absorbs all 2 x 14 partial products of a Montgomery product without overflow:
//       c[i+j] = a[i] * b[j] + c[i+j]        <- one v_mad_u64_u32, destination == addend
// 394 mads + ~125 other instructions per product (519 total, measured 71.5 G products/s as a function
// call).  Additions are 14 independent v_add_u32 (no carry chain); subtraction adds a multiple of p
// laid out so that no limb can underflow.  Values are kept only WEAKLY reduced (below a small multiple
// of p) with per-operation bounds documented below and checked by the host harness
// (tests/host/hostcheck.hip runs this very header on the CPU with bound assertions).
//
// Bounds (k = value < k*p, L = every limb < 2^L):
//   mul(a, b)   needs  14 * 2^(La+Lb) + 14 * 2^56 < 2^64 (La + Lb <= 59.2 suffices) and ka * kb <= 2048
//               gives  k = 2, L = 28 (normalised; top limb < 2^18)
//   add(a, b)   gives  k = ka + kb, limbs add
//   sub<K>(a,b) needs  b normalised (L = 28) and kb < K ; gives k = ka + K, limbs < 2^La + 2^29
//   norm(a)     carry propagation only: same value, L = 28 (value must be < 2^392)
// The memory/ABI form stays 12 x 32-bit Montgomery (R = 2^384); to28 / from28 convert (one product each).
#pragma once
#include "bzk_curve.cuh"
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#endif

namespace bzk {

struct alignas(8) Fp28 {
    uint32_t l[14];
};

namespace fp28 {

static constexpr int N = 14;
static constexpr int W = 28;
static constexpr uint32_t MASK = (1u << W) - 1;
static constexpr uint32_t PINV = 0xffcfffdu;  // -p^-1 mod 2^28

struct Consts {
    uint32_t v[14];
};
// p
static constexpr Consts P = {{0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                              0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u}};
// K*p with limb i (i < 13) raised by 2^28 (-1 for i > 0) and the top limb lowered by 1: same value, every
// limb below the top is >= 2^28 - 1, so `D[i] - b[i]` cannot underflow for a normalised b.
static constexpr Consts D3 = {{0x1fff0001u, 0x1fcffffeu, 0x1bffff2cu, 0x1fffc13eu, 0x126c5c02u, 0x1e2e412du, 0x1e359276u,
                               0x1da8f382u, 0x1d65e28du, 0x1e306861u, 0x12f722c8u, 0x1b3cee14u, 0x15beac7eu, 0x004e032u}};
static constexpr Consts D6 = {{0x1ffe0002u, 0x1f9ffffeu, 0x17fffe5au, 0x1fff827eu, 0x14d8b806u, 0x1c5c825bu, 0x1c6b24eeu,
                               0x1b51e706u, 0x1acbc51cu, 0x1c60d0c4u, 0x15ee4592u, 0x1679dc29u, 0x1b7d58feu, 0x009c065u}};
static constexpr Consts D12 = {{0x1ffc0004u, 0x1f3ffffeu, 0x1ffffcb6u, 0x1fff04fdu, 0x19b1700eu, 0x18b904b7u, 0x18d649deu,
                                0x16a3ce0eu, 0x15978a3au, 0x18c1a18au, 0x1bdc8b26u, 0x1cf3b853u, 0x16fab1fdu, 0x01380ccu}};
static constexpr Consts D24 = {{0x1ff80008u, 0x1e7ffffeu, 0x1ffff96eu, 0x1ffe09fcu, 0x1362e01eu, 0x11720970u, 0x11ac93beu,
                                0x1d479c1eu, 0x1b2f1475u, 0x11834315u, 0x17b9164eu, 0x19e770a8u, 0x1df563fcu, 0x00270199u}};
// 2^400 mod p and 2^384 mod p as plain integers (conversion multipliers), 2^392 mod p (Montgomery one)
static constexpr Consts C_IN = {{0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u,
                                 0xa4e6fe9u, 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau}};
static constexpr Consts C_OUT = {{0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u,
                                  0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u}};
// 2^1176 mod p = R'^3: turns the plain inverse of a Montgomery value into the Montgomery form of the inverse (inv_gcd)
static constexpr Consts R3 = {{0x1f7b890u, 0x294cc4du, 0x9f3af22u, 0xb5ba56cu, 0xcb5c0ccu, 0xc0d975cu, 0xc89a8c5u,
                               0x6c968b4u, 0x22672eau, 0x91de8c9u, 0x35652a6u, 0x84977c8u, 0x424bbb9u, 0x00141abu}};
static constexpr Consts ONE = {{0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu,
                                0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au}};

BZK_HD Fp28 zero() {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = 0;
    return r;
}
BZK_HD Fp28 one() {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = ONE.v[i];
    return r;
}
BZK_HD bool limbs_all_zero(const Fp28& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) o |= a.l[i];
    return o == 0;
}

// ---- the quotient digit of one Montgomery step: (lo * PINV) mod 2^28.  v_mul_lo_u32 issues at 18.3 Top/s on gfx950, v_mad_u64_u32 at
// 31.2 (profiles/r04_ubench_int.txt), and the low half of a multiply-add with a zero addend is the same number - but the compiler narrows
// every C++ spelling of that back to v_mul_lo_u32, and as inline asm the instruction is opaque to the scheduler: the dependent mask lands
// right behind it and costs an s_nop, the 64-bit temporary costs registers.  MEASURED (profiles/r04_run30_inlined_products_ab.txt, same
// box, alternating): G1 accumulation 2.55 / 2.54 ms without, 2.58 / 2.52 ms with - nothing; G2 accumulation 9.08 -> 9.29 ms, G2 reduce
// 1.95 -> 1.99 ms with it (one wave per SIMD, 14 more spilled registers).  Default 0; BZK_MONT_M_MAD = 1 keeps the form for A/B builds.
#ifndef BZK_MONT_M_MAD
#define BZK_MONT_M_MAD 0
#endif
BZK_HD uint32_t mont_m(uint32_t lo) {
#if defined(__HIP_DEVICE_COMPILE__) && BZK_MONT_M_MAD
    uint64_t t, carry_out;
    __asm__("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=&v"(t), "=s"(carry_out) : "v"(lo), "s"(PINV));
    return (uint32_t)t & MASK;
#else
    return (lo * PINV) & MASK;
#endif
}

// ---- the product (see bounds in the header comment)
BZK_HD Fp28 mul_body(const Fp28& a, const Fp28& b) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    {  // host harness: the 64-bit column accumulators must never wrap
        unsigned __int128 worst = 0;
        uint32_t ma = 0, mb = 0;
        for (int i = 0; i < N; ++i) { if (a.l[i] > ma) ma = a.l[i]; if (b.l[i] > mb) mb = b.l[i]; }
        worst = (unsigned __int128)14 * ma * mb + (unsigned __int128)14 * MASK * MASK + ((unsigned __int128)1 << 40);
        assert(worst < ((unsigned __int128)1 << 64));
    }
#endif
    uint64_t c[2 * N];
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = mont_m((uint32_t)c[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)m * P.v[j];
        c[i + 1] += c[i] >> W;
    }
    Fp28 r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        c[k + 1] += c[k] >> W;
        r.l[k - N] = (uint32_t)c[k] & MASK;
    }
    r.l[N - 1] = (uint32_t)c[2 * N - 1];
    return r;
}

// ---- the square: the 91 off-diagonal partial products are taken once against the doubled limbs (2 a_i) a_j, i < j, plus the
// 14 diagonal ones - 105 mads instead of 196 before the (unchanged, 196-mad) Montgomery reduction, 301 instead of 392 in all.
// Same bound as mul(a, a): a column still sums at most 14 x 2^(2 La).  The integer a^2 and therefore every output limb is
// identical to mul_body(a, a) (the reduction only looks at the value), so results do not depend on which one is used.
BZK_HD Fp28 sqr_body(const Fp28& a) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    {
        uint32_t ma = 0;
        for (int i = 0; i < N; ++i) if (a.l[i] > ma) ma = a.l[i];
        assert(ma < (1u << 31));  // the doubled limb must fit 32 bits
        const unsigned __int128 worst = (unsigned __int128)14 * ma * ma + (unsigned __int128)14 * MASK * MASK + ((unsigned __int128)1 << 40);
        assert(worst < ((unsigned __int128)1 << 64));
    }
#endif
    uint64_t c[2 * N];
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        c[2 * i] += (uint64_t)a.l[i] * a.l[i];
        const uint32_t a2 = a.l[i] << 1;
#pragma unroll
        for (int j = i + 1; j < N; ++j) c[i + j] += (uint64_t)a2 * a.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = mont_m((uint32_t)c[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)m * P.v[j];
        c[i + 1] += c[i] >> W;
    }
    Fp28 r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        c[k + 1] += c[k] >> W;
        r.l[k - N] = (uint32_t)c[k] & MASK;
    }
    r.l[N - 1] = (uint32_t)c[2 * N - 1];
    return r;
}

// ---- a b - c d with ONE Montgomery reduction (the Y coordinate of every addition formula is such a difference): the second
// product is accumulated against the non-underflowing negation  6p - d  (d normalised, value < 6p), so the 28 columns only ever
// grow: 2 x 196 + 196 multiply-adds instead of 2 x 392, and the result is a product output (normalised, k = 2) where the
// two-product form needed sub<3> + norm.  Needs  14 (2^(La+Lb) + 2^(Lc+29) + 2^56) < 2^64  and  ka kb + 6 kc <= 2048.
// Inlined (four operands = 56 argument registers do not fit the 32 of a call): used once, in the accumulation's mixed addition.
BZK_HD Fp28 mul_sub2_body(const Fp28& a, const Fp28& b, const Fp28& cc, const Fp28& d) {
    uint32_t nd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
        assert(D6.v[i] >= d.l[i]);
#endif
        nd[i] = D6.v[i] - d.l[i];
    }
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    {
        uint32_t ma = 0, mb = 0, mc = 0, md = 0;
        for (int i = 0; i < N; ++i) {
            if (a.l[i] > ma) ma = a.l[i];
            if (b.l[i] > mb) mb = b.l[i];
            if (cc.l[i] > mc) mc = cc.l[i];
            if (nd[i] > md) md = nd[i];
        }
        const unsigned __int128 worst = (unsigned __int128)14 * ma * mb + (unsigned __int128)14 * mc * md + (unsigned __int128)14 * MASK * MASK +
                                        ((unsigned __int128)1 << 40);
        assert(worst < ((unsigned __int128)1 << 64));
    }
#endif
    uint64_t c[2 * N];
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)cc.l[i] * nd[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = mont_m((uint32_t)c[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)m * P.v[j];
        c[i + 1] += c[i] >> W;
    }
    Fp28 r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        c[k + 1] += c[k] >> W;
        r.l[k - N] = (uint32_t)c[k] & MASK;
    }
    r.l[N - 1] = (uint32_t)c[2 * N - 1];
    return r;
}

// ---- a0 b0 + a1 b1 + a2 b2 + a3 b3 with ONE reduction: a component of  R T - Y PPP  over Fp2 (the G2 mixed addition), the signs
// carried by non-underflowing negations of one factor.  Needs  14 (sum_i 2^(Lai+Lbi) + 2^56) < 2^64  and  sum_i kai kbi <= 2048.
BZK_HD Fp28 mul4_body(const Fp28& a0, const Fp28& b0, const Fp28& a1, const Fp28& b1, const Fp28& a2, const Fp28& b2, const Fp28& a3,
                      const Fp28& b3) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    {
        const Fp28* o[8] = {&a0, &b0, &a1, &b1, &a2, &b2, &a3, &b3};
        unsigned __int128 worst = (unsigned __int128)14 * MASK * MASK + ((unsigned __int128)1 << 40);
        for (int t = 0; t < 4; ++t) {
            uint32_t ma = 0, mb = 0;
            for (int i = 0; i < N; ++i) {
                if (o[2 * t]->l[i] > ma) ma = o[2 * t]->l[i];
                if (o[2 * t + 1]->l[i] > mb) mb = o[2 * t + 1]->l[i];
            }
            worst += (unsigned __int128)14 * ma * mb;
        }
        assert(worst < ((unsigned __int128)1 << 64));
    }
#endif
    uint64_t c[2 * N];
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a0.l[i] * b0.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a1.l[i] * b1.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a2.l[i] * b2.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a3.l[i] * b3.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = mont_m((uint32_t)c[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)m * P.v[j];
        c[i + 1] += c[i] >> W;
    }
    Fp28 r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        c[k + 1] += c[k] >> W;
        r.l[k - N] = (uint32_t)c[k] & MASK;
    }
    r.l[N - 1] = (uint32_t)c[2 * N - 1];
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// one resident copy of the product per kernel image: by-value arguments travel in VGPRs
// The product is a real call (I-cache: see the header).  The operands travel as four-lane VECTOR arguments, not
// as two structs: clang's AMDGPU ABI gives aggregate arguments 16 argument registers in total, so the second
// 14-dword struct of mul_call(Fp28, Fp28) was passed through a scratch copy (56 B stored by the caller and
// re-loaded by the callee on every product - measured as 1.2 GB of HBM writes per 2^20-point accumulate launch).
// Vector arguments are not aggregates and go straight into v0..v27.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __noinline__ static Fp28 mul_call(u32x4 a0, u32x4 a1, u32x4 a2, u32x2 a3, u32x4 b0, u32x4 b1, u32x4 b2, u32x2 b3) {
    Fp28 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a.l[i] = a0[i]; a.l[4 + i] = a1[i]; a.l[8 + i] = a2[i];
        b.l[i] = b0[i]; b.l[4 + i] = b1[i]; b.l[8 + i] = b2[i];
    }
    a.l[12] = a3[0]; a.l[13] = a3[1];
    b.l[12] = b3[0]; b.l[13] = b3[1];
    return mul_body(a, b);
}
#define BZK_FP28_VEC(x) u32x4{x.l[0], x.l[1], x.l[2], x.l[3]}, u32x4{x.l[4], x.l[5], x.l[6], x.l[7]}, \
                        u32x4{x.l[8], x.l[9], x.l[10], x.l[11]}, u32x2{x.l[12], x.l[13]}
BZK_HD Fp28 mul(const Fp28& a, const Fp28& b) { return mul_call(BZK_FP28_VEC(a), BZK_FP28_VEC(b)); }
#if defined(BZK_FP28_NO_SQR)
BZK_HD Fp28 sqr(const Fp28& a) { return mul_call(BZK_FP28_VEC(a), BZK_FP28_VEC(a)); }
#else
// the square is a second resident function (14 argument registers)
__device__ __noinline__ static Fp28 sqr_call(u32x4 a0, u32x4 a1, u32x4 a2, u32x2 a3) {
    Fp28 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a.l[i] = a0[i]; a.l[4 + i] = a1[i]; a.l[8 + i] = a2[i];
    }
    a.l[12] = a3[0]; a.l[13] = a3[1];
    return sqr_body(a);
}
BZK_HD Fp28 sqr(const Fp28& a) { return sqr_call(BZK_FP28_VEC(a)); }
#endif
#else
BZK_HD Fp28 mul(const Fp28& a, const Fp28& b) { return mul_body(a, b); }
BZK_HD Fp28 sqr(const Fp28& a) { return sqr_body(a); }
#endif

BZK_HD Fp28 add(const Fp28& a, const Fp28& b) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a + K*p - b ; b normalised, value(b) < K*p
template <int K>
BZK_HD Fp28 sub(const Fp28& a, const Fp28& b) {
    static_assert(K == 3 || K == 6 || K == 12 || K == 24, "no table for this multiple of p");
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t d = K == 3 ? D3.v[i] : K == 6 ? D6.v[i] : K == 12 ? D12.v[i] : D24.v[i];
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
        assert(d >= b.l[i]);                                 // no limb underflow
        assert((uint64_t)a.l[i] + d - b.l[i] < (1ull << 32));  // no u32 overflow
#endif
        r.l[i] = a.l[i] + d - b.l[i];
    }
    return r;
}

// carry propagation: limbs -> [0, 2^28), value unchanged (value < 2^392 required)
BZK_HD Fp28 norm(const Fp28& a) {
    Fp28 r;
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

// for a product output (normalised, value in [0, 2p)): is it 0 mod p ?
BZK_HD bool mulout_is_zero(const Fp28& a) {
    uint32_t o0 = 0, op = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        o0 |= a.l[i];
        op |= a.l[i] ^ P.v[i];
    }
    return o0 == 0 || op == 0;
}

// ---- conversions with the 12 x 32-bit, R = 2^384 memory form
BZK_HD Fp28 repack_from32(const Fp& a) {  // same integer, 28-bit limbs
    Fp28 r;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int bit = W * k, w = bit >> 5, sh = bit & 31;
        uint32_t v = a.l[w] >> sh;
        if (sh + W > 32 && w + 1 < 12) v |= a.l[w + 1] << (32 - sh);
        r.l[k] = v & MASK;
    }
    return r;
}
BZK_HD Fp repack_to32(const Fp28& a) {  // a normalised and < 2^384
    Fp r;
#pragma unroll
    for (int w = 0; w < 12; ++w) {
        // bits [32w, 32w+32)
        const int lo = (32 * w) / W, sh = 32 * w - W * lo;
        uint64_t v = (uint64_t)a.l[lo] >> sh;
        v |= (uint64_t)a.l[lo + 1] << (W - sh);
        if (2 * W - sh < 32 && lo + 2 < N) v |= (uint64_t)a.l[lo + 2] << (2 * W - sh);
        r.l[w] = (uint32_t)v;
    }
    return r;
}
BZK_HD Fp28 consts_as_fp28(const Consts& c) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = c.v[i];
    return r;
}
// x * 2^384 (12x32)  ->  x * 2^392 (14x28), k = 2, normalised
BZK_HD Fp28 to28(const Fp& a) { return mul(repack_from32(a), consts_as_fp28(C_IN)); }
// any k <= 1024 value -> canonical 12x32 Montgomery-384 limbs
BZK_HD Fp from28(const Fp28& a) {
    Fp28 t = mul(a, consts_as_fp28(C_OUT));  // in [0, 2p), normalised
    // conditional subtraction of p (borrow chain over 28-bit limbs)
    Fp28 s;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t d = t.l[i] - P.v[i] - borrow;
        borrow = d >> 31;  // limbs < 2^28, so a negative difference sets bit 31
        s.l[i] = d & MASK;
    }
    if (!borrow) t = s;
    return repack_to32(t);
}

// ---- strong reduction: any value < 2048 p with limbs < 2^31  ->  normalised, value < 3p
// Quotient estimate from the top limb (p / 2^364 = 0x1a011.ea39...): q = floor(top / 0x1a012) never exceeds
// floor(a / p) and falls short by less than 1.03, so a - q p lies in [0, 2.03 p).
BZK_HD Fp28 reduce(const Fp28& a) {
    Fp28 r = norm(a);
    const uint32_t q = r.l[N - 1] / 0x1a012u;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        const int64_t t = (int64_t)r.l[i] - (int64_t)((uint64_t)q * P.v[i]) + carry;
        r.l[i] = (uint32_t)t & MASK;
        carry = t >> W;
    }
    r.l[N - 1] = (uint32_t)((int64_t)r.l[N - 1] - (int64_t)((uint64_t)q * P.v[N - 1]) + carry);
    return r;
}

// for a reduced value (normalised, < 3p): is it 0 mod p ?
BZK_HD bool reduced_is_zero(const Fp28& a) {
    uint32_t o0 = 0, o1 = 0, o2 = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        o0 |= a.l[i];
        o1 |= a.l[i] ^ P.v[i];
    }
    // 2p in normalised limbs
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t t = 2 * P.v[i] + c;
        const uint32_t limb = i < N - 1 ? (t & MASK) : t;
        c = t >> W;
        o2 |= a.l[i] ^ limb;
    }
    return o0 == 0 || o1 == 0 || o2 == 0;
}

// a^(p-2) ; inv(0) = 0.  Input k <= 45, output a product output (k 2).  Kept as the cross-check of inv (= inv_gcd below).
BZK_HD Fp28 inv_fermat(const Fp28& a) {
    uint32_t e[12];
    {
        uint64_t borrow = 2;
#pragma unroll 1
        for (int i = 0; i < 12; ++i) {
            uint64_t d = (uint64_t)FpParams::MOD[i] - borrow;
            e[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
    }
    Fp28 r = one();
#pragma unroll 1
    for (int i = 383; i >= 0; --i) {
        r = sqr(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, a);
    }
    return r;
}


// ---- inversion by an optimized binary GCD (Pornin 2020, "Optimized Binary GCD for Modular Inversion") on 28-bit limbs.
// The Fermat inversion above is 384 squarings + ~190 products, ~260 k instructions on one lane (0.63 ms measured) - and under
// SIMT a lane that inverts costs its whole wavefront that time.  The binary GCD needs no products: per OUTER iteration 28
// inner steps run on 58-bit approximations (low 28 + top 30 bits of a and b) and produce a 2x2 matrix (f0 g0; f1 g1) with
// |f| + |g| <= 2^28, which is then applied once to the full-width values:
//     (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^28                      exact division (one limb)
//     (u, v) <- (f0 u + g0 v, f1 u + g1 v) / 2^28  mod p               one Montgomery step with the limb-sized modulus word
// so that a = u y and b = v y (mod p) hold throughout with no accumulated power of two.  2 * 381 - 1 = 761 inner steps
// suffice (28 outer iterations; the worst case seen over adversarial and 20 000 random inputs needs 28); 30 are run - further
// steps on a = 0 are the identity.  ~1300 instructions per outer
// iteration, ~38 k in all: 7x fewer than Fermat.  Branch-free apart from uniform loops.
// gcd_inv_plain: y canonical in [1, p) as a plain integer -> y^-1 mod p, canonical; y = 0 -> 0.
BZK_HD Fp28 gcd_inv_plain(const Fp28& y) {
    Fp28 a = y, b, u, v;
#pragma unroll
    for (int i = 0; i < N; ++i) { b.l[i] = P.v[i]; u.l[i] = 0; v.l[i] = 0; }
    u.l[0] = 1;
    auto cond_neg = [](Fp28& x, int64_t top, bool neg) {  // x (limbs 0..12 in [0, 2^28), signed top) <- neg ? -x : x ; returns nothing
        const uint32_t m = neg ? MASK : 0u;
        uint32_t c = neg ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < N - 1; ++i) {
            const uint32_t t = (x.l[i] ^ m) + c;
            x.l[i] = t & MASK;
            c = t >> W;
        }
        const int64_t tt = (neg ? ~top : top) + (int64_t)c;
        x.l[N - 1] = (uint32_t)tt;  // non-negative and < 2^28 by construction (|value| < 2^381)
    };
    // x = (fa * a + fb * b) >> 28, returned with sign handled: out magnitude, *neg = was negative
    auto lin_shift = [&](const Fp28& aa, const Fp28& bb, int64_t fa, int64_t fb, Fp28& out) -> bool {
        int64_t carry = 0;
        {
            const int64_t t = fa * (int64_t)aa.l[0] + fb * (int64_t)bb.l[0];
            carry = t >> W;  // low 28 bits are zero by construction
        }
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const int64_t t = fa * (int64_t)aa.l[i] + fb * (int64_t)bb.l[i] + carry;
            out.l[i - 1] = (uint32_t)t & MASK;
            carry = t >> W;
        }
        const bool neg = carry < 0;
        cond_neg(out, carry, neg);
        return neg;
    };
    // Montgomery step: (|fa| U + |fb| V + t p) >> 28 with U = fa < 0 ? p - u : u (same for V), result in [0, p)
    auto lin_mont = [&](const Fp28& uu, const Fp28& vv, int64_t fa, int64_t fb, bool negate, Fp28& out) {
        Fp28 U, V;
        {
            uint32_t bu = 0, bv = 0;
            const bool nu = fa < 0, nv = fb < 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t du = P.v[i] - uu.l[i] - bu, dv = P.v[i] - vv.l[i] - bv;
                bu = du >> 31;
                bv = dv >> 31;
                U.l[i] = nu ? (du & MASK) : uu.l[i];
                V.l[i] = nv ? (dv & MASK) : vv.l[i];
            }
        }
        const uint64_t ma = (uint64_t)(fa < 0 ? -fa : fa), mb = (uint64_t)(fb < 0 ? -fb : fb);
        uint64_t t0 = ma * U.l[0] + mb * V.l[0];
        const uint32_t q = ((uint32_t)t0 * PINV) & MASK;
        uint64_t carry = (t0 + (uint64_t)q * P.v[0]) >> W;
        Fp28 r;
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const uint64_t t = ma * U.l[i] + mb * V.l[i] + (uint64_t)q * P.v[i] + carry;
            r.l[i - 1] = (uint32_t)t & MASK;
            carry = t >> W;
        }
        r.l[N - 1] = (uint32_t)carry;
        // r < 2p : one conditional subtraction, then the optional negation p - r
        Fp28 sub_;
        uint32_t bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t d = r.l[i] - P.v[i] - bw;
            bw = d >> 31;
            sub_.l[i] = d & MASK;
        }
        if (!bw) r = sub_;
        if (negate) {  // p - r (r = 0 stays 0)
            uint32_t o = 0, b2 = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) o |= r.l[i];
            Fp28 ng;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const uint32_t d = P.v[i] - r.l[i] - b2;
                b2 = d >> 31;
                ng.l[i] = d & MASK;
            }
            if (o) r = ng;
        }
        out = r;
    };
#pragma unroll 1
    for (int it = 0; it < 30; ++it) {
        // 58-bit approximations: exact values while both fit 56 bits, else low 28 bits | top 30 bits (same shift for both)
        int top = 0;
#pragma unroll
        for (int i = 1; i < N; ++i) top = (a.l[i] | b.l[i]) ? i : top;
        uint64_t A, B;
        if (top < 2) {
            A = (uint64_t)a.l[0] | ((uint64_t)a.l[1] << W);
            B = (uint64_t)b.l[0] | ((uint64_t)b.l[1] << W);
        } else {
            const uint32_t hi = a.l[top] | b.l[top];
            int lz = __builtin_clz(hi) - 4;       // zero bits above the leading one inside the 28-bit limb
            if (top == 2 && lz > 26) lz = 26;     // keep the two parts from overlapping (n = max(len, 58))
            const uint64_t wa = ((uint64_t)a.l[top] << W) | a.l[top - 1], wb = ((uint64_t)b.l[top] << W) | b.l[top - 1];
            const uint64_t ha = (((wa << lz) | ((uint64_t)a.l[top - 2] >> (W - lz))) & 0xffffffffffffffull) >> 26;
            const uint64_t hb = (((wb << lz) | ((uint64_t)b.l[top - 2] >> (W - lz))) & 0xffffffffffffffull) >> 26;
            A = (ha << W) | a.l[0];
            B = (hb << W) | b.l[0];
        }
        int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
        for (int j = 0; j < 28; ++j) {
            const bool odd = (A & 1) != 0;
            const bool sw = odd && A < B;
            const uint64_t A2 = sw ? B : A, B2 = sw ? A : B;
            const int64_t tf0 = sw ? f1 : f0, tg0 = sw ? g1 : g0, tf1 = sw ? f0 : f1, tg1 = sw ? g0 : g1;
            A = (A2 - (odd ? B2 : 0)) >> 1;
            B = B2;
            f0 = tf0 - (odd ? tf1 : 0);
            g0 = tg0 - (odd ? tg1 : 0);
            f1 = tf1 << 1;
            g1 = tg1 << 1;
        }
        Fp28 na, nb, nu, nv;
        const bool nega = lin_shift(a, b, f0, g0, na);
        const bool negb = lin_shift(a, b, f1, g1, nb);
        lin_mont(u, v, f0, g0, nega, nu);
        lin_mont(u, v, f1, g1, negb, nv);
        a = na; b = nb; u = nu; v = nv;
    }
    return v;  // b = gcd = 1
}

BZK_HD Fp28 inv_gcd(const Fp28& a_mont) {
    // a = x R' (weakly reduced) -> canonical integer in [0, p)
    Fp28 t = reduce(a_mont);  // < 3p, normalised
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        Fp28 s_;
        uint32_t bw = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t d = t.l[i] - P.v[i] - bw;
            bw = d >> 31;
            s_.l[i] = d & MASK;
        }
        if (!bw) t = s_;
    }
    const Fp28 iv = gcd_inv_plain(t);  // (x R')^-1 as a plain integer
    // x^-1 R' = iv * R'^3 * R'^-1 (one Montgomery product with the constant R'^3)
    return mul(iv, consts_as_fp28(R3));
}
// the inversion every caller uses: inv(0) = 0; input any weakly reduced value (< 2048 p, limbs < 2^31), output a product output
BZK_HD Fp28 inv(const Fp28& a) { return inv_gcd(a); }

}  // namespace fp28

// ------------------------------------------------------------------------------------------------
// Fp2 = Fp[u]/(u^2+1) over the reduced-radix field, with one uniform discipline so that the generic
// XYZZ code (bzk_curve.cuh) can run on it unchanged: every component handed out by these operations is
// normalised with value < 3p (`reduce`), or a product output (< 2p).
// ------------------------------------------------------------------------------------------------
struct Fp2x28 {
    Fp28 c0, c1;
};

struct Fp2x28Ops {
    typedef Fp2x28 T;
    static constexpr int LIMBS = 28;
    // Discipline (k: component < k*p): sums and differences are strongly reduced (< 3p); PRODUCTS are only
    // carry-normalised (mul: c0 < 5p, c1 < 8p; sqr: c0 < 2p, c1 < 4p) - every consumer tolerates k <= 8 operands:
    //   mul / sqr  operate on sums of two components: (8+8) * (8+8) = 256 <= 2048, limbs 29 + 29.6 bits <= 59.2
    //   sub        uses the 12p offset table (subtrahend k < 12)
    //   is_zero    reduces first
    // which saves the ~100-instruction quotient-estimate reduction twice per product.
    BZK_HD static T zero() { return {fp28::zero(), fp28::zero()}; }
    BZK_HD static T one() { return {fp28::one(), fp28::zero()}; }
    BZK_HD static bool is_zero(const T& a) {
        return fp28::reduced_is_zero(fp28::reduce(a.c0)) && fp28::reduced_is_zero(fp28::reduce(a.c1));
    }
    BZK_HD static T add(const T& a, const T& b) { return {fp28::reduce(fp28::add(a.c0, b.c0)), fp28::reduce(fp28::add(a.c1, b.c1))}; }
    BZK_HD static T sub(const T& a, const T& b) {
        return {fp28::reduce(fp28::sub<12>(a.c0, b.c0)), fp28::reduce(fp28::sub<12>(a.c1, b.c1))};
    }
    BZK_HD static T neg(const T& a) { return sub(zero(), a); }
    BZK_HD static T dbl(const T& a) { return add(a, a); }
    BZK_HD static T mul(const T& a, const T& b) {  // Karatsuba: 3 base-field products
        using namespace fp28;
        Fp28 aa = fp28::mul(a.c0, b.c0), bb = fp28::mul(a.c1, b.c1);
        Fp28 s = fp28::mul(fp28::add(a.c0, a.c1), fp28::add(b.c0, b.c1));
        return {norm(fp28::sub<3>(aa, bb)), norm(fp28::sub<3>(fp28::sub<3>(s, aa), bb))};  // k 5, k 8
    }
    BZK_HD static T sqr(const T& a) {  // (c0 + c1)(c0 - c1), 2 c0 c1 : 2 base-field products
        using namespace fp28;
        Fp28 m = fp28::mul(a.c0, a.c1);
        return {fp28::mul(fp28::add(a.c0, a.c1), fp28::sub<12>(a.c0, a.c1)), norm(fp28::add(m, m))};  // k 2, k 4
    }
    BZK_HD static T inv(const T& a) {
        using namespace fp28;
        Fp28 d = fp28::inv(reduce(fp28::add(fp28::sqr(a.c0), fp28::sqr(a.c1))));
        return {fp28::mul(a.c0, d), fp28::mul(reduce(fp28::sub<12>(fp28::zero(), a.c1)), d)};
    }
    BZK_HD static bool eq(const T& a, const T& b) { return is_zero(sub(a, b)); }
};

// ------------------------------------------------------------------------------------------------
// G1 in XYZZ over Fp28.  Invariants of a stored point: X normalised k <= 11, Y normalised k <= 5,
// ZZ / ZZZ product outputs (normalised, k = 2); identity <=> all ZZ limbs zero.
// Affine bases: x, y product outputs of to28 (normalised, k = 2).
// ------------------------------------------------------------------------------------------------
struct G1A28 {
    Fp28 x, y;
};
struct G1X28 {
    Fp28 X, Y, ZZ, ZZZ;
};

namespace g1x28 {
using namespace fp28;
#ifndef BZK_G1_FUSED_Y
#define BZK_G1_FUSED_Y 1  // 0: Y3 of the mixed addition as two reduced products (A/B builds)
#endif

BZK_HD G1X28 identity() { return {zero(), one(), zero(), zero()}; }
BZK_HD bool is_identity(const G1X28& p) { return limbs_all_zero(p.ZZ); }

BZK_HD G1X28 dbl_affine(const G1A28& a) {  // mdbl-2008-s-1
    Fp28 U = add(a.y, a.y);                 // k 4, L 29
    Fp28 V = sqr(U), Wv = mul(U, V), S = mul(a.x, V);
    Fp28 xx = sqr(a.x);
    Fp28 M = add(add(xx, xx), xx);          // k 6, L < 29.6
    G1X28 r;
    r.X = norm(sub<3>(sub<3>(sqr(M), S), S));                 // k 8
    r.Y = norm(sub<3>(mul(M, sub<12>(S, r.X)), mul(Wv, a.y)));  // k 5
    r.ZZ = V;
    r.ZZZ = Wv;
    return r;
}

BZK_HD G1X28 dbl(const G1X28& p) {  // dbl-2008-s-1
    if (is_identity(p)) return p;
    Fp28 U = add(p.Y, p.Y);                 // k 10, L 29
    Fp28 V = sqr(U), Wv = mul(U, V), S = mul(p.X, V);
    Fp28 xx = sqr(p.X);
    Fp28 M = add(add(xx, xx), xx);
    G1X28 r;
    r.X = norm(sub<3>(sub<3>(sqr(M), S), S));
    r.Y = norm(sub<3>(mul(M, sub<12>(S, r.X)), mul(Wv, p.Y)));
    r.ZZ = mul(V, p.ZZ);
    r.ZZZ = mul(Wv, p.ZZZ);
    return r;
}

// acc += q  (q affine, never the identity); neg_q: add -q instead.
// `pre` is called exactly once, after the last product of the formula and before its tail (the fused Y): the accumulation issues
// the NEXT base's loads there.  With products as CALLS a call is a wait-for-all-loads point for the compiler, so loads issued any earlier
// would be waited for at the next product; issued here they fly during the ~3 us of the tail and are complete when the next addition
// starts.  With inlined products (MulInline) the position is no longer forced; it is kept - the loads' 28 registers stay free until there.
struct NoPre {
    BZK_HD void operator()() const {}
};
// how the formula's products are issued: as calls to the resident product functions (every tail kernel), or inlined into the caller
// (the accumulation's hot loop, msm_policy.cuh BZK_G1_ACC_INLINE: no argument / result moves, and the compiler schedules across products)
struct MulCalls {
    BZK_HD static Fp28 mul(const Fp28& a, const Fp28& b) { return fp28::mul(a, b); }
    BZK_HD static Fp28 sqr(const Fp28& a) { return fp28::sqr(a); }
};
struct MulInline {
    BZK_HD static Fp28 mul(const Fp28& a, const Fp28& b) { return mul_body(a, b); }
    BZK_HD static Fp28 sqr(const Fp28& a) { return sqr_body(a); }
};
template <class Pre = NoPre, class M = MulCalls>
BZK_HD void add_mixed(G1X28& acc, const G1A28& q_in, bool neg_q, Pre&& pre = Pre(), M = M()) {
    G1A28 q = q_in;
    if (neg_q) q.y = norm(sub<3>(zero(), q.y));  // 3p - y, k 3
    if (is_identity(acc)) {
        pre();
        acc = {q.x, q.y, one(), one()};
        return;
    }
    Fp28 U2 = M::mul(q.x, acc.ZZ), S2 = M::mul(q.y, acc.ZZZ);
    Fp28 Pp = sub<12>(U2, acc.X);  // k 14
    Fp28 R = sub<6>(S2, acc.Y);    // k 8
    Fp28 PP = M::sqr(Pp);
    if (mulout_is_zero(PP)) {  // same x: doubling or cancellation (rare)
        Fp28 RR = sqr(R);
        if (mulout_is_zero(RR)) acc = dbl_affine(q);
        else acc = identity();
        pre();
        return;
    }
    Fp28 PPP = M::mul(Pp, PP), Q = M::mul(acc.X, PP), RR = M::sqr(R);
    Fp28 X3 = norm(sub<3>(sub<3>(sub<3>(RR, PPP), Q), Q));             // k 11
    acc.ZZ = M::mul(acc.ZZ, PP);
    acc.ZZZ = M::mul(acc.ZZZ, PPP);
    pre();
#if BZK_G1_FUSED_Y
    acc.Y = mul_sub2_body(R, sub<12>(Q, X3), PPP, acc.Y);              // R (Q - X3) - Y PPP, one reduction; k 2
#else
    acc.Y = norm(sub<3>(mul(R, sub<12>(Q, X3)), mul(acc.Y, PPP)));     // k 5
#endif
    acc.X = X3;
}

BZK_HD void add_full(G1X28& acc, const G1X28& q) {  // add-2008-s
    if (is_identity(q)) return;
    if (is_identity(acc)) {
        acc = q;
        return;
    }
    Fp28 U1 = mul(acc.X, q.ZZ), U2 = mul(q.X, acc.ZZ);
    Fp28 S1 = mul(acc.Y, q.ZZZ), S2 = mul(q.Y, acc.ZZZ);
    Fp28 Pp = sub<3>(U2, U1), R = sub<3>(S2, S1);  // k 5
    Fp28 PP = sqr(Pp);
    if (mulout_is_zero(PP)) {
        Fp28 RR = sqr(R);
        if (mulout_is_zero(RR)) acc = dbl(acc);
        else acc = identity();
        return;
    }
    Fp28 PPP = mul(Pp, PP), Q = mul(U1, PP), RR = sqr(R);
    Fp28 X3 = norm(sub<3>(sub<3>(sub<3>(RR, PPP), Q), Q));
    Fp28 Y3 = norm(sub<3>(mul(R, sub<12>(Q, X3)), mul(S1, PPP)));
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = mul(mul(acc.ZZ, q.ZZ), PP);
    acc.ZZZ = mul(mul(acc.ZZZ, q.ZZZ), PPP);
}

// acc += *q with q left in memory (LDS or global): every coordinate of q is read where the formula uses it, so only acc and the
// formula's temporaries stay in registers.  For msm_reduce: with `run` / `acc` / the loaded bucket all resident the kernel needs 354
// registers (one wave per SIMD, 180 of them parked in accumulation registers); in this form it fits the 256-register budget of two
// waves per SIMD and can share a SIMD with an accumulate wave of another MSM.  Same formula and case analysis as add_full.
BZK_HD void add_mem(G1X28& acc, const G1X28* q) {
#if defined(__HIP_DEVICE_COMPILE__)
#define BZK_G1_FENCE() __asm__ volatile("" ::: "memory")
#else
#define BZK_G1_FENCE() ((void)0)
#endif
    {
        const Fp28 qzz = q->ZZ;
        if (limbs_all_zero(qzz)) return;  // q is the identity
        if (is_identity(acc)) {
            acc = *q;
            return;
        }
    }
    BZK_G1_FENCE();
    Fp28 U1 = mul(acc.X, q->ZZ);
    BZK_G1_FENCE();
    Fp28 U2 = mul(q->X, acc.ZZ);
    Fp28 Pp = sub<3>(U2, U1);
    BZK_G1_FENCE();
    Fp28 S1 = mul(acc.Y, q->ZZZ);
    BZK_G1_FENCE();
    Fp28 S2 = mul(q->Y, acc.ZZZ);
    Fp28 R = sub<3>(S2, S1);  // k 5
    Fp28 PP = sqr(Pp);
    if (mulout_is_zero(PP)) {
        Fp28 RR = sqr(R);
        if (mulout_is_zero(RR)) acc = dbl(acc);
        else acc = identity();
        return;
    }
    Fp28 PPP = mul(Pp, PP), Q = mul(U1, PP), RR = sqr(R);
    Fp28 X3 = norm(sub<3>(sub<3>(sub<3>(RR, PPP), Q), Q));
    Fp28 Y3 = norm(sub<3>(mul(R, sub<12>(Q, X3)), mul(S1, PPP)));
    acc.X = X3;
    acc.Y = Y3;
    BZK_G1_FENCE();
    acc.ZZ = mul(mul(acc.ZZ, q->ZZ), PP);
    BZK_G1_FENCE();
    acc.ZZZ = mul(mul(acc.ZZZ, q->ZZZ), PPP);
#undef BZK_G1_FENCE
}

BZK_HD G1X28 mul_u32(const G1X28& p, uint32_t k) {
    G1X28 r = identity();
    for (int i = 31; i >= 0; --i) {
        r = dbl(r);
        if ((k >> i) & 1) add_full(r, p);
    }
    return r;
}

BZK_HD G1A28 affine_to28(const G1Affine& a) { return {to28(a.x), to28(a.y)}; }

// -> standard XYZZ over 12x32 limbs (canonical field elements) for the host-side Horner / packing
BZK_HD G1Xyzz to_std(const G1X28& p) {
    if (is_identity(p)) return xyzz_identity<FpOps>();
    return {from28(p.X), from28(p.Y), from28(p.ZZ), from28(p.ZZZ)};
}

}  // namespace g1x28

// G2 over Fp2x28: generic XYZZ code, plus conversions with the 12 x 32-bit memory form
typedef AffineT<Fp2x28Ops> G2A28;  // 224 B
typedef XyzzT<Fp2x28Ops> G2X28;    // 448 B
namespace g2x28 {
#ifndef BZK_G2_FAST_MIXED
#define BZK_G2_FAST_MIXED 1  // 0: the accumulation uses the generic xyzz_add_mixed<Fp2x28Ops> (A/B builds)
#endif
// acc += q (q affine, never the identity; neg_q: add -q) with STATIC bounds instead of the generic code's uniform "reduce every sum and
// difference below 3p" (16 quotient-estimate reductions per mixed addition; here 2) and with the Y coordinate's two Fp2 products under
// one reduction per component (mul4_body).  Invariants of the accumulator: X, Y components normalised and < 3p; ZZ, ZZZ outputs of
// Fp2x28Ops::mul / sqr (c0 < 5p, c1 < 8p, normalised); identity <=> every ZZ limb zero (the only way an accumulator becomes the
// identity is the explicit assignment below; a non-identity point has ZZ != 0 mod p).  The results satisfy the generic discipline
// (every component < 8p), so the tail kernels consume them unchanged.  Bounds (k = value < k p; all operands of a product normalised
// unless noted):
//   U2 = q.x ZZ: (8 + 8)(5 + 8) = 208, S2 = q.y ZZZ: (12 + 12)(13) = 312 -> (5, 8)     Pp = U2 - X, R = S2 - Y : sub<3> + norm, k <= 11
//   PP, RR = squares: (c0 + c1)(c0 - c1) with sub<12> (k 11 < 12): (22)(23) = 506 <= 2048; 2 c0 c1 -> (2, 4)
//   PPP = Pp PP: (22)(6);  Q = X PP: (6)(6);  X3 = RR - PPP - 2 Q: three sub<12> (limbs < 2^30.8, k <= 40) + reduce -> < 3p
//   T = Q - X3: sub<3> + norm, k <= 11
//   Y3.re = R0 T0 + (12p - R1) T1 + (3p - Y0) PPP0 + Y1 PPP1      (121 + 132 + 15 + 24 = 292 <= 2048; columns 14 * 7 * 2^56 < 2^64)
//   Y3.im = R0 T1 + R1 T0 + (3p - Y0) PPP1 + (3p - Y1) PPP0       (121 + 121 + 24 + 15)                       -> product outputs, k 2
template <class Pre = g1x28::NoPre>
BZK_HD void add_mixed(XyzzT<Fp2x28Ops>& acc, const AffineT<Fp2x28Ops>& q_in, bool neg_q, Pre&& pre = Pre()) {  // pre: see g1x28::add_mixed
    using namespace fp28;
    typedef Fp2x28Ops F;
    // q: components normalised and < 8p - converted bases are product outputs (< 2p), table entries and the group sums of the
    // de-duplication come out of Fp2 products (c0 < 5p, c1 < 8p)
    AffineT<Fp2x28Ops> q = q_in;
    if (neg_q) q.y = {norm(sub<12>(zero(), q.y.c0)), norm(sub<12>(zero(), q.y.c1))};  // 12p - y, k 12
    if (limbs_all_zero(acc.ZZ.c0) && limbs_all_zero(acc.ZZ.c1)) {
        pre();
        acc = {{reduce(q.x.c0), reduce(q.x.c1)}, {reduce(q.y.c0), reduce(q.y.c1)}, F::one(), F::one()};  // X, Y < 3p (once per run)
        return;
    }
    const Fp2x28 U2 = F::mul(q.x, acc.ZZ), S2 = F::mul(q.y, acc.ZZZ);
    const Fp2x28 Pp = {norm(sub<3>(U2.c0, acc.X.c0)), norm(sub<3>(U2.c1, acc.X.c1))};
    const Fp2x28 R = {norm(sub<3>(S2.c0, acc.Y.c0)), norm(sub<3>(S2.c1, acc.Y.c1))};
    Fp2x28 PP;
    PP.c0 = mul(add(Pp.c0, Pp.c1), sub<12>(Pp.c0, Pp.c1));
    {
        const Fp28 m = mul(Pp.c0, Pp.c1);
        PP.c1 = norm(add(m, m));
    }
    if (mulout_is_zero(PP.c0) && F::is_zero(Pp)) {  // same x (exact test only when the cheap one fires): doubling or cancellation
        if (F::is_zero(R)) acc = xyzz_dbl_affine<F>(q);
        else acc = xyzz_identity<F>();
        pre();
        return;
    }
    const Fp2x28 PPP = F::mul(Pp, PP), Q = F::mul(acc.X, PP);
    Fp2x28 RR;
    RR.c0 = mul(add(R.c0, R.c1), sub<12>(R.c0, R.c1));
    {
        const Fp28 m = mul(R.c0, R.c1);
        RR.c1 = norm(add(m, m));
    }
    Fp2x28 X3;
    X3.c0 = reduce(sub<12>(sub<12>(sub<12>(RR.c0, PPP.c0), Q.c0), Q.c0));
    X3.c1 = reduce(sub<12>(sub<12>(sub<12>(RR.c1, PPP.c1), Q.c1), Q.c1));
    const Fp2x28 T = {norm(sub<3>(Q.c0, X3.c0)), norm(sub<3>(Q.c1, X3.c1))};
    acc.ZZ = F::mul(acc.ZZ, PP);
    acc.ZZZ = F::mul(acc.ZZZ, PPP);
    pre();
    const Fp28 nR1 = sub<12>(zero(), R.c1), nY0 = sub<3>(zero(), acc.Y.c0), nY1 = sub<3>(zero(), acc.Y.c1);
    Fp2x28 Y3;
    Y3.c0 = mul4_body(R.c0, T.c0, nR1, T.c1, nY0, PPP.c0, acc.Y.c1, PPP.c1);
    Y3.c1 = mul4_body(R.c0, T.c1, R.c1, T.c0, nY0, PPP.c1, nY1, PPP.c0);
    acc.X = X3;
    acc.Y = Y3;
}
// ---- general addition and doubling of the tail kernels with the same static bounds (folds, bucket reduction, window sums: 4
// quotient-estimate reductions per operation instead of 20 / 16 in the generic code).  Inputs follow the generic discipline (every
// component normalised and < 8p; identity <=> ZZ limbs all zero), outputs do too (X, Y < 3p; ZZ, ZZZ product outputs), so generic and
// static code mix freely.  The Y coordinate keeps its two Fp2 products (the fused form is inlined code, and these functions are
// expanded several times per kernel).  sq(): (c0 + c1)(c0 - c1), 2 c0 c1 for normalised components below kp, K > k the next table.
#ifndef BZK_G2_FAST_TAILS
#define BZK_G2_FAST_TAILS 1  // 0: generic xyzz_add_mem / xyzz_add / xyzz_dbl in the tail kernels (A/B builds)
#endif
template <int K>
BZK_HD Fp2x28 sq(const Fp2x28& a) {
    using namespace fp28;
    Fp2x28 r;
    r.c0 = mul(add(a.c0, a.c1), sub<K>(a.c0, a.c1));
    const Fp28 m = mul(a.c0, a.c1);
    r.c1 = norm(add(m, m));
    return r;
}
BZK_HD bool is_identity(const XyzzT<Fp2x28Ops>& p) { return fp28::limbs_all_zero(p.ZZ.c0) && fp28::limbs_all_zero(p.ZZ.c1); }

BZK_HD XyzzT<Fp2x28Ops> dbl(const XyzzT<Fp2x28Ops>& p) {  // dbl-2008-s-1
    using namespace fp28;
    typedef Fp2x28Ops F;
    if (is_identity(p)) return p;
    const Fp2x28 U = {norm(add(p.Y.c0, p.Y.c0)), norm(add(p.Y.c1, p.Y.c1))};  // k 16
    const Fp2x28 V = sq<24>(U);                                                 // (32)(16 + 24) = 1280
    const Fp2x28 Wv = F::mul(U, V), S = F::mul(p.X, V);                          // (32)(6), (16)(6)
    const Fp2x28 xx = sq<12>(p.X);                                               // (16)(8 + 12)
    const Fp2x28 M = {norm(add(add(xx.c0, xx.c0), xx.c0)), norm(add(add(xx.c1, xx.c1), xx.c1))};  // (6, 12)
    const Fp2x28 MM = sq<24>(M);                                                 // (18)(6 + 24)
    XyzzT<Fp2x28Ops> r;
    r.X = {reduce(sub<12>(sub<12>(MM.c0, S.c0), S.c0)), reduce(sub<12>(sub<12>(MM.c1, S.c1), S.c1))};
    const Fp2x28 T = {norm(sub<3>(S.c0, r.X.c0)), norm(sub<3>(S.c1, r.X.c1))};  // k 11
    const Fp2x28 m1 = F::mul(M, T), m2 = F::mul(Wv, p.Y);                        // (18)(22), (13)(16)
    r.Y = {reduce(sub<12>(m1.c0, m2.c0)), reduce(sub<12>(m1.c1, m2.c1))};
    r.ZZ = F::mul(V, p.ZZ);
    r.ZZZ = F::mul(Wv, p.ZZZ);
    return r;
}

// acc += *q, q left in memory (each coordinate read where the formula uses it, see xyzz_add_mem)
BZK_HD void add_mem(XyzzT<Fp2x28Ops>& acc, const XyzzT<Fp2x28Ops>* q) {
    using namespace fp28;
    typedef Fp2x28Ops F;
#if defined(__HIP_DEVICE_COMPILE__)
#define BZK_G2_FENCE() __asm__ volatile("" ::: "memory")
#else
#define BZK_G2_FENCE() ((void)0)
#endif
    {
        const Fp2x28 qzz = q->ZZ;
        if (limbs_all_zero(qzz.c0) && limbs_all_zero(qzz.c1)) return;  // q is the identity
        if (is_identity(acc)) {
            acc = *q;
            return;
        }
    }
    BZK_G2_FENCE();
    const Fp2x28 U1 = F::mul(acc.X, q->ZZ);  // (16)(16) = 256
    BZK_G2_FENCE();
    const Fp2x28 U2 = F::mul(q->X, acc.ZZ);
    const Fp2x28 Pp = {norm(sub<12>(U2.c0, U1.c0)), norm(sub<12>(U2.c1, U1.c1))};  // k 20
    BZK_G2_FENCE();
    const Fp2x28 S1 = F::mul(acc.Y, q->ZZZ);
    BZK_G2_FENCE();
    const Fp2x28 S2 = F::mul(q->Y, acc.ZZZ);
    const Fp2x28 R = {norm(sub<12>(S2.c0, S1.c0)), norm(sub<12>(S2.c1, S1.c1))};   // k 20
    const Fp2x28 PP = sq<24>(Pp);                                                    // (40)(20 + 24) = 1760
    if (mulout_is_zero(PP.c0) && F::is_zero(Pp)) {  // same x (exact test only when the cheap one fires)
        if (F::is_zero(R)) acc = dbl(acc);
        else acc = xyzz_identity<F>();
        return;
    }
    const Fp2x28 PPP = F::mul(Pp, PP), Q = F::mul(U1, PP);  // (40)(6), (13)(6)
    const Fp2x28 RR = sq<24>(R);
    const Fp2x28 X3 = {reduce(sub<12>(sub<12>(sub<12>(RR.c0, PPP.c0), Q.c0), Q.c0)),
                       reduce(sub<12>(sub<12>(sub<12>(RR.c1, PPP.c1), Q.c1), Q.c1))};
    const Fp2x28 T = {norm(sub<3>(Q.c0, X3.c0)), norm(sub<3>(Q.c1, X3.c1))};        // k 11
    const Fp2x28 m1 = F::mul(R, T), m2 = F::mul(S1, PPP);                            // (40)(22) = 880, (13)(13)
    acc.X = X3;
    acc.Y = {reduce(sub<12>(m1.c0, m2.c0)), reduce(sub<12>(m1.c1, m2.c1))};
    BZK_G2_FENCE();
    acc.ZZ = F::mul(F::mul(acc.ZZ, q->ZZ), PP);
    BZK_G2_FENCE();
    acc.ZZZ = F::mul(F::mul(acc.ZZZ, q->ZZZ), PPP);
#undef BZK_G2_FENCE
}
BZK_HD XyzzT<Fp2x28Ops> mul_u32(const XyzzT<Fp2x28Ops>& p, uint32_t k) {
    XyzzT<Fp2x28Ops> r = xyzz_identity<Fp2x28Ops>();
    for (int i = 31; i >= 0; --i) {
        r = dbl(r);
        if ((k >> i) & 1) add_mem(r, &p);
    }
    return r;
}
BZK_HD Fp2x28 fp2_to28(const Fp2& a) { return {fp28::to28(a.c0), fp28::to28(a.c1)}; }
BZK_HD Fp2 fp2_from28(const Fp2x28& a) { return {fp28::from28(a.c0), fp28::from28(a.c1)}; }
BZK_HD G2A28 affine_to28(const G2Affine& a) { return {fp2_to28(a.x), fp2_to28(a.y)}; }
BZK_HD G2Xyzz to_std(const G2X28& p) {
    if (xyzz_is_identity<Fp2x28Ops>(p)) return xyzz_identity<Fp2Ops>();
    return {fp2_from28(p.X), fp2_from28(p.Y), fp2_from28(p.ZZ), fp2_from28(p.ZZZ)};
}
}  // namespace g2x28
}  // namespace bzk
