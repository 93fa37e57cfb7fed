// Host-side Fr (BLS12-381 scalar field) on 4 x 64-bit limbs for the witness generator and the host hasher: a Groth16 witness of the
// 16-tx Update circuit is ~0.2 CPU-seconds, more than half of it field products (profile of bzk_mpn_update_synthesize), and the host
// CPUs are the scarce resource of the pipelined prover on a quota'd box (HISTORY.md 6b).  Same Montgomery form and byte layout as `Fr`
// (8 x 32-bit limbs, little endian), canonical results - so every value is identical to what fe_mul<FrParams> returns.
//   mul : fully unrolled CIOS product on unsigned __int128
//   dot : sum_k a[k] b[k] with ONE Montgomery reduction (column-wise accumulation of the 512-bit products): the dense MDS rows of
//         Poseidon are dot products of length t = 3 .. 17 - 16 n + 20 word multiplications instead of 36 n
#pragma once
#include <stdint.h>
#include <string.h>

#include "bzk_field.cuh"

namespace bzk {
namespace hfr {
typedef unsigned __int128 u128;
constexpr uint64_t P0 = (uint64_t)FrParams::MOD[0] | ((uint64_t)FrParams::MOD[1] << 32);
constexpr uint64_t P1 = (uint64_t)FrParams::MOD[2] | ((uint64_t)FrParams::MOD[3] << 32);
constexpr uint64_t P2 = (uint64_t)FrParams::MOD[4] | ((uint64_t)FrParams::MOD[5] << 32);
constexpr uint64_t P3 = (uint64_t)FrParams::MOD[6] | ((uint64_t)FrParams::MOD[7] << 32);
constexpr uint64_t X32 = (uint64_t)(uint32_t)(0u - FrParams::INV);  // r^-1 mod 2^32
constexpr uint64_t X64 = X32 * (2 - P0 * X32);                      // one Newton step: r^-1 mod 2^64
constexpr uint64_t INV = (uint64_t)0 - X64;                         // -r^-1 mod 2^64

// t (4 limbs + a possible multiple of r) -> canonical Fr; `top` = limb 4
inline Fr finish(uint64_t t0, uint64_t t1, uint64_t t2, uint64_t t3, uint64_t top) {
    for (;;) {
        const u128 d0 = (u128)t0 - P0;
        const u128 d1 = (u128)t1 - P1 - (uint64_t)((d0 >> 64) & 1);
        const u128 d2 = (u128)t2 - P2 - (uint64_t)((d1 >> 64) & 1);
        const u128 d3 = (u128)t3 - P3 - (uint64_t)((d2 >> 64) & 1);
        const uint64_t bw = (uint64_t)((d3 >> 64) & 1);
        if (top == 0 && bw) break;  // t < r
        t0 = (uint64_t)d0; t1 = (uint64_t)d1; t2 = (uint64_t)d2; t3 = (uint64_t)d3;
        top -= bw;
    }
    Fr o;
    const uint64_t r[4] = {t0, t1, t2, t3};
    memcpy(o.l, r, 32);
    return o;
}

inline Fr mul(const Fr& a_, const Fr& b_) {
    uint64_t a[4], b[4];
    memcpy(a, a_.l, 32);
    memcpy(b, b_.l, 32);
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4;
#define BZK_HFR_ROW(bi)                                                                  \
    {                                                                                    \
        u128 s = (u128)a[0] * (bi) + t0;                                                 \
        const uint64_t lo0 = (uint64_t)s;                                                \
        uint64_t c = (uint64_t)(s >> 64);                                                \
        s = (u128)a[1] * (bi) + t1 + c; t1 = (uint64_t)s; c = (uint64_t)(s >> 64);      \
        s = (u128)a[2] * (bi) + t2 + c; t2 = (uint64_t)s; c = (uint64_t)(s >> 64);      \
        s = (u128)a[3] * (bi) + t3 + c; t3 = (uint64_t)s; t4 = (uint64_t)(s >> 64);     \
        const uint64_t m = lo0 * INV;                                                    \
        s = (u128)m * P0 + lo0; c = (uint64_t)(s >> 64);                                 \
        s = (u128)m * P1 + t1 + c; t0 = (uint64_t)s; c = (uint64_t)(s >> 64);           \
        s = (u128)m * P2 + t2 + c; t1 = (uint64_t)s; c = (uint64_t)(s >> 64);           \
        s = (u128)m * P3 + t3 + c; t2 = (uint64_t)s; c = (uint64_t)(s >> 64);           \
        t3 = t4 + c;                                                                     \
    }
    BZK_HFR_ROW(b[0]) BZK_HFR_ROW(b[1]) BZK_HFR_ROW(b[2]) BZK_HFR_ROW(b[3])
#undef BZK_HFR_ROW
    return finish(t0, t1, t2, t3, 0);  // r < 2^255: the running value stays below 2 r < 2^256
}
inline Fr sqr(const Fr& a) { return mul(a, a); }
// a^(r-2) by the same square-and-multiply ladder as fe_inv<FrParams> (inv(0) = 0), on the 64-bit-limb product: the same canonical value
inline Fr inv(const Fr& a) {
    uint32_t e[8];
    {
        uint64_t borrow = 2;
        for (int i = 0; i < 8; ++i) {
            const uint64_t d = (uint64_t)FrParams::MOD[i] - borrow;
            e[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
    }
    Fr r = Fr::one();
    for (int i = 255; i >= 0; --i) {
        r = mul(r, r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, a);
    }
    return r;
}

// sum_{k < n} a[k * sa] * b[k * sb], n <= 32 (Montgomery forms in, Montgomery form out)
inline Fr dot(const Fr* a, size_t sa, const Fr* b, size_t sb, int n) {
    uint64_t t[9];
    uint64_t acc0 = 0, acc1 = 0, acc2 = 0;
    for (int c = 0; c < 7; ++c) {  // column c of the 8-limb products, 192-bit accumulator
        const int i0 = c > 3 ? c - 3 : 0, i1 = c < 3 ? c : 3;
        for (int k = 0; k < n; ++k) {
            uint64_t x[4], y[4];
            memcpy(x, a[(size_t)k * sa].l, 32);
            memcpy(y, b[(size_t)k * sb].l, 32);
            for (int i = i0; i <= i1; ++i) {
                const u128 pr = (u128)x[i] * y[c - i];
                u128 s = (u128)acc0 + (uint64_t)pr;
                acc0 = (uint64_t)s;
                s = (u128)acc1 + (uint64_t)(pr >> 64) + (uint64_t)(s >> 64);
                acc1 = (uint64_t)s;
                acc2 += (uint64_t)(s >> 64);
            }
        }
        t[c] = acc0;
        acc0 = acc1;
        acc1 = acc2;
        acc2 = 0;
    }
    t[7] = acc0;
    t[8] = acc1;
    // word-by-word Montgomery reduction of the 9-limb value (< 32 r^2 < 2^515)
    const uint64_t p[4] = {P0, P1, P2, P3};
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[i] * INV;
        uint64_t c = 0;
        for (int j = 0; j < 4; ++j) {
            const u128 s = (u128)m * p[j] + t[i + j] + c;
            t[i + j] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
        for (int j = i + 4; c && j < 9; ++j) {
            const u128 s = (u128)t[j] + c;
            t[j] = (uint64_t)s;
            c = (uint64_t)(s >> 64);
        }
    }
    return finish(t[4], t[5], t[6], t[7], t[8]);
}
inline Fr dot(const Fr* a, const Fr* b, int n) { return dot(a, 1, b, 1, n); }

}  // namespace hfr
}  // namespace bzk
