// Dense matrix x vector over Fr on AVX-512 IFMA (vpmadd52luq / vpmadd52huq): the MDS product of a Poseidon round on the HOST.
//
// Why: the witness generator spends ~70 % of its CPU time in the Poseidon gadget's value path (host_r1cs.h g_poseidon_values; measured
// with cycle counters around it), and ~3/4 of that is the dense t x t MDS product of every one of the 64 - 65 rounds - the circuit
// allocates every lane of every round, so the sparse partial-round form of the native hash cannot be used there.  On 64-bit scalar code a
// round is t dot products of ~100 word multiplications each; here the t rows ride in the lanes of a zmm register:
//   * elements as 5 x 52-bit limbs (260 bits), one row per 64-bit lane, 8 rows per register group (t <= 8: one group; t <= 17: up to three)
//   * out_j = sum_k M[j][k] s_k : per k five broadcasts of s_k's limbs and 25 + 25 IFMAs into 10 column accumulators (a column takes at
//     most 2 * 5 * 17 addends below 2^52: no overflow of the 64-bit lanes), then ONE Montgomery reduction per row group - radix 2^52,
//     five steps - a carry sweep, one conditional subtraction of r, and the repack to 4 x 64-bit limbs
//   * values stay in the product's own form (Montgomery, R = 2^256): the table holds 16 M (mod r), so that reducing by 2^260 instead
//     of 2^256 gives exactly the 2^256-form result - every output is the canonical residue, bit-identical to hfr::dot
// Only used where the CPU has the instructions (checked once at run time; the functions carry their own target attribute, the rest of the
// library is compiled for baseline x86-64); everything falls back to hfr::dot otherwise.  BZK_HOST_IFMA=0 forces the fallback (A/B, tests).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "host_fr64.h"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define BZK_HAVE_IFMA_PATH 1
#endif

namespace bzk {
namespace hfr {

struct MdsTable {
    int t = 0, groups = 0;
    bool ifma = false;
    const Fr* mds = nullptr;  // row-major t x t (the scalar fall-back reads it)
    // [group][k][limb][lane]: limb `limb` of 16 * M[8 group + lane][k]; rows beyond t are zero
    alignas(64) uint64_t m[3][17][5][8];
};

inline bool ifma_available() {
#ifdef BZK_HAVE_IFMA_PATH
    static const bool ok = [] {
        const char* e = getenv("BZK_HOST_IFMA");
        if (e && atoi(e) == 0) return false;
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl");
    }();
    return ok;
#else
    return false;
#endif
}

constexpr uint64_t M52 = ((uint64_t)1 << 52) - 1;
inline void to52(const Fr& a, uint64_t o[5]) {
    uint64_t w[4];
    memcpy(w, a.l, 32);
    o[0] = w[0] & M52;
    o[1] = ((w[0] >> 52) | (w[1] << 12)) & M52;
    o[2] = ((w[1] >> 40) | (w[2] << 24)) & M52;
    o[3] = ((w[2] >> 28) | (w[3] << 36)) & M52;
    o[4] = w[3] >> 16;
}

// builds the lane layout of a t x t matrix (row-major Fr in the 2^256 Montgomery form, as the library holds its MDS matrices)
inline void mds_table_build(MdsTable& T, const Fr* mds, int t) {
    T.t = t;
    T.mds = mds;
    T.groups = (t + 7) / 8;
    T.ifma = ifma_available() && t >= 2 && t <= 17;
    memset(T.m, 0, sizeof T.m);
    if (!T.ifma) return;
    for (int j = 0; j < t; ++j)
        for (int k = 0; k < t; ++k) {
            Fr v = mds[(size_t)j * t + k];
            for (int d = 0; d < 4; ++d) v = fe_add<FrParams>(v, v);  // 16 M: the reduction below divides by 2^260, the form is 2^256
            uint64_t l[5];
            to52(v, l);
            for (int q = 0; q < 5; ++q) T.m[j / 8][k][q][j % 8] = l[q];
        }
}

#ifdef BZK_HAVE_IFMA_PATH
__attribute__((target("avx512f,avx512ifma,avx512vl"))) inline void mds_mul_ifma(const MdsTable& T, const Fr* s, Fr* out) {
    const int t = T.t;
    alignas(64) uint64_t s52[17][5];
    for (int k = 0; k < t; ++k) to52(s[k], s52[k]);
    // r in 52-bit limbs and -r^-1 mod 2^52
    uint64_t p52[5];
    {
        Fr pr;
        const uint64_t pw[4] = {P0, P1, P2, P3};
        memcpy(pr.l, pw, 32);
        to52(pr, p52);
    }
    const __m512i NP = _mm512_set1_epi64((long long)(INV & M52));
    const __m512i MASK = _mm512_set1_epi64((long long)M52);
    const __m512i Z = _mm512_setzero_si512();
    __m512i PL[5];
    for (int q = 0; q < 5; ++q) PL[q] = _mm512_set1_epi64((long long)p52[q]);
    for (int g = 0; g < T.groups; ++g) {
        __m512i acc[11];
        for (int c = 0; c < 11; ++c) acc[c] = Z;
        for (int k = 0; k < t; ++k) {
            __m512i b[5];
            for (int q = 0; q < 5; ++q) b[q] = _mm512_set1_epi64((long long)s52[k][q]);
            for (int q = 0; q < 5; ++q) {
                const __m512i a = _mm512_load_si512((const void*)T.m[g][k][q]);
                for (int m = 0; m < 5; ++m) {
                    acc[q + m] = _mm512_madd52lo_epu64(acc[q + m], a, b[m]);
                    acc[q + m + 1] = _mm512_madd52hi_epu64(acc[q + m + 1], a, b[m]);
                }
            }
        }
        // Montgomery reduction, radix 2^52: five steps clear limbs 0..4
        for (int i = 0; i < 5; ++i) {
            const __m512i mq = _mm512_madd52lo_epu64(Z, acc[i], NP);  // (acc_i mod 2^52) * (-r^-1) mod 2^52
            for (int q = 0; q < 5; ++q) {
                acc[i + q] = _mm512_madd52lo_epu64(acc[i + q], mq, PL[q]);
                acc[i + q + 1] = _mm512_madd52hi_epu64(acc[i + q + 1], mq, PL[q]);
            }
            acc[i + 1] = _mm512_add_epi64(acc[i + 1], _mm512_srli_epi64(acc[i], 52));
        }
        // carry sweep over the result limbs (acc[10] is only ever a carry target of limb 9's high halves: zero by the bound sum < 2^520)
        __m512i r0 = acc[5], r1 = acc[6], r2 = acc[7], r3 = acc[8], r4 = acc[9];
        r1 = _mm512_add_epi64(r1, _mm512_srli_epi64(r0, 52)); r0 = _mm512_and_si512(r0, MASK);
        r2 = _mm512_add_epi64(r2, _mm512_srli_epi64(r1, 52)); r1 = _mm512_and_si512(r1, MASK);
        r3 = _mm512_add_epi64(r3, _mm512_srli_epi64(r2, 52)); r2 = _mm512_and_si512(r2, MASK);
        r4 = _mm512_add_epi64(r4, _mm512_srli_epi64(r3, 52)); r3 = _mm512_and_si512(r3, MASK);
        // value < 2 r: subtract r where that does not borrow
        __m512i d0 = _mm512_sub_epi64(r0, PL[0]);
        __m512i d1 = _mm512_sub_epi64(_mm512_sub_epi64(r1, PL[1]), _mm512_srli_epi64(d0, 63));
        __m512i d2 = _mm512_sub_epi64(_mm512_sub_epi64(r2, PL[2]), _mm512_srli_epi64(d1, 63));
        __m512i d3 = _mm512_sub_epi64(_mm512_sub_epi64(r3, PL[3]), _mm512_srli_epi64(d2, 63));
        __m512i d4 = _mm512_sub_epi64(_mm512_sub_epi64(r4, PL[4]), _mm512_srli_epi64(d3, 63));
        const __mmask8 keep = _mm512_cmpneq_epi64_mask(_mm512_srli_epi64(d4, 63), Z);  // borrowed: the value was already below r
        r0 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d0, MASK), r0);
        r1 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d1, MASK), r1);
        r2 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d2, MASK), r2);
        r3 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d3, MASK), r3);
        r4 = _mm512_mask_blend_epi64(keep, d4, r4);
        // 5 x 52 -> 4 x 64
        alignas(64) uint64_t w[4][8];
        _mm512_store_si512((void*)w[0], _mm512_or_si512(r0, _mm512_slli_epi64(r1, 52)));
        _mm512_store_si512((void*)w[1], _mm512_or_si512(_mm512_srli_epi64(r1, 12), _mm512_slli_epi64(r2, 40)));
        _mm512_store_si512((void*)w[2], _mm512_or_si512(_mm512_srli_epi64(r2, 24), _mm512_slli_epi64(r3, 28)));
        _mm512_store_si512((void*)w[3], _mm512_or_si512(_mm512_srli_epi64(r3, 36), _mm512_slli_epi64(r4, 16)));
        for (int lane = 0; lane < 8 && 8 * g + lane < t; ++lane) {
            const uint64_t o[4] = {w[0][lane], w[1][lane], w[2][lane], w[3][lane]};
            memcpy(out[8 * g + lane].l, o, 32);
        }
    }
}
#endif

#ifdef BZK_HAVE_IFMA_PATH
// Eight independent products a_i b_i compared with c_i (all in the 2^256 Montgomery form, canonical), one row per lane: the satisfaction
// scan a_k b_k = c_k over a witness's constraint rows (0.9 M rows for a 16-tx batch - a quarter of the witness generator's CPU time on
// the generic 32-bit-limb product).  a is taken shifted left by four bits (16 a < 2^259 still fits five 52-bit limbs), so the
// reduction by 2^260 yields a b / 2^256 - the canonical product in the same form as c.  Returns the lane mask of rows whose product
// differs from c.  Rows are 32 bytes apart (`Fr` arrays).
__attribute__((target("avx512f,avx512ifma,avx512vl"))) inline unsigned products_mismatch8_ifma(const Fr* a, const Fr* b, const Fr* c) {
    const __m512i IDX = _mm512_set_epi64(28, 24, 20, 16, 12, 8, 4, 0);  // word offsets of eight consecutive rows
    const __m512i MASK = _mm512_set1_epi64((long long)M52);
    const __m512i Z = _mm512_setzero_si512();
    // (no lambdas here: a lambda body does not inherit the enclosing function's target attribute)
    __m512i aw[4], bw[4], cw[4];
    for (int i = 0; i < 4; ++i) {
        aw[i] = _mm512_i64gather_epi64(IDX, (const long long*)a + i, 8);
        bw[i] = _mm512_i64gather_epi64(IDX, (const long long*)b + i, 8);
        cw[i] = _mm512_i64gather_epi64(IDX, (const long long*)c + i, 8);
    }
    __m512i A[5], B[5], C[5];
    // 16 a
    A[0] = _mm512_and_si512(_mm512_slli_epi64(aw[0], 4), MASK);
    A[1] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(aw[0], 48), _mm512_slli_epi64(aw[1], 16)), MASK);
    A[2] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(aw[1], 36), _mm512_slli_epi64(aw[2], 28)), MASK);
    A[3] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(aw[2], 24), _mm512_slli_epi64(aw[3], 40)), MASK);
    A[4] = _mm512_srli_epi64(aw[3], 12);
#define BZK_SPLIT52(w, o)                                                                                         \
    o[0] = _mm512_and_si512(w[0], MASK);                                                                          \
    o[1] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w[0], 52), _mm512_slli_epi64(w[1], 12)), MASK);     \
    o[2] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w[1], 40), _mm512_slli_epi64(w[2], 24)), MASK);     \
    o[3] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w[2], 28), _mm512_slli_epi64(w[3], 36)), MASK);     \
    o[4] = _mm512_srli_epi64(w[3], 16);
    BZK_SPLIT52(bw, B)
    BZK_SPLIT52(cw, C)
#undef BZK_SPLIT52
    __m512i acc[11];
    for (int i = 0; i < 11; ++i) acc[i] = Z;
    for (int q = 0; q < 5; ++q)
        for (int m = 0; m < 5; ++m) {
            acc[q + m] = _mm512_madd52lo_epu64(acc[q + m], A[q], B[m]);
            acc[q + m + 1] = _mm512_madd52hi_epu64(acc[q + m + 1], A[q], B[m]);
        }
    uint64_t p52[5];
    {
        Fr pr;
        const uint64_t pw[4] = {P0, P1, P2, P3};
        memcpy(pr.l, pw, 32);
        to52(pr, p52);
    }
    const __m512i NP = _mm512_set1_epi64((long long)(INV & M52));
    __m512i PL[5];
    for (int q = 0; q < 5; ++q) PL[q] = _mm512_set1_epi64((long long)p52[q]);
    for (int i = 0; i < 5; ++i) {
        const __m512i mq = _mm512_madd52lo_epu64(Z, acc[i], NP);
        for (int q = 0; q < 5; ++q) {
            acc[i + q] = _mm512_madd52lo_epu64(acc[i + q], mq, PL[q]);
            acc[i + q + 1] = _mm512_madd52hi_epu64(acc[i + q + 1], mq, PL[q]);
        }
        acc[i + 1] = _mm512_add_epi64(acc[i + 1], _mm512_srli_epi64(acc[i], 52));
    }
    __m512i r0 = acc[5], r1 = acc[6], r2 = acc[7], r3 = acc[8], r4 = acc[9];
    r1 = _mm512_add_epi64(r1, _mm512_srli_epi64(r0, 52)); r0 = _mm512_and_si512(r0, MASK);
    r2 = _mm512_add_epi64(r2, _mm512_srli_epi64(r1, 52)); r1 = _mm512_and_si512(r1, MASK);
    r3 = _mm512_add_epi64(r3, _mm512_srli_epi64(r2, 52)); r2 = _mm512_and_si512(r2, MASK);
    r4 = _mm512_add_epi64(r4, _mm512_srli_epi64(r3, 52)); r3 = _mm512_and_si512(r3, MASK);
    __m512i d0 = _mm512_sub_epi64(r0, PL[0]);
    __m512i d1 = _mm512_sub_epi64(_mm512_sub_epi64(r1, PL[1]), _mm512_srli_epi64(d0, 63));
    __m512i d2 = _mm512_sub_epi64(_mm512_sub_epi64(r2, PL[2]), _mm512_srli_epi64(d1, 63));
    __m512i d3 = _mm512_sub_epi64(_mm512_sub_epi64(r3, PL[3]), _mm512_srli_epi64(d2, 63));
    __m512i d4 = _mm512_sub_epi64(_mm512_sub_epi64(r4, PL[4]), _mm512_srli_epi64(d3, 63));
    const __mmask8 keep = _mm512_cmpneq_epi64_mask(_mm512_srli_epi64(d4, 63), Z);
    r0 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d0, MASK), r0);
    r1 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d1, MASK), r1);
    r2 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d2, MASK), r2);
    r3 = _mm512_mask_blend_epi64(keep, _mm512_and_si512(d3, MASK), r3);
    r4 = _mm512_mask_blend_epi64(keep, d4, r4);
    return (unsigned)(_mm512_cmpneq_epi64_mask(r0, C[0]) | _mm512_cmpneq_epi64_mask(r1, C[1]) | _mm512_cmpneq_epi64_mask(r2, C[2]) |
                      _mm512_cmpneq_epi64_mask(r3, C[3]) | _mm512_cmpneq_epi64_mask(r4, C[4]));
}
#endif

// first k in [lo, hi) with a[k] b[k] != c[k] (canonical Montgomery values), or -1
inline long products_first_mismatch(const Fr* a, const Fr* b, const Fr* c, size_t lo, size_t hi) {
    size_t k = lo;
#ifdef BZK_HAVE_IFMA_PATH
    if (ifma_available()) {
        for (; k + 8 <= hi; k += 8) {
            const unsigned bad = products_mismatch8_ifma(a + k, b + k, c + k);
            if (bad) return (long)(k + (size_t)__builtin_ctz(bad));
        }
    }
#endif
    for (; k < hi; ++k)
        if (!mul(a[k], b[k]).equals(c[k])) return (long)k;
    return -1;
}

// out[j] = sum_k M[j][k] s[k], j < t (out must not alias s)
inline void mds_mul(const MdsTable& T, const Fr* s, Fr* out) {
#ifdef BZK_HAVE_IFMA_PATH
    if (T.ifma) {
        mds_mul_ifma(T, s, out);
        return;
    }
#endif
    for (int j = 0; j < T.t; ++j) out[j] = dot(T.mds + (size_t)j * T.t, s, T.t);
}

}  // namespace hfr
}  // namespace bzk
