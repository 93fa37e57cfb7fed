// K3: radix-2 NTT over Fr for gfx950.
//
// Replaces bellman 0.14 `EvaluationDomain::{fft, ifft, coset_fft, icoset_fft}` (third-party crate
// reached from `create_random_proof`, /root/reference/src/mpn/circuits/test.rs:135,175,215).
// omega = 7^((r-1)/2^32)^(2^(32-log_n)), coset shift g = 7 (ZkScalar generator,
// /root/reference/src/zk/mod.rs:204).  Natural order in and out.
//
// Structure: the index is split into at most three digits of <= 10 bits (n = R1 * R2 * R3); each pass transforms one
// digit with the whole R-point DFT done in LDS (one HBM round trip per pass instead of one per butterfly stage), the
// inter-pass twiddles come from two 32 KiB tables (w^e = lo[e & 1023] * hi[e >> 10]) and the final pass writes the
// digit-reversed (= natural) order directly in CC-wide chunks - no separate bit-reversal or copy pass.  The coset
// scaling g^i and the 1/n (g^-k) scaling of the inverse are fused into the first load / last store.
// HBM traffic per transform of size n: 64 B * n * passes  (passes = 1, 2, 3 for log_n <= 10, 20, 30).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "bzk_field.cuh"
#include "bzk_fr29.cuh"
#include <mutex>

#include "bzk_internal.h"

namespace bzk {

__device__ __forceinline__ Fr fr_mul(const Fr& a, const Fr& b) { return fe_mul<FrParams>(a, b); }

static Fr host_pow(Fr b, uint64_t e) {
    Fr r = Fr::one();
    while (e) {
        if (e & 1) r = fe_mul<FrParams>(r, b);
        b = fe_sqr<FrParams>(b);
        e >>= 1;
    }
    return r;
}

static Fr host_from_u64(uint64_t v) {
    Fr c = Fr::zero();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return fe_to_mont<FrParams>(c);
}

static Fr host_root_of_unity() {  // 7^((r-1)/2^32)
    uint32_t e[8];
    uint64_t borrow = 1;
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)FrParams::MOD[i] - borrow;
        e[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    // (r-1) >> 32  == drop limb 0
    Fr g = host_from_u64(7), r = Fr::one();
    for (int i = 255; i >= 32; --i) {
        r = fe_sqr<FrParams>(r);
        if ((e[i >> 5] >> (i & 31)) & 1) r = fe_mul<FrParams>(r, g);
    }
    return r;
}

static Fr host_omega(int log_n) {
    Fr w = host_root_of_unity();
    for (int i = log_n; i < 32; ++i) w = fe_sqr<FrParams>(w);
    return w;
}

// table[j] = base^j for j < count, from two small host tables: lo[j & 1023], hi[j >> 10]
__global__ void __launch_bounds__(256) pow_table_kernel(const Fr* __restrict__ lo, const Fr* __restrict__ hi, uint64_t count,
                                                        Fr* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    out[j] = fr_mul(lo[j & 1023], hi[j >> 10]);
}

static int32_t build_pow_table(bzk_ctx* ctx, const Fr& base, uint64_t count, void** out_dev) {
    const uint64_t n_hi = (count + 1023) / 1024;
    std::vector<Fr> lo(1024), hi(n_hi);
    lo[0] = Fr::one();
    for (int i = 1; i < 1024; ++i) lo[i] = fe_mul<FrParams>(lo[i - 1], base);
    Fr step = fe_mul<FrParams>(lo[1023], base);
    hi[0] = Fr::one();
    for (uint64_t i = 1; i < n_hi; ++i) hi[i] = fe_mul<FrParams>(hi[i - 1], step);
    void *dlo = nullptr, *dhi = nullptr, *dt = nullptr;
    BZK_HIP(ctx, hipMalloc(&dlo, 1024 * sizeof(Fr)));
    BZK_HIP(ctx, hipMalloc(&dhi, n_hi * sizeof(Fr)));
    BZK_HIP(ctx, hipMalloc(&dt, (count ? count : 1) * sizeof(Fr)));
    BZK_HIP(ctx, hipMemcpyAsync(dlo, lo.data(), 1024 * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipMemcpyAsync(dhi, hi.data(), n_hi * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    if (count) {
        BZK_LAUNCH(ctx, "ntt_pow_table", pow_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (const Fr*)dlo,
                   (const Fr*)dhi, count, (Fr*)dt);
    }
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    BZK_HIP(ctx, hipFree(dlo));
    BZK_HIP(ctx, hipFree(dhi));
    *out_dev = dt;
    return BZK_OK;
}

__device__ __forceinline__ uint32_t bitrev(uint32_t v, int bits) { return bits ? (__brev(v) >> (32 - bits)) : 0; }

__global__ void __launch_bounds__(256) ntt_scale_kernel(Fr* __restrict__ data, uint64_t n, const Fr* __restrict__ scale,
                                                        const Fr* __restrict__ pw) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = fr_mul(data[i], *scale);
    if (pw) v = fr_mul(v, pw[i]);
    data[i] = v;
}

// ------------------------------------------------------------------------------------------------
// One pass = an R-point DFT (R = 2^b <= 1024) along one digit of the index, for a tile of CC adjacent
// "columns", entirely in LDS, on the 9 x 29-bit reduced-radix field (bzk_fr29.cuh; 163 G products/s against 88 for
// the 8 x 32-bit CIOS).  Decimation in time: rows enter the tile bit-reversed and leave it in natural order; a
// butterfly is  t = v * w ; (u, v) <- (u + t, u - t + 3r)  - the subtrahend is always a fresh product (k 2), so no
// conditional subtraction is needed: values grow by at most 3r per stage (k <= 32 after 10 stages, the product
// accepts k <= 35) and only carries are propagated.  Elements are padded to 48 B (three 16-B accesses) in LDS and in
// the inter-pass buffer.
//   COL   pass: element (a, c) of the tile lives at src[base + a*S + c]; the result row ka is multiplied by the
//               inter-pass twiddle w_N'^(inner * ka) (N' = R*S, two-level table: lo[e & 1023] * hi[e >> 10]) and
//               written to the same position of dst.
//   FINAL pass: rows are contiguous (S = 1); the tile takes CC rows whose OUTPUT indices are adjacent and writes
//               X[(k1 + c) + R1 * (k2 + R2 * ka)]: the digit reversal of the whole transform, in CC-wide chunks.
// Conversions ride on products that are needed anyway: the first load multiplies the raw 8 x 32-bit Montgomery-256
// limbs by 2^266 (or by g^i * 2^266 for a coset transform), the last store by 2^256 (or by 1/n, g^-k / n times it)
// and reduces to canonical limbs.
// ------------------------------------------------------------------------------------------------
struct alignas(16) Fr29P {
    uint32_t l[12];  // 9 used
};
__device__ __forceinline__ Fr29 ld29(const Fr29P& p) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p.l[i];
    return r;
}
__device__ __forceinline__ Fr29P st29(const Fr29& a) {
    Fr29P p;
#pragma unroll
    for (int i = 0; i < 9; ++i) p.l[i] = a.l[i];
    p.l[9] = p.l[10] = p.l[11] = 0;
    return p;
}

struct NttPass {
    const void* src;   // first pass: Fr (8 x 32-bit Montgomery-256); later passes: Fr29P
    void* dst;         // final pass: Fr; earlier passes: Fr29P
    int b, log_cc, final_pass, first_pass;
    // inter-pass elements (round 4): 1 = 32 B (the 256-bit value repacked from / to the nine 29-bit limbs: a product output is < 2 r <
    // 2^256 and normalised) instead of the 48-B padded limb form - a third less HBM traffic between the passes for ~34 shift / mask
    // instructions per element and pass (PMC: 5.1 GB per 2^24-point transform against 1.07 GB algorithmic with 48-B elements)
    int ip32;
    // what the FIRST pass does with the raw 32-byte limbs it loads (x_raw = x * 2^256 mod r, < 2^256):
    //   FIRST_MUL   v = x_raw * (pre[i] or 2^266) * 2^-261          - the value is carried as x * 2^261 ("Montgomery-261")
    //   FIRST_RAW   v = x_raw, no multiplication (round 3)           - carried as x * 2^256; the twiddles stay Montgomery-261, so every
    //               product keeps that scale and the final store's constant is 2^261 instead of 2^256.  Stage 0 multiplies nothing
    //               and takes any operand below 2^256 (sub3 needs its subtrahend < 3 r and the top limb < 2^24.4)
    //   FIRST_PW    the h-polynomial's pointwise step fused into the load (round 3): with a_raw = A 2^256, b_raw = B 2^256 and
    //               c_raw = C 2^251 (the producing transform scales C by 2^-5 in ITS final store, for free)
    //               v = ((a_raw * b_raw) 2^-261 - c_raw + 3 r) * kmul 2^-261 = (A B - C) zinv * 2^261 for kmul = zinv * 2^271
    int first_mode;
    uint32_t xcd_group;   // COL passes: see the block -> tile mapping in the kernel (0 / 1: identity)
    const void* src_b;    // FIRST_PW
    const void* src_c;
    const Fr29P* kmul;
    uint64_t S;           // COL: column stride = size of the inner dimension
    uint32_t R1, R2;      // FINAL: sizes of the digits already transformed
    const Fr29P* tw_r;    // w_R^j * 2^261, j < R/2
    const Fr29P* tlo;     // COL: w_N'^j * 2^261, j < 1024
    const Fr29P* thi;     // COL: w_N'^(1024 j) * 2^261
    const Fr29P* tone;    // COL (round 6, run 31): w_N'^j * 2^261 for EVERY j < N' where N' <= 2^17 (the second boundary of a three-pass plan), or null: one load
                          // from a cache-resident table (<= 6 MB) instead of lo * hi - one of the transform's 15.5 products per element
    const Fr29P* pre;     // FIRST_MUL: g^i * 2^266 by input index (forward coset transform) or null (constant 2^266)
    const Fr29P* post;    // final pass: a multiplier per OUTPUT index ((g^-k / n) 2^256, or (g^k / n) 2^261 for the fused inverse -> coset hand-over) or null
    const Fr29P* post_c;  // final pass: one element (see NttTables::post_one)
};
enum { FIRST_MUL = 0, FIRST_RAW = 1, FIRST_PW = 2 };

// LDS tile: 9 dwords (36 B) per element since round 3 (was 48 B, padded for 16-byte accesses): a 1024-element tile is 36 KiB, so FOUR
// workgroups share a CU's 160 KiB instead of three (4 waves per SIMD; the 2048 tiles of a 2^20-point pass are exactly two rounds of
// the 1024 resident workgroups instead of 2.67), and the odd dword stride spreads neighbouring elements over all 32 banks.
static constexpr uint32_t LDSW = 9;
// Bank swizzle.  With 9-dword elements the bank of limb k of element p is (9 p + k) mod 32 - a bijection of p mod 32 - so the 32 lanes
// of a ds_*_b32 lane group conflict exactly when their element indices agree mod 32.  The kernel's access patterns vary these index
// bits across a lane group: the bit-reversed scatter of the load phase {5..9} (ALL lanes on one bank: 32-way), round 0 {2..6} (4-way),
// round 2 {0,1,4,5,6} (4-way), round 4 {0..3,6} (2-way), later rounds and the store phase {0..4} - PMC: SQ_LDS_BANK_CONFLICT 13.3 M
// cycles per 85 us launch at 2^20, SQ_WAIT_INST_LDS 18 % of the wave cycles (profiles/r03_run5_ntt_pmc.txt).  The element index is
// therefore permuted inside its block of 32 by a GF(2)-linear map of the block number h = idx >> 5:
//     low5 ^= h ^ ((h & 3) << 2) ^ ((h & 2) << 3)
// which is a bijection on every one of those bit sets (checked for every (b, log_cc) the planner produces: tools/ntt_bank_sim.py).
__device__ __forceinline__ uint32_t lds_sw(uint32_t idx) {
    const uint32_t h = idx >> 5;
    return idx ^ ((h ^ ((h & 3u) << 2) ^ ((h & 2u) << 3)) & 31u);
}
__device__ __forceinline__ Fr29 lds_ld(const uint32_t* tile, uint32_t idx) {
    Fr29 r;
    const uint32_t* p = tile + lds_sw(idx) * LDSW;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p[i];
    return r;
}
__device__ __forceinline__ void lds_st(uint32_t* tile, uint32_t idx, const Fr29& v) {
    uint32_t* p = tile + lds_sw(idx) * LDSW;
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = v.l[i];
}

__device__ __forceinline__ Fr29 ntt_first_load(const NttPass& a, uint64_t addr, const Fr29& c_in) {
    const Fr29 x = fr29::repack_from32(((const Fr*)a.src)[addr]);
    if (a.first_mode == FIRST_RAW) return x;
    if (a.first_mode == FIRST_PW) {
        const Fr29 y = fr29::repack_from32(((const Fr*)a.src_b)[addr]);
        const Fr29 z = fr29::repack_from32(((const Fr*)a.src_c)[addr]);
        return fr29::mul(fr29::sub3(fr29::mul(x, y), z), ld29(*a.kmul));
    }
    return fr29::mul(x, a.pre ? ld29(a.pre[addr]) : c_in);
}

// OCC: waves per SIMD the kernel is compiled for (4 = one workgroup per 36 KiB tile slot of the CU, needs <= 128 registers);
// PF: 2 = all three twiddles of a round requested before the LDS reads, 1 = the first-stage twiddle early and the second-stage pair
// after the first two products, 0 = every twiddle loaded where it is used.  A/B builds: env BZK_NTT_VARIANT (ntt_run_ex).
// INL: 1 = the butterflies' products are inlined (fr29::mul_body; ~1.5 KB of code each, eight sites: still inside the 64 KiB instruction
// cache, and two independent products of a butterfly can interleave), 0 = calls to the one resident copy (fr29::mul)
template <int INL>
__device__ __forceinline__ Fr29 nmul(const Fr29& x, const Fr29& y) {
    if constexpr (INL) return fr29::mul_body(x, y);
    else return fr29::mul(x, y);
}
template <int OCC, int PF, int INL>
__global__ void __launch_bounds__(256, OCC) ntt_pass_kernel(NttPass a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    const int b = a.b, lc = a.log_cc;
    const uint32_t R = 1u << b, CC = 1u << lc, tile_n = R << lc;
    uint64_t base = 0, inner0 = 0;
    uint32_t k1 = 0, k2 = 0;
    const Fr29 c_in = fr29::from_consts(fr29::C_IN);
    // load: global row r goes to LDS row bitrev(r) (decimation in time)
    if (!a.final_pass) {
        const uint64_t gpo = a.S >> lc;  // column groups per outer block
        // Tiles narrower than a 128-byte line (CC x 32 B on the first pass): the G = 4 / CC tiles that share each line are given to
        // blocks x, x + 8, x + 16, .. - workgroups are dealt to the 8 XCDs round-robin, so these land on ONE XCD back to back and three
        // of the four reads of a line (and the partial-line stores) meet in that XCD's L2 instead of fetching the line four times
        uint64_t bid = blockIdx.x;
        if (a.xcd_group > 1) {
            const uint64_t G = a.xcd_group, x = bid & 7, y = bid >> 3;
            bid = ((y / G) * 8 + x) * G + (y % G);
        }
        const uint64_t o = bid / gpo, cg = bid % gpo;
        inner0 = cg << lc;
        base = o * R * a.S + inner0;
        for (uint32_t t = threadIdx.x; t < tile_n; t += blockDim.x) {
            const uint32_t c = t & (CC - 1), r = t >> lc;
            const uint64_t addr = base + (uint64_t)r * a.S + c;
            const Fr29 v = a.first_pass ? ntt_first_load(a, addr, c_in)
                                        : (a.ip32 ? fr29::repack_from32(((const Fr*)a.src)[addr]) : ld29(((const Fr29P*)a.src)[addr]));
            lds_st(tile, (bitrev(r, b) << lc) + c, v);
        }
    } else {
        const uint32_t groups = a.R1 >> lc;
        k2 = blockIdx.x / groups;
        k1 = (blockIdx.x % groups) << lc;
        for (uint32_t t = threadIdx.x; t < tile_n; t += blockDim.x) {
            const uint32_t r = t & (R - 1), c = t >> b;
            const uint64_t addr = ((uint64_t)(k1 + c) * a.R2 + k2) * R + r;
            const Fr29 v = a.first_pass ? ntt_first_load(a, addr, c_in)
                                        : (a.ip32 ? fr29::repack_from32(((const Fr*)a.src)[addr]) : ld29(((const Fr29P*)a.src)[addr]));
            lds_st(tile, (bitrev(r, b) << lc) + c, v);
        }
    }
    __syncthreads();
    // Butterfly stages, two per LDS round trip (radix 4 in registers: half the barriers and half the LDS traffic of one stage per
    // round).  Stage 0 has the single twiddle w^0 = one and its operands are fresh products or raw inputs below 2^256 (what sub3
    // needs), so it multiplies nothing.  An odd stage count starts with that lone multiplication-free stage.
    int s0 = 0;
    if (b & 1) {
        for (uint32_t q = threadIdx.x; q < tile_n / 2; q += blockDim.x) {
            const uint32_t c = q & (CC - 1), bq = q >> lc;
            const uint32_t i0 = ((bq << 1) << lc) + c, i1 = (((bq << 1) + 1) << lc) + c;
            const Fr29 u = lds_ld(tile, i0), t = lds_ld(tile, i1);
            lds_st(tile, i0, fr29::norm(fr29::add(u, t)));
            lds_st(tile, i1, fr29::sub3(u, t));
        }
        __syncthreads();
        s0 = 1;
    }
    for (int s = s0; s < b; s += 2) {
        const uint32_t m = 1u << s;
        for (uint32_t q = threadIdx.x; q < tile_n / 4; q += blockDim.x) {
            const uint32_t c = q & (CC - 1), bq = q >> lc;
            const uint32_t j = bq & (m - 1);
            const uint32_t k0 = ((bq >> s) << (s + 2)) + j;
            const uint32_t i0 = (k0 << lc) + c, i1 = ((k0 + m) << lc) + c, i2 = ((k0 + 2 * m) << lc) + c, i3 = ((k0 + 3 * m) << lc) + c;
            // the round's three twiddles depend on the position only: their (L2-resident) loads are issued before the LDS reads they
            // would otherwise queue behind, and complete under them
            const Fr29P* pa = a.tw_r + ((size_t)j << (b - 1 - s));
            const Fr29P* pb = a.tw_r + ((size_t)j << (b - 2 - s));
            const Fr29P* pc = a.tw_r + ((size_t)(j + m) << (b - 2 - s));
            Fr29 wa, wb, wc;
            if (PF >= 1 && s) wa = ld29(*pa);
            if (PF >= 2) { wb = ld29(*pb); wc = ld29(*pc); }
            Fr29 x0 = lds_ld(tile, i0), x1 = lds_ld(tile, i1), x2 = lds_ld(tile, i2), x3 = lds_ld(tile, i3);
            if (s) {  // stage s: both butterflies of the group use w^(j 2^(b-1-s))
                if (PF == 0) wa = ld29(*pa);
                x1 = nmul<INL>(x1, wa);
                x3 = nmul<INL>(x3, wa);
            }
            if (PF < 2) { wb = ld29(*pb); wc = ld29(*pc); }
            const Fr29 y0 = fr29::norm(fr29::add(x0, x1)), y1 = fr29::sub3(x0, x1);
            const Fr29 y2 = fr29::norm(fr29::add(x2, x3)), y3 = fr29::sub3(x2, x3);
            // stage s + 1: (y0, y2) at position j, (y1, y3) at position j + m of their 4m-blocks
            const Fr29 u2 = nmul<INL>(y2, wb);
            const Fr29 u3 = nmul<INL>(y3, wc);
            lds_st(tile, i0, fr29::norm(fr29::add(y0, u2)));
            lds_st(tile, i2, fr29::sub3(y0, u2));
            lds_st(tile, i1, fr29::norm(fr29::add(y1, u3)));
            lds_st(tile, i3, fr29::sub3(y1, u3));
        }
        __syncthreads();
    }
    for (uint32_t t = threadIdx.x; t < tile_n; t += blockDim.x) {
        const uint32_t c = t & (CC - 1), ka = t >> lc;  // natural order after the DIT stages
        Fr29 v = lds_ld(tile, t);
        if (!a.final_pass) {
            const uint64_t e = (inner0 + c) * ka;
            Fr29 w;
            if (a.tone) {  // workgroup-uniform
                w = ld29(a.tone[e]);
            } else {
                w = ld29(a.tlo[e & 1023]);
                if (e >> 10) w = fr29::mul(w, ld29(a.thi[e >> 10]));
            }
            const Fr29 o = fr29::mul(v, w);  // k 2, normalised: below 2^256
            if (a.ip32) ((Fr*)a.dst)[base + (uint64_t)ka * a.S + c] = fr29::repack_to32(o);
            else ((Fr29P*)a.dst)[base + (uint64_t)ka * a.S + c] = st29(o);
        } else {
            const uint64_t k = (uint64_t)(k1 + c) + (uint64_t)a.R1 * (k2 + (uint64_t)a.R2 * ka);
            ((Fr*)a.dst)[k] = fr29::from29_scaled(v, ld29(a.post ? a.post[k] : *a.post_c));
        }
    }
}

// out[i] = 9 x 29-bit limbs of the raw value of (tab[i] * c) in Montgomery-256 form, i.e. (x_i * c') * 2^256 mod r:
// c = Montgomery(2^5) turns Montgomery-256 twiddles into Montgomery-261 ones, c = Montgomery(2^10) gives the x * 2^266
// multipliers of the first load, c = one keeps x * 2^256 (multipliers of the final store)
__global__ void __launch_bounds__(256) ntt_table29_kernel(const Fr* __restrict__ tab, uint64_t count, Fr c, Fr29P* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = st29(fr29::repack_from32(fr_mul(tab[i], c)));
}

// per-size plan: digit split, twiddle tables for both directions, coset tables (all in the 48-byte 9 x 29-bit form)
struct NttTables {
    int nb = 0, b[3] = {0, 0, 0};
    Fr29P* tw_r[2][3] = {};   // [inverse][pass]: w_R^j
    Fr29P* tlo[2][2] = {};    // [inverse][col pass]
    Fr29P* thi[2][2] = {};
    Fr29P* tone[2][2] = {};   // [inverse][col pass]: the single-level table of a boundary with N' <= 2^17 (else null)
    // per-index multiplier tables (n x 48 B each), built on first use:
    Fr29P* pre_g = nullptr;        // g^j * 2^266          first load of a stand-alone forward coset transform
    Fr29P* post_ginv = nullptr;    // (g^-j / n) * 2^256   final store of an inverse coset transform (input carried as x 2^261)
    Fr29P* post_g = nullptr;       // (g^j / n) * 2^261    final store of an inverse transform (input carried raw) whose output goes
                                   //                      straight into a forward COSET transform: that one then loads raw, no multiplication
    Fr29P* post_one = nullptr;     // [0] = 2^256, [1] = 2^256 / n (input carried as x 2^261) ; [2] = 2^261, [3] = 2^261 / n (carried raw)
    Fr29P* kmul = nullptr;         // [0] = 2^271 / (g^n - 1): the h polynomial's pointwise step fused into a first load (FIRST_PW)
    uint64_t n = 0;
    std::mutex lazy;
};

static int ntt_bmax() {
    static int v = [] {
        const char* e = getenv("BZK_NTT_BMAX");
        int x = e ? atoi(e) : 10;
        return x < 4 ? 4 : (x > 10 ? 10 : x);
    }();
    return v;
}
static uint32_t ntt_tile_elems() {  // LDS tile in field elements (48 B each); default 1024 = 48 KiB -> 3 workgroups per CU
    static uint32_t v = [] {
        const char* e = getenv("BZK_NTT_TILE");
        uint32_t x = e ? (uint32_t)atoi(e) : 1024u;
        return x < 1024u ? 1024u : (x > 2048u ? 2048u : x);
    }();
    return v;
}

static Fr host_pow_u64(const Fr& base, uint64_t e) { return host_pow(base, e); }

// device table of `count` powers of `base`, converted to the 29-bit form with the extra factor `c`
static int32_t build_table29(bzk_ctx* ctx, const Fr& base, uint64_t count, const Fr& c, Fr29P** out) {
    void* raw = nullptr;
    BZK_TRY(build_pow_table(ctx, base, count, &raw));
    void* p = nullptr;
    BZK_HIP(ctx, hipMalloc(&p, (count ? count : 1) * sizeof(Fr29P)));
    BZK_LAUNCH(ctx, "ntt_table29", ntt_table29_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (const Fr*)raw, count, c, (Fr29P*)p);
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    BZK_HIP(ctx, hipFree(raw));
    *out = (Fr29P*)p;
    return BZK_OK;
}

static int32_t ntt_tables(bzk_ctx* ctx, int log_n, NttTables** out) {
    if (ctx->ntt_tw[log_n][1]) {
        *out = (NttTables*)ctx->ntt_tw[log_n][1];
        return BZK_OK;
    }
    const uint64_t n = (uint64_t)1 << log_n;
    NttTables* T = new NttTables();
    const int bm = ntt_bmax();
    if (log_n <= bm) { T->nb = 1; T->b[0] = log_n; }
    else if (log_n <= 2 * bm) { T->nb = 2; T->b[0] = (log_n + 1) / 2; T->b[1] = log_n / 2; }
    else {
        T->nb = 3;
        T->b[0] = (log_n + 2) / 3;
        const int rest = log_n - T->b[0];
        T->b[1] = (rest + 1) / 2;
        T->b[2] = rest / 2;
        if (T->b[0] > 10) { delete T; return BZK_E_ARG; }
    }
    const Fr w = host_omega(log_n), g = host_from_u64(7);
    const Fr dirs[2] = {w, fe_inv<FrParams>(w)};
    const Fr c5 = host_from_u64(32), c10 = host_from_u64(1024), c0 = Fr::one();
    for (int d = 0; d < 2; ++d) {
        for (int k = 0; k < T->nb; ++k) {
            const uint64_t R = (uint64_t)1 << T->b[k];
            BZK_TRY(build_table29(ctx, host_pow_u64(dirs[d], n / R), R / 2 ? R / 2 : 1, c5, &T->tw_r[d][k]));
        }
        // inter-pass twiddles: pass 0 over the whole transform (N' = n), pass 1 (three-digit plans) over N' = n / R1
        for (int k = 0; k + 1 < T->nb; ++k) {
            const uint64_t np = k == 0 ? n : n >> T->b[0];
            const Fr base = k == 0 ? dirs[d] : host_pow_u64(dirs[d], (uint64_t)1 << T->b[0]);
            BZK_TRY(build_table29(ctx, base, 1024, c5, &T->tlo[d][k]));
            BZK_TRY(build_table29(ctx, host_pow_u64(base, 1024), (np + 1023) / 1024, c5, &T->thi[d][k]));
            static const bool tone_off = env_on("BZK_NTT_NO_TONE");  // A/B runs
            if (!tone_off && k > 0 && np <= ((uint64_t)1 << 17)) BZK_TRY(build_table29(ctx, base, np, c5, &T->tone[d][k]));
        }
    }
    {
        const Fr ninv = fe_inv<FrParams>(host_from_u64(n));
        // Z(g) = g^n - 1 on the coset; kmul = 2^271 / Z as a raw value (c = Montgomery(2^15) on top of the Montgomery-256 limbs)
        const Fr zinv = fe_inv<FrParams>(fe_sub<FrParams>(host_pow(g, n), Fr::one()));
        Fr five[5] = {Fr::one(), ninv, c5, fe_mul<FrParams>(ninv, c5), fe_mul<FrParams>(zinv, host_from_u64(32768))};
        void* raw = nullptr;
        void* p = nullptr;
        BZK_HIP(ctx, hipMalloc(&raw, sizeof five));
        BZK_HIP(ctx, hipMalloc(&p, 5 * sizeof(Fr29P)));
        BZK_HIP(ctx, hipMemcpyAsync(raw, five, sizeof five, hipMemcpyHostToDevice, ctx->stream));
        BZK_LAUNCH(ctx, "ntt_table29", ntt_table29_kernel, dim3(1), dim3(256), 0, (const Fr*)raw, (uint64_t)5, c0, (Fr29P*)p);
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BZK_HIP(ctx, hipFree(raw));
        T->post_one = (Fr29P*)p;
        T->kmul = (Fr29P*)p + 4;
        (void)c10;
    }
    T->n = n;
    ctx->ntt_tw[log_n][1] = T;
    *out = T;
    return BZK_OK;
}

// the three per-index tables are only built when a transform asks for them (a prover needs post_g and post_ginv, a stand-alone
// forward coset transform pre_g): 48 B x n each
static int32_t ntt_lazy_table(bzk_ctx* ctx, NttTables* T, int which, Fr29P** out) {
    std::lock_guard<std::mutex> lk(T->lazy);
    Fr29P** slot = which == 0 ? &T->pre_g : (which == 1 ? &T->post_ginv : &T->post_g);
    if (!*slot) {
        const Fr g = host_from_u64(7), ninv = fe_inv<FrParams>(host_from_u64(T->n));
        if (which == 0) BZK_TRY(build_table29(ctx, g, T->n, host_from_u64(1024), slot));                            // g^j 2^266
        else if (which == 1) BZK_TRY(build_table29(ctx, fe_inv<FrParams>(g), T->n, ninv, slot));                    // g^-j / n 2^256
        else BZK_TRY(build_table29(ctx, g, T->n, fe_mul<FrParams>(ninv, host_from_u64(32)), slot));                 // g^j / n 2^261
    }
    *out = *slot;
    return BZK_OK;
}

// fuse: 0 = the transform bzk_ntt describes.  The h-polynomial chain (groth16_h) uses
//   NTT_INV_TO_COSET  inverse plain transform whose final store also applies the coset scaling g^k of the FOLLOWING forward coset
//                     transform (table post_g) - that one is then run as NTT_FWD_RAW
//   NTT_FWD_RAW       forward transform of already coset-scaled input: nothing to multiply on load
//   NTT_FWD_RAW_S5    the same with the output scaled by 2^-5 (the C vector of the pointwise step, see FIRST_PW)
//   NTT_INV_COSET_PW  inverse coset transform of (A B - C) / Z, the pointwise step fused into its first load (src_b, src_c)
enum { NTT_PLAIN = 0, NTT_INV_TO_COSET = 1, NTT_FWD_RAW = 2, NTT_FWD_RAW_S5 = 3, NTT_INV_COSET_PW = 4 };

static int32_t ntt_run_ex(bzk_ctx* ctx, void* data_dev, uint32_t log_n, int inverse, int coset, int fuse, const void* src_b,
                          const void* src_c) {
    if (log_n > 28) return BZK_E_ARG;
    const uint64_t n = (uint64_t)1 << log_n;
    if (log_n == 0) return BZK_OK;
    NttTables* T;
    BZK_TRY(ntt_tables(ctx, (int)log_n, &T));
    // first-load / final-store plan
    int first_mode = FIRST_RAW;
    const Fr29P *pre = nullptr, *post = nullptr, *post_c = nullptr;
    switch (fuse) {
        case NTT_INV_TO_COSET: inverse = 1; BZK_TRY(ntt_lazy_table(ctx, T, 2, (Fr29P**)&post)); break;
        case NTT_FWD_RAW: inverse = 0; post_c = T->post_one + 2; break;
        case NTT_FWD_RAW_S5: inverse = 0; post_c = T->post_one + 0; break;
        case NTT_INV_COSET_PW: inverse = 1; first_mode = FIRST_PW; BZK_TRY(ntt_lazy_table(ctx, T, 1, (Fr29P**)&post)); break;
        default:
            if (coset && !inverse) {  // g^i on load: the value is carried as x 2^261
                first_mode = FIRST_MUL;
                BZK_TRY(ntt_lazy_table(ctx, T, 0, (Fr29P**)&pre));
                post_c = T->post_one + 0;
            } else if (coset) {       // g^-k / n on store (table in the x 2^261 convention)
                first_mode = FIRST_MUL;
                BZK_TRY(ntt_lazy_table(ctx, T, 1, (Fr29P**)&post));
            } else {
                post_c = T->post_one + (inverse ? 3 : 2);  // carried raw: no multiplication on load
            }
    }
    static const int ip32 = [] { const char* e = getenv("BZK_NTT_IP32"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }();  // 0: 48-B inter-pass elements (A/B runs)
    void* tmp = nullptr;
    if (T->nb > 1) {
        BZK_TRY(ws_reserve(ctx, ws_pad(n * (ip32 ? sizeof(Fr) : sizeof(Fr29P))) + 512));
        tmp = ctx->ws;
    }
    const int d = inverse ? 1 : 0;
    const uint32_t tile_max = ntt_tile_elems();
    // tiles above 64 KiB of dynamic LDS need the opt-in; the attribute is per DEVICE, so remember it per device id (a process
    // may hold contexts on several GPUs, and prover slots call this from concurrent threads)
    typedef void (*pass_fn)(NttPass);
    // (variants 1 and 2 - <4, 2, 0> and <4, 1, 0>, the forms of round 3 that spilled 19 / 3 registers at four waves per SIMD and lost their A/B to
    // variant 4 - are no longer instantiated: the indices select the default, so every NTT kernel in the code object is spill-free)
    static const pass_fn variants[6] = {ntt_pass_kernel<3, 2, 0>, ntt_pass_kernel<4, 1, 1>, ntt_pass_kernel<4, 1, 1>, ntt_pass_kernel<4, 0, 0>,
                                        ntt_pass_kernel<4, 1, 1>, ntt_pass_kernel<3, 2, 1>};
    static const int variant = [] {
        const char* e = getenv("BZK_NTT_VARIANT");
        const int v = e ? atoi(e) : 4;  // 4 waves per SIMD, first-stage twiddle requested early, products inlined (profiles/r03_run5...)
        return v < 0 || v > 5 ? 4 : v;
    }();
    static const bool xcd_swz = [] { const char* e = getenv("BZK_NTT_NO_XCD"); return !(e && atoi(e) != 0); }();
    const pass_fn kern = variants[variant];
    static std::once_flag lds_attr_once[64];
    std::call_once(lds_attr_once[ctx->device & 63], [] {
        for (pass_fn f : variants) (void)hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    });
    uint64_t S = n;
    for (int k = 0; k < T->nb; ++k) {
        const int b = T->b[k];
        const uint64_t R = (uint64_t)1 << b;
        S >>= b;
        NttPass a;
        memset(&a, 0, sizeof a);
        a.b = b;
        a.first_pass = k == 0;
        a.final_pass = k == T->nb - 1;
        a.first_mode = first_mode;
        a.ip32 = ip32;
        a.src_b = src_b;
        a.src_c = src_c;
        a.kmul = T->kmul;
        a.src = k == 0 ? data_dev : (const void*)tmp;
        a.dst = a.final_pass ? data_dev : tmp;
        a.tw_r = T->tw_r[d][k];
        a.pre = pre;
        uint64_t lanes;  // how many adjacent columns exist
        if (!a.final_pass) {
            a.S = S;
            a.tlo = T->tlo[d][k];
            a.thi = T->thi[d][k];
            a.tone = T->tone[d][k];
            lanes = S;
        } else {
            a.R1 = T->nb >= 2 ? 1u << T->b[0] : 1u;
            a.R2 = T->nb == 3 ? 1u << T->b[1] : 1u;
            a.post = post;
            a.post_c = post_c;
            lanes = a.R1;
        }
        int lc = 0;
        while (((uint64_t)2 << lc) <= lanes && (R << (lc + 1)) <= tile_max) ++lc;
        a.log_cc = lc;
        const uint64_t tiles = n >> (b + lc);
        if (!a.final_pass && xcd_swz && lc < 2 && tiles % (8u << (2 - lc)) == 0) a.xcd_group = 4u >> lc;
        if (tiles > 0x7fffffffull) return BZK_E_ARG;
        const size_t lds = (size_t)(LDSW * 4) << (b + lc);
        if (a.final_pass) {
            BZK_LAUNCH(ctx, "ntt_final", kern, dim3((unsigned)tiles), dim3(256), lds, a);
        } else {
            BZK_LAUNCH(ctx, "ntt_col", kern, dim3((unsigned)tiles), dim3(256), lds, a);
        }
    }
    return BZK_OK;
}

int32_t ntt_run(bzk_ctx* ctx, void* data_dev, uint32_t log_n, int inverse, int coset) {
    return ntt_run_ex(ctx, data_dev, log_n, inverse, coset, NTT_PLAIN, nullptr, nullptr);
}

// h(x) = (A(x) B(x) - C(x)) / Z(x) from the evaluations a, b, c (zero-padded to m): on return a holds the coefficients of h.
// bellman `create_proof`'s seven transforms + two pointwise passes (a4); round 3: 3 x (inverse -> coset forward) with the coset
// scaling riding on the inverse transform's final store, and the pointwise (a b - c) / Z riding on the last transform's first load:
// no stand-alone pointwise kernel, 7 of the 14 load / store conversions multiply nothing.
int32_t ntt_h_chain(bzk_ctx* ctx, void* a, void* b, void* c, uint32_t log_m) {
    void* v[3] = {a, b, c};
    for (int k = 0; k < 3; ++k) {
        BZK_TRY(ntt_run_ex(ctx, v[k], log_m, 1, 0, NTT_INV_TO_COSET, nullptr, nullptr));
        BZK_TRY(ntt_run_ex(ctx, v[k], log_m, 0, 1, k == 2 ? NTT_FWD_RAW_S5 : NTT_FWD_RAW, nullptr, nullptr));
    }
    return ntt_run_ex(ctx, a, log_m, 1, 1, NTT_INV_COSET_PW, b, c);
}

void ntt_free_tables(bzk_ctx* ctx) {
    for (int i = 0; i <= 32; ++i) {
        NttTables* T = (NttTables*)ctx->ntt_tw[i][1];
        if (!T) continue;
        for (int d = 0; d < 2; ++d) {
            for (int k = 0; k < 3; ++k)
                if (T->tw_r[d][k]) (void)hipFree(T->tw_r[d][k]);
            for (int k = 0; k < 2; ++k) {
                if (T->tlo[d][k]) (void)hipFree(T->tlo[d][k]);
                if (T->thi[d][k]) (void)hipFree(T->thi[d][k]);
                if (T->tone[d][k]) (void)hipFree(T->tone[d][k]);
            }
        }
        if (T->pre_g) (void)hipFree(T->pre_g);
        if (T->post_ginv) (void)hipFree(T->post_ginv);
        if (T->post_g) (void)hipFree(T->post_g);
        (void)hipFree(T->post_one);
        delete T;
        ctx->ntt_tw[i][1] = nullptr;
    }
}

}  // namespace bzk

using namespace bzk;

extern "C" {

int32_t bzk_ntt_dev(bzk_ctx* ctx, void* data_dev, uint32_t log_n, int32_t inverse, int32_t coset) {
    if (!ctx || !data_dev) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    return ntt_run(ctx, data_dev, log_n, inverse != 0, coset != 0);
}

int32_t bzk_ntt(bzk_ctx* ctx, uint8_t* data, uint32_t log_n, int32_t inverse, int32_t coset) {
    if (!ctx || !data || log_n > 28) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    const size_t bytes = ((size_t)1 << log_n) * 32;
    void* d = nullptr;
    BZK_HIP(ctx, hipMalloc(&d, bytes));
    int32_t st = BZK_OK;
    if (hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
    if (st == BZK_OK) st = ntt_run(ctx, d, log_n, inverse != 0, coset != 0);
    if (st == BZK_OK && hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return st;
}

}  // extern "C"
