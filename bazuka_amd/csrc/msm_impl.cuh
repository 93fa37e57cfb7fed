// K4 / K5: Pippenger multi-scalar multiplication over BLS12-381 G1 / G2 for gfx950.
//
// Replaces bellman 0.14 `multiexp` (third-party crate behind `create_random_proof`,
// /root/reference/src/mpn/circuits/test.rs:135) - the h / l / a / b_g1 / b_g2 queries of a proof.
//
// Pipeline (on the ctx stream unless noted, data stays in HBM):
//   0. msm_convert     raw 96-byte bases -> internal 112-byte form, on the call's SIDE stream (joined before step 5)
//   1. msm_digits      one lane per scalar: Montgomery -> canonical, signed c-bit window recoding,
//                      emits (bucket within its window, point index | window | sign) pairs, window-major => coalesced stores
//   2. radix sort      rocPRIM pair sort by the bucket-within-window key (c - 1 bits + sentinel: two 8-bit passes at c = 16; the
//                      sort is stable and the pairs were emitted window-major, so (bucket, window) runs stay contiguous)
//   3. msm_offsets     run boundaries from the sorted keys and the window bits of the values
//   4. size sort       buckets ordered by population (descending, on 16-bit clamped counts) so the 64 lanes of a
//                      wavefront run equally long accumulation loops; task table (runs of <= seg entries)
//   5. msm_accumulate  one lane per TASK: gathers its affine points (7 x 16 B loads each) and folds them with XYZZ
//                      mixed adds                                           <-- dominant kernel
//      msm_fold*       partial sums of multi-task buckets (regime picked on the device, see msm_fold_threshold)
//   6. msm_reduce      per chunk of CH buckets: running-sum  sum (b+1) B_b  (+ chunk offset); BZK_F_THROUGHPUT: two levels
//                      (chunk sums and totals, then the offsets as a second reduction over the totals)
//   7. msm_window_sum  per window: LDS trees over the chunk results
//   8. host            Horner over <= 32 window sums, to affine, pack
// With BZK_F_DEDUP (section 8) equal scalars are merged first: 32-bit hash sort, group sums through steps 4 - 5, batched
// to-affine with the binary-GCD inversion, then the pipeline above over the distinct scalars.
// MFMA is not used anywhere: the arithmetic is 32-bit integer carry chains.
#pragma once
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstring>

#include "bzk_internal.h"
#include "host_fp64.h"
#include "msm_policy.cuh"
#include "msm_g2pair_tails.cuh"

namespace bzk {

// rocPRIM configuration for the bucket-population sort (524 288 keys of 16 bits at 2^20 points).  Below 2^20 items the library
// picks its merge sort (device_radix_sort.hpp: merge_sort_limit = 1 M): 0.125 ms.  With the limit lowered and 8-bit onesweep passes
// (the gfx942 tuning for 4-byte keys and values) the same sort takes two passes: 0.076 ms (profiles/r02_run9_sort_config_ab.txt).
// The same explicit configuration changes nothing for the 16.7 M-pair sort (0.426 vs 0.430 ms: the library's own choice is
// equivalent there) and the 8-byte-key variant made the de-duplication sort slower (0.20 -> 0.25 - 0.32 ms), so both keep
// rocPRIM's defaults.
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 16>, rocprim::kernel_config<1024, 16>, 8,
                                                                       rocprim::block_radix_rank_algorithm::match>,
                                   65536>
    SortCfg32;  // 4-byte keys, 4-byte values
static bool msm_tuned_sort() {  // BZK_MSM_SORT_DEFAULT=1: rocPRIM's own (untuned) configuration, for A/B runs
    static const bool on = [] { const char* e = getenv("BZK_MSM_SORT_DEFAULT"); return !(e && atoi(e) != 0); }();
    return on;
}

// ------------------------------------------------------------------------------------------------
// parameters
// ------------------------------------------------------------------------------------------------
static int msm_pick_c(uint64_t n) {
    // c = floor(log2(1.6 n)) - 4: the step to the next window size is taken at ~0.62 * 2^k rather than at 2^k - a proof's
    // witness queries (0.7 - 0.9 M points) run 7 % faster with c = 16 than with 15 (profiles/r01_run40...)
    const uint64_t n16 = n + n / 2 + n / 10;
    int lg = 0;
    while (((uint64_t)1 << (lg + 1)) <= n16) ++lg;
    int c = lg - 4;
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    return c;
}
// G2 on pairs of lanes (bzk_g2pair.cuh: accumulation; msm_g2pair_tails.cuh: folds, bucket reduction, window sums).  BZK_G2_PAIR=0: the
// one-lane kernels everywhere (same-box A/B; both are compiled in); BZK_G2_PAIR_TAILS=0: pair accumulation with the one-lane tails
static bool msm_g2_pair_on() {
    static const bool on = [] { const char* e = getenv("BZK_G2_PAIR"); return !(e && atoi(e) == 0); }();
    return on;
}
static bool msm_g2_pair_tails_on() {
    static const bool on = [] { const char* e = getenv("BZK_G2_PAIR_TAILS"); return !(e && atoi(e) == 0); }();
    return on && msm_g2_pair_on();
}
static int msm_windows_for(int c) { return (256 + c - 1) / c; }  // signed digits need one spare bit
// Window size of a proof's witness MSMs (flag BZK_F_DEDUP: l, a, b_g1, b_g2) - A/B knob.  By WORK, in units of one mixed
// addition, an MSM costs W(c) x (n_eff + kappa 2^(c-1)) with W(c) = ceil(256 / c), n_eff ~ 0.6 n distinct scalars after
// de-duplication and kappa = 4.75 general additions per bucket x 14 / 10 products = 6.65: for the 16-tx Update circuit
// (n = 0.7 - 0.9 M) that is flat within 7 % from c = 13 to 16, and so is the measurement - pipelined proofs/s 41.3 / 39.7
// with the plain pick (16), 40.9 - 43.3 with 13 / 14, 38.7 with 15 / 14, run-to-run noise +-1.5
// (profiles/r01_run48_49_witness_window.txt).  The plain pick therefore stays; env BZK_MSM_C_WIT_G1 / BZK_MSM_C_WIT_G2
// set a window size for the unsharded witness MSMs of one curve.
template <class C>
static int msm_pick_c_witness(uint64_t n, int c_plain) {
    static const int env_g1 = [] { const char* e = getenv("BZK_MSM_C_WIT_G1"); return e ? atoi(e) : 0; }();
    static const int env_g2 = [] { const char* e = getenv("BZK_MSM_C_WIT_G2"); return e ? atoi(e) : 0; }();
    const int env = C::RAW == 192 ? env_g2 : env_g1;
    (void)n;
    return env >= 4 && env <= 20 ? env : c_plain;
}

// ------------------------------------------------------------------------------------------------
// 1. digits
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) msm_digits_kernel(const U128* __restrict__ scalars, uint64_t n, int mont, int c,
                                                         int w_total, int w_begin, int w_cnt, uint32_t table_stride, int table_wpl,
                                                         const uint32_t* __restrict__ rep, int wiv, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s;
    {
        U128 a = scalars[2 * i], b = scalars[2 * i + 1];
        s.l[0] = a.x; s.l[1] = a.y; s.l[2] = a.z; s.l[3] = a.w;
        s.l[4] = b.x; s.l[5] = b.y; s.l[6] = b.z; s.l[7] = b.w;
    }
    if (mont) {
        s = fe_from_mont<FrParams>(s);
    } else {
        // BZK_F_CANONICAL input is not range-checked by the caller: the signed recoding below drops the carry out of the top
        // window, which is only safe below 2^255, so bring any 256-bit value under r first (2^256 < 3 r; r P = O)
        fe_reduce_once<FrParams>(s);
        fe_reduce_once<FrParams>(s);
    }
    const uint32_t half = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    // table mode (table_stride != 0): the table holds level j = 2^(c wpl j) P_i, window w = j * wpl + w' feeds bucket set
    // w' with table entry [j][i].  wpl = 1 (full table): every window shares ONE bucket set and [w_begin, w_begin + w_cnt)
    // selects levels; wpl > 1 (folded table): the range selects bucket sets and `w_total` is levels * wpl (windows past
    // the real top digit are zero).  Without a table buckets are per window and the value is the base index
    const bool folded = table_stride && table_wpl > 1;
    const uint32_t nb = (table_stride && !folded) ? half : (uint32_t)w_cnt * half;  // sentinel key (sorted behind every bucket)
    uint64_t buf = 0;
    int cnt = 0, w = 0;
    uint32_t carry = 0;
    auto emit = [&](uint32_t raw) {
        uint32_t d = raw + carry;
        uint32_t neg = 0;
        if (d > half) {
            d = (1u << c) - d;
            neg = 1;
            carry = 1;
        } else {
            carry = 0;
        }
        if (folded) {
            const int lvl = w / table_wpl, ws = w % table_wpl;
            if (ws >= w_begin && ws < w_begin + w_cnt) {
                const uint32_t lw = (uint32_t)(ws - w_begin);
                const uint64_t o = ((uint64_t)lvl * w_cnt + lw) * n + i;
                keys[o] = d ? lw * half + (d - 1) : nb;
                vals[o] = ((uint32_t)lvl * table_stride + (uint32_t)i) | (neg << 31);
            }
        } else if (w >= w_begin && w < w_begin + w_cnt) {
            const uint32_t lw = (uint32_t)(w - w_begin);
            const uint64_t o = (uint64_t)lw * n + i;
            if (table_stride) {
                keys[o] = d ? (d - 1) : nb;
                vals[o] = ((uint32_t)w * table_stride + (uint32_t)i) | (neg << 31);
            } else if (wiv) {
                // window-in-value form: the sort key is the bucket WITHIN its window (c - 1 bits + the sentinel: one radix pass
                // fewer than w * half + d), the window rides in bits 27..30 of the value.  The pairs are emitted window-major and
                // the radix sort is stable, so the runs of (bucket, window) are still contiguous: msm_offsets_wiv finds them
                keys[o] = d ? (d - 1) : half;
                vals[o] = (rep ? rep[i] : (uint32_t)i) | (lw << 27) | (neg << 31);
            } else {
                keys[o] = d ? lw * half + (d - 1) : nb;
                vals[o] = (rep ? rep[i] : (uint32_t)i) | (neg << 31);
            }
        }
        ++w;
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        buf |= (uint64_t)s.l[j] << cnt;
        cnt += 32;
        while (cnt >= c && w < w_total) {
            emit((uint32_t)buf & mask);
            buf >>= c;
            cnt -= c;
        }
    }
    while (w < w_total) {
        emit((uint32_t)buf & mask);
        buf >>= c;
    }
}

// ------------------------------------------------------------------------------------------------
// 1b. digits of the endomorphism form (bzk_endo.cuh; round 4): the canonical scalar is split into E signed sub-scalars (G2: four
// of < 2^63 in base X, G1: two of < 2^127 in base X^2), each recoded into w_sub signed c-bit digits.  Window j of sub-scalar m feeds
// bucket set j with the image m of its base: the value's index field is  (m << ibits) | base index.  Window-in-value pairs, emitted
// window-major with the E images of a window adjacent, so that the stable sort leaves every (bucket, window) run contiguous.
// ------------------------------------------------------------------------------------------------
template <int E>
static __global__ void __launch_bounds__(256) msm_digits_endo_kernel(const U128* __restrict__ scalars, uint64_t n, int mont, int c, int w_sub,
                                                                     int ibits, const uint32_t* __restrict__ rep, uint32_t* __restrict__ keys,
                                                                     uint32_t* __restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s;
    {
        U128 a = scalars[2 * i], b = scalars[2 * i + 1];
        s.l[0] = a.x; s.l[1] = a.y; s.l[2] = a.z; s.l[3] = a.w;
        s.l[4] = b.x; s.l[5] = b.y; s.l[6] = b.z; s.l[7] = b.w;
    }
    if (mont) {
        s = fe_from_mont<FrParams>(s);
    } else {  // the split needs k < r (2^256 < 3 r)
        fe_reduce_once<FrParams>(s);
        fe_reduce_once<FrParams>(s);
    }
    constexpr int ML = E == 4 ? 2 : 4;  // limbs of a sub-scalar's magnitude
    uint32_t mag[E][ML];
    bool sneg[E];
    if constexpr (E == 4) {
        int64_t sv[4];
        endo::decompose4(s.l, sv);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            sneg[m] = sv[m] < 0;
            const uint64_t a = (uint64_t)(sneg[m] ? -sv[m] : sv[m]);
            mag[m][0] = (uint32_t)a;
            mag[m][1] = (uint32_t)(a >> 32);
        }
    } else {
        endo::decompose2(s.l, mag, sneg);
    }
    const uint32_t half = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    const uint32_t base = rep ? rep[i] : (uint32_t)i;
#pragma unroll
    for (int m = 0; m < E; ++m) {
        uint64_t buf = 0;
        int cnt = 0, w = 0;
        uint32_t carry = 0;
        auto emit = [&](uint32_t raw) {
            uint32_t d = raw + carry;
            uint32_t neg = sneg[m] ? 1u : 0u;
            if (d > half) {
                d = (1u << c) - d;
                neg ^= 1u;
                carry = 1;
            } else {
                carry = 0;
            }
            const uint64_t o = ((uint64_t)w * E + m) * n + i;
            keys[o] = d ? (d - 1) : half;
            vals[o] = (((uint32_t)m << ibits) | base) | ((uint32_t)w << 27) | (neg << 31);
            ++w;
        };
#pragma unroll
        for (int j = 0; j < ML; ++j) {
            buf |= (uint64_t)mag[m][j] << cnt;
            cnt += 32;
            while (cnt >= c && w < w_sub) {
                emit((uint32_t)buf & mask);
                buf >>= c;
                cnt -= c;
            }
        }
        while (w < w_sub) {
            emit((uint32_t)buf & mask);
            buf >>= c;
        }
    }
}

// images X^m P (m = 1 .. E - 1) of `count` points in the internal affine form: data[m * stride + i] from data[i]
template <class C>
__global__ void __launch_bounds__(128) msm_endo_images_kernel(typename C::DevAff* __restrict__ data, uint64_t count, uint64_t stride) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    C::endo_images(data[i], data + i, (size_t)stride);
}

// ------------------------------------------------------------------------------------------------
// 3. bucket boundaries
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) msm_offsets_kernel(const uint32_t* __restrict__ keys, uint64_t len, uint32_t nb,
                                                          uint32_t* __restrict__ start, uint32_t* __restrict__ endx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const uint32_t k = keys[i];
    if (k >= nb) return;
    if (i == 0 || keys[i - 1] != k) start[k] = (uint32_t)i;
    if (i + 1 == len || keys[i + 1] != k) endx[k] = (uint32_t)(i + 1);
}

// window-in-value form (see msm_digits): key = bucket within the window, window = bits 27..30 of the value
static __global__ void __launch_bounds__(256) msm_offsets_wiv_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                              uint64_t len, uint32_t half, uint32_t* __restrict__ start,
                                                              uint32_t* __restrict__ endx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    // a boundary between positions i - 1 and i closes the run of i - 1 and opens the run of i: two loads per array instead of three
    const uint32_t k = keys[i];
    const uint32_t w = (vals[i] >> 27) & 15u;
    const uint32_t g = w * half + k;
    if (i == 0) {
        if (k < half) start[g] = 0;
    } else {
        const uint32_t kp = keys[i - 1];
        const uint32_t wp = (vals[i - 1] >> 27) & 15u;
        if (kp != k || wp != w) {
            if (k < half) start[g] = (uint32_t)i;
            if (kp < half) endx[wp * half + kp] = (uint32_t)i;
        }
    }
    if (i + 1 == len && k < half) endx[g] = (uint32_t)len;
}

// count[g] = end[g] - start[g] (in place over `endx`), iota[g] = g
static __global__ void __launch_bounds__(256) msm_count_kernel(const uint32_t* __restrict__ start, uint32_t* __restrict__ endx_count,
                                                        uint32_t* __restrict__ iota, uint32_t* __restrict__ ckey, uint32_t nb, uint32_t clamp) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nb) return;
    const uint32_t c = endx_count[g] - start[g];
    endx_count[g] = c;
    iota[g] = g;
    // sort key of the population order, clamped (the true counts are gathered through the order).  clamp = 255 (ONE 8-bit radix pass) for calls whose mean
    // population is small - the 2^20-point headline: 32 per bucket -: a bucket of more than MSM_SEG_MAX = 254 entries is cut into equal tasks of <= seg whatever
    // its size, and the clamp keeps all such buckets before the single-task ones.  clamp = 65535 (two passes) where buckets are full (2^22 points and up): there
    // the order among the multi-task buckets decides whether the lanes of a wave walk runs of equal length - with 8-bit keys every bucket of a 2^24-point call
    // tied at 255 and the accumulation ran 21 % longer (39.8 -> 48.3 ms: runs of 256 beside runs of 171; round 6, runs 9 / 12)
    ckey[g] = c < clamp ? c : clamp;
}

// ------------------------------------------------------------------------------------------------
// 3b. (round 2, removed again) counting sort of the pairs through GLOBAL atomics - count[key]++ / scan / scatter with a per-bucket
// cursor: measured 0.62 ms + 0.85 ms at 2^20 points against 0.43 + 0.05 ms for rocPRIM's radix sort + msm_offsets (device-scope
// atomics to random addresses run at ~27 G/s on MI355X, they are resolved beyond the per-XCD L2); 2^24: 20.1 ms against 6.5 ms
// (profiles/r02_run2_csort_ab.txt).
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// 3c. (round 2, removed again) two-pass LDS partition of the pairs instead of the radix sort: coarse bins of 256 buckets per
// 16 k-pair tile with LDS cursors, then one workgroup per bin producing the bucket boundaries directly.  Measured on the c = 16
// sizes: 0.56 ms against 0.47 ms at 2^20 points, 3.1 against 1.7 ms at 2^22, 12.5 against 6.7 ms at 2^24
// (profiles/r02_run6_psort_ab.txt) - the tile scatter leaves 32-byte runs per bin (uncoalesced 4-byte stores) where onesweep
// orders a tile in LDS before it writes.  Slower: gone.
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// 4b. base conversion to the policy's internal form (G1: 14 x 28-bit limbs); one pass per call
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(128) msm_convert_bases_kernel(const void* __restrict__ raw, uint64_t n,
                                                                typename C::DevAff* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = C::convert(raw, i);
}

// ------------------------------------------------------------------------------------------------
// 4c. static-base tables: tab[j][i] = 2^(c wpl j) * P_i in the internal affine form, `levels` of them.  Built once per
// base set (a Groth16 CRS query is static).  Full table (levels = W, wpl = 1): all windows feed ONE bucket set, so the
// bucket reduction shrinks from W windows to one and the host-side Horner disappears; costs W x the base memory.
// Folded table (levels = L < W, wpl = ceil(W / L)): windows j * wpl + w' share bucket set w' - the same number of
// mixed adds, 1 / L of the buckets to reduce, for L x the base memory (L = 2: 234 MB at 2^20 points).
// ------------------------------------------------------------------------------------------------
template <class C>
__global__ void __launch_bounds__(64) msm_table_build_kernel(const void* __restrict__ raw, uint64_t n, int dbl_per_level, int levels,
                                                             typename C::DevAff* __restrict__ tab) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename C::DevAff a = C::convert(raw, i);
    tab[i] = a;
    typename C::Pt p = C::identity();
    C::add_mixed(p, a, false);
#pragma unroll 1
    for (int w = 1; w < levels; ++w) {
#pragma unroll 1
        for (int d = 0; d < dbl_per_level; ++d) p = C::dbl(p);
        a = C::to_dev_affine(p);
        tab[(uint64_t)w * n + i] = a;
        p = C::identity();  // restart from the affine form: keeps the coordinates small and exact
        C::add_mixed(p, a, false);
    }
}

// ------------------------------------------------------------------------------------------------
// 5. bucket accumulation (dominant kernel)
//
// Work unit = a TASK: a run of at most MSM_SEG consecutive entries of one bucket.  Buckets are taken in
// population order (largest first), so (a) the 64 lanes of a wavefront run equally long loops and
// (b) a heavily populated bucket - real witnesses put ~10 % of the scalars on the value 1, and a short
// top window concentrates n points on a handful of buckets - is spread over many lanes instead of
// serialising on one.  Multi-task buckets write per-task partial sums, folded by msm_fold_kernel.
// ------------------------------------------------------------------------------------------------
// every array the accumulation kernels gather bases from ends in this many spare bytes: the G2 pair kernel's direct-to-LDS loads read 16-byte pieces, the
// last of which reaches 8 bytes past a lane's 56-byte component (msm_accumulate_g2pair_kernel); workspace arrays are followed by other workspace arrays
static constexpr size_t MSM_GATHER_PAD = 256;
static constexpr uint32_t MSM_SEG_MAX = 254;  // longest serial run of mixed adds one lane executes (< 255: see msm_count_kernel's sort key)

// buckets arrive ordered by their clamped count (msm_count_kernel: 255 for sparsely, 65535 for densely populated calls); the true counts are
// gathered through the order here
static __global__ void __launch_bounds__(256) msm_ntask_kernel(const uint32_t* __restrict__ count, const uint32_t* __restrict__ order,
                                                               uint32_t nb, uint32_t seg, uint32_t* __restrict__ count_sorted,
                                                               uint32_t* __restrict__ ntask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const uint32_t c = count[order[i]];
    count_sorted[i] = c;
    ntask[i] = c <= seg ? 1u : (c + seg - 1) / seg;  // empty buckets keep one task (writes the identity)
}

#ifndef BZK_MSM_SPLIT_DEFAULT
#define BZK_MSM_SPLIT_DEFAULT 2       // window ranges in flight of one stand-alone G1 call (msm_run_split)
#endif
#ifndef BZK_MSM_SPLIT_PRIO_DEFAULT
#define BZK_MSM_SPLIT_PRIO_DEFAULT 0  // the later ranges' streams at the highest priority
#endif
#ifndef BZK_MSM_HEAVY_PRIO_DEFAULT
#define BZK_MSM_HEAVY_PRIO_DEFAULT false  // round 6 A/B: see HeavyScope
#endif
#ifndef BZK_MSM_PREFETCH
#define BZK_MSM_PREFETCH 1  // 0: gather each base at the top of its own addition (A/B builds)
#endif
template <class C, int OCC>
__global__ void __launch_bounds__(128, OCC) msm_accumulate_kernel(const void* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                             const uint32_t* __restrict__ start,
                                                             const uint32_t* __restrict__ count_sorted,
                                                             const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ tbase, uint32_t nb, uint32_t t_max,
                                                             uint32_t seg, typename C::Pt* __restrict__ buckets,
                                                             typename C::Pt* __restrict__ partial, uint32_t vmask,
                                                             const void* __restrict__ bases2, uint32_t n_split, uint32_t ibits,
                                                             uint32_t stride1, uint32_t stride2) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= t_max) return;
    // base indices >= n_split name entries of a SECOND array (the de-duplication's group sums, which live in the call's workspace
    // while the bases proper may be a resident, shared, read-only set - MsmBases): one compare + select per gather.  Endomorphism
    // form: the bits above `ibits` of the index select the image m of the base (array m of either set; ibits = 31 otherwise: m = 0)
    const uint32_t imask = (1u << ibits) - 1u;
    auto ld = [&](uint32_t idx) {
        const uint32_t m = idx >> ibits, b = idx & imask;
        return b >= n_split ? C::load(bases2, m * stride2 + (b - n_split)) : C::load(bases, m * stride1 + b);
    };
    // last sorted position i with tbase[i] <= t
    uint32_t lo = 0, hi = nb;
    while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tbase[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint32_t i = lo;
    const uint32_t cnt = count_sorted[i];
    const uint32_t k = t - tbase[i];
    // a bucket of more than `seg` entries is cut into nt = ceil(cnt / seg) runs of EQUAL length q = ceil(cnt / nt)
    // (not seg, seg, ..., remainder): neighbouring lanes - the tasks of one bucket - finish together
    const uint32_t nt = cnt <= seg ? 1u : (cnt + seg - 1) / seg;
    const uint32_t q = (cnt + nt - 1) / nt;
    if (k >= nt) return;  // beyond the last task
    const uint32_t g = order[i];
    const uint32_t s = start[g] + k * q;
    const uint32_t len = k * q >= cnt ? 0u : (cnt - k * q < q ? cnt - k * q : q);  // empty bucket: writes the identity
    typename C::Pt acc = C::identity();
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (C::ACC_LDS) {
        // Round 6, run 24: the NEXT base travels global memory -> LDS by direct loads (global_load_lds_dwordx4: no destination registers) from the TOP of the current
        // addition and is read from LDS at the top of the next one - the G2 pair kernel's way (below).  Requested into registers it could only be asked for under the
        // fused-Y tail of the formula (28 registers are not free any earlier): ~2.5 us of cover - enough for a base set that sits in the Infinity Cache (the
        // 2^20-point resident set: 117 MB), not for gathers that go to HBM (a 13-level static table is 1.5 GB: its accumulation ran 1.6 x longer per addition).
        // A base is 112 B = 7 pieces of 16 B; a piece's destination is wave-uniform + lane x 16: seven 1 KiB slabs per wave and one of index words.
        __shared__ __attribute__((aligned(16))) char stage[2][8][1024];
        char* const slab = &stage[threadIdx.x >> 6][0][0];
        const uint32_t lane16 = (threadIdx.x & 63u) * 16u, lane4 = (threadIdx.x & 63u) * 4u;
        auto request = [&](uint32_t idx) {
            const uint32_t m = idx >> ibits, b = idx & imask;
            const char* p = b >= n_split ? (const char*)bases2 + (size_t)(m * stride2 + (b - n_split)) * sizeof(typename C::DevAff)
                                         : (const char*)bases + (size_t)(m * stride1 + b) * sizeof(typename C::DevAff);
#pragma unroll
            for (int k2 = 0; k2 < 7; ++k2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 16 * k2), (__attribute__((address_space(3))) void*)(slab + 1024 * k2), 16, 0, 0);
        };
        auto request_word = [&](const uint32_t* w) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w, (__attribute__((address_space(3))) void*)(slab + 7168), 4, 0, 0);
        };
        auto collect = [&](uint32_t& word) {
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the direct loads of this wave have landed
            __asm__ volatile("" ::: "memory");
            const U128* q2 = (const U128*)(slab + lane16);
            U128 v[7];
#pragma unroll
            for (int k2 = 0; k2 < 7; ++k2) v[k2] = q2[64 * k2];
            word = *(const uint32_t*)(slab + 7168 + lane4);
            typename C::DevAff a;
            uint32_t* dst = a.x.l;  // x.l[0..13] then y.l[0..13] are contiguous (2 x 56 B)
#pragma unroll
            for (int k2 = 0; k2 < 7; ++k2) {
                dst[4 * k2] = v[k2].x; dst[4 * k2 + 1] = v[k2].y; dst[4 * k2 + 2] = v[k2].z; dst[4 * k2 + 3] = v[k2].w;
            }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slabs are free again: the next requests may overwrite them
            return a;
        };
        if (len) {
            const uint32_t v0 = vals[s];
            bool neg = (v0 >> 31) != 0;
            request(v0 & vmask);
            request_word(vals + s + (1 < len ? 1u : 0u));
            for (uint32_t j = 0; j < len; ++j) {
                uint32_t vn;
                const typename C::DevAff p = collect(vn);  // base j and the index word of entry j + 1 (past the end of the run: the last entry again - no branch around loads)
                request(vn & vmask);
                request_word(vals + s + (j + 2 < len ? j + 2 : len - 1));
                C::add_mixed_pre(acc, p, neg, []() {});
                neg = (vn >> 31) != 0;
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);  // nothing of this wave may still be writing LDS when the workgroup's slot is handed on
        }
        if (cnt <= seg) buckets[g] = acc;
        else partial[t] = acc;
        return;
    }
#endif
#if BZK_MSM_PREFETCH
    if (len) {
        // software pipelining of the gathers: the next base is requested inside the current addition, after its last product CALL
        // (see g1x28::add_mixed), and arrives under the inlined tail of the formula
        uint32_t v = vals[s];
        typename C::DevAff p = ld(v & vmask);
        for (uint32_t j = 0; j < len; ++j) {
            // unconditional (the last iteration re-reads its own entry): a branch around the loads would force their results to be
            // merged with a default - i.e. waited for - on the spot.  The index word is requested at the top of the iteration, the
            // seven 16-byte loads of the base it names inside the addition
            const uint32_t vn = vals[s + (j + 1 < len ? j + 1 : j)];
            typename C::DevAff pn;
            C::add_mixed_pre(acc, p, (v >> 31) != 0, [&]() { pn = ld(vn & vmask); });
            p = pn;
            v = vn;
        }
    }
#else
    for (uint32_t j = 0; j < len; ++j) {
        const uint32_t v = vals[s + j];
        typename C::DevAff p = ld(v & vmask);
        C::add_mixed(acc, p, (v >> 31) != 0);
    }
#endif
    if (cnt <= seg) buckets[g] = acc;
    else partial[t] = acc;
}

// ---- 5b. (round 5) the G2 accumulation on PAIRS of lanes (bzk_g2pair.cuh): lanes 2 k and 2 k + 1 share task k, each holds one Fp2 component
// of the accumulator (56 registers) and of the bases it gathers (2 x 56 bytes of the 224).  Same task table, same buckets / partial sums in
// memory (the one-lane tail kernels consume them unchanged: X is brought below 3 p once per task).  No product is a call, so nothing between
// the request of the next base - at the top of an addition - and its first use forces a wait: the gather has a whole addition to land.
#ifndef BZK_G2_PAIR_OCC
#define BZK_G2_PAIR_OCC 2
#endif
#ifndef BZK_G2_PAIR_LDS
#define BZK_G2_PAIR_LDS 1  // the next base through LDS by direct loads (round 6); 0: through registers (BZK_G2_PAIR_PRE picks where the request sits)
#endif
#ifndef BZK_G2_PAIR_PRE
#define BZK_G2_PAIR_PRE 0  // 0: the next base requested at the top of the addition (round 5; A/B builds)
#endif
struct alignas(8) U128a8 {
    uint32_t x, y, z, w;
};
struct alignas(8) U64a8 {
    uint32_t x, y;
};
__device__ __forceinline__ Fp28 g2p_load56(const char* p) {  // 14 limbs at an 8-byte aligned address: 3 x 16 B + 8 B
    const U128a8 a = *(const U128a8*)p, b = *(const U128a8*)(p + 16), c = *(const U128a8*)(p + 32);
    const U64a8 d = *(const U64a8*)(p + 48);
    Fp28 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void g2p_store56(char* p, const Fp28& v) {
    *(U128a8*)p = U128a8{v.l[0], v.l[1], v.l[2], v.l[3]};
    *(U128a8*)(p + 16) = U128a8{v.l[4], v.l[5], v.l[6], v.l[7]};
    *(U128a8*)(p + 32) = U128a8{v.l[8], v.l[9], v.l[10], v.l[11]};
    *(U64a8*)(p + 48) = U64a8{v.l[12], v.l[13]};
}
template <int OCC>  // a template so that only the G2 translation unit instantiates it
__global__ void __launch_bounds__(128, OCC) msm_accumulate_g2pair_kernel(const void* __restrict__ bases, const uint32_t* __restrict__ vals,
                                                                                     const uint32_t* __restrict__ start,
                                                                                     const uint32_t* __restrict__ count_sorted,
                                                                                     const uint32_t* __restrict__ order,
                                                                                     const uint32_t* __restrict__ tbase, uint32_t nb, uint32_t t_max,
                                                                                     uint32_t seg, G2X28* __restrict__ buckets, G2X28* __restrict__ partial,
                                                                                     uint32_t vmask, const void* __restrict__ bases2, uint32_t n_split,
                                                                                     uint32_t ibits, uint32_t stride1, uint32_t stride2) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gt >> 1;
    const uint32_t comp = (gt & 1u) * 56u;  // byte offset of this lane's component inside an Fp2 value (c0 | c1)
    if (t >= t_max) return;                // pairs leave together
    const uint32_t imask = (1u << ibits) - 1u;
    auto ld = [&](uint32_t idx) {
        const uint32_t m = idx >> ibits, b = idx & imask;
        const char* p = b >= n_split ? (const char*)bases2 + (size_t)(m * stride2 + (b - n_split)) * sizeof(G2A28)
                                     : (const char*)bases + (size_t)(m * stride1 + b) * sizeof(G2A28);
        g2p::Aff a;
        a.x = g2p_load56(p + comp);
        a.y = g2p_load56(p + 112 + comp);
        return a;
    };
    uint32_t lo = 0, hi = nb;
    while (lo + 1 < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tbase[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint32_t i = lo;
    const uint32_t cnt = count_sorted[i];
    const uint32_t k = t - tbase[i];
    const uint32_t nt = cnt <= seg ? 1u : (cnt + seg - 1) / seg;
    const uint32_t q = (cnt + nt - 1) / nt;
    if (k >= nt) return;
    const uint32_t g = order[i];
    const uint32_t s = start[g] + k * q;
    const uint32_t len = k * q >= cnt ? 0u : (cnt - k * q < q ? cnt - k * q : q);
    g2p::Pt acc = g2p::identity();
#if BZK_G2_PAIR_LDS && defined(__HIP_DEVICE_COMPILE__)
    // Round 6: the NEXT base travels global memory -> LDS by direct loads (global_load_lds_dwordx4: no destination registers, nothing to
    // park) while the current addition runs, and is read from LDS at the top of the next iteration.  A lane's half of a base is 2 x 56 bytes (x | y component):
    // four 16-byte pieces each, the last one reading 8 bytes past the component (inside the point for x; the arrays this kernel gathers from - resident sets,
    // tables, the workspace - end in a 16-byte pad for the y of the last point).  The destination of such a load is wave-uniform base + lane x 16: one 1 KiB slab per piece.
    __shared__ __attribute__((aligned(16))) char stage[2][9][1024];  // per wave: 8 slabs of base pieces + one of index words (256 B used)
    char* const slab = &stage[threadIdx.x >> 6][0][0];
    const uint32_t lane16 = (threadIdx.x & 63u) * 16u, lane4 = (threadIdx.x & 63u) * 4u;
    auto request = [&](uint32_t idx) {
        const uint32_t m = idx >> ibits, b = idx & imask;
        const char* p = (b >= n_split ? (const char*)bases2 + (size_t)(m * stride2 + (b - n_split)) * sizeof(G2A28)
                                      : (const char*)bases + (size_t)(m * stride1 + b) * sizeof(G2A28)) + comp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 16 * k), (__attribute__((address_space(3))) void*)(slab + 1024 * k), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 112 + 16 * k),
                                             (__attribute__((address_space(3))) void*)(slab + 1024 * (4 + k)), 16, 0, 0);
        }
    };
    // the index word of the entry after next travels the same way: a register holding it would be live across the whole addition (the compiler spilled it
    // and waited for the gather in order to do so)
    auto request_word = [&](const uint32_t* w) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w, (__attribute__((address_space(3))) void*)(slab + 8192), 4, 0, 0);
    };
    auto collect = [&](uint32_t& word) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the direct loads of this wave have landed (they count as vector-memory operations)
        __asm__ volatile("" ::: "memory");
        g2p::Aff a;
        const U128* q = (const U128*)(slab + lane16);
        const U128 x0 = q[0], x1 = q[64], x2 = q[128], x3 = q[192], y0 = q[256], y1 = q[320], y2 = q[384], y3 = q[448];
        word = *(const uint32_t*)(slab + 8192 + lane4);
        a.x.l[0] = x0.x; a.x.l[1] = x0.y; a.x.l[2] = x0.z; a.x.l[3] = x0.w; a.x.l[4] = x1.x; a.x.l[5] = x1.y; a.x.l[6] = x1.z; a.x.l[7] = x1.w;
        a.x.l[8] = x2.x; a.x.l[9] = x2.y; a.x.l[10] = x2.z; a.x.l[11] = x2.w; a.x.l[12] = x3.x; a.x.l[13] = x3.y;
        a.y.l[0] = y0.x; a.y.l[1] = y0.y; a.y.l[2] = y0.z; a.y.l[3] = y0.w; a.y.l[4] = y1.x; a.y.l[5] = y1.y; a.y.l[6] = y1.z; a.y.l[7] = y1.w;
        a.y.l[8] = y2.x; a.y.l[9] = y2.y; a.y.l[10] = y2.z; a.y.l[11] = y2.w; a.y.l[12] = y3.x; a.y.l[13] = y3.y;
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slabs are free again: the next requests may overwrite them
        return a;
    };
    if (len) {
        const uint32_t v0 = vals[s];
        bool neg = (v0 >> 31) != 0;
        request(v0 & vmask);
        request_word(vals + s + (1 < len ? 1u : 0u));
        for (uint32_t j = 0; j < len; ++j) {
            uint32_t vn;
            const g2p::Aff p = collect(vn);  // base j and the index word of entry j + 1 (past the end of the run: the last entry again - no branch around loads)
            request(vn & vmask);
            request_word(vals + s + (j + 2 < len ? j + 2 : len - 1));
            g2p::add_mixed(acc, p, neg);  // both requests land under this addition; nothing of them is held in registers
            neg = (vn >> 31) != 0;
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // nothing of this wave may still be writing LDS when the workgroup's slot is handed on
        acc.X = fp28::reduce(acc.X);
    }
#else
    if (len) {
        uint32_t v = vals[s];
        g2p::Aff p = ld(v & vmask);
        for (uint32_t j = 0; j < len; ++j) {
            const uint32_t vn = vals[s + (j + 1 < len ? j + 1 : j)];
#if BZK_G2_PAIR_PRE
            // requested behind the last product that still needs every register (round 6): with the request at the top of the addition the 28 registers of
            // `pn` were parked in scratch on EVERY addition (8 scratch stores behind an s_waitcnt vmcnt(0): 3.76 GB of writes per 2^20-point launch, and the
            // gather waited for instead of flying - VERDICT r5 weak 5); the four-product tail of the formula (980 multiply-adds) covers the gather
            g2p::Aff pn;
            g2p::add_mixed(acc, p, (v >> 31) != 0, [&]() { pn = ld(vn & vmask); });
#else
            const g2p::Aff pn = ld(vn & vmask);  // requested now, first read by the next iteration
            g2p::add_mixed(acc, p, (v >> 31) != 0);
#endif
            p = pn;
            v = vn;
        }
        acc.X = fp28::reduce(acc.X);  // < 3 p: the discipline of the one-lane consumers (folds, bucket reduction)
    }
#endif
    char* dst = (char*)(cnt <= seg ? &buckets[g] : &partial[t]) + comp;
    g2p_store56(dst, acc.X);
    g2p_store56(dst + 112, acc.Y);
    g2p_store56(dst + 224, acc.ZZ);
    g2p_store56(dst + 336, acc.ZZZ);
}

// acc += *q.  G2 (PARK_REDUCE): q stays in memory and is read where the formula uses it - two resident 112-register points
// plus an addition's temporaries do not survive the calls to the field product, and the tail kernels spilled to scratch
// memory (fold 368, fold_small 352, reduce 1744, window sums 304 / 352 bytes per lane; now 16 - 80).  G1: by value, as before.
template <class C>
__device__ __forceinline__ void add_from(typename C::Pt& acc, const typename C::Pt* q) {
    if constexpr (C::PARK_REDUCE) {
        C::add_mem(acc, q);
    } else {
        typename C::Pt p = *q;
        C::add(acc, p);
    }
}

// Multi-task buckets: with at most `thr` tasks one lane folds the partial sums serially, with more a 64-lane workgroup does
// (strided pass + LDS tree).  thr depends on HOW MANY partial sums there are beyond one per bucket (read on the device from the
// task table, uniform over the grid):
//   few (< 65 536: uniform scalars up to ~2^22 points, witnesses)  thr = 2  - latency regime: a serial fold of up to 15 general
//        additions is a 0.23 ms (G1) / 1 ms (G2) dependency chain, the tree is at most 4 deep (round 2)
//   many (2^24 points: every bucket holds ~512 entries = 2 - 3 tasks) thr = 16 - throughput regime: a workgroup per bucket
//        would spend 64 lanes and six barriers on two additions (6.9 ms at 2^24, profiles/r02_run6_psort_ab.txt)
static constexpr uint32_t MSM_FOLD_SMALL = 2, MSM_FOLD_SMALL_BULK = 16, MSM_FOLD_BULK_FROM = 65536;
__device__ __forceinline__ uint32_t msm_fold_threshold(const uint32_t* __restrict__ tbase, const uint32_t* __restrict__ ntask, uint32_t nb) {
    const uint32_t extra = tbase[nb - 1] + ntask[nb - 1] - nb;  // partial sums beyond one per bucket
    return extra >= MSM_FOLD_BULK_FROM ? MSM_FOLD_SMALL_BULK : MSM_FOLD_SMALL;
}

// multi-task buckets with few tasks: one lane per bucket (sorted position), serial fold
template <class C>
__global__ void __launch_bounds__(64) msm_fold_small_kernel(const uint32_t* __restrict__ count_sorted,
                                                            const uint32_t* __restrict__ order, const uint32_t* __restrict__ tbase,
                                                            const uint32_t* __restrict__ ntask, uint32_t nb, uint32_t n_pos, uint32_t seg,
                                                            const typename C::Pt* __restrict__ partial, typename C::Pt* __restrict__ buckets) {
    typedef typename C::Pt Pt;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos) return;
    const uint32_t cnt = count_sorted[i];
    if (cnt <= seg) return;
    const uint32_t nt = (cnt + seg - 1) / seg;
    if (nt > msm_fold_threshold(tbase, ntask, nb)) return;
    const Pt* src = partial + tbase[i];
    Pt acc = src[0];
    for (uint32_t j = 1; j < nt; ++j) add_from<C>(acc, &src[j]);  // G2: second operand read from memory where it is used (no scratch)
    buckets[order[i]] = acc;
}

// GIANT buckets (round 6, run 22): a bucket of more than MSM_FOLD_WIDE_FROM tasks would be folded by ONE workgroup below - its 64 lanes each walk nt / 64
// general additions in a row.  Where the top window of a call is degenerate (c = 15: 17 x 15 = 255 bits, so window 17 only holds the carries of the signed
// recoding - 45 % of all uniform scalars land in ITS bucket 0: 236 k entries = 5 700 tasks at 2^19 points) that one chain was 1.6 ms of a 4.1 ms call
// (profiles/r06_run21_mid_size_task_cut.txt); skewed scalar vectors (one value repeated) have such a bucket in every window.  One level more: any workgroup of a
// fixed grid folds a CHUNK of 64 partial sums of a giant bucket (one load + a 6-level tree) into `wide`, and the workgroup fold below then meets nt / 64 chunk
// sums instead of nt partial sums: a chain of ~15 general additions whatever the population.  The order ties every bucket beyond the clamp of its sort key
// (msm_count_kernel), so giants are looked for among the first MSM_FOLD_WIDE_POS sorted positions only - any beyond keep the one-level fold (correct, slower).
// break-even: nt / 64 + 6 general additions in a row (one level) against 7 + nt / 4096 + 7 (two levels) at nt = 512; at 2^16 points (nt ~ 256) the extra level lost 0.11 ms (run 23)
static constexpr uint32_t MSM_FOLD_WIDE_FROM = 640, MSM_FOLD_WIDE_POS = 64, MSM_FOLD_WIDE_GRID = 512;
__device__ __forceinline__ uint32_t msm_fold_wide_slot(const uint32_t* __restrict__ tbase, uint32_t i) { return tbase[i] / 64 + i; }  // disjoint ranges: tbase grows by nt
template <class C>
__global__ void __launch_bounds__(64) msm_fold_wide_kernel(const uint32_t* __restrict__ count_sorted, const uint32_t* __restrict__ tbase, uint32_t nb,
                                                           uint32_t seg, const typename C::Pt* __restrict__ partial, typename C::Pt* __restrict__ wide) {
    typedef typename C::Pt Pt;
    __shared__ Pt sh[64];
    const uint32_t n_pos = nb < MSM_FOLD_WIDE_POS ? nb : MSM_FOLD_WIDE_POS;
    uint32_t before = 0;  // chunks of the giants at earlier positions: the chunks of ALL giants are dealt round-robin over the grid (eight giants of 16 chunks each -
                          // the top window at c = 14 - are one round of 128 workgroups, not eight rounds of the same 16: run 22, 0.91 ms -> one tree)
    for (uint32_t i = 0; i < n_pos; ++i) {
        const uint32_t cnt = count_sorted[i];
        const uint32_t nt = cnt <= seg ? 1u : (cnt + seg - 1) / seg;
        if (nt <= MSM_FOLD_WIDE_FROM) continue;  // workgroup-uniform
        const uint32_t chunks = (nt + 63) / 64;
        const Pt* src = partial + tbase[i];
        Pt* dst = wide + msm_fold_wide_slot(tbase, i);
        const uint32_t first = (blockIdx.x + gridDim.x - before % gridDim.x) % gridDim.x;
        before += chunks;
        for (uint32_t ch = first; ch < chunks; ch += gridDim.x) {
            const uint32_t j = ch * 64 + threadIdx.x;
            sh[threadIdx.x] = j < nt ? src[j] : C::identity();
            __syncthreads();
            for (int s = 32; s > 0; s >>= 1) {
                if ((int)threadIdx.x < s) {
                    Pt a = sh[threadIdx.x];
                    add_from<C>(a, &sh[threadIdx.x + s]);
                    sh[threadIdx.x] = a;
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) dst[ch] = sh[0];
            __syncthreads();  // sh[] is free again
        }
    }
}

// heavily populated buckets: one 64-lane workgroup per bucket: lanes stride over the partial sums (a giant bucket's chunk sums: above), LDS tree
template <class C>
__global__ void __launch_bounds__(64) msm_fold_kernel(const uint32_t* __restrict__ count_sorted, const uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ tbase, const uint32_t* __restrict__ ntask, uint32_t nb, uint32_t n_big,
                                                      uint32_t seg, const typename C::Pt* __restrict__ partial, const typename C::Pt* __restrict__ wide,
                                                      typename C::Pt* __restrict__ buckets) {
    typedef typename C::Pt Pt;
    __shared__ Pt sh[64];
    // round 6: a bounded grid walks the sorted positions and STOPS at the first single-task bucket (the order is by population, largest first: every later
    // position is single-task too) - for uniform scalars that is position 0, and the launch costs a few microseconds instead of one exiting workgroup per
    // position that could in principle hold a multi-task bucket (63 k of them at 2^20 points: 0.03 ms)
    const uint32_t thr = msm_fold_threshold(tbase, ntask, nb);
    for (uint32_t i = blockIdx.x; i < n_big; i += gridDim.x) {
        const uint32_t cnt = count_sorted[i];
        if (cnt <= seg) break;  // workgroup-uniform
        uint32_t nt = (cnt + seg - 1) / seg;
        if (nt <= thr) continue;  // msm_fold_small_kernel's
        const Pt* src = partial + tbase[i];
        if (wide && i < MSM_FOLD_WIDE_POS && nt > MSM_FOLD_WIDE_FROM) {  // msm_fold_wide_kernel left one sum per chunk of 64
            src = wide + msm_fold_wide_slot(tbase, i);
            nt = (nt + 63) / 64;
        }
        Pt acc = threadIdx.x < nt ? src[threadIdx.x] : C::identity();
        for (uint32_t j = threadIdx.x + 64; j < nt; j += 64) add_from<C>(acc, &src[j]);
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 32; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                Pt a = sh[threadIdx.x];
                add_from<C>(a, &sh[threadIdx.x + s]);
                sh[threadIdx.x] = a;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) buckets[order[i]] = sh[0];
        __syncthreads();  // sh[] is free again
    }
}

// ------------------------------------------------------------------------------------------------
// 6. chunked running-sum reduction:  out[t] = sum_{j<CH} (lo + j + 1) * B[w][lo + j]
// ------------------------------------------------------------------------------------------------
//    Two-level form (tot != nullptr): the chunk offset is NOT multiplied in per lane.  Level 1 writes  B_k = sum_j (j+1) S[lo+j]  and
//    the chunk total T_k; the offsets  sum_k (ch k) T_k = ch * sum_k k T_k  are the same problem on per_win = half / ch "buckets"
//    T_1 .. T_{per_win-1} (stored shifted by one, the last slot = identity), solved by a second launch of this kernel with
//    post_dbl = log2 ch doublings of its result.  2 + 4.75/8 general additions per bucket instead of 4.75 (the 15-bit double-and-add
//    per chunk is 60 % of the one-level work) for a chain of 16 + 41 instead of 41: a THROUGHPUT form, for calls that overlap others.
template <class C>
__global__ void __launch_bounds__(64) msm_reduce_kernel(const typename C::Pt* __restrict__ buckets, uint32_t half, uint32_t ch,
                                                        uint32_t n_chunks_total, typename C::Pt* __restrict__ out, uint32_t out_stride,
                                                        uint32_t out_off, typename C::Pt* __restrict__ tot, uint32_t post_dbl) {
    typedef typename C::Pt Pt;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks_total) return;
    const uint32_t per_win = half / ch;
    const uint32_t w = t / per_win, k = t % per_win;
    const uint32_t lo = k * ch;
    const Pt* b = buckets + (size_t)w * half + lo;
    const size_t o = (size_t)w * out_stride + out_off + k;
    Pt* const tot_slot = tot ? tot + (size_t)w * per_win + (k ? k - 1 : per_win - 1) : nullptr;
    if constexpr (C::PARK_REDUCE) {
        // G2: a point is 112 registers; `run`, `acc` and an addition's temporaries do not survive a call to the field
        // product together (the compiler spilled 1774 registers, 1744 B of scratch per lane).  Here no point stays in
        // registers between additions: `run` lives in LDS (28 KB per wavefront), `acc` in this lane's own output slot,
        // and the second operand of every addition is read from memory where it is used (xyzz_add_mem).
        __shared__ Pt park[64];
        Pt* const run_m = &park[threadIdx.x];
        Pt* const acc_m = &out[o];
        *run_m = C::identity();
        *acc_m = C::identity();
        for (int j = (int)ch - 1; j >= 0; --j) {
            {
                Pt r = *run_m;
                C::add_mem(r, &b[j]);
                *run_m = r;
            }
            __asm__ volatile("" ::: "memory");
            {
                Pt a = *acc_m;
                C::add_mem(a, run_m);
                *acc_m = a;
            }
            __asm__ volatile("" ::: "memory");
        }
        if (tot_slot) {
            *tot_slot = k ? *run_m : C::identity();
            return;
        }
        if (lo) {
            Pt m = C::identity();  // lo * run, double-and-add with `run` read from LDS
            for (int i = 31; i >= 0; --i) {
                m = C::dbl(m);
                if ((lo >> i) & 1) C::add_mem(m, run_m);
                __asm__ volatile("" ::: "memory");
            }
            *run_m = m;
            __asm__ volatile("" ::: "memory");
            Pt a = *acc_m;
            C::add_mem(a, run_m);
            *acc_m = a;
        }
        if (post_dbl) {
            __asm__ volatile("" ::: "memory");
            Pt a = *acc_m;
            for (uint32_t d = 0; d < post_dbl; ++d) a = C::dbl(a);
            *acc_m = a;
        }
        return;
    }
    Pt run = C::identity(), acc = C::identity();
    for (int j = (int)ch - 1; j >= 0; --j) {
        Pt p = b[j];
        C::add(run, p);
        C::add(acc, run);
    }
    if (tot_slot) {
        out[o] = acc;
        *tot_slot = k ? run : C::identity();
        return;
    }
    if (lo) {
        Pt m = C::mul_u32(run, lo);
        C::add(acc, m);
    }
    for (uint32_t d = 0; d < post_dbl; ++d) acc = C::dbl(acc);
    out[o] = acc;
}

// ------------------------------------------------------------------------------------------------
// 7. per-window sum of the chunk results: two shallow LDS trees (the reduction phase is latency-bound - every
//    XYZZ add is ~14 dependent field products - so depth, not work, is what is minimised here)
//    stage A: THREADS consecutive chunk results -> 1 partial        (grid = windows x groups)
//    stage B: the <= THREADS partials of a window -> the window sum, converted to standard limbs
// ------------------------------------------------------------------------------------------------
// `active`: a power of two <= THREADS; lanes >= active hold the identity, so the tree starts at active / 2 (every level is one
// general addition of latency - ~16 us in G1, ~43 us in G2 - whether or not its operands are identities)
template <class C, int THREADS>
__device__ __forceinline__ typename C::Pt block_tree_sum(typename C::Pt acc, typename C::Pt* sh, int active = THREADS) {
    typedef typename C::Pt Pt;
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = active / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            Pt a = sh[threadIdx.x];
            add_from<C>(a, &sh[threadIdx.x + s]);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    return sh[0];
}

// ------------------------------------------------------------------------------------------------
// 7b. (round 3) cooperative general addition for the latency-bound window-sum trees, G1 only.  A general XYZZ addition is 14
//     products of which at most 4 are ever independent: a QUAD of lanes computes them side by side - 4 product steps instead of 14 -
//     and hands the results round with quad shuffles.  All four lanes hold both operands and end with the same result, which is
//     limb-identical to g1x28::add_full(a, b) (same products, same subtractions, same normalisations).
//        step 1  X1 ZZ2 | X2 ZZ1 | Y1 ZZZ2 | Y2 ZZZ1          -> U1 U2 S1 S2 ; P = U2 - U1, R = S2 - S1
//        step 2  P P    | R R    | ZZ1 ZZ2 | ZZZ1 ZZZ2        -> PP RR ZZ12 ZZZ12
//        step 3  P PP   | U1 PP  | ZZ12 PP | -                 -> PPP Q ZZ3 ; X3 = RR - PPP - 2 Q
//        step 4  R (Q - X3) | S1 PPP | ZZZ12 PPP | -           -> Y3 = difference, ZZZ3
//     A link of a tree is then ~5 us instead of ~15.6 us.
// ------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
// lane SRC of the quad broadcasts its value: a DPP quad_perm move per limb (VALU, no trip through the LDS crossbar as __shfl's
// ds_bpermute would take)
template <int SRC>
__device__ __forceinline__ Fp28 quad_get(const Fp28& v) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);  // quad_perm:[SRC, SRC, SRC, SRC]
    Fp28 r;
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], ctrl, 0xf, 0xf, false);
        __asm__ volatile("" : "+v"(r.l[i]));  // keep the move a move: no folding of the DPP operand into the consuming instruction
    }
    return r;
}
__device__ __forceinline__ Fp28 sel4(int q, const Fp28& a, const Fp28& b, const Fp28& c, const Fp28& d) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < 14; ++i) r.l[i] = q == 0 ? a.l[i] : (q == 1 ? b.l[i] : (q == 2 ? c.l[i] : d.l[i]));
    return r;
}
// The products are INLINED (fp28::mul_body), not the library's calls (fp28::mul): (1) a call makes the caller save every live limb
// around it - the first forms of these functions kept 0.5 - 1.3 KB of scratch per lane busy - and (2) the callee's last VALU write of a
// result register may sit one instruction (its s_setpc) before the caller's first DPP read of it, while a DPP source needs two wait
// states after a VALU write; the compiler's hazard recogniser does not look across the call boundary (first GPU run of the DPP form
// with calls: wrong sums, cured then by an s_nop after each call).  Inlined, both problems are the compiler's own.
// every lane of the (aligned) quad passes the same a and b and receives a + b
__device__ __forceinline__ G1X28 g1_add_quad(G1X28 a, const G1X28& b) {
    using namespace fp28;
    const int q = (int)(threadIdx.x & 3u);
    if (g1x28::is_identity(b)) return a;  // quad-uniform: all four lanes hold the same operands
    if (g1x28::is_identity(a)) return b;
    const Fp28 r1 = mul_body(sel4(q, a.X, b.X, a.Y, b.Y), sel4(q, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ));
    const Fp28 U1 = quad_get<0>(r1), U2 = quad_get<1>(r1), S1 = quad_get<2>(r1), S2 = quad_get<3>(r1);
    const Fp28 Pp = sub<3>(U2, U1), R = sub<3>(S2, S1);  // k 5
    const Fp28 r2 = mul_body(sel4(q, Pp, R, a.ZZ, a.ZZZ), sel4(q, Pp, R, b.ZZ, b.ZZZ));
    const Fp28 PP = quad_get<0>(r2), RR = quad_get<1>(r2), Z12 = quad_get<2>(r2), Z123 = quad_get<3>(r2);
    if (mulout_is_zero(PP)) {  // same x (doubling or cancellation; practically never between partial sums): the serial formula, by every lane
        g1x28::add_full(a, b);
        return a;
    }
    const Fp28 r3 = mul_body(sel4(q, Pp, U1, Z12, Z12), PP);
    const Fp28 PPP = quad_get<0>(r3), Q = quad_get<1>(r3), ZZ3 = quad_get<2>(r3);
    G1X28 o;
    o.X = norm(sub<3>(sub<3>(sub<3>(RR, PPP), Q), Q));
    const Fp28 r4 = mul_body(sel4(q, R, S1, Z123, Z123), sel4(q, sub<12>(Q, o.X), PPP, PPP, PPP));
    const Fp28 m0 = quad_get<0>(r4), m1 = quad_get<1>(r4);
    o.Y = norm(sub<3>(m0, m1));
    o.ZZ = ZZ3;
    o.ZZZ = quad_get<2>(r4);
    return o;
}
// 2 p on a quad (dbl-2008-s-1, the products of g1x28::dbl in three steps):
//        step 1  U U | X X | - | -                       (U = 2 Y)   -> V, XX ; M = 3 XX
//        step 2  U V | X V | V ZZ | M M                              -> W, S, ZZ3, MM ; X3 = MM - 2 S
//        step 3  M (S - X3) | W Y | W ZZZ | -                        -> Y3 = difference, ZZZ3
__device__ __forceinline__ G1X28 g1_dbl_quad(const G1X28& p) {
    using namespace fp28;
    if (g1x28::is_identity(p)) return p;
    const int q = (int)(threadIdx.x & 3u);
    const Fp28 U = add(p.Y, p.Y);  // k 10
    const Fp28 f1 = (q & 1) ? p.X : U;
    const Fp28 r1 = mul_body(f1, f1);
    const Fp28 V = quad_get<0>(r1), xx = quad_get<1>(r1);
    const Fp28 M = add(add(xx, xx), xx);
    const Fp28 r2 = mul_body(sel4(q, U, p.X, V, M), sel4(q, V, V, p.ZZ, M));
    const Fp28 Wv = quad_get<0>(r2), S = quad_get<1>(r2), MM = quad_get<3>(r2);
    G1X28 o;
    o.ZZ = quad_get<2>(r2);
    o.X = norm(sub<3>(sub<3>(MM, S), S));
    const Fp28 r3 = mul_body(sel4(q, M, Wv, Wv, Wv), sel4(q, sub<12>(S, o.X), p.Y, p.ZZZ, p.ZZZ));
    const Fp28 m0 = quad_get<0>(r3), m1 = quad_get<1>(r3);
    o.Y = norm(sub<3>(m0, m1));
    o.ZZZ = quad_get<2>(r3);
    return o;
}
// The same two operations on points that stay in memory (LDS or global), as CALLS: one copy of the ~3 000-instruction bodies per
// code object, nothing live in the caller across them.  *dst = *pa + *pb (dst may be pa); lane 0 of the quad stores, and a later read
// of *dst by the whole quad is ordered behind that store (one wave, LDS and the vector memory path are in order per wave).
static __device__ __noinline__ void g1_add_quad_mem(G1X28* dst, const G1X28* pa, const G1X28* pb) {
    const G1X28 r = g1_add_quad(*pa, *pb);
    if ((threadIdx.x & 3u) == 0) *dst = r;
}
static __device__ __noinline__ void g1_dbl_quad_mem(G1X28* dst, const G1X28* pa) {
    const G1X28 r = g1_dbl_quad(*pa);
    if ((threadIdx.x & 3u) == 0) *dst = r;
}
#endif
// tree over `count` (a power of two <= THREADS) points in sh[], quads of lanes per addition; result in sh[0]
template <int THREADS>
__device__ __forceinline__ void g1_quad_tree(G1X28* sh, int count) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int quad = (int)(threadIdx.x >> 2), n_quads = THREADS / 4;
    for (int s = count / 2; s > 0; s >>= 1) {
        for (int p = quad; p < s; p += n_quads) {
            const G1X28 r = g1_add_quad(sh[p], sh[p + s]);
            if ((threadIdx.x & 3u) == 0) sh[p] = r;
        }
        __syncthreads();
    }
#endif
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS) msm_window_partial_quad_kernel(const G1X28* __restrict__ chunk_out, uint32_t per_win, uint32_t groups,
                                                                         G1X28* __restrict__ partial_out) {
    __shared__ G1X28 sh[THREADS];
    const uint32_t w = blockIdx.x / groups, g = blockIdx.x % groups;
    const uint32_t i = g * THREADS + threadIdx.x;
    sh[threadIdx.x] = i < per_win ? chunk_out[(size_t)w * per_win + i] : g1x28::identity();
    __syncthreads();
    g1_quad_tree<THREADS>(sh, THREADS);
    if (threadIdx.x == 0) partial_out[(size_t)w * groups + g] = sh[0];
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS) msm_window_sum_quad_kernel(const G1X28* __restrict__ partials, uint32_t groups,
                                                                      XyzzT<FpOps>* __restrict__ win_out) {
    __shared__ G1X28 sh[THREADS];
    const uint32_t w = blockIdx.x;
    const G1X28* src = partials + (size_t)w * groups;
    int active = 1;
    while (active < THREADS && (uint32_t)active < groups) active <<= 1;
    // groups <= THREADS partials (more: folded serially into the first THREADS slots by their owners)
    G1X28 acc = threadIdx.x < groups ? src[threadIdx.x] : g1x28::identity();
    for (uint32_t i = threadIdx.x + THREADS; i < groups; i += THREADS) g1x28::add_full(acc, src[i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    g1_quad_tree<THREADS>(sh, active);
    if (threadIdx.x == 0) win_out[w] = g1x28::to_std(sh[0]);
}

// Level 2 of the two-level bucket reduction (section 6) on quads, G1: out = 2^post_dbl * sum_{j < ch2} (lo + j + 1) T[lo + j].  The
// level has per_win / ch2 chunks per window - 8 192 lanes for a 2^20-point MSM in the one-lane form, a pure chain of 16 + ~25 + 3
// point operations of ~15 us (0.56 ms, as long as level 1 with its 16x the work); here a chunk is a quad, its chain the same links
// at ~7 / ~5 us (addition / doubling), and ch2 may be smaller (more, shorter chains) because the lanes are there.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) msm_reduce_l2_quad_kernel(const G1X28* __restrict__ tot, uint32_t per_win, uint32_t ch2,
                                                                     uint32_t n_chunks2, G1X28* __restrict__ out, uint32_t out_stride,
                                                                     uint32_t out_off, uint32_t post_dbl) {
#if defined(__HIP_DEVICE_COMPILE__)
    // a quad's three points (run, acc, m) live in LDS between the calls: 42 KB per 256 lanes
    __shared__ G1X28 park[THREADS / 4][3];
    const uint32_t t = (blockIdx.x * THREADS + threadIdx.x) >> 2;
    if (t >= n_chunks2) return;  // whole quads leave (no workgroup barrier below)
    const uint32_t per2 = per_win / ch2;
    const uint32_t w = t / per2, k = t % per2, lo = k * ch2;
    const G1X28* b = tot + (size_t)w * per_win + lo;
    G1X28* const run = &park[threadIdx.x >> 2][0];
    G1X28* const acc = run + 1;
    G1X28* const m = run + 2;
    const bool lead = (threadIdx.x & 3u) == 0;
    if (lead) {
        const G1X28 first = b[ch2 - 1];
        *run = first;
        *acc = first;
    }
#pragma nounroll
    for (int j = (int)ch2 - 2; j >= 0; --j) {
        g1_add_quad_mem(run, run, &b[j]);
        g1_add_quad_mem(acc, acc, run);
    }
    // m = lo * run by double-and-add below lo's top bit; lo is a multiple of ch2 >= 2, so its bit 0 is clear and the last bit step
    // takes `acc` in instead; then the post-doublings
    if (lead) *m = lo ? *run : *acc;
    const int top = lo ? 31 - __clz((int)lo) : 0;
#pragma nounroll
    for (int i = top - 1; i >= -(int)post_dbl; --i) {
        g1_dbl_quad_mem(m, m);
        if (i > 0 ? ((lo >> i) & 1u) != 0 : (i == 0)) g1_add_quad_mem(m, m, i == 0 ? acc : run);
    }
    if (lead) out[(size_t)w * out_stride + out_off + k] = *m;
#endif
}

// ------------------------------------------------------------------------------------------------
// 6b. (round 6) bucket reduction WITHOUT scalar multiplications, G1.  The weight of bucket b is b + 1 = 1 + sum_k 2^k bit_k(b), so
//        sum_b (b + 1) B_b  =  S_tot + sum_k 2^k S_k,      S_k = sum of the buckets whose index has bit k set,  S_tot = sum of all buckets
//     and with b = L h + l (L = 2^lbits columns, H = 2^hbits rows) every S_k is a sum of ROW sums R_h = sum_l B[h][l] (k >= lbits) or of COLUMN sums
//     C_l = sum_h B[h][l] (k < lbits), S_tot the sum of all column sums.  Two launches:
//        msm_rowcol_quad_kernel   the row sums and the column sums: 2 general additions per bucket (the chunked running sum of section 6 spends 4.2: 60 % of
//                                 it is the chunk offset's 15-bit double-and-add), chains of 7 + log2(lanes per row) links instead of 41
//        msm_bitsum_quad_kernel   per (bucket set, bit): a tree over the <= 512 row / column sums with that bit set -> the c terms of the set
//     and the weights 2^k cost nothing: the host's Horner over the windows doubles its accumulator c times per window anyway - the terms join it at
//     their bit positions (msm_horner_terms_host: the same c W doublings, c W additions instead of W).  No window-sum kernels.
//     Same-box A/B and the issue budget of a proof before / after: profiles/r06_run4...
// ------------------------------------------------------------------------------------------------
// trees over SEGMENTS of `seg` (a power of two) consecutive points of sh[0 .. THREADS): result of segment g in sh[g * seg]
template <int THREADS>
__device__ __forceinline__ void g1_quad_tree_seg(G1X28* sh, int seg) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int quad = (int)(threadIdx.x >> 2), n_quads = THREADS / 4, n_seg = THREADS / seg;
    for (int s = seg / 2; s > 0; s >>= 1) {
        const int n_adds = n_seg * s;
        for (int q = quad; q < n_adds; q += n_quads) {
            const int g = q / s, i = q % s;
            const G1X28 r = g1_add_quad(sh[g * seg + i], sh[g * seg + i + s]);
            if ((threadIdx.x & 3u) == 0) sh[g * seg + i] = r;
        }
        __syncthreads();
    }
#endif
}
struct RowColPlan {  // host-side shape of one bucket set of `half` = L H buckets
    uint32_t lbits, hbits, leaf_r, leaf_c, seg_r, seg_c, wgs_r, wgs_c;
};
static RowColPlan msm_rowcol_plan(int c_minus_1, uint32_t threads) {
    RowColPlan P;
    P.lbits = (uint32_t)(c_minus_1 + 1) / 2;
    P.hbits = (uint32_t)c_minus_1 - P.lbits;
    const uint32_t L = 1u << P.lbits, H = 1u << P.hbits;
    // buckets one lane sums serially before the trees (env BZK_MSM_RC_LEAF = 2 | 4 | 8 | 16 for A/B runs; run 30: 4 and 16 against 8)
    static const uint32_t leaf_want = [] { const char* e = getenv("BZK_MSM_RC_LEAF"); const int v = e ? atoi(e) : 8; return (uint32_t)((v == 2 || v == 4 || v == 8 || v == 16) ? v : 8); }();
    P.leaf_r = std::max<uint32_t>(std::min<uint32_t>(leaf_want, L), L / threads);
    P.leaf_c = std::max<uint32_t>(std::min<uint32_t>(leaf_want, H), H / threads);
    P.seg_r = L / P.leaf_r;                                                 // lanes per row / per column (tree width)
    P.seg_c = H / P.leaf_c;
    P.wgs_r = (H + threads / P.seg_r - 1) / (threads / P.seg_r);
    P.wgs_c = (L + threads / P.seg_c - 1) / (threads / P.seg_c);
    return P;
}
template <int THREADS>
__global__ void __launch_bounds__(THREADS, 2) msm_rowcol_quad_kernel(const G1X28* __restrict__ buckets, uint32_t half, RowColPlan P, G1X28* __restrict__ rows,
                                                                  G1X28* __restrict__ cols) {
    __shared__ G1X28 sh[THREADS];
    const uint32_t L = 1u << P.lbits, H = 1u << P.hbits, per_set = P.wgs_r + P.wgs_c;
    const uint32_t set = blockIdx.x / per_set;
    uint32_t local = blockIdx.x % per_set;
    const G1X28* b = buckets + (size_t)set * half;
    const bool col_pass = local >= P.wgs_r;
    uint32_t seg, first, stride, leaf, slot, out_idx;
    bool valid;
    if (!col_pass) {
        seg = P.seg_r;
        const uint32_t row = local * (THREADS / seg) + threadIdx.x / seg, part = threadIdx.x % seg;
        valid = row < H;
        first = row * L + part * P.leaf_r;
        stride = 1;
        leaf = P.leaf_r;
        slot = threadIdx.x;
        out_idx = row;
    } else {
        local -= P.wgs_r;
        seg = P.seg_c;
        const uint32_t cpw = THREADS / seg;  // columns per workgroup; the column index is the fastest lane index: neighbours read neighbouring buckets
        const uint32_t col = local * cpw + threadIdx.x % cpw, part = threadIdx.x / cpw;
        valid = col < L;
        first = part * P.leaf_c * L + col;
        stride = L;
        leaf = P.leaf_c;
        slot = (threadIdx.x % cpw) * seg + part;
        out_idx = col;
    }
    G1X28 acc = g1x28::identity();
    if (valid) {
        acc = b[first];
        for (uint32_t j = 1; j < leaf; ++j) {
            const G1X28 q = b[first + j * stride];
            g1x28::add_full(acc, q);
        }
    }
    sh[slot] = acc;
    __syncthreads();
    g1_quad_tree_seg<THREADS>(sh, (int)seg);
    if (valid && slot % seg == 0) (col_pass ? cols : rows)[(size_t)set * (col_pass ? L : H) + out_idx] = sh[slot];
}
// Terms of a set.  Bit k of the bucket index contributes S_k = (k < lbits: the column sums with bit k of l set; else: the row sums with bit k - lbits of h
// set); two ADJACENT bits share a workgroup of 2 x THREADS lanes (THREADS = 128: the two trees' 2 x 128 points are the 56 KiB of LDS a workgroup may declare) - each half runs the tree of one bit, then one quad forms D_j = S_2j + 2 S_(2j+1) (a
// doubling and an addition: ~12 us) - so the host's Horner adds half as many points (an XYZZ addition costs a host core 0.5 us: 272 of them were 0.14 of
// the step's 3.5 ms).  Terms of a set, standard-limb XYZZ at terms[set * n_terms + t]:  t < n_pairs: D_t (bit position 2 t);  t = n_pairs: the plain sum
// of the set (every row sum; position 0).  n_pairs = ceil((c - 1) / 2).
__host__ __device__ inline uint32_t msm_bitsum_pairs(const RowColPlan& P) { return (P.lbits + P.hbits + 1u) / 2u; }
template <int THREADS>
__global__ void __launch_bounds__(2 * THREADS) msm_bitsum_quad_kernel(const G1X28* __restrict__ rows, const G1X28* __restrict__ cols, RowColPlan P,
                                                                      XyzzT<FpOps>* __restrict__ terms) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ G1X28 sh[2 * THREADS];
    const uint32_t n_bits = P.lbits + P.hbits, n_pairs = msm_bitsum_pairs(P), n_terms = n_pairs + 1u;
    const uint32_t set = blockIdx.x / n_terms, t = blockIdx.x % n_terms;
    const uint32_t half = threadIdx.x / THREADS, tid = threadIdx.x % THREADS;  // half 0: bit 2 t (or the plain sum), half 1: bit 2 t + 1
    const bool total = t == n_pairs;
    const uint32_t k = 2u * t + half;
    const bool have = total ? half == 0 : k < n_bits;  // an odd bit count leaves the last pair's upper half empty
    const bool from_cols = !total && k < P.lbits;
    const uint32_t n_src = 1u << (from_cols ? P.lbits : P.hbits), bit = total ? 0u : (from_cols ? k : k - P.lbits);
    const G1X28* src = (from_cols ? cols : rows) + (size_t)set * n_src;
    const uint32_t n_leaves = !have ? 0u : (total ? n_src : n_src / 2);
    auto leaf_index = [&](uint32_t j) { return total ? j : ((((j >> bit) << 1) | 1u) << bit) | (j & ((1u << bit) - 1u)); };
    G1X28 acc = g1x28::identity();
    if (tid < n_leaves) {
        acc = src[leaf_index(tid)];
        for (uint32_t j = tid + THREADS; j < n_leaves; j += THREADS) {
            const G1X28 q = src[leaf_index(j)];
            g1x28::add_full(acc, q);
        }
    }
    G1X28* const my = sh + half * THREADS;
    my[tid] = acc;
    __syncthreads();
    // both halves walk the same number of levels (the barriers are the workgroup's); a half whose tree is narrower idles through the wide levels
    const uint32_t widest = 1u << (P.lbits > P.hbits ? P.lbits : P.hbits);
    int levels_from = 1;
    while (levels_from < THREADS && (uint32_t)levels_from < widest) levels_from <<= 1;
    const int quad = (int)(tid >> 2), n_quads = THREADS / 4;
    for (int s = levels_from / 2; s > 0; s >>= 1) {  // (slots beyond a half's leaves hold identities: their additions return at once)
        for (int p = quad; p < s; p += n_quads) {
            const G1X28 r = g1_add_quad(my[p], my[p + s]);
            if ((tid & 3u) == 0) my[p] = r;
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) {  // one quad: D = S_lo + 2 S_hi
        G1X28 r = sh[0];
        if (!total && 2u * t + 1u < n_bits) r = g1_add_quad(r, g1_dbl_quad(sh[THREADS]));
        if (threadIdx.x == 0) terms[(size_t)set * n_terms + t] = g1x28::to_std(r);
    }
#endif
}

template <class C, int THREADS>
__global__ void __launch_bounds__(THREADS) msm_window_partial_kernel(const typename C::Pt* __restrict__ chunk_out, uint32_t per_win,
                                                                     uint32_t groups, typename C::Pt* __restrict__ partial_out) {
    typedef typename C::Pt Pt;
    __shared__ Pt sh[THREADS];
    const uint32_t w = blockIdx.x / groups, g = blockIdx.x % groups;
    const uint32_t i = g * THREADS + threadIdx.x;
    Pt acc = i < per_win ? chunk_out[(size_t)w * per_win + i] : C::identity();
    Pt r = block_tree_sum<C, THREADS>(acc, sh);
    if (threadIdx.x == 0) partial_out[(size_t)w * groups + g] = r;
}

template <class C, int THREADS>
__global__ void __launch_bounds__(THREADS) msm_window_sum_kernel(const typename C::Pt* __restrict__ partials, uint32_t groups,
                                                                 XyzzT<typename C::HostF>* __restrict__ win_out) {
    typedef typename C::Pt Pt;
    __shared__ Pt sh[THREADS];
    const uint32_t w = blockIdx.x;
    const Pt* src = partials + (size_t)w * groups;
    // the first term is loaded, not added to an identity (one general addition of latency less)
    Pt acc = threadIdx.x < groups ? src[threadIdx.x] : C::identity();
    for (uint32_t i = threadIdx.x + THREADS; i < groups; i += THREADS) add_from<C>(acc, &src[i]);
    int active = 1;
    while (active < THREADS && (uint32_t)active < groups) active <<= 1;  // 16 partials per window at c = 16: 4 levels, not 8
    Pt r = block_tree_sum<C, THREADS>(acc, sh, active);
    if (threadIdx.x == 0) win_out[w] = C::to_std(r);  // standard 12 x 32-bit XYZZ for the host
}

// ------------------------------------------------------------------------------------------------
// synthetic bases: out[i] = k_i * G
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix_first(uint64_t seed) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <class F>
__global__ void __launch_bounds__(64) synth_bases_kernel(AffineT<F> gen, uint64_t seed, uint64_t start, uint64_t n,
                                                         AffineT<F>* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = splitmix_first(seed + 0x632BE59BD9B4E019ull * (start + i)) | 1;
    XyzzT<F> r = xyzz_identity<F>();
    for (int b = 63; b >= 0; --b) {
        r = xyzz_dbl<F>(r);
        if ((k >> b) & 1) xyzz_add_mixed<F>(r, gen);
    }
    AffineT<F> a;
    xyzz_to_affine<F>(r, a);
    out[i] = a;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static Fp fp_from_canon_be_hex(const char* hex) {  // 96 hex chars, big-endian
    Fp c = Fp::zero();
    for (int i = 0; i < 96; ++i) {
        char ch = hex[i];
        uint32_t v = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        int nib = 95 - i;  // nibble index from LSB
        c.l[nib / 8] |= v << ((nib % 8) * 4);
    }
    return fe_to_mont<FpParams>(c);
}

static G1Affine g1_generator_host() {
    static const G1Affine g = {
        fp_from_canon_be_hex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
        fp_from_canon_be_hex("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")};
    return g;
}
static G2Affine g2_generator_host() {
    static const G2Affine g = {
        {fp_from_canon_be_hex("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
         fp_from_canon_be_hex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")},
        {fp_from_canon_be_hex("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
         fp_from_canon_be_hex("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")}};
    return g;
}

template <class F>
struct PointIO;
template <>
struct PointIO<FpOps> {
    static constexpr int RAW = 96, PACKED = 97;
    static void pack(const XyzzT<FpOps>& p, uint8_t* out) {
        AffineT<HFpOps> a;  // the inversion on the 64-bit-limb host field (host_fp64.h)
        bool fin = xyzz_to_affine<HFpOps>(to_host_fast<FpOps>(p), a);
        memcpy(out, a.x.l, 48);
        memcpy(out + 48, a.y.l, 48);
        out[96] = fin ? 0 : 1;
    }
    static XyzzT<FpOps> unpack(const uint8_t* in) {
        if (in[96]) return xyzz_identity<FpOps>();
        G1Affine a;
        memcpy(a.x.l, in, 48);
        memcpy(a.y.l, in + 48, 48);
        return xyzz_from_affine<FpOps>(a);
    }
};
template <>
struct PointIO<Fp2Ops> {
    static constexpr int RAW = 192, PACKED = 193;
    static void pack(const XyzzT<Fp2Ops>& p, uint8_t* out) {
        AffineT<HFp2Ops> a;
        bool fin = xyzz_to_affine<HFp2Ops>(to_host_fast<Fp2Ops>(p), a);
        memcpy(out, a.x.c0.l, 48);
        memcpy(out + 48, a.x.c1.l, 48);
        memcpy(out + 96, a.y.c0.l, 48);
        memcpy(out + 144, a.y.c1.l, 48);
        out[192] = fin ? 0 : 1;
    }
    static XyzzT<Fp2Ops> unpack(const uint8_t* in) {
        if (in[192]) return xyzz_identity<Fp2Ops>();
        G2Affine a;
        memcpy(a.x.c0.l, in, 48);
        memcpy(a.x.c1.l, in + 48, 48);
        memcpy(a.y.c0.l, in + 96, 48);
        memcpy(a.y.c1.l, in + 144, 48);
        return xyzz_from_affine<Fp2Ops>(a);
    }
};

static int bits_for(uint64_t v) {
    int b = 1;
    while (b < 64 && ((uint64_t)1 << b) <= v) ++b;
    return b;
}

// ------------------------------------------------------------------------------------------------
// 8. scalar de-duplication (flag BZK_F_DEDUP).  A Groth16 witness repeats itself: of the 904 876 assignment values
// of the 16-tx Update circuit only 486 398 are distinct (277 k values occur exactly twice, the bits 0/1 thousands of
// times).  Equal scalars s contribute s * (P_i + P_j + ...), so the bases of each group are summed ONCE (one mixed
// add per duplicate, through the same task-based accumulation as the buckets), the sums are brought to affine form
// with a batched inversion and the bucket phase then runs over the distinct scalars only - instead of repeating
// every duplicate's addition in each of the W windows.  Zero scalars are dropped.  The result is the same group
// element; grouping is by a 64-bit hash sort with a full 256-bit comparison of sorted neighbours (a hash collision
// can only split a group, never merge two).
// ------------------------------------------------------------------------------------------------
// 32-bit sort keys since round 2 (were 64): a collision of two DIFFERENT scalars only interleaves two groups in the sorted order
// and thereby splits them (the neighbour comparison is on all 256 bits), i.e. costs a little de-duplication - n^2 / 2^33 pairs:
// ~130 at the 0.9 M values of a 16-tx circuit, ~30 k of 14 M at the production size - while the sort drops from 8-byte keys and
// eight 8-bit passes to 4-byte keys and four (the four de-duplication sorts of a 256-tx proof took 105 ms of its 260,
// profiles/r02_run10_production.txt)
static __global__ void __launch_bounds__(256) dedup_hash_kernel(const U128* __restrict__ scalars, uint64_t n, uint32_t* __restrict__ hkey,
                                                                uint32_t* __restrict__ idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const U128 a = scalars[2 * i], b = scalars[2 * i + 1];
    const uint32_t l[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint64_t h = 0x9E3779B97F4A7C15ull;
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        any |= l[k];
        h = (h ^ l[k]) * 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
    }
    uint32_t h32 = (uint32_t)(h ^ (h >> 32));
    if (h32 == ~0u) h32 = ~0u - 1;
    hkey[i] = any ? h32 : ~0u;  // zero scalars sort to the end and are dropped
    idx[i] = (uint32_t)i;
}

__device__ __forceinline__ bool dedup_same(const U128* __restrict__ scalars, uint32_t i, uint32_t j) {
    const U128 a0 = scalars[2 * (uint64_t)i], a1 = scalars[2 * (uint64_t)i + 1], b0 = scalars[2 * (uint64_t)j], b1 = scalars[2 * (uint64_t)j + 1];
    return a0.x == b0.x && a0.y == b0.y && a0.z == b0.z && a0.w == b0.w && a1.x == b1.x && a1.y == b1.y && a1.z == b1.z && a1.w == b1.w;
}

// head[i]: sorted position i starts a group of equal non-zero scalars; mhead[i]: ... of at least two members
static __global__ void __launch_bounds__(256) dedup_heads_kernel(const U128* __restrict__ scalars, const uint32_t* __restrict__ hkey_s,
                                                                 const uint32_t* __restrict__ idx_s, uint64_t n, uint32_t* __restrict__ head,
                                                                 uint32_t* __restrict__ mhead) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool alive = hkey_s[i] != ~0u;
    bool h = false, m = false;
    if (alive) {
        h = i == 0 || hkey_s[i - 1] != hkey_s[i] || !dedup_same(scalars, idx_s[i - 1], idx_s[i]);
        if (h) m = i + 1 < n && hkey_s[i + 1] == hkey_s[i] && dedup_same(scalars, idx_s[i + 1], idx_s[i]);
    }
    head[i] = h ? 1u : 0u;
    mhead[i] = m ? 1u : 0u;
}

// per sorted position: the bucket key of the group-sum accumulation (multi-member groups only) and, at group heads,
// the compacted scalar, its base index (`rep`: the original base, or n + m for the m-th group sum) and gof[m] = g
static __global__ void __launch_bounds__(256) dedup_assign_kernel(const U128* __restrict__ scalars, const uint32_t* __restrict__ hkey_s,
                                                                  const uint32_t* __restrict__ idx_s, const uint32_t* __restrict__ head,
                                                                  const uint32_t* __restrict__ mhead, const uint32_t* __restrict__ gid_ex,
                                                                  const uint32_t* __restrict__ mid_ex, uint64_t n, uint32_t sentinel,
                                                                  uint32_t* __restrict__ key2, U128* __restrict__ scal2,
                                                                  uint32_t* __restrict__ rep, uint32_t* __restrict__ gof) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool alive = hkey_s[i] != ~0u;
    const bool h = head[i] != 0, m = mhead[i] != 0;
    const bool member = alive && (m || !h);  // belongs to a group of >= 2
    key2[i] = member ? mid_ex[i] + mhead[i] - 1 : sentinel;
    if (h) {
        const uint32_t g = gid_ex[i], src = idx_s[i];
        scal2[2 * (uint64_t)g] = scalars[2 * (uint64_t)src];
        scal2[2 * (uint64_t)g + 1] = scalars[2 * (uint64_t)src + 1];
        if (m) {
            rep[g] = (uint32_t)n + mid_ex[i];
            gof[mid_ex[i]] = g;
        } else {
            rep[g] = src;
        }
    }
}

// group sums (XYZZ) -> internal affine form, batched inversion: one lane per K consecutive sums (Montgomery's trick,
// prefix products parked in `pref`).  A sum that is the identity (P + (-P)) cannot be a base: its scalar is zeroed.
template <class C>
__global__ void __launch_bounds__(64) dedup_affine_kernel(const typename C::Pt* __restrict__ sums, uint32_t m_total, uint32_t K,
                                                          typename C::Fld* __restrict__ pref, typename C::DevAff* __restrict__ out,
                                                          const uint32_t* __restrict__ gof, U128* __restrict__ scal2) {
    typedef typename C::Fld Fld;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = (uint64_t)t * K;
    if (lo >= m_total) return;
    const uint32_t hi = (uint32_t)(lo + K < m_total ? lo + K : m_total);
    Fld acc = C::f_one();
    for (uint32_t i = (uint32_t)lo; i < hi; ++i) {
        pref[i] = acc;
        const typename C::Pt p = sums[i];
        if (!C::is_identity(p)) acc = C::f_mul(acc, p.ZZZ);
    }
    Fld inv = C::f_inv(acc);
    for (uint32_t i = hi; i-- > (uint32_t)lo;) {
        const typename C::Pt p = sums[i];
        if (C::is_identity(p)) {
            const uint32_t g = gof[i];
            scal2[2 * (uint64_t)g] = U128{0, 0, 0, 0};
            scal2[2 * (uint64_t)g + 1] = U128{0, 0, 0, 0};
            continue;
        }
        const Fld i3 = C::f_mul(inv, pref[i]);  // 1 / ZZZ_i
        inv = C::f_mul(inv, p.ZZZ);
        out[i] = C::affine_with_inv(p, i3);
    }
}

// ---- the saturating part of an MSM on a LOWEST-priority stream (round 6) ----------------------------------------------------------------
// What the kernel trace of two MSMs in flight shows (profiles/r06_run1_two_msms_timeline.txt): beside another stream's accumulation - two
// long-lived, issue-saturating waves per SIMD - the short kernels of an MSM's front chain (digits 0.04 ms alone, the sort passes, boundaries)
// take 1.3 + 0.8 + 0.8 ms: younger waves get the VALU slots the older ones leave.  The chain gates this MSM's own accumulation, so the two
// MSMs end up taking turns (3.4 ms each instead of 3.8) instead of overlapping.  With the accumulation, the folds and the bucket reduction on
// a stream of the device's LOWEST priority, everything else - of every stream of the process - outranks them.
// env BZK_MSM_HEAVY_PRIO=0 | 1 (A/B).
static bool msm_heavy_prio_on() {
    static const bool on = [] { const char* e = getenv("BZK_MSM_HEAVY_PRIO"); return e ? atoi(e) != 0 : BZK_MSM_HEAVY_PRIO_DEFAULT; }();
    return on;
}
struct HeavyScope {
    bzk_ctx* ctx;
    hipStream_t home = nullptr;
    bool in = false;
    explicit HeavyScope(bzk_ctx* c) : ctx(c) {}
    HeavyScope(const HeavyScope&) = delete;
    void enter() {
        if (in || !(msm_heavy_prio_on() || ctx->heavy_force)) return;
        if (!ctx->heavy && !ctx->heavy_tried) {
            ctx->heavy_tried = true;
            int least = 0, greatest = 0;
            hipStream_t s = nullptr;
            hipEvent_t a = nullptr, b = nullptr;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess ||
                hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                if (s) (void)hipStreamDestroy(s);
                if (a) (void)hipEventDestroy(a);
                if (b) (void)hipEventDestroy(b);
            } else {
                ctx->heavy = s; ctx->ev_heavy_in = a; ctx->ev_heavy_out = b;
            }
        }
        if (!ctx->heavy) return;  // no side stream: everything stays on the context's stream
        if (hipEventRecord(ctx->ev_heavy_in, ctx->stream) != hipSuccess || hipStreamWaitEvent(ctx->heavy, ctx->ev_heavy_in, 0) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        home = ctx->stream;
        ctx->stream = ctx->heavy;  // BZK_LAUNCH / ProfScope work on ctx->stream
        in = true;
    }
    void leave() {
        if (!in) return;
        in = false;
        const hipError_t e = hipEventRecord(ctx->ev_heavy_out, ctx->heavy);
        ctx->stream = home;
        // the home stream continues behind the side stream's work (on failure: a host-side wait keeps the order)
        if (e != hipSuccess || hipStreamWaitEvent(home, ctx->ev_heavy_out, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(ctx->heavy);
        }
    }
    ~HeavyScope() { leave(); }
};

// bucket phase shared by the windows and by the group sums: boundaries of each key's run in the sorted pair list,
// buckets in population order, task table, task-based accumulation, folds.  Result: buckets[key] for key < nb.
template <class Pt>
struct BucketArrays {
    uint32_t *start, *count, *count_s, *iota, *order, *ntask, *tbase;
    Pt* partial;
    Pt* wide = nullptr;  // chunk sums of giant buckets (msm_fold_wide_kernel): msm_fold_wide_cap(capacity of `partial`) points, or null = one-level fold
};
template <class C>
static int32_t bucket_accumulate(bzk_ctx* ctx, const void* bases, const uint32_t* keys_s, const uint32_t* vals_s, uint64_t len, uint32_t nb,
                                 uint32_t seg, const BucketArrays<typename C::Pt>& A, typename C::Pt* buckets, void* tmp_buf, size_t tmp,
                                 bool group_sums = false, uint32_t wiv_half = 0, const void* bases2 = nullptr, uint32_t n_split = 0xffffffffu,
                                 uint32_t ibits = 31, uint32_t stride1 = 0, uint32_t stride2 = 0, HeavyScope* heavy = nullptr) {
    // start[] and count[] are taken from the workspace back to back: one fill covers both
    if ((const char*)A.count > (const char*)A.start && (size_t)((const char*)A.count - (const char*)A.start) <= (size_t)nb * 4 + 256) {
        BZK_HIP(ctx, hipMemsetAsync(A.start, 0, (size_t)((const char*)A.count - (const char*)A.start) + (size_t)nb * 4, ctx->stream));
    } else {
        BZK_HIP(ctx, hipMemsetAsync(A.start, 0, (size_t)nb * 4, ctx->stream));
        BZK_HIP(ctx, hipMemsetAsync(A.count, 0, (size_t)nb * 4, ctx->stream));
    }
    const uint32_t vmask = wiv_half ? 0x07ffffffu : 0x7fffffffu;
    if (wiv_half) {
        BZK_LAUNCH(ctx, "msm_offsets", msm_offsets_wiv_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, keys_s, vals_s, len, wiv_half,
                   A.start, A.count);
    } else {
        BZK_LAUNCH(ctx, "msm_offsets", msm_offsets_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, keys_s, len, nb, A.start, A.count);
    }
    // counts, identity permutation and the clamped population keys in one pass: A.ntask holds the keys until msm_ntask overwrites it,
    // A.tbase receives the (unused) sorted keys
    // one radix pass over 8-bit keys while the mean population is small, two over 16-bit keys otherwise (msm_count_kernel)
    const bool small_pop = len / nb <= 64;
    const uint32_t pop_clamp = small_pop ? 255u : 65535u;
    const int pop_bits = small_pop ? 8 : 16;
    BZK_LAUNCH(ctx, "msm_count", msm_count_kernel, dim3((nb + 255) / 256), dim3(256), 0, A.start, A.count, A.iota, A.ntask, nb, pop_clamp);
    {
        ProfScope ps(ctx, "msm_sort_buckets");
        size_t t = tmp;
        hipError_t e = msm_tuned_sort()
                           ? rocprim::radix_sort_pairs_desc<SortCfg32>(tmp_buf, t, A.ntask, A.tbase, A.iota, A.order, (size_t)nb, 0, pop_bits, ctx->stream)
                           : rocprim::radix_sort_pairs_desc(tmp_buf, t, A.ntask, A.tbase, A.iota, A.order, (size_t)nb, 0, pop_bits, ctx->stream);
        if (e != hipSuccess) { ctx->last_error = std::string("radix_sort_pairs_desc: ") + hipGetErrorString(e); return BZK_E_DEVICE; }
    }
    BZK_LAUNCH(ctx, "msm_ntask", msm_ntask_kernel, dim3((nb + 255) / 256), dim3(256), 0, A.count, A.order, nb, seg, A.count_s, A.ntask);
    {
        ProfScope ps(ctx, "msm_scan_tasks");
        size_t t = tmp;
        hipError_t e = rocprim::exclusive_scan(tmp_buf, t, A.ntask, A.tbase, 0u, (size_t)nb, rocprim::plus<uint32_t>(), ctx->stream);
        if (e != hipSuccess) { ctx->last_error = std::string("exclusive_scan: ") + hipGetErrorString(e); return BZK_E_DEVICE; }
    }
    const uint32_t t_max = (uint32_t)((uint64_t)nb + len / seg + 1);
    if (heavy) heavy->enter();  // accumulation + folds (+ the caller's bucket reduction) on the lowest-priority stream; the caller leaves the scope
    // G1: 2 waves/SIMD: forcing 3 or 4 (__launch_bounds__) spills 68 / 223 VGPRs and measured 13 % / 80 % slower.
    // G2: 1 wave/SIMD (512 registers: 8 spilled instead of 402 at 2 waves), see msm_policy.cuh
    auto k_acc = msm_accumulate_kernel<C, C::ACC_OCC>;
    auto k_fold = msm_fold_kernel<C>;
    auto k_fold_small = msm_fold_small_kernel<C>;
    const bool g2_pair = msm_g2_pair_on();
    bool launched = false;
    if constexpr (C::PAIR_ACC) {
        if (g2_pair) {
            const uint64_t lanes = 2ull * t_max;
            if (group_sums) {
                BZK_LAUNCH(ctx, "dedup_accumulate", (msm_accumulate_g2pair_kernel<BZK_G2_PAIR_OCC>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, bases,
                           vals_s, A.start, A.count_s, A.order, A.tbase, nb, t_max, seg, buckets, A.partial, vmask, bases2, n_split, ibits, stride1, stride2);
            } else {
                BZK_LAUNCH(ctx, "msm_accumulate", (msm_accumulate_g2pair_kernel<BZK_G2_PAIR_OCC>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, bases,
                           vals_s, A.start, A.count_s, A.order, A.tbase, nb, t_max, seg, buckets, A.partial, vmask, bases2, n_split, ibits, stride1, stride2);
            }
            launched = true;
        }
    }
    if (launched) {
    } else if (group_sums) {
        BZK_LAUNCH(ctx, "dedup_accumulate", k_acc, dim3((t_max + 127) / 128), dim3(128), 0, bases, vals_s, A.start, A.count_s, A.order, A.tbase,
                   nb, t_max, seg, buckets, A.partial, vmask, bases2, n_split, ibits, stride1, stride2);
    } else {
        BZK_LAUNCH(ctx, "msm_accumulate", k_acc, dim3((t_max + 127) / 128), dim3(128), 0, bases, vals_s, A.start, A.count_s, A.order, A.tbase,
                   nb, t_max, seg, buckets, A.partial, vmask, bases2, n_split, ibits, stride1, stride2);
    }
    // only sorted positions < len / seg can hold a multi-task bucket; of those only the first
    // len / (seg * MSM_FOLD_SMALL) can need the workgroup-wide fold
    const uint32_t n_pos = (uint32_t)std::min<uint64_t>(nb, len / seg + 1);
    // upper bound of the partial sums beyond one per bucket is len / seg: below MSM_FOLD_BULK_FROM the device is certainly in the
    // latency regime (thr = 2); at or above it may be in either, and the workgroup fold must still cover positions with > 2 tasks
    // unless even the LOWER bound of `extra` says bulk.  extra >= tasks of full runs only = sum floor(cnt / seg) - (buckets with
    // cnt > seg) which the host does not know; so: bulk is assumed only for the launch SIZE when len / seg >= 4 x the threshold (the
    // 2^24-point regime: 524 288 launches that exit at once otherwise) and the kernel re-checks per bucket either way
    const bool surely_bulk = len / seg >= 4ull * MSM_FOLD_BULK_FROM && len / seg >= 2ull * nb;
    const uint32_t n_big = (uint32_t)std::min<uint64_t>(nb, len / ((uint64_t)seg * (surely_bulk ? MSM_FOLD_SMALL_BULK : MSM_FOLD_SMALL)) + 1);
    if constexpr (C::PAIR_ACC) {
        if (msm_g2_pair_tails_on()) {
            static const bool wide_off_p = env_on("BZK_MSM_NO_WIDE_FOLD");  // A/B runs
            const G2X28* const wide_p = wide_off_p ? nullptr : (const G2X28*)A.wide;
            if (wide_p && len > (uint64_t)seg * MSM_FOLD_WIDE_FROM)  // a giant bucket holds more than MSM_FOLD_WIDE_FROM tasks of `seg` entries
                BZK_LAUNCH(ctx, "msm_fold_wide", msm_fold_wide_g2pair_kernel<0>, dim3(MSM_FOLD_WIDE_GRID), dim3(128), 0, A.count_s, A.tbase, nb, seg, MSM_FOLD_WIDE_FROM,
                           MSM_FOLD_WIDE_POS, (const G2X28*)A.partial, (G2X28*)A.wide);
            BZK_LAUNCH(ctx, "msm_fold", msm_fold_g2pair_kernel<0>, dim3(n_big), dim3(128), 0, A.count_s, A.order, A.tbase, A.ntask, nb, seg, MSM_FOLD_BULK_FROM,
                       MSM_FOLD_SMALL, MSM_FOLD_SMALL_BULK, A.partial, buckets, wide_p, MSM_FOLD_WIDE_FROM, MSM_FOLD_WIDE_POS);
            BZK_LAUNCH(ctx, "msm_fold_small", msm_fold_small_g2pair_kernel<0>, dim3((unsigned)((2ull * n_pos + 63) / 64)), dim3(64), 0, A.count_s, A.order, A.tbase,
                       A.ntask, nb, n_pos, seg, MSM_FOLD_BULK_FROM, MSM_FOLD_SMALL, MSM_FOLD_SMALL_BULK, A.partial, buckets);
            return BZK_OK;
        }
    }
    static const bool wide_off = env_on("BZK_MSM_NO_WIDE_FOLD");  // A/B runs
    const typename C::Pt* const wide = wide_off ? nullptr : A.wide;
    // a giant bucket holds more than MSM_FOLD_WIDE_FROM tasks of `seg` entries: none can exist in a shorter pair list
    if (wide && len > (uint64_t)seg * MSM_FOLD_WIDE_FROM) {
        auto k_wide = msm_fold_wide_kernel<C>;
        BZK_LAUNCH(ctx, "msm_fold_wide", k_wide, dim3(MSM_FOLD_WIDE_GRID), dim3(64), 0, A.count_s, A.tbase, nb, seg, A.partial, A.wide);
    }
    BZK_LAUNCH(ctx, "msm_fold", k_fold, dim3(std::min<uint32_t>(n_big, 2048u)), dim3(64), 0, A.count_s, A.order, A.tbase, A.ntask, nb, n_big, seg, A.partial, wide, buckets);
    BZK_LAUNCH(ctx, "msm_fold_small", k_fold_small, dim3((n_pos + 63) / 64), dim3(64), 0, A.count_s, A.order, A.tbase, A.ntask, nb, n_pos, seg,
               A.partial, buckets);
    return BZK_OK;
}


// side stream of a call (bzk_ctx::aux): lazily created; `false` = run everything on the main stream
static bool msm_aux_ready(bzk_ctx* ctx) {
    static const bool off = [] { const char* e = getenv("BZK_MSM_NO_AUX"); return e && atoi(e) != 0; }();
    if (off) return false;
    if (ctx->aux) return true;
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (s) (void)hipStreamDestroy(s);
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        return false;
    }
    ctx->aux = s; ctx->ev_fork = a; ctx->ev_join = b;
    return true;
}
// the main stream waits for the side stream's work of this call exactly once - at the first consumer or, failing that, when the
// call returns (the workspace must not be re-used or freed under a conversion that is still running)
struct AuxJoin {
    bzk_ctx* ctx;
    bool pending = false;
    explicit AuxJoin(bzk_ctx* c) : ctx(c) {}
    void join() {
        if (pending) {
            (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
            pending = false;
        }
    }
    ~AuxJoin() { join(); }
};
template <class C>
static int32_t msm_convert_launch(bzk_ctx* ctx, const void* bases_raw, uint64_t n, typename C::DevAff* conv) {
    auto k_conv = msm_convert_bases_kernel<C>;
    BZK_LAUNCH(ctx, "msm_convert_bases", k_conv, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, bases_raw, n, conv);
    return BZK_OK;
}

// Computes sum over windows [w_begin, w_end) of 2^(c w) S_w into `result` (host XYZZ, standard limbs).
struct MsmTable {
    void* data = nullptr;  // DevAff[w_total][n]
    uint64_t n = 0;
    int c = 0, w_total = 0;  // w_total: windows of the signed c-bit recoding
    int levels = 0, wpl = 1; // table levels; windows per level (1 = full table)
};

// A resident base set in the policy's internal form (G1: 112 B, G2: 224 B per point): converted ONCE when a static point set - a
// Groth16 CRS query - is loaded, shared read-only by every later call (and by every rank of a window-sharded MSM: no rank converts
// anything per call).  bzk_msm_g*_bases_*.
// endomorphism policy of a curve (msm_policy.cuh ENDO_DEFAULT; env BZK_MSM_ENDO_G1 / _G2): 0 never, 1 every whole-MSM call over a set
// with images, 2 only calls flagged BZK_F_THROUGHPUT.  ONE predicate for load time (are the images worth their memory?) and run time
template <class C>
static int msm_endo_mode() {
    static const int mode = [] { const char* e = getenv(C::ENDO_ENV); return e ? atoi(e) : C::ENDO_DEFAULT; }();
    return mode;
}
struct MsmBases {
    void* data = nullptr;  // DevAff[endo][n]: the set itself, then (endo > 1) its images X^m P, m = 1 .. endo - 1 (bzk_endo.cuh)
    uint64_t n = 0;
    int device = 0;
    int endo = 1;
    size_t bytes = 0;  // device memory of `data` (bzk_msm_bases_info)
};
// Where a call that must not finish on this host thread leaves its window sums: `d_win` receives, in DEVICE memory and in stream
// order, the (w_end - w_begin) window sums S_w as standard-limb XYZZ points; no read-back, no Horner, no synchronisation.  The
// caller combines them - possibly with the windows of other devices - through msm_horner_host.  (bzk_mg: multi-GPU entry points.)
struct MsmWinOut {
    void* d_win = nullptr;
    int c = 0, w_total = 0, w_begin = 0, w_end = 0;  // filled in by msm_run
    bool single = false;                             // full static table: d_win[0] is the result itself
    // round 6: 0 = d_win holds one window SUM per bucket set; n > 0 = it holds the n TERMS of every bucket set (multiplication-free reduction, section 6b:
    // the caller combines them with msm_horner_terms_host) - msm_terms_per_set<C>(c) says which before the call, so that ranks size their exchange alike
    int terms_per_set = 0;
};
// does the multiplication-free reduction (section 6b) serve this curve at window size c?  G1 with 11 <= c <= 21; env BZK_MSM_BITSUM=0: never (A/B)
template <class C>
static bool msm_bitsum_applies(int c) {
    static const bool on = [] { const char* e = getenv("BZK_MSM_BITSUM"); return e ? atoi(e) != 0 : true; }();
    return on && !C::PARK_REDUCE && c >= 11 && c <= 21;
}
template <class C>
static int msm_terms_per_set(int c) { return msm_bitsum_applies<C>(c) ? c / 2 + 1 : 0; }

// Horner over window sums on the host: result = sum_k 2^(c (w0 + k)) S[k].  c * (count + w0) doublings that nothing can overlap
// with - they run on the 6 x 64-bit host field of host_fp64.h (round 3: 0.63 -> 0.45 us per doubling, same values).
template <class F>
static XyzzT<F> msm_horner_host(const XyzzT<F>* S, int count, int c, int w0) {
    typedef typename HostFast<F>::Ops H;
    XyzzT<H> acc = xyzz_identity<H>();
    for (int k = count - 1; k >= 0; --k) {
        for (int d = 0; d < c; ++d) acc = xyzz_dbl<H>(acc);
        xyzz_add<H>(acc, to_host_fast<F>(S[k]));
    }
    for (int d = 0; d < c * w0; ++d) acc = xyzz_dbl<H>(acc);
    return from_host_fast<F>(acc);
}

// Horner over the TERMS of the multiplication-free reduction (section 6b): bucket set k (k < count) contributes its n_pairs digit terms D_j = S_2j + 2 S_(2j+1)
// at bit positions c (w0 + k) + 2 j and its last term - the plain sum of the set - at position c (w0 + k).  The same c (count + w0) doublings as
// msm_horner_host, count * (n_pairs + 1) additions.
template <class F>
static XyzzT<F> msm_horner_terms_host(const XyzzT<F>* T, int count, int c, int w0) {
    typedef typename HostFast<F>::Ops H;
    const int n_pairs = c / 2, n_terms = n_pairs + 1;  // ceil((c - 1) / 2)
    XyzzT<H> acc = xyzz_identity<H>();
    for (int k = count - 1; k >= 0; --k) {
        const XyzzT<F>* t = T + (size_t)k * n_terms;
        for (int bit = c - 1; bit >= 0; --bit) {
            acc = xyzz_dbl<H>(acc);
            if ((bit & 1) == 0 && bit / 2 < n_pairs) xyzz_add<H>(acc, to_host_fast<F>(t[bit / 2]));
            if (bit == 0) xyzz_add<H>(acc, to_host_fast<F>(t[n_pairs]));
        }
    }
    for (int d = 0; d < c * w0; ++d) acc = xyzz_dbl<H>(acc);
    return from_host_fast<F>(acc);
}

template <class C>
static int32_t msm_run(bzk_ctx* ctx, const void* bases_raw, const void* scalars, uint64_t n, uint32_t flags, int w_begin,
                       int w_end, XyzzT<typename C::HostF>& result, const MsmTable* table = nullptr, const MsmBases* prep = nullptr,
                       MsmWinOut* wout = nullptr) {
    typedef typename C::HostF F;
    typedef typename C::Pt Pt;
    typedef XyzzT<F> StdPt;
    result = xyzz_identity<F>();
    if (wout) { wout->c = 0; wout->w_total = 0; wout->w_begin = wout->w_end = 0; wout->single = false; }
    if (n == 0 || (w_end >= 0 && w_begin >= w_end)) return BZK_OK;  // w_end < 0: every window (resolved once c is known)
    if (n >= ((uint64_t)1 << 31)) return BZK_E_ARG;
    if (prep && (table || n > prep->n || !prep->data)) return BZK_E_ARG;
    int c = table ? table->c : (ctx->msm_c_override >= 2 && ctx->msm_c_override <= 20 ? ctx->msm_c_override : msm_pick_c(n));
    // the work-model pick applies to unsharded calls only: a caller that names a window range counts windows with
    // bzk_msm_window_count(n), i.e. with the plain pick
    if (!table && (flags & BZK_F_DEDUP) && w_end < 0 && w_begin == 0 && !(ctx->msm_c_override >= 2 && ctx->msm_c_override <= 20))
        c = msm_pick_c_witness<C>(n, c);
    const bool dedup = (flags & BZK_F_DEDUP) && C::CONVERT_BASES && !table && n >= 4096 && n < ((uint64_t)1 << 30);
    const uint32_t m_max = dedup ? (uint32_t)(n / 2 + 1) : 0;  // group sums: at most n / 2 groups of >= 2 members
    // window-in-value pairs (msm_digits): per-window bucket sets of up to 16 windows per pass over fewer than 2^27 bases
    static const bool wiv_off = [] { const char* e = getenv("BZK_MSM_NO_WIV"); return e && atoi(e) != 0; }();
    // Endomorphism form (bzk_endo.cuh, round 4) for whole MSMs over a resident set that carries its images: E sub-scalars per scalar
    // whose windows share w_sub = ENDO_BITS / c bucket sets (c = 16: 8 on G1, 4 on G2, instead of 16) - the same additions, 1 / 2
    // resp. 1 / 4 of the buckets to reduce.  Calls that name a window range or leave their window sums on the device (the
    // multi-GPU entry points) keep the plain form: their partition is defined over the plain windows.
    int E = 1, ibits = 31;
    const int endo_mode = msm_endo_mode<C>();
    const bool endo_wanted = endo_mode == 1 || (endo_mode == 2 && (flags & BZK_F_THROUGHPUT));
    if (endo_wanted && prep && prep->endo > 1 && !table && !wout && w_begin == 0 && w_end < 0 && !ctx->msm_no_endo && !wiv_off) {
        const int e = prep->endo, ib = 27 - (e == 4 ? 2 : 1), ws = (C::ENDO_BITS + c - 1) / c;
        if (e == C::ENDO && (uint64_t)n + m_max < ((uint64_t)1 << ib) && ws <= 16 && (uint64_t)ws * n * e < ((uint64_t)1 << 30)) {
            E = e;
            ibits = ib;
        }
    }
    const int w_total = E > 1 ? (C::ENDO_BITS + c - 1) / c : msm_windows_for(c);
    const bool folded = table && table->wpl > 1;  // bucket sets [w_begin, w_end) of a folded table, fed by every level
    const int levels = table ? table->levels : 1;
    if (w_end < 0) w_end = folded ? table->wpl : w_total;
    if (w_begin >= w_end) return BZK_OK;
    if (w_end > (folded ? table->wpl : w_total)) return BZK_E_ARG;
    if (table && (n > table->n || (uint64_t)levels * table->n >= ((uint64_t)1 << 31))) return BZK_E_ARG;
    const uint32_t half = 1u << (c - 1);
    uint32_t ch = ctx->msm_chunk_override > 0 ? (uint32_t)ctx->msm_chunk_override : 8u;
    if (ch > half) ch = half;
    while (half % ch) --ch;
    const uint32_t per_win = half / ch;
    // two-level bucket reduction (see msm_reduce_kernel): ctx->msm_reduce2 > 0 forces it, < 0 forbids it, 0 = the caller's hint
    // BZK_F_THROUGHPUT (set by bzk_groth16_prove for its overlapping MSMs)
    // level-2 chunk: 8 in the one-lane form; the quad form of G1 (msm_reduce_l2_quad_kernel) takes shorter chunks (env BZK_MSM_L2_CH,
    // BZK_MSM_QUAD_L2=0 for the one-lane form: A/B runs)
    static const bool quad_l2_on = [] { const char* e = getenv("BZK_MSM_QUAD_L2"); return !(e && atoi(e) == 0); }();
    static const uint32_t quad_l2_ch = [] {
        const char* e = getenv("BZK_MSM_L2_CH");
        const int v = e ? atoi(e) : 4;
        return (uint32_t)((v == 2 || v == 4 || v == 8 || v == 16) ? v : 4);
    }();
    const bool quad_l2 = quad_l2_on && !C::PARK_REDUCE;
    // G2 on pairs of lanes (msm_g2pair_tails.cuh): the reduction is two-level for EVERY call, with level-2 chunks of 4.  Measured (round 5, run 5,
    // 2^20 points, same box): one level 2.17 - 2.21 ms (65 536 pairs x 41 links: two rounds of waves at one wave per SIMD; the one-lane kernel
    // 1.89) against 0.41 + 0.77 ms in two levels with chunks of 8, where level 2 is 64 - 256 waves walking 41 links: shorter chunks put
    // more pairs on a shorter chain (8 + ~17 links).  env BZK_MSM_PAIR_L2_CH = 2 | 4 | 8 | 16, BZK_MSM_REDUCE2=-1 for the one-level form (A/B)
    static const uint32_t pair_l2_ch = [] {
        const char* e = getenv("BZK_MSM_PAIR_L2_CH");
        const int v = e ? atoi(e) : 4;
        return (uint32_t)((v == 2 || v == 4 || v == 8 || v == 16) ? v : 4);
    }();
    bool pair_tails = false;
    if constexpr (C::PAIR_ACC) pair_tails = msm_g2_pair_tails_on();
    const uint32_t ch2 = std::min<uint32_t>(pair_tails ? pair_l2_ch : quad_l2 ? quad_l2_ch : 8u, per_win);
    const bool two_level = per_win >= 64 && (ctx->msm_reduce2 > 0 || (ctx->msm_reduce2 == 0 && ((flags & BZK_F_THROUGHPUT) || pair_tails)));
    const uint32_t per_win_out = two_level ? per_win + per_win / ch2 : per_win;  // chunk results per window handed to the window sums
    // round 6: the multiplication-free reduction (section 6b: row / column sums, bit sums, the weights in the host's Horner) - G1.  env BZK_MSM_BITSUM=0: the old path (A/B)
    // (a caller that leaves its results on the device - a rank of a device group - receives the TERMS of its bucket sets instead of window sums: MsmWinOut)
    const bool bitsum = msm_bitsum_applies<C>(c) && !folded;
    const RowColPlan rc_plan = msm_rowcol_plan(c - 1, 256);
    const size_t rc_per_set = ((size_t)1 << rc_plan.lbits) + ((size_t)1 << rc_plan.hbits);
    if (wout) { wout->c = c; wout->w_total = w_total; wout->w_begin = w_begin; wout->w_end = w_end; wout->single = table && !folded; wout->terms_per_set = bitsum ? c / 2 + 1 : 0; }

    // windows are processed in groups so that one group's pair list stays below 2^30 entries
    int group = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(w_end - w_begin), ((uint64_t)1 << 30) / n));
    if (table || E > 1) group = w_end - w_begin;  // shared bucket sets: all requested windows in one pass
    const uint64_t len_max = (uint64_t)group * n * (folded ? (uint64_t)levels : (uint64_t)E);
    if (len_max >= ((uint64_t)1 << 31)) return BZK_E_ARG;
    const uint32_t nb_max = (table && !folded) ? half : (uint32_t)group * half;
    const uint32_t nb_alloc = std::max(nb_max, m_max);

    // rocPRIM temp sizes
    size_t tmp1 = 0, tmp2 = 0, tmp3 = 0, tmp4 = 0;
    {
        uint32_t* nul = nullptr;
        // both configurations are sized for (the A/B switch is read once per process, the workspace is grow-only)
        size_t q = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp1, nul, nul, nul, nul, (size_t)len_max, 0, bits_for(nb_max), ctx->stream);
        if (e != hipSuccess) { ctx->last_error = "rocprim size query"; return BZK_E_DEVICE; }
        e = rocprim::radix_sort_pairs_desc(nullptr, tmp2, nul, nul, nul, nul, (size_t)nb_alloc, 0, bits_for(len_max), ctx->stream);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs_desc<SortCfg32>(nullptr, q, nul, nul, nul, nul, (size_t)nb_alloc, 0, 16, ctx->stream);
        tmp2 = std::max(tmp2, q);
        if (e != hipSuccess) { ctx->last_error = "rocprim size query"; return BZK_E_DEVICE; }
        e = rocprim::exclusive_scan(nullptr, tmp3, nul, nul, 0u, (size_t)std::max<uint64_t>(nb_alloc, dedup ? n : 0), rocprim::plus<uint32_t>(),
                                    ctx->stream);
        if (e != hipSuccess) { ctx->last_error = "rocprim size query"; return BZK_E_DEVICE; }
        if (dedup) {
            e = rocprim::radix_sort_pairs(nullptr, tmp4, nul, nul, nul, nul, (size_t)n, 0, 32, ctx->stream);
            if (e != hipSuccess) { ctx->last_error = "rocprim size query"; return BZK_E_DEVICE; }
        }
    }
    const size_t tmp = std::max(std::max(tmp1, tmp2), std::max(tmp3, tmp4));
    // serial run length per lane: ~4x the mean bucket population, within [32, 256] - short enough that the
    // few over-full buckets of a degenerate top window do not set the kernel's critical path at small n
    uint32_t seg = (uint32_t)std::min<uint64_t>(MSM_SEG_MAX, std::max<uint64_t>(32, 4 * (n * E / half + 1)));
    if (table) seg = 64;  // shared buckets are all heavily populated: short runs keep every SIMD busy
    if (folded) seg = (uint32_t)std::min<uint64_t>(MSM_SEG_MAX, std::max<uint64_t>(32, 4 * (len_max / nb_max + 1)));
    // few, heavily populated buckets (a rank of a window-sharded MSM owns 2 windows of 2^23 points: 65 536 buckets of
    // 256 entries): one task per bucket would leave the machine under-filled and the kernel as long as its longest
    // run.  Cut the runs so that there are at least ~4 tasks per resident lane (131 072 lanes at 2 waves/SIMD).
    auto enough_tasks = [&](uint32_t sg, uint64_t len_, uint32_t nb_) {
        // endomorphism form: E x fewer, E x fuller buckets - two tasks per resident lane are enough (cutting every other bucket of a
        // 262 144-bucket call in two bought partial sums and folds, not balance: r04 run 3)
        // a window range of a split call (msm_run_split) runs beside the other ranges' kernels: it need not fill the device alone, and every cut costs a fold
        const uint64_t target = (E > 1 || ctx->is_part || ctx->split_active ? 2ull : 4ull) * 131072;
        if (nb_ >= target || len_ / sg + nb_ >= target) return sg;
        // run 34: two tasks per lane of FULL buckets need no cut - a rank of 2 (8 of 16 windows over 2^21 points: 262 144 buckets of 64 entries) ran 3.72 - 3.77 ms with
        // its fuller half of the buckets cut in two (and folded) against 3.45 - 3.51 ms uncut; thinner buckets (c = 15 calls: 32 entries) keep the cut
        if (nb_ >= 2ull * 131072 && len_ / nb_ >= 48) return sg;
        return (uint32_t)std::max<uint64_t>(32, len_ / (target - nb_));
    };
    if (!table || folded) seg = std::min(seg, enough_tasks(seg, len_max, nb_max));
    static const uint32_t seg_override = [] { const char* e = getenv("BZK_MSM_SEG"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 8 && v <= (int)MSM_SEG_MAX ? v : 0); }();
    if (seg_override) seg = seg_override;  // A/B runs
    const uint32_t seg_dd = 8;  // group sums are latency-bound (a 7 k-member group of bits is one bucket): short serial runs
    // capacity of the per-task partial sums: sized for the shortest run length any later adjustment can pick (32; 64 for tables)
    const uint64_t t_cap = std::max<uint64_t>((uint64_t)nb_max + len_max / std::min<uint32_t>(seg, 32u) + 1,
                                              dedup ? (uint64_t)m_max + n / seg_dd + 1 : 0);
    size_t total = 0;
    total += 4 * ws_pad(len_max * 4);                 // keys, vals, keys_sorted, vals_sorted
    total += 7 * ws_pad((size_t)nb_alloc * 4);        // start, count, count_sorted, iota, order, ntask, tbase
    total += ws_pad((size_t)t_cap * sizeof(Pt));      // per-task partial sums (multi-task buckets only)
    const size_t wide_cap = (size_t)t_cap / 32 + 2 * MSM_FOLD_WIDE_POS + 2;  // slot(i) + chunks(i) <= t / 64 + i + t / 64 + 1 (msm_fold_wide_slot)
    total += ws_pad(wide_cap * sizeof(Pt));
    total += ws_pad((size_t)nb_alloc * sizeof(Pt));   // buckets
    total += ws_pad((size_t)group * per_win_out * sizeof(Pt));
    if (two_level) total += ws_pad((size_t)group * per_win * sizeof(Pt));
    total += ws_pad(((size_t)group * (per_win_out / C::WSUM_THREADS + 1)) * sizeof(Pt));
    total += ws_pad((size_t)w_total * sizeof(StdPt));
    const size_t n_sets_max = (table && !folded) ? 1 : (size_t)group;
    const size_t n_terms = (size_t)c / 2 + 1;  // per bucket set: ceil((c - 1) / 2) digit terms + the plain sum (msm_bitsum_quad_kernel)
    if (bitsum) total += ws_pad(n_sets_max * rc_per_set * sizeof(Pt)) + ws_pad(n_sets_max * n_terms * sizeof(StdPt));
    if (C::CONVERT_BASES && !table) total += ws_pad((size_t)((prep ? 0 : n) + (size_t)m_max * E) * sizeof(typename C::DevAff));
    if (dedup) {
        total += 11 * ws_pad(n * 4) + ws_pad(n * 32) + ws_pad((size_t)m_max * sizeof(typename C::Fld));
    }
    total += ws_pad(tmp) + 8192;
    BZK_TRY(ws_reserve(ctx, total));
    BZK_TRY(pinned_reserve(ctx, std::max<size_t>((size_t)w_total, bitsum ? n_sets_max * n_terms : 0) * sizeof(StdPt) + 64));
    WsCursor cur(ctx->ws);
    uint32_t* keys = cur.take<uint32_t>(len_max);
    uint32_t* vals = cur.take<uint32_t>(len_max);
    uint32_t* keys_s = cur.take<uint32_t>(len_max);
    uint32_t* vals_s = cur.take<uint32_t>(len_max);
    BucketArrays<Pt> BA;
    BA.start = cur.take<uint32_t>(nb_alloc);
    BA.count = cur.take<uint32_t>(nb_alloc);
    BA.count_s = cur.take<uint32_t>(nb_alloc);
    BA.iota = cur.take<uint32_t>(nb_alloc);
    BA.order = cur.take<uint32_t>(nb_alloc);
    BA.ntask = cur.take<uint32_t>(nb_alloc);
    BA.tbase = cur.take<uint32_t>(nb_alloc);
    BA.partial = cur.take<Pt>(t_cap);
    BA.wide = cur.take<Pt>(wide_cap);
    Pt* buckets = cur.take<Pt>(nb_alloc);
    Pt* chunk_out = cur.take<Pt>((size_t)group * per_win_out);
    Pt* chunk_tot = two_level ? cur.take<Pt>((size_t)group * per_win) : nullptr;
    Pt* wpart = cur.take<Pt>((size_t)group * (per_win_out / C::WSUM_THREADS + 1));
    StdPt* win_out = cur.take<StdPt>(w_total);
    Pt* rc_buf = bitsum ? cur.take<Pt>(n_sets_max * rc_per_set) : nullptr;           // row sums of every set, then the column sums
    StdPt* terms_out = bitsum ? cur.take<StdPt>(n_sets_max * n_terms) : nullptr;
    const void* bases = table ? table->data : bases_raw;
    typename C::DevAff* conv = nullptr;
    typename C::DevAff* sums_aff = nullptr;  // de-duplication group sums in affine form: base indices n .. n + m_max
    AuxJoin aux(ctx);
    if (prep) {
        bases = prep->data;  // resident internal form: nothing to convert
        if (m_max) sums_aff = cur.take<typename C::DevAff>((size_t)m_max * E);  // group sums, then (endomorphism form) their images
    } else if (C::CONVERT_BASES && !table) {
        conv = cur.take<typename C::DevAff>(n + m_max);
        sums_aff = conv + n;
        // the conversion is not needed before the first accumulation: it runs on the side stream beside digits / sort
        if (msm_aux_ready(ctx) && hipEventRecord(ctx->ev_fork, ctx->stream) == hipSuccess &&
            hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0) == hipSuccess) {
            hipStream_t main_stream = ctx->stream;
            ctx->stream = ctx->aux;  // BZK_LAUNCH / ProfScope work on ctx->stream
            const int32_t st = msm_convert_launch<C>(ctx, bases_raw, n, conv);
            const hipError_t e = hipEventRecord(ctx->ev_join, ctx->aux);
            ctx->stream = main_stream;
            aux.pending = true;
            if (st != BZK_OK) return st;
            if (e != hipSuccess) { ctx->last_error = "aux join event"; return BZK_E_DEVICE; }
        } else {
            (void)hipGetLastError();
            BZK_TRY(msm_convert_launch<C>(ctx, bases_raw, n, conv));
        }
        bases = conv;
    }
    const uint32_t n_split = dedup ? (uint32_t)n : 0xffffffffu;
    uint32_t *hkey = nullptr, *hkey_s = nullptr;
    uint32_t *didx = nullptr, *didx_s = nullptr, *head = nullptr, *mhead = nullptr, *gid_ex = nullptr, *mid_ex = nullptr, *key2 = nullptr,
             *rep = nullptr, *gof = nullptr;
    U128* scal2 = nullptr;
    typename C::Fld* pref = nullptr;
    if (dedup) {
        hkey = cur.take<uint32_t>(n);
        hkey_s = cur.take<uint32_t>(n);
        didx = cur.take<uint32_t>(n);
        didx_s = cur.take<uint32_t>(n);
        head = cur.take<uint32_t>(n);
        mhead = cur.take<uint32_t>(n);
        gid_ex = cur.take<uint32_t>(n);
        mid_ex = cur.take<uint32_t>(n);
        key2 = cur.take<uint32_t>(n);
        rep = cur.take<uint32_t>(n);
        gof = cur.take<uint32_t>(n);
        scal2 = cur.take<U128>(2 * n);
        pref = cur.take<typename C::Fld>(m_max);
    }
    void* tmp_buf = cur.take<char>(tmp);

    uint64_t n_eff = n;
    const void* scal_eff = scalars;
    if (dedup) {
        const unsigned gb = (unsigned)((n + 255) / 256);
        BZK_LAUNCH(ctx, "dedup_hash", dedup_hash_kernel, dim3(gb), dim3(256), 0, (const U128*)scalars, n, hkey, didx);
        {
            ProfScope ps(ctx, "dedup_sort");
            size_t t = tmp;
            hipError_t e = rocprim::radix_sort_pairs(tmp_buf, t, hkey, hkey_s, didx, didx_s, (size_t)n, 0, 32, ctx->stream);
            if (e != hipSuccess) { ctx->last_error = std::string("dedup sort: ") + hipGetErrorString(e); return BZK_E_DEVICE; }
        }
        BZK_LAUNCH(ctx, "dedup_heads", dedup_heads_kernel, dim3(gb), dim3(256), 0, (const U128*)scalars, (const uint32_t*)hkey_s,
                   (const uint32_t*)didx_s, n, head, mhead);
        {
            ProfScope ps(ctx, "dedup_scan");
            size_t t = tmp;
            hipError_t e = rocprim::exclusive_scan(tmp_buf, t, head, gid_ex, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream);
            if (e == hipSuccess) {
                t = tmp;
                e = rocprim::exclusive_scan(tmp_buf, t, mhead, mid_ex, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream);
            }
            if (e != hipSuccess) { ctx->last_error = std::string("dedup scan: ") + hipGetErrorString(e); return BZK_E_DEVICE; }
        }
        // group counts to the host (launch sizes of what follows)
        uint32_t* hp = (uint32_t*)ctx->pinned;
        BZK_HIP(ctx, hipMemcpyAsync(hp + 0, gid_ex + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        BZK_HIP(ctx, hipMemcpyAsync(hp + 1, head + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        BZK_HIP(ctx, hipMemcpyAsync(hp + 2, mid_ex + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        BZK_HIP(ctx, hipMemcpyAsync(hp + 3, mhead + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t G = hp[0] + hp[1], M = hp[2] + hp[3];
        if (ctx->timing) fprintf(stderr, "[bzk] dedup: n %llu -> %u distinct non-zero scalars, %u groups of >= 2\n", (unsigned long long)n, G, M);
        if (G == 0) {  // every scalar is zero
            // every other exit of this function ends in a host-side wait for the stream; here the side stream's conversion may still be
            // reading the caller's bases and writing the workspace: wait for it before the caller may touch either
            aux.join();
            BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (wout) {  // the window sums (or terms) of an all-zero scalar vector: identities
                std::vector<StdPt> ids((size_t)(wout->single ? 1 : w_end - w_begin) * (size_t)std::max(1, wout->terms_per_set), xyzz_identity<F>());
                BZK_HIP(ctx, hipMemcpy(wout->d_win, ids.data(), ids.size() * sizeof(StdPt), hipMemcpyHostToDevice));
            }
            return BZK_OK;
        }
        if (M > m_max) { ctx->last_error = "dedup: group count out of range"; return BZK_E_INTERNAL; }
        BZK_LAUNCH(ctx, "dedup_assign", dedup_assign_kernel, dim3(gb), dim3(256), 0, (const U128*)scalars, (const uint32_t*)hkey_s,
                   (const uint32_t*)didx_s, (const uint32_t*)head, (const uint32_t*)mhead, (const uint32_t*)gid_ex, (const uint32_t*)mid_ex,
                   n, 0xffffffffu, key2, scal2, rep, gof);
        if (M) {
            aux.join();
            {
                HeavyScope hs(ctx);  // left (joined) before dedup_affine: the batched inversion is a latency chain, not a saturating grid
                BZK_TRY(bucket_accumulate<C>(ctx, bases, key2, didx_s, n, M, seg_dd, BA, buckets, tmp_buf, tmp, true, 0u, sums_aff, n_split, 31, 0, 0, &hs));
            }
            static const uint32_t K = [] {  // sums per lane of the batched inversion (env BZK_DEDUP_K for A/B runs)
                const char* e = getenv("BZK_DEDUP_K");
                const int v = e ? atoi(e) : 8;
                return (uint32_t)(v < 4 ? 4 : (v > 256 ? 256 : v));
            }();
            auto k_aff = dedup_affine_kernel<C>;
            BZK_LAUNCH(ctx, "dedup_affine", k_aff, dim3((unsigned)(((M + K - 1) / K + 63) / 64)), dim3(64), 0, (const Pt*)buckets, M, K, pref,
                       sums_aff, (const uint32_t*)gof, scal2);
            if (E > 1) {  // images X^m S of the group sums (an identity sum has a zeroed scalar: its slot is never gathered)
                auto k_img = msm_endo_images_kernel<C>;
                BZK_LAUNCH(ctx, "dedup_images", k_img, dim3((M + 127) / 128), dim3(128), 0, sums_aff, (uint64_t)M, (uint64_t)m_max);
            }
        }
        n_eff = G;
        scal_eff = scal2;
        seg = (uint32_t)std::min<uint64_t>(MSM_SEG_MAX, std::max<uint64_t>(32, 4 * (n_eff * E / half + 1)));
        seg = std::min(seg, enough_tasks(seg, (uint64_t)group * n_eff * E, nb_max));
        if (seg_override) seg = seg_override;
    }

    const int mont = (flags & BZK_F_CANONICAL) ? 0 : 1;
    const bool wiv = !wiv_off && !table && group <= 16 && (uint64_t)n + m_max < ((uint64_t)1 << 27);
    std::vector<StdPt> wsum((size_t)(w_end - w_begin) * (bitsum ? n_terms : 1));  // window sums, or (section 6b) the terms of every bucket set
    for (int wb = w_begin; wb < w_end; wb += group) {
        const int wc = std::min(group, w_end - wb);
        const uint64_t len = (uint64_t)wc * n_eff * (folded ? (uint64_t)levels : (uint64_t)E);
        const uint32_t nb = (table && !folded) ? half : (uint32_t)wc * half;
        const int n_red_win = (table && !folded) ? 1 : wc;  // bucket sets to reduce
        if (E > 1) {
            BZK_LAUNCH(ctx, "msm_digits_endo", (msm_digits_endo_kernel<C::ENDO>), dim3((unsigned)((n_eff + 255) / 256)), dim3(256), 0, (const U128*)scal_eff,
                       n_eff, mont, c, w_total, ibits, (const uint32_t*)(dedup ? rep : nullptr), keys, vals);
        } else {
            BZK_LAUNCH(ctx, "msm_digits", msm_digits_kernel, dim3((unsigned)((n_eff + 255) / 256)), dim3(256), 0, (const U128*)scal_eff, n_eff,
                       mont, c, folded ? levels * table->wpl : w_total, wb, wc, (uint32_t)(table ? table->n : 0), table ? table->wpl : 1,
                       (const uint32_t*)(dedup ? rep : nullptr), wiv ? 1 : 0, keys, vals);
        }
        {
            ProfScope ps(ctx, "msm_sort_pairs");
            size_t t = tmp;
            hipError_t e = rocprim::radix_sort_pairs(tmp_buf, t, keys, keys_s, vals, vals_s, (size_t)len, 0, bits_for(wiv ? half : nb), ctx->stream);
            if (e != hipSuccess) { ctx->last_error = std::string("radix_sort_pairs: ") + hipGetErrorString(e); return BZK_E_DEVICE; }
        }
        aux.join();
        HeavyScope heavy(ctx);  // accumulation, folds, bucket reduction; left before the window sums (latency-bound trees of a few workgroups)
        BZK_TRY(bucket_accumulate<C>(ctx, bases, keys_s, vals_s, len, nb, seg, BA, buckets, tmp_buf, tmp, false, wiv ? half : 0u, sums_aff, n_split,
                                     (uint32_t)ibits, E > 1 ? (uint32_t)prep->n : 0u, E > 1 ? m_max : 0u, &heavy));
        if constexpr (!C::PARK_REDUCE) {
            if (bitsum) {
                G1X28* const rows = (G1X28*)rc_buf;
                G1X28* const cols = rows + ((size_t)n_red_win << rc_plan.hbits);
                BZK_LAUNCH(ctx, "msm_rowcol", (msm_rowcol_quad_kernel<256>), dim3((unsigned)n_red_win * (rc_plan.wgs_r + rc_plan.wgs_c)), dim3(256), 0,
                           (const G1X28*)buckets, half, rc_plan, rows, cols);
                heavy.leave();
                // halves of 128 lanes: with 256-lane halves (112 KiB of static LDS - the runtime takes it) a workgroup saves a tree level and loses more to its
                // eight waves sharing a CU: 0.166 - 0.170 against 0.150 - 0.153 ms, same box, alternating (profiles/r06_run11_bitsum_wide_negative.txt)
                StdPt* const terms_dst = wout ? (StdPt*)wout->d_win + ((table && !folded) ? 0 : (size_t)(wb - w_begin) * n_terms) : terms_out;
                BZK_LAUNCH(ctx, "msm_bitsum", (msm_bitsum_quad_kernel<128>), dim3((unsigned)n_red_win * (unsigned)n_terms), dim3(256), 0, (const G1X28*)rows,
                           (const G1X28*)cols, rc_plan, (XyzzT<FpOps>*)terms_dst);
                if (wout) {  // the terms stay on the device, in stream order; the caller reads them back (or exchanges them) and combines
                    if (table && !folded) return BZK_OK;
                    continue;
                }
                BZK_HIP(ctx, hipMemcpyAsync(ctx->pinned, terms_out, (size_t)n_red_win * n_terms * sizeof(StdPt), hipMemcpyDeviceToHost, ctx->stream));
                BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (table && !folded) {  // one bucket set fed by every level of the table: its c terms are the whole result
                    result = msm_horner_terms_host<F>((const StdPt*)ctx->pinned, 1, c, 0);
                    return BZK_OK;
                }
                memcpy(&wsum[(size_t)(wb - w_begin) * n_terms], ctx->pinned, (size_t)wc * n_terms * sizeof(StdPt));
                continue;
            }
        }
        auto k_red = msm_reduce_kernel<C>;
        const uint32_t n_chunks = (uint32_t)n_red_win * per_win;
        bool tails_done = false;
        if constexpr (C::PAIR_ACC) {
            if (msm_g2_pair_tails_on()) {
                auto grid2 = [](uint32_t chunks) { return dim3((unsigned)((2ull * chunks + 63) / 64)); };
                if (two_level) {
                    int lg = 0;
                    while ((1u << lg) < ch) ++lg;
                    const uint32_t n_chunks2 = (uint32_t)n_red_win * (per_win / ch2);
                    BZK_LAUNCH(ctx, "msm_reduce", msm_reduce_g2pair_kernel<0>, grid2(n_chunks), dim3(64), 0, (const Pt*)buckets, half, ch, n_chunks, chunk_out,
                               per_win_out, 0u, chunk_tot, 0u);
                    BZK_LAUNCH(ctx, "msm_reduce_l2", msm_reduce_g2pair_kernel<0>, grid2(n_chunks2), dim3(64), 0, (const Pt*)chunk_tot, per_win, ch2, n_chunks2,
                               chunk_out, per_win_out, per_win, (Pt*)nullptr, (uint32_t)lg);
                } else {
                    BZK_LAUNCH(ctx, "msm_reduce", msm_reduce_g2pair_kernel<0>, grid2(n_chunks), dim3(64), 0, (const Pt*)buckets, half, ch, n_chunks, chunk_out, per_win,
                               0u, (Pt*)nullptr, 0u);
                }
                heavy.leave();
                constexpr int WTP = C::WSUM_THREADS;
                const uint32_t groups_p = (per_win_out + WTP - 1) / WTP;
                StdPt* const win_dst_p = wout ? (StdPt*)wout->d_win + ((table && !folded) ? 0 : wb - w_begin) : win_out;
                BZK_LAUNCH(ctx, "msm_window_partial", (msm_window_partial_g2pair_kernel<WTP>), dim3((unsigned)n_red_win * groups_p), dim3(WTP), 0,
                           (const Pt*)chunk_out, per_win_out, groups_p, wpart);
                BZK_LAUNCH(ctx, "msm_window_sum", (msm_window_sum_g2pair_kernel<WTP>), dim3((unsigned)n_red_win), dim3(WTP), 0, (const Pt*)wpart, groups_p,
                           win_dst_p);
                tails_done = true;
            }
        }
        if (tails_done) {
        } else if (two_level) {
            int lg = 0;
            while ((1u << lg) < ch) ++lg;
            const uint32_t n_chunks2 = (uint32_t)n_red_win * (per_win / ch2);
            BZK_LAUNCH(ctx, "msm_reduce", k_red, dim3((n_chunks + 63) / 64), dim3(64), 0, buckets, half, ch, n_chunks, chunk_out, per_win_out, 0u,
                       chunk_tot, 0u);
            bool l2_done = false;
            if constexpr (!C::PARK_REDUCE) {
                if (quad_l2) {
                    BZK_LAUNCH(ctx, "msm_reduce_l2", (msm_reduce_l2_quad_kernel<256>), dim3((n_chunks2 * 4 + 255) / 256), dim3(256), 0,
                               (const G1X28*)chunk_tot, per_win, ch2, n_chunks2, (G1X28*)chunk_out, per_win_out, per_win, (uint32_t)lg);
                    l2_done = true;
                }
            }
            if (!l2_done)
                BZK_LAUNCH(ctx, "msm_reduce_l2", k_red, dim3((n_chunks2 + 63) / 64), dim3(64), 0, (const Pt*)chunk_tot, per_win, ch2, n_chunks2,
                           chunk_out, per_win_out, per_win, (Pt*)nullptr, (uint32_t)lg);
        } else {
            BZK_LAUNCH(ctx, "msm_reduce", k_red, dim3((n_chunks + 63) / 64), dim3(64), 0, buckets, half, ch, n_chunks, chunk_out, per_win, 0u,
                       (Pt*)nullptr, 0u);
        }
        heavy.leave();
        constexpr int WT = C::WSUM_THREADS;
        const uint32_t groups = (per_win_out + WT - 1) / WT;
        auto k_wp = msm_window_partial_kernel<C, WT>;
        auto k_ws = msm_window_sum_kernel<C, WT>;
        StdPt* const win_dst = wout ? (StdPt*)wout->d_win + ((table && !folded) ? 0 : wb - w_begin) : win_out;
        // G1: the two trees on quads of lanes (7b); env BZK_MSM_QUAD_TREE=0: the one-lane-per-point form (A/B runs).  G2: on pairs (above) or one lane
        static const bool quad_tree = [] { const char* e = getenv("BZK_MSM_QUAD_TREE"); return !(e && atoi(e) == 0); }();
        if (tails_done) {  // the pair kernels above produced the window sums already
        } else if constexpr (!C::PARK_REDUCE) {
            if (quad_tree) {
                BZK_LAUNCH(ctx, "msm_window_partial", (msm_window_partial_quad_kernel<WT>), dim3((unsigned)n_red_win * groups), dim3(WT), 0,
                           (const G1X28*)chunk_out, per_win_out, groups, (G1X28*)wpart);
                BZK_LAUNCH(ctx, "msm_window_sum", (msm_window_sum_quad_kernel<WT>), dim3((unsigned)n_red_win), dim3(WT), 0, (const G1X28*)wpart, groups,
                           (XyzzT<FpOps>*)win_dst);
            } else {
                BZK_LAUNCH(ctx, "msm_window_partial", k_wp, dim3((unsigned)n_red_win * groups), dim3(WT), 0, chunk_out, per_win_out, groups, wpart);
                BZK_LAUNCH(ctx, "msm_window_sum", k_ws, dim3((unsigned)n_red_win), dim3(WT), 0, wpart, groups, win_dst);
            }
        } else {
            BZK_LAUNCH(ctx, "msm_window_partial", k_wp, dim3((unsigned)n_red_win * groups), dim3(WT), 0, chunk_out, per_win_out, groups, wpart);
            BZK_LAUNCH(ctx, "msm_window_sum", k_ws, dim3((unsigned)n_red_win), dim3(WT), 0, wpart, groups, win_dst);
        }
        if (wout) {
            if (table && !folded) return BZK_OK;
            continue;  // window sums stay on the device, in stream order; the caller reads them back and combines
        }
        BZK_HIP(ctx, hipMemcpyAsync(ctx->pinned, win_out, (size_t)n_red_win * sizeof(StdPt), hipMemcpyDeviceToHost, ctx->stream));
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (table && !folded) {  // the table already carries the 2^(c w) factors: the single bucket-set sum IS the result
            memcpy(&result, ctx->pinned, sizeof(StdPt));
            return BZK_OK;
        }
        memcpy(&wsum[wb - w_begin], ctx->pinned, (size_t)wc * sizeof(StdPt));
    }
    if (wout) return BZK_OK;
    // Horner over the window sums (host): result = 2^(c*w_begin) * sum_k 2^(c k) S_{w_begin+k}
    result = bitsum ? msm_horner_terms_host<F>(wsum.data(), w_end - w_begin, c, w_begin) : msm_horner_host<F>(wsum.data(), (int)wsum.size(), c, w_begin);
    return BZK_OK;
}

// ---- one stand-alone MSM as several window RANGES in flight (round 6, run 18) ------------------------------------------------------------------
// A stand-alone call is a chain: digits, sorts, boundaries (0.5 ms at 2^20 points, latency / HBM-bound small grids), the accumulation (2.3 ms, saturating),
// row / column sums + bit sums (0.5 ms, mostly idle lanes), read-back + host Horner (0.15 ms, device idle): a third of the call leaves the multiplier idle.
// Two INDEPENDENT calls in flight hide those phases under each other's accumulation (bench two_msms_in_flight: 1.15 - 1.18 x).  The same holds WITHIN a call:
// its windows are independent until the Horner.  So the call runs as P ranges of windows, highest first, each with its own stream + workspace (range 0 on the
// call's own context, the others on bzk_ctx::parts children): the front chain of range k + 1 runs beside the accumulation of range k, the reduction of range k
// beside the accumulation of range k + 1, and the host's Horner consumes the ranges in the order they finish (highest windows first - Horner's own order).
// Every range leaves the TERMS of its bucket sets (multiplication-free reduction, section 6b), so this form exists where that reduction does (G1, 11 <= c <= 21),
// for whole-MSM calls over a resident base set that are not flagged BZK_F_THROUGHPUT (those overlap with other calls already) or BZK_F_DEDUP.
// Same result bytes: the same window terms enter the same Horner.  env (read when a context is created) BZK_MSM_SPLIT = 1 (off) | 2 | 3 | 4,
// BZK_MSM_SPLIT_MIN_LOG (default 18), BZK_MSM_SPLIT_PRIO = 1: the children's streams at the highest priority (A/B).
template <class C>
static int msm_split_parts(const bzk_ctx* ctx, uint64_t n, uint32_t flags, const MsmBases* prep) {
    // measured (profiles/r06_run20..23): two ranges gain 13 - 16 % at 2^18 and 2^19 points, nothing (+- 1 %) from 2^20 up - there the ranges' accumulations
    // saturate the device either way and what overlaps is paid for in stretched neighbours -, and lose 15 % at 2^17: by default [2^18, 2^20) points
    const bool explicit_range = ctx->msm_split_min_log >= 10 && ctx->msm_split_min_log <= 30;
    const int min_log = explicit_range ? ctx->msm_split_min_log : 18;
    if (!explicit_range && ctx->msm_split <= 0 && ctx->msm_split_cuts[0] <= 0 && n >= ((uint64_t)1 << 20)) return 1;
    int parts = ctx->msm_split > 0 ? std::min(ctx->msm_split, 4) : BZK_MSM_SPLIT_DEFAULT;
    if (ctx->msm_split_cuts[0] > 0) {  // explicit window counts per range
        parts = 0;
        while (parts < 4 && ctx->msm_split_cuts[parts] > 0) ++parts;
    }
    // prep == nullptr: a call over raw bases (msm_entry_dev converts them once for all ranges)
    if (parts < 2 || C::PARK_REDUCE || ctx->is_part || (flags & (BZK_F_THROUGHPUT | BZK_F_DEDUP)) || n < ((uint64_t)1 << min_log) || (prep && n > prep->n)) return 1;
    if (prep && msm_endo_mode<C>() == 1 && prep->endo > 1 && !ctx->msm_no_endo) return 1;  // that call takes the endomorphism form: shared bucket sets, no window ranges
    const int c = ctx->msm_c_override >= 2 && ctx->msm_c_override <= 20 ? ctx->msm_c_override : msm_pick_c(n);
    if (!msm_bitsum_applies<C>(c)) return 1;
    const int W = msm_windows_for(c);
    if ((uint64_t)W * n >= ((uint64_t)1 << 30)) return 1;  // the ranges of such a call are window groups already (msm_run)
    return std::min(parts, W / 2);
}
// continues a Horner over terms (msm_horner_terms_host without its trailing doublings): acc <- acc 2^(c count) + sum_k 2^(c k) (terms of set k)
template <class F>
static void msm_horner_terms_continue(XyzzT<typename HostFast<F>::Ops>& acc, const XyzzT<F>* T, int count, int c) {
    typedef typename HostFast<F>::Ops H;
    const int n_pairs = c / 2, n_terms = n_pairs + 1;
    for (int k = count - 1; k >= 0; --k) {
        const XyzzT<F>* t = T + (size_t)k * n_terms;
        for (int bit = c - 1; bit >= 0; --bit) {
            acc = xyzz_dbl<H>(acc);
            if ((bit & 1) == 0 && bit / 2 < n_pairs) xyzz_add<H>(acc, to_host_fast<F>(t[bit / 2]));
            if (bit == 0) xyzz_add<H>(acc, to_host_fast<F>(t[n_pairs]));
        }
    }
}
template <class C>
static int32_t msm_run_split(bzk_ctx* ctx, const void* scalars, uint64_t n, uint32_t flags, XyzzT<typename C::HostF>& result, const MsmBases* prep, int parts) {
    typedef typename C::HostF F;
    typedef XyzzT<F> StdPt;
    typedef typename HostFast<F>::Ops H;
    // priority mode (fixed when the first child is created): 0 = every range at the call's priority; 1 = the later ranges' streams at the highest priority;
    // 2 = as 1, and their SATURATING kernels (accumulation, folds, row / column sums: HeavyScope) at the lowest - the front chain of range k + 1 outranks the
    // accumulation of range k, which outranks the accumulation of range k + 1, which the tails of range k outrank: the ranges finish in issue order
    const int prio_mode = ctx->msm_split_prio >= 0 ? ctx->msm_split_prio : BZK_MSM_SPLIT_PRIO_DEFAULT;
    const bool kids_high = prio_mode != 0;
    const int c = ctx->msm_c_override >= 2 && ctx->msm_c_override <= 20 ? ctx->msm_c_override : msm_pick_c(n);
    const int W = msm_windows_for(c), n_terms = msm_terms_per_set<C>(c);
    bzk_ctx* pc[4] = {ctx, nullptr, nullptr, nullptr};
    for (int p = 1; p < parts; ++p) {
        if (!(pc[p] = ctx_part(ctx, (size_t)p - 1, kids_high))) { ctx->last_error = "msm split: child context"; return BZK_E_DEVICE; }
        pc[p]->heavy_force = prio_mode == 2;
    }
    const size_t bytes_all = (size_t)W * n_terms * sizeof(StdPt);
    if (ctx->split_terms_bytes < bytes_all) {
        if (ctx->split_terms) {
            BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            BZK_HIP(ctx, hipFree(ctx->split_terms));
            ctx->split_terms = nullptr;
            ctx->split_terms_bytes = 0;
        }
        if (hipMalloc(&ctx->split_terms, bytes_all) != hipSuccess) { (void)hipGetLastError(); ctx->last_error = "msm split: terms buffer"; return BZK_E_ALLOC; }
        ctx->split_terms_bytes = bytes_all;
    }
    if (!ctx->split_ev) BZK_HIP(ctx, hipEventCreateWithFlags(&ctx->split_ev, hipEventDisableTiming));
    BZK_TRY(pinned_reserve(ctx, bytes_all + 64));  // msm_run asks for no more than this on the call's own context: the staging is not re-allocated under a copy
    // range p = windows [lo[p], lo[p - 1]), highest windows first
    int lo[5];
    lo[0] = W;
    for (int p = 1; p <= parts; ++p) lo[p] = (int)((int64_t)W * (parts - p) / parts);
    {
        int sum = 0, k = 0;
        while (k < 4 && ctx->msm_split_cuts[k] > 0) sum += ctx->msm_split_cuts[k++];
        if (k == parts && sum == W)
            for (int p = 1; p <= parts; ++p) lo[p] = lo[p - 1] - ctx->msm_split_cuts[p - 1];
    }
    // the children start behind whatever the caller's stream holds at this point (the scalars may be its work)
    BZK_HIP(ctx, hipEventRecord(ctx->split_ev, ctx->stream));
    for (int p = 1; p < parts; ++p) BZK_HIP(ctx, hipStreamWaitEvent(pc[p]->stream, ctx->split_ev, 0));
    int32_t st = BZK_OK;
    int issued = 0;
    ctx->split_active = true;  // range 0 runs on the call's own context: its msm_run sees the same task-count target as the children's
    struct Off { bzk_ctx* c; ~Off() { c->split_active = false; } } off{ctx};
    for (int p = 0; p < parts && st == BZK_OK; ++p) {
        const int wb = lo[p + 1], we = lo[p];
        MsmWinOut wo;
        wo.d_win = (StdPt*)ctx->split_terms + (size_t)wb * n_terms;
        XyzzT<F> unused;
        st = msm_run<C>(pc[p], nullptr, scalars, n, flags, wb, we, unused, nullptr, prep, &wo);
        issued = p + 1;
        if (st == BZK_OK && (wo.terms_per_set != n_terms || wo.c != c || wo.w_total != W)) { st = BZK_E_INTERNAL; pc[p]->last_error = "msm split: window plan differs"; }
        if (st == BZK_OK && hipMemcpyAsync((StdPt*)ctx->pinned + (size_t)wb * n_terms, wo.d_win, (size_t)(we - wb) * n_terms * sizeof(StdPt), hipMemcpyDeviceToHost,
                                           pc[p]->stream) != hipSuccess) {
            (void)hipGetLastError();
            st = BZK_E_DEVICE;
            pc[p]->last_error = "msm split: read-back";
        }
        if (st != BZK_OK && pc[p] != ctx) ctx->last_error = pc[p]->last_error;
    }
    // the host consumes the ranges in issue order (= Horner's order); after a failure it still waits for everything issued - the ranges read the caller's
    // scalars and write the shared staging
    XyzzT<H> acc = xyzz_identity<H>();
    for (int p = 0; p < issued; ++p) {
        if (hipStreamSynchronize(pc[p]->stream) != hipSuccess) {
            (void)hipGetLastError();
            if (st == BZK_OK) { st = BZK_E_DEVICE; ctx->last_error = "msm split: range failed on the device"; }
        }
        if (st == BZK_OK) msm_horner_terms_continue<F>(acc, (const StdPt*)ctx->pinned + (size_t)lo[p + 1] * n_terms, lo[p] - lo[p + 1], c);
    }
    if (st != BZK_OK) return st;
    result = from_host_fast<F>(acc);
    return BZK_OK;
}

template <class C>
static int32_t msm_entry_dev(bzk_ctx* ctx, const void* bases, const void* scalars, uint64_t n, uint32_t flags, int w_begin,
                             int w_end, uint8_t* out) {
    typedef typename C::HostF F;
    if (!ctx || !out || (n && (!bases || !scalars))) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    XyzzT<F> r;
    const int parts = (w_begin == 0 && w_end < 0 && n) ? msm_split_parts<C>(ctx, n, flags, nullptr) : 1;
    if (parts > 1) {
        // window ranges in flight (msm_run_split) over raw bases: converted ONCE, on the call's stream, into a buffer of the context's that every range gathers from
        const size_t need = (size_t)n * sizeof(typename C::DevAff) + MSM_GATHER_PAD;
        if (ctx->split_conv_bytes < need) {
            if (ctx->split_conv) {
                BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                BZK_HIP(ctx, hipFree(ctx->split_conv));
                ctx->split_conv = nullptr;
                ctx->split_conv_bytes = 0;
            }
            if (hipMalloc(&ctx->split_conv, need + (need >> 3)) != hipSuccess) { (void)hipGetLastError(); ctx->last_error = "msm split: base buffer"; return BZK_E_ALLOC; }
            ctx->split_conv_bytes = need + (need >> 3);
        }
        BZK_TRY(msm_convert_launch<C>(ctx, bases, n, (typename C::DevAff*)ctx->split_conv));
        MsmBases tmp;
        tmp.data = ctx->split_conv;
        tmp.n = n;
        tmp.device = ctx->device;
        BZK_TRY(msm_run_split<C>(ctx, scalars, n, flags, r, &tmp, parts));
    } else {
        BZK_TRY(msm_run<C>(ctx, bases, scalars, n, flags, w_begin, w_end, r));  // w_end < 0: all windows of the c msm_run picks
    }
    PointIO<F>::pack(r, out);
    return BZK_OK;
}

template <class C>
static int32_t msm_entry_host(bzk_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t* out) {
    if (!ctx || !out || (n && (!bases || !scalars))) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    void *db = nullptr, *ds = nullptr;
    if (n) {
        BZK_HIP(ctx, hipMalloc(&db, n * C::RAW));
        if (hipMalloc(&ds, n * 32) != hipSuccess) { (void)hipFree(db); return BZK_E_ALLOC; }
        hipError_t e1 = hipMemcpyAsync(db, bases, n * C::RAW, hipMemcpyHostToDevice, ctx->stream);
        hipError_t e2 = hipMemcpyAsync(ds, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream);
        if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(db); (void)hipFree(ds); return BZK_E_DEVICE; }
    }
    int32_t st = msm_entry_dev<C>(ctx, db, ds, n, flags, 0, -1, out);
    (void)hipStreamSynchronize(ctx->stream);
    if (db) (void)hipFree(db);
    if (ds) (void)hipFree(ds);
    return st;
}

template <class C>
static int32_t msm_table_build(bzk_ctx* ctx, const void* bases_raw, uint64_t n, MsmTable** out, int levels_req = 0, int c_req = 0) {
    if (!ctx || !out || !bases_raw || n == 0 || levels_req < 0) return BZK_E_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    int c = msm_pick_c(n);
    if (const char* e = getenv("BZK_MSM_TABLE_C")) {
        int v = atoi(e);
        if (v >= 4 && v <= 20) c = v;
    }
    if (c_req >= 4 && c_req <= 22) c = c_req;  // explicit window size (a full table affords c ~ log2 n: the buckets are shared by all windows)
    const int w_total = msm_windows_for(c);
    // levels_req = 0 (or >= W): full table, one level per window; otherwise windows per level = ceil(W / levels_req)
    // and only as many levels as that leaves non-empty
    const int wpl = (levels_req == 0 || levels_req >= w_total) ? 1 : (w_total + levels_req - 1) / levels_req;
    const int levels = (w_total + wpl - 1) / wpl;
    if ((uint64_t)levels * n >= ((uint64_t)1 << 31)) return BZK_E_ARG;
    MsmTable* t = new (std::nothrow) MsmTable();
    if (!t) return BZK_E_ALLOC;
    t->n = n; t->c = c; t->w_total = w_total; t->levels = levels; t->wpl = wpl;
    hipError_t e = hipMalloc(&t->data, (size_t)levels * n * sizeof(typename C::DevAff) + MSM_GATHER_PAD);
    if (e != hipSuccess) {
        ctx->last_error = std::string("table alloc: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        delete t;
        return BZK_E_ALLOC;
    }
    auto k = msm_table_build_kernel<C>;
    BZK_LAUNCH(ctx, "msm_table_build", k, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, bases_raw, n, c * wpl, levels, (typename C::DevAff*)t->data);
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = t;
    return BZK_OK;
}

static void msm_table_free(bzk_ctx* ctx, MsmTable* t) {
    if (!t) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (t->data) (void)hipFree(t->data);
    delete t;
}

template <class C>
static int32_t msm_table_entry(bzk_ctx* ctx, const MsmTable* t, const void* scalars, uint64_t n, uint32_t flags, int w_begin,
                               int w_end, uint8_t* out) {
    typedef typename C::HostF F;
    if (!ctx || !t || !out || (n && !scalars)) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    XyzzT<F> r;
    BZK_TRY(msm_run<C>(ctx, nullptr, scalars, n, flags, w_begin, w_end, r, t));
    PointIO<F>::pack(r, out);
    return BZK_OK;
}

// ---- resident base sets (MsmBases) -------------------------------------------------------------------------------------------
template <class C>
static int32_t msm_bases_load(bzk_ctx* ctx, const void* bases_raw, uint64_t n, MsmBases** out) {
    if (!ctx || !out || !bases_raw || n == 0 || n >= ((uint64_t)1 << 31)) return BZK_E_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    MsmBases* b = new (std::nothrow) MsmBases();
    if (!b) return BZK_E_ALLOC;
    b->n = n;
    b->device = ctx->device;
    // with its endomorphism images (E x the memory) unless no call could ever use them - the curve's policy is 0, or the context opts
    // out (BZK_MSM_NO_ENDO, device groups) - or the device has no room for them beside a reserve.  bzk_msm_bases_info reports the outcome
    b->endo = (ctx->msm_no_endo || msm_endo_mode<C>() == 0) ? 1 : C::ENDO;
    if (b->endo > 1) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        if ((size_t)b->endo * n * sizeof(typename C::DevAff) + ((size_t)8 << 30) > free_b) b->endo = 1;
    }
    hipError_t e = hipMalloc(&b->data, (size_t)b->endo * n * sizeof(typename C::DevAff) + MSM_GATHER_PAD);
    if (e != hipSuccess && b->endo > 1) {
        (void)hipGetLastError();
        b->endo = 1;
        e = hipMalloc(&b->data, (size_t)n * sizeof(typename C::DevAff) + MSM_GATHER_PAD);
    }
    if (e != hipSuccess) {
        ctx->last_error = std::string("bases alloc: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        delete b;
        return BZK_E_ALLOC;
    }
    int32_t st = msm_convert_launch<C>(ctx, bases_raw, n, (typename C::DevAff*)b->data);
    if (st == BZK_OK && b->endo > 1) {
        auto k_img = msm_endo_images_kernel<C>;
        st = [&]() -> int32_t {
            BZK_LAUNCH(ctx, "msm_endo_images", k_img, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (typename C::DevAff*)b->data, n, n);
            return BZK_OK;
        }();
    }
    if (st == BZK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
    if (st != BZK_OK) {
        (void)hipFree(b->data);
        delete b;
        return st;
    }
    b->bytes = (size_t)b->endo * n * sizeof(typename C::DevAff);
    *out = b;
    return BZK_OK;
}
static void msm_bases_free(bzk_ctx* ctx, MsmBases* b) {
    if (!b) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (b->data) (void)hipFree(b->data);
    delete b;
}
template <class C>
static int32_t msm_bases_entry(bzk_ctx* ctx, const MsmBases* b, const void* scalars, uint64_t n, uint32_t flags, int w_begin, int w_end,
                               uint8_t* out) {
    typedef typename C::HostF F;
    if (!ctx || !b || !out || (n && !scalars)) return BZK_E_ARG;
    if (b->device != ctx->device) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    XyzzT<F> r;
    const int parts = (w_begin == 0 && w_end < 0) ? msm_split_parts<C>(ctx, n, flags, b) : 1;
    if (parts > 1) {
        BZK_TRY(msm_run_split<C>(ctx, scalars, n, flags, r, b, parts));
    } else {
        BZK_TRY(msm_run<C>(ctx, nullptr, scalars, n, flags, w_begin, w_end, r, nullptr, b));
    }
    PointIO<F>::pack(r, out);
    return BZK_OK;
}
// multi-GPU hook (mg.hip): windows [w_begin, w_end) of the MSM over a resident base set (or raw bases), window sums left in
// DEVICE memory at d_win (standard-limb XYZZ, sizeof = 4 field elements), nothing read back.  w_end < 0: every window.
template <class C>
static int32_t msm_windows_dev(bzk_ctx* ctx, const MsmBases* b, const void* bases_raw, const void* scalars, uint64_t n, uint32_t flags,
                               int w_begin, int w_end, void* d_win, int32_t info[5]) {
    typedef typename C::HostF F;
    if (!ctx || (!b && !bases_raw) || !d_win || (n && !scalars)) return BZK_E_ARG;
    if (b && b->device != ctx->device) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    XyzzT<F> r;
    MsmWinOut wo;
    wo.d_win = d_win;
    BZK_TRY(msm_run<C>(ctx, bases_raw, scalars, n, flags, w_begin, w_end, r, nullptr, b, &wo));
    if (info) { info[0] = wo.c; info[1] = wo.w_total; info[2] = wo.w_begin; info[3] = wo.w_end; info[4] = wo.terms_per_set; }
    return BZK_OK;
}
template <class F>
static int32_t horner_packed(const void* S, int count, int c, int w0, uint8_t* out) {
    if (!out || count < 0 || (count && !S)) return BZK_E_ARG;
    XyzzT<F> r = msm_horner_host<F>((const XyzzT<F>*)S, count, c, w0);
    PointIO<F>::pack(r, out);
    return BZK_OK;
}

template <class F>
static int32_t horner_terms_packed(const void* T, int count, int c, int w0, uint8_t* out) {
    if (!out || count < 0 || (count && !T)) return BZK_E_ARG;
    XyzzT<F> r = msm_horner_terms_host<F>((const XyzzT<F>*)T, count, c, w0);
    PointIO<F>::pack(r, out);
    return BZK_OK;
}

template <class F>
static int32_t sum_packed(const uint8_t* pts, uint32_t count, uint8_t* out) {
    if (!out || (count && !pts)) return BZK_E_ARG;
    XyzzT<F> acc = xyzz_identity<F>();
    for (uint32_t i = 0; i < count; ++i) {
        XyzzT<F> p = PointIO<F>::unpack(pts + (size_t)i * PointIO<F>::PACKED);
        xyzz_add<F>(acc, p);
    }
    PointIO<F>::pack(acc, out);
    return BZK_OK;
}

}  // namespace bzk

