// The latency-bound G2 tail kernels on PAIRS of lanes (round 5; arithmetic: bzk_g2pair.cuh): folds of multi-task buckets, the chunked
// bucket reduction, the window-sum trees.  The one-lane forms hold two 112-register points plus an addition's temporaries - they ran at 512
// registers with 94 / 33 / 31 spilled ones (profiles/r04_kernel_resources.txt) and every link of their chains was ~43 us (42 dependent field
// products on one lane).  Here a lane holds half of each point (56 registers), a link is 21 product-times, and the additions are two
// no-inline bodies per code object (the G1 quad kernels' lesson: many inlined copies of a 10 k-instruction body run at instruction-cache
// speed) whose operands travel through private memory - ~2 k cycles of traffic beside a ~60 k-cycle addition.
//
// Memory layout unchanged: G2X28 = X.c0 X.c1 Y.c0 Y.c1 ZZ.c0 ZZ.c1 ZZZ.c0 ZZZ.c1 (8 x 56 B); lane parity picks the component.  Every point
// these kernels STORE satisfies both disciplines (X, Y normalised and < 3 p; ZZ / ZZZ product outputs), so one-lane consumers (dedup_affine,
// to_std) read them unchanged.  They READ points written by the pair accumulation or by themselves only (msm_impl.cuh dispatches the
// pair tails together with the pair accumulation).
#pragma once
#include "bzk_g2pair.cuh"

namespace bzk {

struct alignas(8) G2pU128 {
    uint32_t x, y, z, w;
};
struct alignas(8) G2pU64 {
    uint32_t x, y;
};
__device__ __forceinline__ Fp28 g2p_ld56(const char* p) {  // 14 limbs at an 8-byte aligned address: 3 x 16 B + 8 B
    const G2pU128 a = *(const G2pU128*)p, b = *(const G2pU128*)(p + 16), c = *(const G2pU128*)(p + 32);
    const G2pU64 d = *(const G2pU64*)(p + 48);
    Fp28 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void g2p_st56(char* p, const Fp28& v) {
    *(G2pU128*)p = G2pU128{v.l[0], v.l[1], v.l[2], v.l[3]};
    *(G2pU128*)(p + 16) = G2pU128{v.l[4], v.l[5], v.l[6], v.l[7]};
    *(G2pU128*)(p + 32) = G2pU128{v.l[8], v.l[9], v.l[10], v.l[11]};
    *(G2pU64*)(p + 48) = G2pU64{v.l[12], v.l[13]};
}
// this lane's half of a stored point (comp = 0 | 56: byte offset of the component inside an Fp2 value)
__device__ __forceinline__ g2p::Pt g2p_ld_pt(const G2X28* p, uint32_t comp) {
    const char* b = (const char*)p + comp;
    return {g2p_ld56(b), g2p_ld56(b + 112), g2p_ld56(b + 224), g2p_ld56(b + 336)};
}
__device__ __forceinline__ void g2p_st_pt(G2X28* p, uint32_t comp, const g2p::Pt& v) {
    char* b = (char*)p + comp;
    g2p_st56(b, fp28::reduce(v.X));  // < 3 p: the discipline every consumer accepts (an identity stays an identity: ZZ decides)
    g2p_st56(b + 112, v.Y);
    g2p_st56(b + 224, v.ZZ);
    g2p_st56(b + 336, v.ZZZ);
}
// the two bodies (one copy per code object).  The addition CALLS the doubling in its P + P branch (see g2p::add: with the doubling inlined the
// body outgrows a conditional branch's reach and the compiler's long-branch expansion destroyed the return address)
static __device__ __noinline__ void g2p_dbl_ni(g2p::Pt* p) { *p = g2p::dbl(*p); }
struct G2pDblCall {
    __device__ __forceinline__ void operator()(g2p::Pt& p) const { g2p_dbl_ni(&p); }
};
static __device__ __noinline__ void g2p_add_ni(g2p::Pt* acc, const g2p::Pt* q) {
    g2p::Pt a = *acc;
    g2p::add(a, *q, G2pDblCall());
    *acc = a;
}

// ---- multi-task buckets with few tasks: one PAIR per bucket (sorted position), serial fold.  thr: see msm_fold_threshold
template <int UNIT = 0>  // a template so that only the G2 translation unit instantiates it
__global__ void __launch_bounds__(64) msm_fold_small_g2pair_kernel(const uint32_t* __restrict__ count_sorted, const uint32_t* __restrict__ order,
                                                                   const uint32_t* __restrict__ tbase, const uint32_t* __restrict__ ntask, uint32_t nb,
                                                                   uint32_t n_pos, uint32_t seg, uint32_t bulk_from, uint32_t thr_small, uint32_t thr_bulk,
                                                                   const G2X28* __restrict__ partial, G2X28* __restrict__ buckets) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gt >> 1, comp = (gt & 1u) * 56u;
    if (i >= n_pos) return;
    const uint32_t cnt = count_sorted[i];
    if (cnt <= seg) return;
    const uint32_t nt = (cnt + seg - 1) / seg;
    const uint32_t extra = tbase[nb - 1] + ntask[nb - 1] - nb;
    if (nt > (extra >= bulk_from ? thr_bulk : thr_small)) return;
    const G2X28* src = partial + tbase[i];
    g2p::Pt acc = g2p_ld_pt(&src[0], comp);
    for (uint32_t j = 1; j < nt; ++j) {
        const g2p::Pt q = g2p_ld_pt(&src[j], comp);
        g2p_add_ni(&acc, &q);
    }
    g2p_st_pt(&buckets[order[i]], comp, acc);
}

// ---- GIANT buckets (round 6, run 27; msm_impl.cuh msm_fold_wide_kernel is the one-lane twin): a bucket of more than `wide_from` tasks - the carry-only top
// window of a recoding narrower than 16 bits holds 45 % of all uniform scalars in ONE bucket - is cut into chunks of 64 partial sums; any workgroup of a fixed
// grid folds a chunk (one load + a 6-level tree of 64 pairs) into `wide`, the chunks of all giants among the first `wide_pos` sorted positions dealt round-robin;
// the workgroup fold below then meets nt / 64 chunk sums.  At 2^19 points the one-level fold was 2.66 ms of a 9.5 ms G2 call (90 pair additions in a row per pair).
__device__ __forceinline__ uint32_t g2p_fold_wide_slot(const uint32_t* __restrict__ tbase, uint32_t i) { return tbase[i] / 64 + i; }
template <int UNIT = 0>
__global__ void __launch_bounds__(128) msm_fold_wide_g2pair_kernel(const uint32_t* __restrict__ count_sorted, const uint32_t* __restrict__ tbase, uint32_t nb,
                                                                   uint32_t seg, uint32_t wide_from, uint32_t wide_pos, const G2X28* __restrict__ partial,
                                                                   G2X28* __restrict__ wide) {
    __shared__ G2X28 sh[64];
    const uint32_t pr = threadIdx.x >> 1, comp = (threadIdx.x & 1u) * 56u;
    const uint32_t n_pos = nb < wide_pos ? nb : wide_pos;
    uint32_t before = 0;
    for (uint32_t i = 0; i < n_pos; ++i) {
        const uint32_t cnt = count_sorted[i];
        const uint32_t nt = cnt <= seg ? 1u : (cnt + seg - 1) / seg;
        if (nt <= wide_from) continue;  // workgroup-uniform
        const uint32_t chunks = (nt + 63) / 64;
        const G2X28* src = partial + tbase[i];
        G2X28* dst = wide + g2p_fold_wide_slot(tbase, i);
        const uint32_t first = (blockIdx.x + gridDim.x - before % gridDim.x) % gridDim.x;
        before += chunks;
        for (uint32_t ch = first; ch < chunks; ch += gridDim.x) {
            const uint32_t j = ch * 64 + pr;
            g2p_st_pt(&sh[pr], comp, j < nt ? g2p_ld_pt(&src[j], comp) : g2p::identity());
            __syncthreads();
            for (uint32_t s = 32; s > 0; s >>= 1) {
                if (pr < s) {
                    g2p::Pt a = g2p_ld_pt(&sh[pr], comp);
                    const g2p::Pt q = g2p_ld_pt(&sh[pr + s], comp);
                    g2p_add_ni(&a, &q);
                    g2p_st_pt(&sh[pr], comp, a);
                }
                __syncthreads();
            }
            if (pr == 0) g2p_st_pt(&dst[ch], comp, g2p_ld_pt(&sh[0], comp));
            __syncthreads();  // sh[] is free again
        }
    }
}

// ---- heavily populated buckets: one workgroup of 64 PAIRS per bucket: pairs stride over the partial sums (a giant's chunk sums: above), then a tree through LDS
template <int UNIT = 0>
__global__ void __launch_bounds__(128) msm_fold_g2pair_kernel(const uint32_t* __restrict__ count_sorted, const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ tbase, const uint32_t* __restrict__ ntask, uint32_t nb,
                                                              uint32_t seg, uint32_t bulk_from, uint32_t thr_small, uint32_t thr_bulk,
                                                              const G2X28* __restrict__ partial, G2X28* __restrict__ buckets,
                                                              const G2X28* __restrict__ wide, uint32_t wide_from, uint32_t wide_pos) {
    __shared__ G2X28 sh[64];
    const uint32_t i = blockIdx.x;
    if (i >= nb) return;
    const uint32_t cnt = count_sorted[i];
    if (cnt <= seg) return;  // uniform: the whole workgroup leaves
    uint32_t nt = (cnt + seg - 1) / seg;
    const uint32_t extra = tbase[nb - 1] + ntask[nb - 1] - nb;
    if (nt <= (extra >= bulk_from ? thr_bulk : thr_small)) return;
    const uint32_t pr = threadIdx.x >> 1, comp = (threadIdx.x & 1u) * 56u;
    const G2X28* src = partial + tbase[i];
    if (wide && i < wide_pos && nt > wide_from) {  // msm_fold_wide_g2pair_kernel left one sum per chunk of 64
        src = wide + g2p_fold_wide_slot(tbase, i);
        nt = (nt + 63) / 64;
    }
    g2p::Pt acc = pr < nt ? g2p_ld_pt(&src[pr], comp) : g2p::identity();
    for (uint32_t j = pr + 64; j < nt; j += 64) {
        const g2p::Pt q = g2p_ld_pt(&src[j], comp);
        g2p_add_ni(&acc, &q);
    }
    g2p_st_pt(&sh[pr], comp, acc);
    __syncthreads();
    for (uint32_t s = 32; s > 0; s >>= 1) {
        if (pr < s) {
            g2p::Pt a = g2p_ld_pt(&sh[pr], comp);
            const g2p::Pt q = g2p_ld_pt(&sh[pr + s], comp);
            g2p_add_ni(&a, &q);
            g2p_st_pt(&sh[pr], comp, a);
        }
        __syncthreads();
    }
    if (pr == 0) g2p_st_pt(&buckets[order[i]], comp, g2p_ld_pt(&sh[0], comp));
}

// ---- chunked running-sum reduction (msm_reduce_kernel's three forms: one level with the chunk offset multiplied in, level 1 of the
// two-level form (tot != nullptr), level 2 (post_dbl)); one PAIR per chunk
template <int UNIT = 0>  // a template so that only the G2 translation unit instantiates it
__global__ void __launch_bounds__(64) msm_reduce_g2pair_kernel(const G2X28* __restrict__ buckets, uint32_t half, uint32_t ch, uint32_t n_chunks_total,
                                                               G2X28* __restrict__ out, uint32_t out_stride, uint32_t out_off, G2X28* __restrict__ tot,
                                                               uint32_t post_dbl) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gt >> 1, comp = (gt & 1u) * 56u;
    if (t >= n_chunks_total) return;
    const uint32_t per_win = half / ch;
    const uint32_t w = t / per_win, k = t % per_win;
    const uint32_t lo = k * ch;
    const G2X28* b = buckets + (size_t)w * half + lo;
    const size_t o = (size_t)w * out_stride + out_off + k;
    // the first step loads instead of adding to an identity
    g2p::Pt run = g2p_ld_pt(&b[ch - 1], comp);
    g2p::Pt acc = run;
#pragma nounroll
    for (int j = (int)ch - 2; j >= 0; --j) {
        const g2p::Pt p = g2p_ld_pt(&b[j], comp);
        g2p_add_ni(&run, &p);
        g2p_add_ni(&acc, &run);
    }
    if (tot) {
        g2p_st_pt(&out[o], comp, acc);
        g2p_st_pt(&tot[(size_t)w * per_win + (k ? k - 1 : per_win - 1)], comp, k ? run : g2p::identity());
        return;
    }
    if (lo) {  // acc += lo * run: double-and-add below lo's top bit
        g2p::Pt m = run;
        const int top = 31 - __clz((int)lo);
#pragma nounroll
        for (int i = top - 1; i >= 0; --i) {
            g2p_dbl_ni(&m);
            if ((lo >> i) & 1u) g2p_add_ni(&m, &run);
        }
        g2p_add_ni(&acc, &m);
    }
#pragma nounroll
    for (uint32_t d = 0; d < post_dbl; ++d) g2p_dbl_ni(&acc);
    g2p_st_pt(&out[o], comp, acc);
}

// ---- window sums: trees over ELEMS points in LDS, one pair per addition (2 * ELEMS / 2 = ELEMS threads)
template <int ELEMS>
__device__ __forceinline__ void g2p_tree(G2X28* sh, int count) {  // count: a power of two <= ELEMS; result in sh[0]
    const uint32_t pr = threadIdx.x >> 1, comp = (threadIdx.x & 1u) * 56u;
    for (int s = count / 2; s > 0; s >>= 1) {
        if ((int)pr < s) {
            g2p::Pt a = g2p_ld_pt(&sh[pr], comp);
            const g2p::Pt q = g2p_ld_pt(&sh[pr + s], comp);
            g2p_add_ni(&a, &q);
            g2p_st_pt(&sh[pr], comp, a);
        }
        __syncthreads();
    }
}
template <int ELEMS>
__global__ void __launch_bounds__(ELEMS) msm_window_partial_g2pair_kernel(const G2X28* __restrict__ chunk_out, uint32_t per_win, uint32_t groups,
                                                                          G2X28* __restrict__ partial_out) {
    __shared__ G2X28 sh[ELEMS];
    const uint32_t w = blockIdx.x / groups, g = blockIdx.x % groups;
    const uint32_t i = g * ELEMS + threadIdx.x;
    sh[threadIdx.x] = i < per_win ? chunk_out[(size_t)w * per_win + i] : xyzz_identity<Fp2x28Ops>();
    __syncthreads();
    g2p_tree<ELEMS>(sh, ELEMS);
    if (threadIdx.x < 2) g2p_st_pt(&partial_out[(size_t)w * groups + g], (threadIdx.x & 1u) * 56u, g2p_ld_pt(&sh[0], (threadIdx.x & 1u) * 56u));
}
template <int ELEMS>
__global__ void __launch_bounds__(ELEMS) msm_window_sum_g2pair_kernel(const G2X28* __restrict__ partials, uint32_t groups, XyzzT<Fp2Ops>* __restrict__ win_out) {
    __shared__ G2X28 sh[ELEMS];
    const uint32_t w = blockIdx.x;
    const G2X28* src = partials + (size_t)w * groups;
    const uint32_t pr = threadIdx.x >> 1, comp = (threadIdx.x & 1u) * 56u;
    int active = 1;
    while (active < ELEMS && (uint32_t)active < groups) active <<= 1;
    // groups <= ELEMS partials straight into LDS (more: folded by the pairs, serially, first)
    sh[threadIdx.x] = threadIdx.x < groups ? src[threadIdx.x] : xyzz_identity<Fp2x28Ops>();
    __syncthreads();
    if (groups > (uint32_t)ELEMS) {
        for (uint32_t e = pr; e < (uint32_t)ELEMS; e += ELEMS / 2) {
            g2p::Pt a = g2p_ld_pt(&sh[e], comp);
            for (uint32_t i = e + ELEMS; i < groups; i += ELEMS) {
                const g2p::Pt q = g2p_ld_pt(&src[i], comp);
                g2p_add_ni(&a, &q);
            }
            g2p_st_pt(&sh[e], comp, a);
        }
        __syncthreads();
    }
    g2p_tree<ELEMS>(sh, active);
    if (threadIdx.x == 0) win_out[w] = g2x28::to_std(sh[0]);  // standard 12 x 32-bit XYZZ for the host
}

}  // namespace bzk
