// K1 (batched Poseidon) and K2 (dense 4-ary ZkState tree re-hash) for gfx950.
//
// Replaces, for bulk inputs, the reference's
//   PoseidonHasher::hash -> poseidon::poseidon      /root/reference/src/zk/mod.rs:496-511,
//                                                   /root/reference/src/zk/poseidon/mod.rs:24-84
//   KvStoreStateManager::set_data level loop        /root/reference/src/zk/state/mod.rs:353-391
// with identical outputs: state = [0, inputs...], R_F/2 full rounds, R_P partial rounds (S-box on
// element 0 only), R_F/2 full rounds, dense MDS product every round, result = state[1].
//
// Parameters are NOT copied from the reference's params/*.txt: they are re-derived at first use
// with the public hadeshash Grain-LFSR procedure (src/zk/poseidon/params/README.md names the
// generator script and its arguments `1 0 255 t 5 128`); tests compare them with the files and the
// 16 reference KATs (src/zk/poseidon/mod.rs:114-149) gate the result.
//
// Kernel shape: one lane per hash, state in VGPRs (t x 8 limbs), round constants and the MDS matrix
// are wave-uniform and come through the scalar cache (SGPR operands of v_mad_u64_u32).  The MDS row
// products are accumulated UNREDUCED in 17 limbs and Montgomery-reduced once per row: t*t half
// products + t reductions per round instead of t*t full products.  The work is integer-ALU bound
// (about 1.9k Fr products per arity-4 hash against 160 B of traffic), so coalescing matters little.
#include <string.h>

#include <mutex>
#include <algorithm>
#include <map>
#include <vector>

#include "bzk_poseidon29.cuh"
#include "bzk_poseidon29_coop.cuh"
#include "bzk_poseidon_opt.h"
#include "bzk_internal.h"
#include "host_zk.h"

namespace bzk {

// ------------------------------------------------------------------------------------------------
// host: parameter generation (Grain LFSR, hadeshash)
// ------------------------------------------------------------------------------------------------
namespace {

struct GrainLfsr {
    uint8_t s[80];
    int head = 0;
    GrainLfsr(int t, int rf, int rp) {
        int k = 0;
        auto put = [&](int v, int w) {
            for (int i = w - 1; i >= 0; --i) s[k++] = (uint8_t)((v >> i) & 1);
        };
        put(1, 2);     // field = GF(p)
        put(0, 4);     // s-box = x^alpha
        put(255, 12);  // n
        put(t, 12);
        put(rf, 10);
        put(rp, 10);
        while (k < 80) s[k++] = 1;
        for (int i = 0; i < 160; ++i) clock();
    }
    int clock() {
        auto at = [&](int i) { return s[(head + i) % 80]; };
        uint8_t nb = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
        s[head] = nb;
        head = (head + 1) % 80;
        return nb;
    }
    int next_bit() {  // shrinking: keep the 2nd bit of a pair iff the 1st is set
        for (;;) {
            int a = clock(), b = clock();
            if (a) return b;
        }
    }
    Fr next_raw(bool* below_modulus) {  // 255 bits, MSB first, canonical limbs (not reduced)
        Fr v = Fr::zero();
        for (int i = 254; i >= 0; --i)
            if (next_bit()) v.l[i >> 5] |= 1u << (i & 31);
        bool lt = false;
        for (int i = 7; i >= 0; --i) {
            if (v.l[i] != FrParams::MOD[i]) {
                lt = v.l[i] < FrParams::MOD[i];
                break;
            }
        }
        *below_modulus = lt;
        return v;
    }
};

struct HostParams {
    int t = 0, rf = 8, rp = 0;
    std::vector<Fr> rc, mds;  // Montgomery
};

std::mutex g_mu;
HostParams g_params[18];

const HostParams& host_params(int t) {
    std::lock_guard<std::mutex> lk(g_mu);
    HostParams& P = g_params[t];
    if (P.t) return P;
    P.rf = 8;
    P.rp = t <= 5 ? 56 : 57;
    GrainLfsr g(t, P.rf, P.rp);
    while ((int)P.rc.size() < t * (P.rf + P.rp)) {
        bool lt;
        Fr v = g.next_raw(&lt);
        if (lt) P.rc.push_back(fe_to_mont<FrParams>(v));  // rejection sampling
    }
    std::vector<Fr> xy;
    for (int i = 0; i < 2 * t; ++i) {
        bool lt;
        Fr v = g.next_raw(&lt);
        if (!lt) fe_reduce_once<FrParams>(v);  // MDS draws are reduced, not rejected
        xy.push_back(fe_to_mont<FrParams>(v));
    }
    P.mds.resize((size_t)t * t);
    for (int i = 0; i < t; ++i)
        for (int j = 0; j < t; ++j) P.mds[(size_t)i * t + j] = fe_inv<FrParams>(fe_add<FrParams>(xy[i], xy[t + j]));
    P.t = t;
    return P;
}

}  // namespace

PoseidonHostParams poseidon_host_params(int t) {
    const HostParams& P = host_params(t);
    return {P.t, P.rf, P.rp, P.rc.data(), P.mds.data()};
}

// ------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------
__device__ __noinline__ Fr fr_mul_call(Fr a, Fr b) { return fe_mul<FrParams>(a, b); }

__device__ __forceinline__ Fr fr_sbox(const Fr& x) {
    Fr x2 = fr_mul_call(x, x);
    Fr x4 = fr_mul_call(x2, x2);
    return fr_mul_call(x4, x);
}

// acc(17 limbs) += a * b   (plain 8x8 limb product, no reduction)
__device__ __forceinline__ void wide_mac(uint32_t (&acc)[17], const Fr& a, const uint32_t* __restrict__ b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t bi = b[i];
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t s = (uint64_t)a.l[j] * bi + acc[i + j] + c;
            acc[i + j] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        // propagate into the upper limbs
#pragma unroll
        for (int k = i + 8; k < 17; ++k) {
            uint64_t s = (uint64_t)acc[k] + c;
            acc[k] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
    }
}

// Montgomery reduction of a 17-limb value v < t * r^2 (t <= 17, so v < 2^5 * 2^510 < 2^544) to
// v * R^-1 mod r, fully reduced.
struct Wide17 {
    uint32_t v[17];
};
// by-value argument (17 VGPRs): never hand a noinline device function a pointer into the caller's
// private arrays - that form hung on gfx950 / ROCm 7.2 in the first GPU session (HISTORY.md 3.1)
__device__ __noinline__ Fr wide_reduce(Wide17 acc_in) {
    uint32_t a[18];
#pragma unroll
    for (int i = 0; i < 17; ++i) a[i] = acc_in.v[i];
    a[17] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t m = a[i] * FrParams::INV;
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t s = (uint64_t)m * FrParams::MOD[j] + a[i + j] + c;
            a[i + j] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
#pragma unroll
        for (int k = i + 8; k < 18; ++k) {
            uint64_t s = (uint64_t)a[k] + c;
            a[k] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
    }
    // value = a[8..17] (10 limbs), < (t r^2 + R r)/R < (t+1) * r  with t <= 17: subtract r while >= r.
    // a[16], a[17] can be non-zero only transiently (value < 18 r < 2^260): fold by repeated subtraction.
    uint32_t v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = a[8 + i];
    // binary descent: subtract 16r, 8r, 4r, 2r, r when possible
#pragma unroll
    for (int sh = 4; sh >= 0; --sh) {
        uint32_t t[9];
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            // limb i of (r << sh)
            uint32_t lo = i < 8 ? FrParams::MOD[i] : 0u;
            uint32_t prev = (i > 0 && i <= 8) ? FrParams::MOD[i - 1] : 0u;
            uint32_t ri = sh ? ((lo << sh) | (prev >> (32 - sh))) : lo;
            uint64_t d = (uint64_t)v[i] - ri - borrow;
            t[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
        if (!borrow) {
#pragma unroll
            for (int i = 0; i < 9; ++i) v[i] = t[i];
        }
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = v[i];
    return r;
}

template <int T>
__global__ void __launch_bounds__(128) poseidon_kernel(const Fr* __restrict__ in, uint64_t n, const Fr* __restrict__ consts,
                                                       int rf, int rp, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr st[T];
    st[0] = Fr::zero();
#pragma unroll
    for (int k = 1; k < T; ++k) st[k] = in[i * (T - 1) + (k - 1)];
    const Fr* rc = consts;
    const Fr* mds = consts + (size_t)T * (rf + rp);
    const int half_f = rf / 2;
    for (int rnd = 0; rnd < rf + rp; ++rnd) {
#pragma unroll
        for (int k = 0; k < T; ++k) st[k] = fe_add<FrParams>(st[k], rc[rnd * T + k]);
        const bool full = rnd < half_f || rnd >= half_f + rp;
        if (full) {
#pragma unroll
            for (int k = 0; k < T; ++k) st[k] = fr_sbox(st[k]);
        } else {
            st[0] = fr_sbox(st[0]);
        }
        Fr nw[T];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            Wide17 acc;
#pragma unroll
            for (int q = 0; q < 17; ++q) acc.v[q] = 0;
#pragma unroll
            for (int k = 0; k < T; ++k) wide_mac(acc.v, st[k], mds[j * T + k].l);
            nw[j] = wide_reduce(acc);
        }
#pragma unroll
        for (int k = 0; k < T; ++k) st[k] = nw[k];
    }
    out[i] = st[1];
}

// ---- reduced-radix kernel (widths 2..8, i.e. every arity the MPN circuits and trees use)
// State in 9 x 29-bit limbs (bzk_fr29.cuh).  Per round: add constant + carry-normalise; S-box through the
// resident product function; each MDS row = up to 6 products accumulated UNREDUCED in 64-bit columns and ONE
// Montgomery reduction (rows wider than 6 use two groups).  Bounds: see the header of bzk_fr29.cuh.
template <int T>
__global__ void __launch_bounds__(128) poseidon29_kernel(const Fr* __restrict__ in, uint64_t n, const Fr29* __restrict__ consts,
                                                         int rf, int rp, Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = poseidon29_hash<T>(in + i * (T - 1), consts, rf, rp);
}

typedef void (*poseidon29_fn)(const Fr*, uint64_t, const Fr29*, int, int, Fr*);
static poseidon29_fn poseidon29_table(int t) {
    switch (t) {
        case 2: return poseidon29_kernel<2>;
        case 3: return poseidon29_kernel<3>;
        case 4: return poseidon29_kernel<4>;
        case 5: return poseidon29_kernel<5>;
        case 6: return poseidon29_kernel<6>;
        case 7: return poseidon29_kernel<7>;
        case 8: return poseidon29_kernel<8>;
        default: return nullptr;
    }
}

typedef void (*poseidon_fn)(const Fr*, uint64_t, const Fr*, int, int, Fr*);
static poseidon_fn poseidon_table(int t) {
    switch (t) {
        case 9: return poseidon_kernel<9>;
        case 10: return poseidon_kernel<10>;
        case 11: return poseidon_kernel<11>;
        case 12: return poseidon_kernel<12>;
        case 13: return poseidon_kernel<13>;
        case 14: return poseidon_kernel<14>;
        case 15: return poseidon_kernel<15>;
        case 16: return poseidon_kernel<16>;
        case 17: return poseidon_kernel<17>;
        default: return nullptr;
    }
}

// host evaluation of the very function the kernel runs (it is __host__ __device__), for the start-up self check
static Fr host_hash29(int t, const Fr* in, const Fr29* c, int rf, int rp) {
    switch (t) {
        case 2: return poseidon29_hash<2>(in, c, rf, rp);
        case 3: return poseidon29_hash<3>(in, c, rf, rp);
        case 4: return poseidon29_hash<4>(in, c, rf, rp);
        case 5: return poseidon29_hash<5>(in, c, rf, rp);
        case 6: return poseidon29_hash<6>(in, c, rf, rp);
        case 7: return poseidon29_hash<7>(in, c, rf, rp);
        default: return poseidon29_hash<8>(in, c, rf, rp);
    }
}
// the reference's plain round function on the host (8 x 32-bit limbs), the yardstick of the self check
static Fr host_hash_plain(const HostParams& P, const Fr* in) {
    const int T = P.t, half = P.rf / 2;
    std::vector<Fr> st((size_t)T, Fr::zero()), nw((size_t)T);
    for (int k = 1; k < T; ++k) st[k] = in[k - 1];
    for (int r = 0; r < P.rf + P.rp; ++r) {
        for (int k = 0; k < T; ++k) st[k] = fe_add<FrParams>(st[k], P.rc[(size_t)r * T + k]);
        const bool full = r < half || r >= half + P.rp;
        for (int k = 0; k < (full ? T : 1); ++k) {
            Fr x2 = fe_sqr<FrParams>(st[k]);
            st[k] = fe_mul<FrParams>(fe_sqr<FrParams>(x2), st[k]);
        }
        for (int j = 0; j < T; ++j) {
            Fr acc = Fr::zero();
            for (int k = 0; k < T; ++k) acc = fe_add<FrParams>(acc, fe_mul<FrParams>(P.mds[(size_t)j * T + k], st[k]));
            nw[j] = acc;
        }
        st.swap(nw);
    }
    return st[1];
}

// device copy of the constants: widths <= 8 in the sparse-partial-round form (bzk_poseidon_opt.h), 9 x 29-bit
// internal representation (rc are plain addends: x * 2^261); wider ones in the 8 x 32-bit form of the generic kernel
static int32_t poseidon_consts_dev(bzk_ctx* ctx, int t, const void** out, int* rf, int* rp) {
    const HostParams& P = host_params(t);
    *rf = P.rf;
    *rp = P.rp;
    if (!ctx->poseidon_dev[t]) {
        void* d = nullptr;
        if (t <= 8) {
            std::vector<Fr> flat;
            if (!poseidon_optimize(t, P.rf, P.rp, P.rc, P.mds, flat)) {
                ctx->last_error = "poseidon: no sparse form for width " + std::to_string(t);
                return BZK_E_INTERNAL;
            }
            std::vector<Fr29> f29(flat.size());
            for (size_t i = 0; i < flat.size(); ++i) f29[i] = fr29::norm(fr29::to29(flat[i]));
            // self check: the derived constants reproduce the plain round function (fails loudly, no fallback)
            Fr probe[8];
            for (int k = 0; k < 8; ++k) probe[k] = fe_mul<FrParams>(P.mds[(size_t)(k % (t * t))], P.rc[(size_t)k]);
            if (!host_hash29(t, probe, f29.data(), P.rf, P.rp).equals(host_hash_plain(P, probe))) {
                ctx->last_error = "poseidon: sparse-round constants failed the self check for width " + std::to_string(t);
                return BZK_E_INTERNAL;
            }
            BZK_HIP(ctx, hipMalloc(&d, f29.size() * sizeof(Fr29)));
            BZK_HIP(ctx, hipMemcpyAsync(d, f29.data(), f29.size() * sizeof(Fr29), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            std::vector<Fr> flat(P.rc);
            flat.insert(flat.end(), P.mds.begin(), P.mds.end());
            BZK_HIP(ctx, hipMalloc(&d, flat.size() * sizeof(Fr)));
            BZK_HIP(ctx, hipMemcpyAsync(d, flat.data(), flat.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        ctx->poseidon_dev[t] = d;
    }
    *out = ctx->poseidon_dev[t];
    return BZK_OK;
}
// the same table for witfill.hip (pass 1 of the deferred witness values hashes with poseidon29_hash too)
int32_t poseidon_consts_dev_shared(bzk_ctx* ctx, int t, const void** out, int* rf, int* rp) { return poseidon_consts_dev(ctx, t, out, rf, rp); }


// ------------------------------------------------------------------------------------------------
// Cooperative hash for LATENCY-bound launches (the top levels of a tree, small batched tree updates): eight lanes per node instead of one lane
// per node - bzk_poseidon29_coop.cuh; used only when there are too few nodes to fill the machine anyway.
// ------------------------------------------------------------------------------------------------
static constexpr uint64_t COOP_MAX_NODES = 8192;

// node i of the launch: children at in[4 i .. 4 i + 3] (or, with a parent list, at child[4 p] -> parent[p])
__global__ void __launch_bounds__(64) poseidon29_coop5_kernel(const Fr* __restrict__ in, uint64_t n, const uint64_t* __restrict__ parents,
                                                              const Fr29* __restrict__ consts, int rf, int rp, Fr* __restrict__ out) {
    const uint64_t node = (uint64_t)blockIdx.x * 8 + (threadIdx.x >> 3);
    const uint64_t clamped = node < n ? node : n - 1;  // whole groups stay convergent for the shuffles
    const uint64_t p = parents ? parents[clamped] : clamped;
    const Fr h = poseidon29_coop<5>(in + 4 * p, consts, rf, rp);
    if (node < n && (threadIdx.x & 7) == 1) out[p] = h;
}

// the same for any width: node i hashes in[(T - 1) i .. (T - 1) i + T - 2]
template <int T>
__global__ void __launch_bounds__(64) poseidon29_coop_kernel(const Fr* __restrict__ in, uint64_t n, const Fr29* __restrict__ consts, int rf, int rp,
                                                             Fr* __restrict__ out) {
    const uint64_t node = (uint64_t)blockIdx.x * 8 + (threadIdx.x >> 3);
    const uint64_t p = node < n ? node : n - 1;  // whole groups stay convergent for the shuffles
    const Fr h = poseidon29_coop<T>(in + (uint64_t)(T - 1) * p, consts, rf, rp);
    if (node < n && (threadIdx.x & 7) == 1) out[p] = h;
}
typedef void (*poseidon29_coop_fn)(const Fr*, uint64_t, const Fr29*, int, int, Fr*);
static poseidon29_coop_fn poseidon29_coop_table(int t) {
    switch (t) {
        case 2: return poseidon29_coop_kernel<2>;
        case 3: return poseidon29_coop_kernel<3>;
        case 4: return poseidon29_coop_kernel<4>;
        case 6: return poseidon29_coop_kernel<6>;
        case 7: return poseidon29_coop_kernel<7>;
        case 8: return poseidon29_coop_kernel<8>;
        default: return nullptr;
    }
}

// env BZK_POSEIDON_COOP_ANY=0: widths other than 5 keep the one-lane-per-hash kernel for small batches too (A/B runs)
static bool poseidon_coop_any_off() {
    static const bool off = [] { const char* e = getenv("BZK_POSEIDON_COOP_ANY"); return e && atoi(e) == 0; }();
    return off;
}

int32_t poseidon_launch(bzk_ctx* ctx, const void* in_dev, uint32_t arity, uint64_t n, void* out_dev) {
    if (arity < 1 || arity > 16) return BZK_E_ARG;
    if (n == 0) return BZK_OK;
    const int t = (int)arity + 1;
    const void* consts;
    int rf, rp;
    BZK_TRY(poseidon_consts_dev(ctx, t, &consts, &rf, &rp));
    const uint64_t blocks = (n + 127) / 128;
    if (blocks > 0x7fffffffull) return BZK_E_ARG;
    if (t == 5 && n <= COOP_MAX_NODES && !ctx->no_coop) {  // too few nodes to fill the machine: shorten the chain instead
        BZK_LAUNCH(ctx, "poseidon_coop", poseidon29_coop5_kernel, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, (const Fr*)in_dev, n,
                   (const uint64_t*)nullptr, (const Fr29*)consts, rf, rp, (Fr*)out_dev);
        return BZK_OK;
    }
    if (t <= 8 && n <= COOP_MAX_NODES && !ctx->no_coop && !poseidon_coop_any_off()) {  // the struct hashes of small batches: same reasoning
        poseidon29_coop_fn k = poseidon29_coop_table(t);
        BZK_LAUNCH(ctx, "poseidon_coop", k, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, (const Fr*)in_dev, n, (const Fr29*)consts, rf, rp, (Fr*)out_dev);
        return BZK_OK;
    }
    if (t <= 8) {
        poseidon29_fn k = poseidon29_table(t);
        BZK_LAUNCH(ctx, "poseidon", k, dim3((unsigned)blocks), dim3(128), 0, (const Fr*)in_dev, n, (const Fr29*)consts, rf, rp, (Fr*)out_dev);
    } else {
        poseidon_fn k = poseidon_table(t);
        BZK_LAUNCH(ctx, "poseidon", k, dim3((unsigned)blocks), dim3(128), 0, (const Fr*)in_dev, n, (const Fr*)consts, rf, rp, (Fr*)out_dev);
    }
    return BZK_OK;
}


// ------------------------------------------------------------------------------------------------
// device-resident dense 4-ary tree with batched updates and proofs (bzk_tree4_*): all levels in ONE heap-order
// array, node i of depth k at (4^k - 1)/3 + i, leaves at depth log4 - the reference's aux numbering
// (src/zk/state/mod.rs:355, 382-383) extended by the leaf level.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tree4_fill_kernel(Fr* __restrict__ dst, uint64_t count, Fr v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = v;
}
__global__ void __launch_bounds__(256) tree4_iota_kernel(uint64_t* __restrict__ dst, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = i;
}
__global__ void __launch_bounds__(256) tree4_scatter_kernel(Fr* __restrict__ level, const uint64_t* __restrict__ idx,
                                                            const Fr* __restrict__ vals, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) level[idx[i]] = vals[i];
}
// parents[j] of the level at `parent`: re-hash from its four children (contiguous in `child`)
__global__ void __launch_bounds__(128) tree4_rehash_kernel(const Fr* __restrict__ child, Fr* __restrict__ parent,
                                                           const uint64_t* __restrict__ parents, uint64_t n,
                                                           const Fr29* __restrict__ consts, int rf, int rp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = parents[i];
    parent[p] = poseidon29_hash<5>(child + 4 * p, consts, rf, rp);
}
// out[(q * log4 + layer) * 3 + s] = s-th sibling (index order, self skipped) of query q at depth log4 - layer
__global__ void __launch_bounds__(256) tree4_prove_kernel(const Fr* __restrict__ nodes, uint32_t log4, const uint64_t* __restrict__ idx,
                                                          uint64_t n, Fr* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * log4) return;
    const uint64_t q = t / log4;
    const uint32_t layer = (uint32_t)(t % log4);  // 0 = leaf level
    const uint32_t depth = log4 - layer;
    const uint64_t off = ((((uint64_t)1) << (2 * depth)) - 1) / 3;
    const uint64_t cur = idx[q] >> (2 * layer);
    const uint64_t base = cur & ~(uint64_t)3;
    int s = 0;
    for (uint64_t j = base; j < base + 4; ++j)
        if (j != cur) out[t * 3 + s++] = nodes[off + j];
}

// ------------------------------------------------------------------------------------------------
// Device-resident MPN account state (second half; bzk_mpn_tree_*): the production state model
//     List{L, Struct{tx_nonce, withdraw_nonce, pub_x, pub_y, List{T, Struct{token_id, balance}}}}   src/mpn/mod.rs:218-241
// as `KvStoreStateManager::{get,set}_mpn_account / prove` see it (src/zk/state/mod.rs:93-264, 310-420), kept in HBM:
//   * the account level is the dense heap-order tree above (bzk_tree4) over the account LEAF HASHES
//     H5(nonce, wnonce, x, y, tokens_root) - 46 GB at the production depth L = 15;
//   * account CONTENTS exist only for populated accounts, in a slot pool (the dense form of 4^15 accounts x 4^3 token
//     slots would be 4.5 TB): per slot the four cells, the 4^T (token_id, balance) pairs and the token sub-tree as levels
//     of a forest - level k holds cap x 4^k hashes, node (slot, i) at slot 4^k + i, so "parent = index >> 2" holds across
//     the whole forest and one launch re-hashes a level for all touched accounts.  Slot 0 is the default (empty) account.
// The account-index -> slot map lives on the host side of the handle (an unordered_map): kernels get slot lists.
// ------------------------------------------------------------------------------------------------
// in5[a] = cells[slot_a][0..3], token_root[slot_a]
__global__ void __launch_bounds__(256) mpn_leaf_inputs_kernel(const Fr* __restrict__ cells, const Fr* __restrict__ tok_root,
                                                              const uint64_t* __restrict__ slots, uint64_t n, Fr* __restrict__ in5) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 5) return;
    const uint64_t a = t / 5, j = t % 5;
    const uint64_t s = slots[a];
    in5[t] = j < 4 ? cells[s * 4 + j] : tok_root[s];
}
// out record of account a: nonce, wnonce, x, y, tokens_root, then 4^T x (token_id, balance)
__global__ void __launch_bounds__(256) mpn_get_kernel(const Fr* __restrict__ cells, const Fr* __restrict__ tok, const Fr* __restrict__ tok_root,
                                                      const uint64_t* __restrict__ slots, uint64_t n, uint32_t rec, uint32_t tslots,
                                                      Fr* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * rec) return;
    const uint64_t a = t / rec, j = t % rec;
    const uint64_t s = slots[a];
    out[t] = j < 4 ? cells[s * 4 + j] : j == 4 ? tok_root[s] : tok[s * 2 * tslots + (j - 5)];
}
struct ForestLevels {
    const Fr* lvl[9];
};
// token-level `prove`: out[(q * T + layer) * 3 + s] = s-th sibling (index order, self skipped) at forest depth T - layer
__global__ void __launch_bounds__(256) mpn_prove_token_kernel(ForestLevels F, uint32_t T, const uint64_t* __restrict__ gidx, uint64_t n,
                                                              Fr* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * T) return;
    const uint64_t q = t / T;
    const uint32_t layer = (uint32_t)(t % T);
    const uint32_t depth = T - layer;
    const uint64_t cur = gidx[q] >> (2 * layer);
    const uint64_t base = cur & ~(uint64_t)3;
    const Fr* L = F.lvl[depth];
    int s = 0;
    for (uint64_t j = base; j < base + 4; ++j)
        if (j != cur) out[t * 3 + s++] = L[j];
}

}  // namespace bzk

using namespace bzk;

extern "C" {

int32_t bzk_poseidon_batch_dev(bzk_ctx* ctx, const void* in_dev, uint32_t arity, uint64_t n, void* out_dev) {
    if (!ctx || (n && (!in_dev || !out_dev))) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    return poseidon_launch(ctx, in_dev, arity, n, out_dev);
}

int32_t bzk_poseidon_batch(bzk_ctx* ctx, const uint8_t* in, uint32_t arity, uint64_t n, uint8_t* out) {
    if (!ctx || arity < 1 || arity > 16 || (n && (!in || !out))) return BZK_E_ARG;
    if (n == 0) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    const size_t in_b = (size_t)n * arity * 32, out_b = (size_t)n * 32;
    BZK_TRY(ws_reserve(ctx, ws_pad(in_b) + ws_pad(out_b) + 512));
    WsCursor cur(ctx->ws);
    uint8_t* din = cur.take<uint8_t>(in_b);
    uint8_t* dout = cur.take<uint8_t>(out_b);
    BZK_HIP(ctx, hipMemcpyAsync(din, in, in_b, hipMemcpyHostToDevice, ctx->stream));
    BZK_TRY(poseidon_launch(ctx, din, arity, n, dout));
    BZK_HIP(ctx, hipMemcpyAsync(out, dout, out_b, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

// Dense tree: level k (k = log4-1 .. 0) has 4^k nodes, each the arity-4 hash of 4 consecutive
// children of level k+1 - exactly the batched layout, so a level is one poseidon launch.
int32_t bzk_merkle4_root_dev(bzk_ctx* ctx, const void* leaves_dev, uint32_t log4, uint8_t root[32], void* nodes_opt_dev) {
    if (!ctx || !leaves_dev || !root || log4 > 15) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    if (log4 == 0) {
        BZK_HIP(ctx, hipMemcpyAsync(root, leaves_dev, 32, hipMemcpyDeviceToHost, ctx->stream));
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return BZK_OK;
    }
    const uint64_t n_internal = ((((uint64_t)1) << (2 * log4)) - 1) / 3;
    uint8_t* nodes = (uint8_t*)nodes_opt_dev;
    if (!nodes) {
        BZK_TRY(ws_reserve(ctx, ws_pad(n_internal * 32) + 512));
        nodes = (uint8_t*)ctx->ws;
    }
    const uint8_t* child = (const uint8_t*)leaves_dev;
    for (int k = (int)log4 - 1; k >= 0; --k) {
        const uint64_t cnt = (uint64_t)1 << (2 * k);
        uint8_t* dst = nodes + ((cnt - 1) / 3) * 32;  // heap offset (4^k - 1)/3
        BZK_TRY(poseidon_launch(ctx, child, 4, cnt, dst));
        child = dst;
    }
    BZK_HIP(ctx, hipMemcpyAsync(root, nodes, 32, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

int32_t bzk_merkle4_root(bzk_ctx* ctx, const uint8_t* leaves, uint32_t log4, uint8_t root[32], uint8_t* nodes_opt) {
    if (!ctx || !leaves || !root || log4 > 15) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    const uint64_t n_leaves = (uint64_t)1 << (2 * log4), n_internal = (n_leaves - 1) / 3;
    void *dl = nullptr, *dn = nullptr;
    BZK_HIP(ctx, hipMalloc(&dl, n_leaves * 32));
    if (hipMalloc(&dn, n_internal ? n_internal * 32 : 32) != hipSuccess) {
        (void)hipFree(dl);
        return BZK_E_ALLOC;
    }
    int32_t st = BZK_OK;
    if (hipMemcpyAsync(dl, leaves, n_leaves * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
    if (st == BZK_OK) st = bzk_merkle4_root_dev(ctx, dl, log4, root, dn);
    if (st == BZK_OK && nodes_opt && n_internal) {
        if (hipMemcpyAsync(nodes_opt, dn, n_internal * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dl);
    (void)hipFree(dn);
    return st;
}

// ---- device-resident tree (see the kernels above)
struct bzk_tree4 {
    uint32_t log4 = 0;
    Fr* nodes = nullptr;
    uint64_t n_nodes = 0;
};
static inline uint64_t tree4_off(uint32_t depth) { return ((((uint64_t)1) << (2 * depth)) - 1) / 3; }

int32_t bzk_tree4_create(bzk_ctx* ctx, uint32_t log4, const void* leaves_dev, const uint8_t default_leaf[32], bzk_tree4** out) {
    if (!ctx || !out || log4 == 0 || log4 > 15 || (!leaves_dev && !default_leaf)) return BZK_E_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    bzk_tree4* t = new (std::nothrow) bzk_tree4();
    if (!t) return BZK_E_ALLOC;
    t->log4 = log4;
    t->n_nodes = tree4_off(log4 + 1);
    if (hipMalloc((void**)&t->nodes, t->n_nodes * sizeof(Fr)) != hipSuccess) {
        (void)hipGetLastError();
        delete t;
        return BZK_E_ALLOC;
    }
    auto fail = [&](int32_t st) {
        (void)hipFree(t->nodes);
        delete t;
        return st;
    };
    if (leaves_dev) {
        const uint64_t n_leaves = (uint64_t)1 << (2 * log4);
        if (hipMemcpyAsync(t->nodes + tree4_off(log4), leaves_dev, n_leaves * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            return fail(BZK_E_DEVICE);
        for (int k = (int)log4 - 1; k >= 0; --k) {
            const int32_t st = poseidon_launch(ctx, t->nodes + tree4_off(k + 1), 4, (uint64_t)1 << (2 * k), t->nodes + tree4_off(k));
            if (st != BZK_OK) return fail(st);
        }
    } else {
        // empty tree: every node of a level equals that level's default (src/zk/state/mod.rs:240-257: the default chain)
        ZkScalar d;
        memcpy(d.v.l, default_leaf, 32);
        for (int k = (int)log4; k >= 0; --k) {
            const uint64_t cnt = (uint64_t)1 << (2 * k);
            hipLaunchKernelGGL(tree4_fill_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, t->nodes + tree4_off(k), cnt, d.v);
            if (hipGetLastError() != hipSuccess) return fail(BZK_E_DEVICE);
            ZkScalar c[4] = {d, d, d, d};
            d = poseidon_hash(c, 4);
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(BZK_E_DEVICE);
    *out = t;
    return BZK_OK;
}

void bzk_tree4_free(bzk_ctx* ctx, bzk_tree4* t) {
    if (!t) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(t->nodes);
    delete t;
}

int32_t bzk_tree4_root(bzk_ctx* ctx, const bzk_tree4* t, uint8_t root[32]) {
    if (!ctx || !t || !root) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    BZK_HIP(ctx, hipMemcpyAsync(root, t->nodes, 32, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

// Batched `set_data` on leaves (src/zk/state/mod.rs:310-420): write the n leaves, then re-hash, level by level, exactly
// the parents that have a changed child (each once).  A later entry for the same index wins.
int32_t bzk_tree4_update(bzk_ctx* ctx, bzk_tree4* t, const uint64_t* idx, const uint8_t* leaves, uint64_t n) {
    if (!ctx || !t || (n && (!idx || !leaves))) return BZK_E_ARG;
    if (n == 0) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    const uint64_t n_leaves = (uint64_t)1 << (2 * t->log4);
    std::map<uint64_t, uint64_t> last;  // index -> position of its last occurrence
    for (uint64_t i = 0; i < n; ++i) {
        if (idx[i] >= n_leaves) return BZK_E_ARG;
        last[idx[i]] = i;
    }
    std::vector<uint64_t> uidx;
    std::vector<Fr> uval;
    uidx.reserve(last.size());
    uval.reserve(last.size());
    for (auto& kv : last) {
        uidx.push_back(kv.first);
        Fr v;
        memcpy(v.l, leaves + 32 * kv.second, 32);
        uval.push_back(v);
    }
    const uint64_t m = uidx.size();
    const void* consts;
    int rf, rp;
    BZK_TRY(poseidon_consts_dev(ctx, 5, &consts, &rf, &rp));
    // parent lists of every level, computed on the host (sorted unique), uploaded in one piece
    std::vector<uint64_t> all(uidx);  // level log4 (the leaves) first
    std::vector<uint64_t> level_off{0}, level_cnt{m};
    {
        std::vector<uint64_t> parents(uidx);  // sorted (map order)
        for (int k = (int)t->log4 - 1; k >= 0; --k) {
            for (auto& p : parents) p >>= 2;
            parents.erase(std::unique(parents.begin(), parents.end()), parents.end());
            level_off.push_back(all.size());
            level_cnt.push_back(parents.size());
            all.insert(all.end(), parents.begin(), parents.end());
        }
    }
    BZK_TRY(ws_reserve(ctx, ws_pad(all.size() * 8) + ws_pad(m * 32) + 512));
    WsCursor cur(ctx->ws);
    uint64_t* d_idx = cur.take<uint64_t>(all.size());
    Fr* d_val = cur.take<Fr>(m);
    BZK_HIP(ctx, hipMemcpyAsync(d_idx, all.data(), all.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipMemcpyAsync(d_val, uval.data(), m * 32, hipMemcpyHostToDevice, ctx->stream));
    BZK_LAUNCH(ctx, "tree4_scatter", tree4_scatter_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, t->nodes + tree4_off(t->log4),
               (const uint64_t*)d_idx, (const Fr*)d_val, m);
    for (int k = (int)t->log4 - 1, lv = 1; k >= 0; --k, ++lv) {
        const uint64_t np = level_cnt[lv];
        if (np <= COOP_MAX_NODES && !ctx->no_coop) {
            BZK_LAUNCH(ctx, "tree4_rehash_coop", poseidon29_coop5_kernel, dim3((unsigned)((np + 7) / 8)), dim3(64), 0,
                       (const Fr*)(t->nodes + tree4_off(k + 1)), np, (const uint64_t*)(d_idx + level_off[lv]), (const Fr29*)consts, rf, rp,
                       t->nodes + tree4_off(k));
        } else {
            BZK_LAUNCH(ctx, "tree4_rehash", tree4_rehash_kernel, dim3((unsigned)((np + 127) / 128)), dim3(128), 0,
                       (const Fr*)(t->nodes + tree4_off(k + 1)), t->nodes + tree4_off(k), (const uint64_t*)(d_idx + level_off[lv]), np,
                       (const Fr29*)consts, rf, rp);
        }
    }
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `all` / `uval` are pageable host memory: keep them alive until here
    return BZK_OK;
}

// Batched `prove` (src/zk/state/mod.rs:218-264): for every index the log4 sibling triples, leaf level first, siblings
// in index order with the node itself left out.  out: n * log4 * 3 scalars.
int32_t bzk_tree4_prove(bzk_ctx* ctx, const bzk_tree4* t, const uint64_t* idx, uint64_t n, uint8_t* out) {
    if (!ctx || !t || (n && (!idx || !out))) return BZK_E_ARG;
    if (n == 0) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    const uint64_t n_leaves = (uint64_t)1 << (2 * t->log4);
    for (uint64_t i = 0; i < n; ++i)
        if (idx[i] >= n_leaves) return BZK_E_ARG;
    const uint64_t cells = n * t->log4;
    BZK_TRY(ws_reserve(ctx, ws_pad(n * 8) + ws_pad(cells * 96) + 512));
    WsCursor cur(ctx->ws);
    uint64_t* d_idx = cur.take<uint64_t>(n);
    Fr* d_out = cur.take<Fr>(cells * 3);
    BZK_HIP(ctx, hipMemcpyAsync(d_idx, idx, n * 8, hipMemcpyHostToDevice, ctx->stream));
    BZK_LAUNCH(ctx, "tree4_prove", tree4_prove_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, (const Fr*)t->nodes, t->log4,
               (const uint64_t*)d_idx, n, d_out);
    BZK_HIP(ctx, hipMemcpyAsync(out, d_out, cells * 96, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

// node i of depth k (0 = root, log4 = leaves)
int32_t bzk_tree4_node(bzk_ctx* ctx, const bzk_tree4* t, uint32_t depth, uint64_t index, uint8_t out[32]) {
    if (!ctx || !t || !out || depth > t->log4 || index >= ((uint64_t)1 << (2 * depth))) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    BZK_HIP(ctx, hipMemcpyAsync(out, t->nodes + tree4_off(depth) + index, 32, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

// ---- device-resident MPN account state (kernels and layout: see above) -------------------------------------------------
}  // extern "C"
#include <unordered_map>
struct bzk_mpn_tree {
    uint32_t L = 0, T = 0;
    uint64_t cap = 0, used = 1;  // slot 0 = the default account
    bzk_tree4* acct = nullptr;
    Fr* cells = nullptr;         // cap x 4
    Fr* tok = nullptr;           // cap x 4^T x 2
    Fr* tl[9] = {};              // token forest: tl[k] = cap x 4^k hashes (k = 0: the accounts' tokens_root)
    std::unordered_map<uint64_t, uint64_t> slot_of;
    // a device error in the middle of set_accounts leaves contents, leaf hashes and inner nodes out of step: the handle refuses
    // every later call instead of answering from a half-updated state
    bool poisoned = false;
};
namespace {
// re-hash, level by level, exactly the parents with a changed child.  lvl[d] = array of depth d (d = 0 .. depth_leaf);
// `leaf_idx`: sorted unique indices at depth_leaf whose values were already written.
int32_t rehash_paths(bzk_ctx* ctx, Fr* const* lvl, int depth_leaf, int depth_top, const std::vector<uint64_t>& leaf_idx) {
    if (leaf_idx.empty() || depth_leaf <= depth_top) return BZK_OK;
    const void* consts;
    int rf, rp;
    BZK_TRY(poseidon_consts_dev(ctx, 5, &consts, &rf, &rp));
    std::vector<uint64_t> all, level_off, level_cnt;
    std::vector<uint64_t> parents(leaf_idx);
    for (int k = depth_leaf - 1; k >= depth_top; --k) {
        for (auto& p : parents) p >>= 2;
        parents.erase(std::unique(parents.begin(), parents.end()), parents.end());
        level_off.push_back(all.size());
        level_cnt.push_back(parents.size());
        all.insert(all.end(), parents.begin(), parents.end());
    }
    uint64_t* d_idx = nullptr;
    BZK_HIP(ctx, hipMalloc((void**)&d_idx, all.size() * 8));  // not the ctx workspace: the caller's staging lives there
    hipError_t e = hipMemcpyAsync(d_idx, all.data(), all.size() * 8, hipMemcpyHostToDevice, ctx->stream);
    int32_t st = e == hipSuccess ? BZK_OK : BZK_E_DEVICE;
    auto level = [&](int k, size_t lv) -> int32_t {
        const uint64_t np = level_cnt[lv];
        if (np <= COOP_MAX_NODES && !ctx->no_coop) {
            BZK_LAUNCH(ctx, "tree4_rehash_coop", poseidon29_coop5_kernel, dim3((unsigned)((np + 7) / 8)), dim3(64), 0, (const Fr*)lvl[k + 1], np,
                       (const uint64_t*)(d_idx + level_off[lv]), (const Fr29*)consts, rf, rp, lvl[k]);
        } else {
            BZK_LAUNCH(ctx, "tree4_rehash", tree4_rehash_kernel, dim3((unsigned)((np + 127) / 128)), dim3(128), 0, (const Fr*)lvl[k + 1], lvl[k],
                       (const uint64_t*)(d_idx + level_off[lv]), np, (const Fr29*)consts, rf, rp);
        }
        return BZK_OK;
    };
    size_t lv = 0;
    for (int k = depth_leaf - 1; k >= depth_top && st == BZK_OK; --k, ++lv) st = level(k, lv);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == BZK_OK) st = BZK_E_DEVICE;  // `all` is pageable host memory
    (void)hipFree(d_idx);
    return st;
}
struct DevBuf {  // scoped device staging
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int32_t alloc(bzk_ctx* ctx, size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 32) != hipSuccess) {
            (void)hipGetLastError();
            ctx->last_error = "mpn_tree: staging allocation failed";
            return BZK_E_ALLOC;
        }
        return BZK_OK;
    }
};
}  // namespace
extern "C" {

int32_t bzk_mpn_tree_create(bzk_ctx* ctx, uint32_t log4_tree, uint32_t log4_token_tree, uint64_t capacity, bzk_mpn_tree** out) {
    if (!ctx || !out || log4_tree == 0 || log4_tree > 15 || log4_token_tree == 0 || log4_token_tree > 8 || capacity == 0) return BZK_E_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    bzk_mpn_tree* t = new (std::nothrow) bzk_mpn_tree();
    if (!t) return BZK_E_ALLOC;
    t->L = log4_tree;
    t->T = log4_token_tree;
    t->cap = capacity + 1;
    const uint64_t ts = (uint64_t)1 << (2 * t->T);
    auto fail = [&](int32_t st) {
        bzk_mpn_tree_free(ctx, t);
        return st;
    };
    if (hipMalloc((void**)&t->cells, t->cap * 4 * sizeof(Fr)) != hipSuccess || hipMalloc((void**)&t->tok, t->cap * ts * 2 * sizeof(Fr)) != hipSuccess)
        return fail(BZK_E_ALLOC);
    for (uint32_t k = 0; k <= t->T; ++k)
        if (hipMalloc((void**)&t->tl[k], (t->cap << (2 * k)) * sizeof(Fr)) != hipSuccess) return fail(BZK_E_ALLOC);
    // defaults (`compress_default`, src/zk/mod.rs:401-423): zero cells, H2(0, 0) token leaves, the default chain above them
    if (hipMemsetAsync(t->cells, 0, t->cap * 4 * sizeof(Fr), ctx->stream) != hipSuccess ||
        hipMemsetAsync(t->tok, 0, t->cap * ts * 2 * sizeof(Fr), ctx->stream) != hipSuccess)
        return fail(BZK_E_DEVICE);
    ZkScalar z2[2] = {ZkScalar::zero(), ZkScalar::zero()};
    ZkScalar d = poseidon_hash(z2, 2);
    for (int k = (int)t->T; k >= 0; --k) {
        const uint64_t cnt = t->cap << (2 * k);
        hipLaunchKernelGGL(tree4_fill_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, t->tl[k], cnt, d.v);
        if (hipGetLastError() != hipSuccess) return fail(BZK_E_DEVICE);
        if (k > 0) {
            ZkScalar c[4] = {d, d, d, d};
            d = poseidon_hash(c, 4);
        }
    }
    ZkScalar leaf_in[5] = {ZkScalar::zero(), ZkScalar::zero(), ZkScalar::zero(), ZkScalar::zero(), d};
    const ZkScalar leaf = poseidon_hash(leaf_in, 5);
    const int32_t st = bzk_tree4_create(ctx, t->L, nullptr, (const uint8_t*)leaf.v.l, &t->acct);
    if (st != BZK_OK) return fail(st);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(BZK_E_DEVICE);
    *out = t;
    return BZK_OK;
}

void bzk_mpn_tree_free(bzk_ctx* ctx, bzk_mpn_tree* t) {
    if (!t) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (t->acct) bzk_tree4_free(ctx, t->acct);
    if (t->cells) (void)hipFree(t->cells);
    if (t->tok) (void)hipFree(t->tok);
    for (auto& p : t->tl)
        if (p) (void)hipFree(p);
    delete t;
}

int32_t bzk_mpn_tree_root(bzk_ctx* ctx, const bzk_mpn_tree* t, uint8_t root[32]) {
    if (!ctx || !t) return BZK_E_ARG;
    if (t->poisoned) return BZK_E_INTERNAL;
    return bzk_tree4_root(ctx, t->acct, root);
}

uint64_t bzk_mpn_tree_accounts(const bzk_mpn_tree* t) { return t ? t->used - 1 : 0; }

// Batched `set_mpn_account` (src/zk/state/mod.rs:158-208): account a := (cells[a], its token slots tok_index[tok_off[a] ..
// tok_off[a+1]) := tok_vals).  As in the reference, token slots that are not named keep their contents.
static int32_t mpn_tree_set_accounts_body(bzk_ctx* ctx, bzk_mpn_tree* t, const uint64_t* idx, const uint8_t* cells, const uint8_t* tok_vals,
                                          uint64_t n, uint64_t m, const std::vector<uint64_t>& slots, const std::vector<uint64_t>& cell_pos,
                                          const std::vector<uint64_t>& tok_pos, const std::vector<uint64_t>& g, DevBuf& d_pos, DevBuf& d_val,
                                          DevBuf& d_h);

int32_t bzk_mpn_tree_set_accounts(bzk_ctx* ctx, bzk_mpn_tree* t, const uint64_t* idx, const uint8_t* cells, const uint64_t* tok_off,
                                  const uint64_t* tok_index, const uint8_t* tok_vals, uint64_t n) {
    if (!ctx || !t || (n && (!idx || !cells || !tok_off))) return BZK_E_ARG;
    if (t->poisoned) return BZK_E_INTERNAL;
    if (n == 0) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    const uint64_t n_acct = (uint64_t)1 << (2 * t->L), ts = (uint64_t)1 << (2 * t->T);
    // tok_off is a CSR row pointer: it starts at 0 (entries before tok_off[0] would belong to no account and were once scattered into
    // pool slot 0, the shared default account - ADVICE r2)
    if (tok_off[0] != 0) return BZK_E_ARG;
    const uint64_t m = tok_off[n];
    if (m && (!tok_index || !tok_vals)) return BZK_E_ARG;
    // validation before anything is changed
    uint64_t fresh = 0;
    {
        std::vector<uint64_t> seen(idx, idx + n);
        std::sort(seen.begin(), seen.end());
        if (std::adjacent_find(seen.begin(), seen.end()) != seen.end() || seen.back() >= n_acct) return BZK_E_ARG;
        for (uint64_t a = 0; a < n; ++a) {
            if (tok_off[a] > tok_off[a + 1] || tok_off[a + 1] > m) return BZK_E_ARG;
            std::vector<uint64_t> ti(tok_index + tok_off[a], tok_index + tok_off[a + 1]);
            std::sort(ti.begin(), ti.end());
            if (std::adjacent_find(ti.begin(), ti.end()) != ti.end() || (!ti.empty() && ti.back() >= ts)) return BZK_E_ARG;
            if (!t->slot_of.count(idx[a])) ++fresh;
        }
    }
    if (t->used + fresh > t->cap) {
        ctx->last_error = "mpn_tree: account pool exhausted";
        return BZK_E_ALLOC;
    }
    // staging (positions, values, intermediate hashes) is allocated BEFORE the slot map changes: an allocation failure leaves the
    // handle exactly as it was
    DevBuf d_pos, d_val, d_h;
    const size_t pos_cnt = std::max<size_t>(n * 4, m * 2);
    BZK_TRY(d_pos.alloc(ctx, pos_cnt * 8));
    BZK_TRY(d_val.alloc(ctx, std::max<size_t>(n * 5, m * 2) * sizeof(Fr)));
    BZK_TRY(d_h.alloc(ctx, std::max<size_t>(n, m) * sizeof(Fr)));
    // new slots are handed out from a local cursor and committed to the handle only together with the first device write
    std::vector<uint64_t> slots(n), cell_pos(n * 4), tok_pos(m * 2), g(m);
    std::vector<std::pair<uint64_t, uint64_t>> fresh_slots;
    uint64_t next = t->used;
    for (uint64_t a = 0; a < n; ++a) {
        auto it = t->slot_of.find(idx[a]);
        uint64_t s;
        if (it != t->slot_of.end()) s = it->second;
        else { s = next++; fresh_slots.push_back({idx[a], s}); }
        slots[a] = s;
        for (int j = 0; j < 4; ++j) cell_pos[a * 4 + j] = s * 4 + j;
        for (uint64_t q = tok_off[a]; q < tok_off[a + 1]; ++q) {
            g[q] = s * ts + tok_index[q];
            tok_pos[2 * q] = 2 * g[q];
            tok_pos[2 * q + 1] = 2 * g[q] + 1;
        }
    }
    for (auto& fs : fresh_slots) t->slot_of[fs.first] = fs.second;
    t->used = next;
    const int32_t st = mpn_tree_set_accounts_body(ctx, t, idx, cells, tok_vals, n, m, slots, cell_pos, tok_pos, g, d_pos, d_val, d_h);
    if (st != BZK_OK) t->poisoned = true;  // contents / leaf hashes / inner nodes may disagree from here on
    return st;
}

static int32_t mpn_tree_set_accounts_body(bzk_ctx* ctx, bzk_mpn_tree* t, const uint64_t* idx, const uint8_t* cells, const uint8_t* tok_vals,
                                          uint64_t n, uint64_t m, const std::vector<uint64_t>& slots, const std::vector<uint64_t>& cell_pos,
                                          const std::vector<uint64_t>& tok_pos, const std::vector<uint64_t>& g, DevBuf& d_pos, DevBuf& d_val,
                                          DevBuf& d_h) {
    // 1. cells
    BZK_HIP(ctx, hipMemcpyAsync(d_pos.p, cell_pos.data(), n * 4 * 8, hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipMemcpyAsync(d_val.p, cells, n * 4 * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    BZK_LAUNCH(ctx, "mpn_scatter_cells", tree4_scatter_kernel, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, t->cells,
               (const uint64_t*)d_pos.p, (const Fr*)d_val.p, n * 4);
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffers are re-used below
    if (m) {
        // 2. token slots, their H2 leaf hashes, the token sub-trees of the touched accounts
        BZK_HIP(ctx, hipMemcpyAsync(d_pos.p, tok_pos.data(), m * 2 * 8, hipMemcpyHostToDevice, ctx->stream));
        BZK_HIP(ctx, hipMemcpyAsync(d_val.p, tok_vals, m * 2 * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        BZK_LAUNCH(ctx, "mpn_scatter_tokens", tree4_scatter_kernel, dim3((unsigned)((m * 2 + 255) / 256)), dim3(256), 0, t->tok,
                   (const uint64_t*)d_pos.p, (const Fr*)d_val.p, m * 2);
        BZK_TRY(poseidon_launch(ctx, d_val.p, 2, m, d_h.p));
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BZK_HIP(ctx, hipMemcpyAsync(d_pos.p, g.data(), m * 8, hipMemcpyHostToDevice, ctx->stream));
        BZK_LAUNCH(ctx, "mpn_scatter_token_hashes", tree4_scatter_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, t->tl[t->T],
                   (const uint64_t*)d_pos.p, (const Fr*)d_h.p, m);
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> gs(g);
        std::sort(gs.begin(), gs.end());
        BZK_TRY(rehash_paths(ctx, t->tl, (int)t->T, 0, gs));
    }
    // 3. account leaves H5(nonce, wnonce, x, y, tokens_root) and the account tree above them
    BZK_HIP(ctx, hipMemcpyAsync(d_pos.p, slots.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    BZK_LAUNCH(ctx, "mpn_leaf_inputs", mpn_leaf_inputs_kernel, dim3((unsigned)((n * 5 + 255) / 256)), dim3(256), 0, (const Fr*)t->cells,
               (const Fr*)t->tl[0], (const uint64_t*)d_pos.p, n, (Fr*)d_val.p);
    BZK_TRY(poseidon_launch(ctx, d_val.p, 5, n, d_h.p));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<std::pair<uint64_t, uint64_t>> order(n);  // (account index, position): the tree update wants sorted indices
    for (uint64_t a = 0; a < n; ++a) order[a] = {idx[a], a};
    std::sort(order.begin(), order.end());
    std::vector<uint64_t> sidx(n), perm(n);
    for (uint64_t a = 0; a < n; ++a) { sidx[a] = order[a].first; perm[a] = order[a].second; }
    BZK_HIP(ctx, hipMemcpyAsync(d_pos.p, idx, n * 8, hipMemcpyHostToDevice, ctx->stream));
    Fr* leaves = t->acct->nodes + tree4_off(t->L);
    BZK_LAUNCH(ctx, "mpn_scatter_leaves", tree4_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, leaves, (const uint64_t*)d_pos.p,
               (const Fr*)d_h.p, n);
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<Fr*> lvl(t->L + 1);
    for (uint32_t k = 0; k <= t->L; ++k) lvl[k] = t->acct->nodes + tree4_off(k);
    return rehash_paths(ctx, lvl.data(), (int)t->L, 0, sidx);
}

// Batched `get_mpn_account` (src/zk/state/mod.rs:93-137): per account 5 + 2 * 4^T scalars - tx_nonce, withdraw_nonce, pub_x,
// pub_y, tokens_root (= `MpnAccount::tokens_hash`, the before_balances_hash of the transitions), then (token_id, balance) of
// every token slot.  Accounts that were never set read as the default account (all zero, default tokens_root).
int32_t bzk_mpn_tree_get_accounts(bzk_ctx* ctx, const bzk_mpn_tree* t, const uint64_t* idx, uint64_t n, uint8_t* out) {
    if (!ctx || !t || (n && (!idx || !out))) return BZK_E_ARG;
    if (t->poisoned) return BZK_E_INTERNAL;
    if (n == 0) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    const uint64_t n_acct = (uint64_t)1 << (2 * t->L), ts = (uint64_t)1 << (2 * t->T);
    const uint32_t rec = (uint32_t)(5 + 2 * ts);
    std::vector<uint64_t> slots(n);
    for (uint64_t a = 0; a < n; ++a) {
        if (idx[a] >= n_acct) return BZK_E_ARG;
        auto it = t->slot_of.find(idx[a]);
        slots[a] = it == t->slot_of.end() ? 0 : it->second;
    }
    DevBuf d_s, d_o;
    BZK_TRY(d_s.alloc(ctx, n * 8));
    BZK_TRY(d_o.alloc(ctx, n * rec * sizeof(Fr)));
    BZK_HIP(ctx, hipMemcpyAsync(d_s.p, slots.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    BZK_LAUNCH(ctx, "mpn_get", mpn_get_kernel, dim3((unsigned)((n * rec + 255) / 256)), dim3(256), 0, (const Fr*)t->cells, (const Fr*)t->tok,
               (const Fr*)t->tl[0], (const uint64_t*)d_s.p, n, rec, (uint32_t)ts, (Fr*)d_o.p);
    BZK_HIP(ctx, hipMemcpyAsync(out, d_o.p, n * rec * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

// `prove(tree_loc = [], index)`: L sibling triples per account (the src_proof / dst_proof / proof of the transitions)
int32_t bzk_mpn_tree_prove(bzk_ctx* ctx, const bzk_mpn_tree* t, const uint64_t* idx, uint64_t n, uint8_t* out) {
    if (!ctx || !t) return BZK_E_ARG;
    if (t->poisoned) return BZK_E_INTERNAL;
    return bzk_tree4_prove(ctx, t->acct, idx, n, out);
}

// `prove(tree_loc = [account, 4], token_index)`: T sibling triples (the *_balance_proof of the transitions)
int32_t bzk_mpn_tree_prove_token(bzk_ctx* ctx, const bzk_mpn_tree* t, const uint64_t* account_idx, const uint64_t* token_idx, uint64_t n,
                                 uint8_t* out) {
    if (!ctx || !t || (n && (!account_idx || !token_idx || !out))) return BZK_E_ARG;
    if (t->poisoned) return BZK_E_INTERNAL;
    if (n == 0) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    const uint64_t n_acct = (uint64_t)1 << (2 * t->L), ts = (uint64_t)1 << (2 * t->T);
    std::vector<uint64_t> g(n);
    for (uint64_t a = 0; a < n; ++a) {
        if (account_idx[a] >= n_acct || token_idx[a] >= ts) return BZK_E_ARG;
        auto it = t->slot_of.find(account_idx[a]);
        g[a] = (it == t->slot_of.end() ? 0 : it->second) * ts + token_idx[a];
    }
    const uint64_t cells = n * t->T;
    DevBuf d_g, d_o;
    BZK_TRY(d_g.alloc(ctx, n * 8));
    BZK_TRY(d_o.alloc(ctx, cells * 3 * sizeof(Fr)));
    BZK_HIP(ctx, hipMemcpyAsync(d_g.p, g.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    ForestLevels F;
    for (int k = 0; k < 9; ++k) F.lvl[k] = t->tl[k];
    BZK_LAUNCH(ctx, "mpn_prove_token", mpn_prove_token_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, F, t->T,
               (const uint64_t*)d_g.p, n, (Fr*)d_o.p);
    BZK_HIP(ctx, hipMemcpyAsync(out, d_o.p, cells * 3 * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

// `ZkStateModel::compress` (src/zk/mod.rs:392-423, `ZkStateBuilder::compress` src/zk/state/mod.rs) for a DENSE state of the MPN shape
// List{L, Struct{S, S, S, S, List{T, Struct{S, S}}}} resident in HBM - BASELINE configs[4]'s secondary instance (4^9
// accounts x (64 H2 + 21 H4 + 1 H5)).  cells_dev: 4^L x 4 scalars (nonce, withdraw nonce, x, y); tokens_dev: 4^L x 4^T x 2 scalars
// (token id, balance).  Every level is one dense launch: H2 over all token slots, T levels of H4, H5 per account, L levels of H4.
int32_t bzk_mpn_state_compress_dev(bzk_ctx* ctx, uint32_t log4_tree, uint32_t log4_token_tree, const void* cells_dev, const void* tokens_dev,
                                   uint8_t root[32]) {
    if (!ctx || !cells_dev || !tokens_dev || !root || log4_tree == 0 || log4_tree > 12 || log4_token_tree == 0 || log4_token_tree > 8)
        return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    const uint64_t n_acct = (uint64_t)1 << (2 * log4_tree), ts = (uint64_t)1 << (2 * log4_token_tree);
    if (n_acct * ts > ((uint64_t)1 << 32)) return BZK_E_ARG;
    // workspace: two ping-pong hash buffers of n_acct * ts scalars, the H5 inputs (n_acct x 5) and the account tree's nodes
    const uint64_t n_nodes = tree4_off(log4_tree);
    BZK_TRY(ws_reserve(ctx, 2 * ws_pad(n_acct * ts * sizeof(Fr)) + ws_pad(n_acct * 5 * sizeof(Fr)) + ws_pad((n_acct + n_nodes) * sizeof(Fr)) + 1024));
    WsCursor cur(ctx->ws);
    Fr* h0 = cur.take<Fr>(n_acct * ts);
    Fr* h1 = cur.take<Fr>(n_acct * ts);
    Fr* in5 = cur.take<Fr>(n_acct * 5);
    Fr* leaves = cur.take<Fr>(n_acct);
    Fr* nodes = cur.take<Fr>(n_nodes ? n_nodes : 1);
    BZK_TRY(poseidon_launch(ctx, tokens_dev, 2, n_acct * ts, h0));                                  // Struct{token, balance}
    Fr *src = h0, *dst = h1;
    for (int k = (int)log4_token_tree - 1; k >= 0; --k) {                                          // List{T, ..}: per-account sub-trees
        BZK_TRY(poseidon_launch(ctx, src, 4, n_acct << (2 * k), dst));
        std::swap(src, dst);
    }
    // account a: H5(cells[a][0..3], token root a); slot = account index in the dense layout
    {
        uint64_t* ident = (uint64_t*)dst;  // the free ping-pong buffer holds the identity slot list for the gather kernel
        hipLaunchKernelGGL(tree4_iota_kernel, dim3((unsigned)((n_acct + 255) / 256)), dim3(256), 0, ctx->stream, ident, n_acct);
        BZK_LAUNCH(ctx, "mpn_leaf_inputs", mpn_leaf_inputs_kernel, dim3((unsigned)((n_acct * 5 + 255) / 256)), dim3(256), 0, (const Fr*)cells_dev,
                   (const Fr*)src, (const uint64_t*)ident, n_acct, in5);
    }
    BZK_TRY(poseidon_launch(ctx, in5, 5, n_acct, leaves));
    const Fr* child = leaves;
    for (int k = (int)log4_tree - 1; k >= 0; --k) {
        Fr* out = nodes + tree4_off(k);
        BZK_TRY(poseidon_launch(ctx, child, 4, (uint64_t)1 << (2 * k), out));
        child = out;
    }
    BZK_HIP(ctx, hipMemcpyAsync(root, nodes, 32, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

}  // extern "C"
