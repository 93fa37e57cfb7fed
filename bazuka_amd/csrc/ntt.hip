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
// "columns", entirely in LDS: decimation-in-frequency butterflies (output rows bit-reversed inside the tile,
// undone by the store), compact twiddle table w_R^j.
//   COL   pass: element (a, c) of the tile lives at src[base + a*S + c]; the result row ka is multiplied by the
//               inter-pass twiddle w_N'^(inner * ka) (N' = R*S, two-level table: lo[e & 1023] * hi[e >> 10]) and
//               written to the same position of dst.
//   FINAL pass: rows are contiguous (S = 1); the tile takes CC rows whose OUTPUT indices are adjacent and writes
//               X[(k1 + c) + R1 * (k2 + R2 * ka)]: the digit reversal of the whole transform, in CC-wide chunks.
// The coset pre-scale (g^i, first pass) and the 1/n (and g^-k) post-scale (final pass) are fused into load / store.
// ------------------------------------------------------------------------------------------------
struct NttPass {
    const Fr* src;
    Fr* dst;
    int b, log_cc, final_pass;
    uint64_t S;        // COL: column stride = size of the inner dimension
    uint32_t R1, R2;   // FINAL: sizes of the digits already transformed
    const Fr* tw_r;    // w_R^j, j < R/2
    const Fr* tlo;     // COL: w_N'^j, j < 1024
    const Fr* thi;     // COL: w_N'^(1024 j)
    const Fr* pre;     // g^i by input index (first pass of a forward coset transform) or null
    const Fr* post;    // g^-k / n by output index (inverse coset transform) or null
    const Fr* post_c;  // 1/n (plain inverse transform) or null
};

__global__ void __launch_bounds__(256) ntt_pass_kernel(NttPass a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Fr* tile = (Fr*)smem;
    const int b = a.b, lc = a.log_cc;
    const uint32_t R = 1u << b, CC = 1u << lc, tile_n = R << lc;
    uint64_t base = 0, inner0 = 0;
    uint32_t k1 = 0, k2 = 0;
    if (!a.final_pass) {
        const uint64_t gpo = a.S >> lc;  // column groups per outer block
        const uint64_t o = blockIdx.x / gpo, cg = blockIdx.x % gpo;
        inner0 = cg << lc;
        base = o * R * a.S + inner0;
        for (uint32_t t = threadIdx.x; t < tile_n; t += blockDim.x) {
            const uint32_t c = t & (CC - 1), r = t >> lc;
            const uint64_t addr = base + (uint64_t)r * a.S + c;
            Fr v = a.src[addr];
            if (a.pre) v = fr_mul(v, a.pre[addr]);
            tile[t] = v;
        }
    } else {
        const uint32_t groups = a.R1 >> lc;
        k2 = blockIdx.x / groups;
        k1 = (blockIdx.x % groups) << lc;
        for (uint32_t t = threadIdx.x; t < tile_n; t += blockDim.x) {
            const uint32_t r = t & (R - 1), c = t >> b;
            const uint64_t addr = ((uint64_t)(k1 + c) * a.R2 + k2) * R + r;
            Fr v = a.src[addr];
            if (a.pre) v = fr_mul(v, a.pre[addr]);
            tile[(r << lc) + c] = v;
        }
    }
    __syncthreads();
    for (int s = b - 1; s >= 0; --s) {
        const uint32_t m = 1u << s;
        for (uint32_t q = threadIdx.x; q < tile_n / 2; q += blockDim.x) {
            const uint32_t c = q & (CC - 1), bq = q >> lc;
            const uint32_t j = bq & (m - 1);
            const uint32_t k = ((bq >> s) << (s + 1)) + j;
            const uint32_t i0 = (k << lc) + c, i1 = ((k + m) << lc) + c;
            const Fr u = tile[i0], v = tile[i1];
            tile[i0] = fe_add<FrParams>(u, v);
            Fr d = fe_sub<FrParams>(u, v);
            if (j) d = fr_mul(d, a.tw_r[j << (b - 1 - s)]);
            tile[i1] = d;
        }
        __syncthreads();
    }
    for (uint32_t t = threadIdx.x; t < tile_n; t += blockDim.x) {
        const uint32_t c = t & (CC - 1), p = t >> lc;
        const uint32_t ka = bitrev(p, b);
        Fr v = tile[t];
        if (!a.final_pass) {
            const uint64_t e = (inner0 + c) * ka;
            if (e) {
                Fr w = a.tlo[e & 1023];
                if (e >> 10) w = fr_mul(w, a.thi[e >> 10]);
                v = fr_mul(v, w);
            }
            a.dst[base + (uint64_t)ka * a.S + c] = v;
        } else {
            const uint64_t k = (uint64_t)(k1 + c) + (uint64_t)a.R1 * (k2 + (uint64_t)a.R2 * ka);
            if (a.post) v = fr_mul(v, a.post[k]);
            else if (a.post_c) v = fr_mul(v, *a.post_c);
            a.dst[k] = v;
        }
    }
}

// per-size plan: digit split, twiddle tables for both directions, coset tables
struct NttTables {
    int nb = 0, b[3] = {0, 0, 0};
    Fr* tw_r[2][3] = {};   // [inverse][pass]: w_R^j
    Fr* tlo[2][2] = {};    // [inverse][col pass]
    Fr* thi[2][2] = {};
    Fr* g_pow = nullptr;        // g^j, j < n
    Fr* ginv_scaled = nullptr;  // g^-j / n
    Fr* n_inv = nullptr;        // single element 1/n
};

static int ntt_bmax() {
    static int v = [] {
        const char* e = getenv("BZK_NTT_BMAX");
        int x = e ? atoi(e) : 10;
        return x < 4 ? 4 : (x > 10 ? 10 : x);
    }();
    return v;
}
static uint32_t ntt_tile_elems() {  // LDS tile in field elements (32 B each); default 1024 = 32 KiB -> 5 workgroups per CU (measured best, run 19)
    static uint32_t v = [] {
        const char* e = getenv("BZK_NTT_TILE");
        uint32_t x = e ? (uint32_t)atoi(e) : 1024u;
        return x < 1024u ? 1024u : (x > 4096u ? 4096u : x);
    }();
    return v;
}

static Fr host_pow_u64(const Fr& base, uint64_t e) { return host_pow(base, e); }

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
    void* p;
    for (int d = 0; d < 2; ++d) {
        for (int k = 0; k < T->nb; ++k) {
            const uint64_t R = (uint64_t)1 << T->b[k];
            BZK_TRY(build_pow_table(ctx, host_pow_u64(dirs[d], n / R), R / 2 ? R / 2 : 1, &p));
            T->tw_r[d][k] = (Fr*)p;
        }
        // inter-pass twiddles: pass 0 over the whole transform (N' = n), pass 1 (three-digit plans) over N' = n / R1
        for (int k = 0; k + 1 < T->nb; ++k) {
            const uint64_t np = k == 0 ? n : n >> T->b[0];
            const Fr base = k == 0 ? dirs[d] : host_pow_u64(dirs[d], (uint64_t)1 << T->b[0]);
            BZK_TRY(build_pow_table(ctx, base, 1024, &p));
            T->tlo[d][k] = (Fr*)p;
            BZK_TRY(build_pow_table(ctx, host_pow_u64(base, 1024), (np + 1023) / 1024, &p));
            T->thi[d][k] = (Fr*)p;
        }
    }
    BZK_TRY(build_pow_table(ctx, g, n, &p));
    T->g_pow = (Fr*)p;
    // g^-j / n: scale the table's first-level factors instead of a pass over the table
    {
        const Fr gi = fe_inv<FrParams>(g), ninv = fe_inv<FrParams>(host_from_u64(n));
        BZK_TRY(build_pow_table(ctx, gi, n, &p));
        T->ginv_scaled = (Fr*)p;
        BZK_HIP(ctx, hipMalloc(&p, sizeof(Fr)));
        BZK_HIP(ctx, hipMemcpyAsync(p, &ninv, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        T->n_inv = (Fr*)p;
        BZK_LAUNCH(ctx, "ntt_scale", ntt_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, T->ginv_scaled, n,
                   (const Fr*)T->n_inv, (const Fr*)nullptr);
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->ntt_tw[log_n][1] = T;
    *out = T;
    return BZK_OK;
}

int32_t ntt_run(bzk_ctx* ctx, void* data_dev, uint32_t log_n, int inverse, int coset) {
    if (log_n > 28) return BZK_E_ARG;
    const uint64_t n = (uint64_t)1 << log_n;
    if (log_n == 0) return BZK_OK;
    NttTables* T;
    BZK_TRY(ntt_tables(ctx, (int)log_n, &T));
    Fr* data = (Fr*)data_dev;
    Fr* tmp = data;
    if (T->nb > 1) {
        BZK_TRY(ws_reserve(ctx, ws_pad(n * sizeof(Fr)) + 512));
        tmp = (Fr*)ctx->ws;
    }
    const int d = inverse ? 1 : 0;
    const uint32_t tile_max = ntt_tile_elems();
    static bool lds_attr_set = false;  // tiles above 64 KiB of dynamic LDS need the opt-in
    if (!lds_attr_set) {
        (void)hipFuncSetAttribute((const void*)ntt_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        lds_attr_set = true;
    }
    uint64_t S = n;
    for (int k = 0; k < T->nb; ++k) {
        const int b = T->b[k];
        const uint64_t R = (uint64_t)1 << b;
        S >>= b;
        NttPass a;
        memset(&a, 0, sizeof a);
        a.b = b;
        a.final_pass = k == T->nb - 1;
        a.src = k == 0 ? data : tmp;
        a.dst = a.final_pass ? data : tmp;
        a.tw_r = T->tw_r[d][k];
        a.pre = (k == 0 && coset && !inverse) ? T->g_pow : nullptr;
        uint64_t lanes;  // how many adjacent columns exist
        if (!a.final_pass) {
            a.S = S;
            a.tlo = T->tlo[d][k];
            a.thi = T->thi[d][k];
            lanes = S;
        } else {
            a.R1 = T->nb >= 2 ? 1u << T->b[0] : 1u;
            a.R2 = T->nb == 3 ? 1u << T->b[1] : 1u;
            a.post = (inverse && coset) ? T->ginv_scaled : nullptr;
            a.post_c = (inverse && !coset) ? T->n_inv : nullptr;
            lanes = a.R1;
        }
        int lc = 0;
        while (((uint64_t)2 << lc) <= lanes && (R << (lc + 1)) <= tile_max) ++lc;
        a.log_cc = lc;
        const uint64_t tiles = n >> (b + lc);
        if (tiles > 0x7fffffffull) return BZK_E_ARG;
        if (a.final_pass) {
            BZK_LAUNCH(ctx, "ntt_final", ntt_pass_kernel, dim3((unsigned)tiles), dim3(256), (size_t)sizeof(Fr) << (b + lc), a);
        } else {
            BZK_LAUNCH(ctx, "ntt_col", ntt_pass_kernel, dim3((unsigned)tiles), dim3(256), (size_t)sizeof(Fr) << (b + lc), a);
        }
    }
    return BZK_OK;
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
            }
        }
        (void)hipFree(T->g_pow);
        (void)hipFree(T->ginv_scaled);
        (void)hipFree(T->n_inv);
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
