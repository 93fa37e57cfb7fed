// K3: radix-2 NTT over Fr for gfx950.
//
// Replaces bellman 0.14 `EvaluationDomain::{fft, ifft, coset_fft, icoset_fft}` (third-party crate
// reached from `create_random_proof`, /root/reference/src/mpn/circuits/test.rs:135,175,215).
// omega = 7^((r-1)/2^32)^(2^(32-log_n)), coset shift g = 7 (ZkScalar generator,
// /root/reference/src/zk/mod.rs:204).  Natural order in and out.
//
// Structure: bit-reversal permutation (the coset pre-scale g^j is fused into it), then the
// butterfly stages.  The first LOCAL_LOG stages (spans < 2^LOCAL_LOG) run inside one workgroup on
// an LDS-resident tile (one HBM round trip for all of them); the remaining stages are one
// streaming pass each.  The 1/n (and g^-j) scaling of the inverse transforms is fused into the last
// pass.  Twiddles w^j come from an HBM/L2-resident table built once per (log_n, direction) and
// cached in the context.  HBM traffic per transform of size n: 64 B * n * (2 + max(0, log_n - 10)).
#include <string.h>

#include <vector>

#include "bzk_field.cuh"
#include "bzk_internal.h"

namespace bzk {

static constexpr int LOCAL_LOG = 10;  // 1024-point LDS tile = 32 KiB

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

// out[rev(i)] = in[i] * (coset ? g^i : 1)
__global__ void __launch_bounds__(256) ntt_bitrev_kernel(const Fr* __restrict__ in, Fr* __restrict__ out, int log_n,
                                                         const Fr* __restrict__ gpow /*or null*/) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((uint64_t)1 << log_n)) return;
    Fr v = in[i];
    if (gpow) v = fr_mul(v, gpow[i]);
    out[bitrev((uint32_t)i, log_n)] = v;
}

// Stages 0 .. local_log-1 on tiles of 2^local_log consecutive (bit-reversed-order) elements in LDS.
// tw has n/2 entries w^j; stage s uses w^(j * n / 2^(s+1)).
__global__ void __launch_bounds__(256) ntt_local_kernel(Fr* __restrict__ data, int log_n, int local_log,
                                                        const Fr* __restrict__ tw, const Fr* __restrict__ final_scale,
                                                        const Fr* __restrict__ ginv_pow, int last) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Fr* tile = (Fr*)smem;
    const uint32_t tile_n = 1u << local_log;
    const uint64_t base = (uint64_t)blockIdx.x * tile_n;
    for (uint32_t i = threadIdx.x; i < tile_n; i += blockDim.x) tile[i] = data[base + i];
    __syncthreads();
    for (int s = 0; s < local_log; ++s) {
        const uint32_t m = 1u << s;
        const int tw_shift = log_n - 1 - s;
        for (uint32_t b = threadIdx.x; b < tile_n / 2; b += blockDim.x) {
            const uint32_t j = b & (m - 1);
            const uint32_t k = ((b >> s) << (s + 1)) + j;
            Fr u = tile[k];
            Fr v = fr_mul(tile[k + m], tw[(uint64_t)j << tw_shift]);
            tile[k] = fe_add<FrParams>(u, v);
            tile[k + m] = fe_sub<FrParams>(u, v);
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < tile_n; i += blockDim.x) {
        Fr v = tile[i];
        if (last && final_scale) {
            v = fr_mul(v, *final_scale);
            if (ginv_pow) v = fr_mul(v, ginv_pow[base + i]);
        }
        data[base + i] = v;
    }
}

// one global stage s (m = 2^s)
__global__ void __launch_bounds__(256) ntt_stage_kernel(Fr* __restrict__ data, int log_n, int s, const Fr* __restrict__ tw,
                                                        const Fr* __restrict__ final_scale, const Fr* __restrict__ ginv_pow,
                                                        int last) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= ((uint64_t)1 << (log_n - 1))) return;
    const uint64_t m = (uint64_t)1 << s;
    const uint64_t j = b & (m - 1);
    const uint64_t k = ((b >> s) << (s + 1)) + j;
    Fr u = data[k];
    Fr v = fr_mul(data[k + m], tw[j << (log_n - 1 - s)]);
    Fr x = fe_add<FrParams>(u, v), y = fe_sub<FrParams>(u, v);
    if (last && final_scale) {
        x = fr_mul(x, *final_scale);
        y = fr_mul(y, *final_scale);
        if (ginv_pow) {
            x = fr_mul(x, ginv_pow[k]);
            y = fr_mul(y, ginv_pow[k + m]);
        }
    }
    data[k] = x;
    data[k + m] = y;
}

__global__ void __launch_bounds__(256) ntt_scale_kernel(Fr* __restrict__ data, uint64_t n, const Fr* __restrict__ scale,
                                                        const Fr* __restrict__ pw) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = fr_mul(data[i], *scale);
    if (pw) v = fr_mul(v, pw[i]);
    data[i] = v;
}

struct NttTables {
    Fr* tw_fwd;    // w^j, j < n/2
    Fr* tw_inv;    // w^-j
    Fr* g_pow;     // g^j, j < n
    Fr* ginv_pow;  // g^-j
    Fr* n_inv;     // single element 1/n
};

// ctx->ntt_tw[log_n][0] -> device NttTables payload pointers (host struct kept in [1])
static int32_t ntt_tables(bzk_ctx* ctx, int log_n, NttTables** out) {
    if (ctx->ntt_tw[log_n][1]) {
        *out = (NttTables*)ctx->ntt_tw[log_n][1];
        return BZK_OK;
    }
    const uint64_t n = (uint64_t)1 << log_n;
    NttTables* T = new NttTables();
    Fr w = host_omega(log_n), g = host_from_u64(7);
    Fr wi = fe_inv<FrParams>(w), gi = fe_inv<FrParams>(g);
    void* p;
    BZK_TRY(build_pow_table(ctx, w, n / 2, &p));   T->tw_fwd = (Fr*)p;
    BZK_TRY(build_pow_table(ctx, wi, n / 2, &p));  T->tw_inv = (Fr*)p;
    BZK_TRY(build_pow_table(ctx, g, n, &p));       T->g_pow = (Fr*)p;
    BZK_TRY(build_pow_table(ctx, gi, n, &p));      T->ginv_pow = (Fr*)p;
    Fr ninv = fe_inv<FrParams>(host_from_u64(n));
    BZK_HIP(ctx, hipMalloc(&p, sizeof(Fr)));
    BZK_HIP(ctx, hipMemcpyAsync(p, &ninv, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    T->n_inv = (Fr*)p;
    ctx->ntt_tw[log_n][1] = T;
    // remember one device pointer in slot [0] so ctx_destroy can free at least the biggest table;
    // the others are released in ntt_free_tables (called from bzk_ctx_destroy through slot [1])
    *out = T;
    return BZK_OK;
}

int32_t ntt_run(bzk_ctx* ctx, void* data_dev, uint32_t log_n, int inverse, int coset) {
    if (log_n > 28) return BZK_E_ARG;
    const uint64_t n = (uint64_t)1 << log_n;
    if (log_n == 0) return BZK_OK;
    NttTables* T;
    BZK_TRY(ntt_tables(ctx, (int)log_n, &T));
    BZK_TRY(ws_reserve(ctx, ws_pad(n * sizeof(Fr)) + 512));
    Fr* tmp = (Fr*)ctx->ws;
    Fr* data = (Fr*)data_dev;
    const Fr* tw = inverse ? T->tw_inv : T->tw_fwd;
    const unsigned gb = (unsigned)((n + 255) / 256);
    BZK_LAUNCH(ctx, "ntt_bitrev", ntt_bitrev_kernel, dim3(gb), dim3(256), 0, (const Fr*)data, tmp, (int)log_n,
               (const Fr*)((coset && !inverse) ? T->g_pow : nullptr));
    const Fr* fscale = inverse ? T->n_inv : nullptr;
    const Fr* gip = (inverse && coset) ? T->ginv_pow : nullptr;
    const int local = (int)log_n < LOCAL_LOG ? (int)log_n : LOCAL_LOG;
    {
        const int last = local == (int)log_n;
        BZK_LAUNCH(ctx, "ntt_local", ntt_local_kernel, dim3((unsigned)(n >> local)), dim3(256), (size_t)sizeof(Fr) << local, tmp,
                   (int)log_n, local, tw, fscale, gip, last);
    }
    for (int s = local; s < (int)log_n; ++s) {
        const int last = s == (int)log_n - 1;
        BZK_LAUNCH(ctx, "ntt_stage", ntt_stage_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, tmp, (int)log_n, s, tw,
                   fscale, gip, last);
    }
    BZK_HIP(ctx, hipMemcpyAsync(data, tmp, n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    return BZK_OK;
}

void ntt_free_tables(bzk_ctx* ctx) {
    for (int i = 0; i <= 32; ++i) {
        NttTables* T = (NttTables*)ctx->ntt_tw[i][1];
        if (!T) continue;
        (void)hipFree(T->tw_fwd);
        (void)hipFree(T->tw_inv);
        (void)hipFree(T->g_pow);
        (void)hipFree(T->ginv_pow);
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
