// Micro-benchmark: ONE round of batched-affine point addition (shared inversion, Montgomery's trick) against the XYZZ mixed
// addition the bucket accumulation uses - the measurement VERDICT r1 item 6 asks for instead of the paper estimate of DESIGN 6b.
// Uses the product's own field (bzk_fp28.cuh: 14 x 28-bit limbs, the product as a resident function, binary-GCD inversion).
//
//   xyzz     : what msm_accumulate does: every lane folds a run of affine points into an XYZZ accumulator in registers
//              (8 products + 2 squares per addition, 112 B read per addition, nothing written)
//   ba<G>    : the most favourable form of a batched-affine round: the pairs to be added lie CONTIGUOUS in memory (no sorted-index
//              gathers, no bucket bookkeeping, no P = +-Q cases), a workgroup of 256 lanes shares one inversion over 256 x G pairs:
//                phase 1  per pair d = x2 - x1, running product (1 product), prefix parked in global scratch
//                phase 2  prefix and suffix product scans over the 256 lane products through LDS (2 x 8 products), lane 0 inverts
//                         the total (binary GCD, ~75 product-times during which the other 255 lanes wait)
//                phase 3  back-substitution (2 products), lambda = dy / dx, x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1
//                         (2 products + 1 square): 6 products per addition + (16 + 75 + 3) / G per lane
//              memory per addition: 2 x 112 B read twice (second time from L2), 56 B prefix written + read, 112 B written
//   ba_free  : the same with the inversion replaced by a constant (the bound if inversions were decoupled into another kernel)
// Reported: additions per second.  A real MSM round adds the pairing bookkeeping, gathers through the sorted index list in the
// first round and one kernel triple per halving of the bucket populations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_batched_affine tools/ubench_batched_affine.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define BZK_FP_NOINLINE 1
#include "../bazuka_amd/csrc/bzk_field.cuh"
#include "../bazuka_amd/csrc/bzk_fp28.cuh"
using namespace bzk;
using namespace bzk::fp28;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fill(G1A28* pts, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = i * 0x9E3779B97F4A7C15ull + 12345;
    G1A28 p;
    for (int k = 0; k < 14; ++k) {
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        p.x.l[k] = (uint32_t)z & (k == 13 ? 0xffffu : MASK);
        p.y.l[k] = (uint32_t)(z >> 32) & (k == 13 ? 0xffffu : MASK);
    }
    pts[i] = p;  // arbitrary field elements below p: timing does not need curve points (distinct x are what matters)
}

__device__ __forceinline__ G1A28 load_pt(const G1A28* base, uint64_t idx) {
    const uint4* p = (const uint4*)(base + idx);
    uint4 v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = p[k];
    G1A28 a;
    uint32_t* dst = a.x.l;
#pragma unroll
    for (int k = 0; k < 7; ++k) { dst[4 * k] = v[k].x; dst[4 * k + 1] = v[k].y; dst[4 * k + 2] = v[k].z; dst[4 * k + 3] = v[k].w; }
    return a;
}

// baseline: runs of `run` mixed additions per lane, points gathered through a pseudo-random index (like the sorted pair list)
__global__ void __launch_bounds__(128, 2) k_xyzz(const G1A28* __restrict__ pts, uint64_t n_pts, uint32_t run, G1X28* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    G1X28 acc = g1x28::identity();
    uint64_t z = t * 0x9E3779B97F4A7C15ull + 7;
    for (uint32_t j = 0; j < run; ++j) {
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        const G1A28 p = load_pt(pts, z % n_pts);
        g1x28::add_mixed(acc, p, (z >> 63) != 0);
    }
    out[t] = acc;
}

template <int G, bool FREE_INV>
__global__ void __launch_bounds__(256, 2) k_ba(const G1A28* __restrict__ pairs, Fp28* __restrict__ pref, G1A28* __restrict__ out) {
    __shared__ Fp28 sh[256];
    const uint64_t lane = threadIdx.x;
    const uint64_t blk_base = (uint64_t)blockIdx.x * 256 * G;  // pair index of the workgroup's first pair
    // phase 1: pair (blk_base + g * 256 + lane): consecutive lanes read consecutive pairs
    Fp28 pr = one();
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const uint64_t q = blk_base + (uint64_t)g * 256 + lane;
        const G1A28 a = load_pt(pairs, 2 * q), b = load_pt(pairs, 2 * q + 1);
        const Fp28 d = norm(sub<3>(b.x, a.x));
        pref[q] = pr;
        pr = mul(pr, d);
    }
    // phase 2: prefix E_l (exclusive) and suffix F_l (exclusive) products of the lane totals, total T
    Fp28 inc = pr;
    sh[lane] = inc;
    __syncthreads();
#pragma unroll 1
    for (int d = 1; d < 256; d <<= 1) {
        Fp28 o = sh[lane >= (uint64_t)d ? lane - d : lane];
        __syncthreads();
        if (lane >= (uint64_t)d) inc = mul(inc, o);
        sh[lane] = inc;
        __syncthreads();
    }
    const Fp28 T = sh[255];
    const Fp28 E = lane ? sh[lane - 1] : one();
    __syncthreads();
    Fp28 suf = pr;
    sh[lane] = suf;
    __syncthreads();
#pragma unroll 1
    for (int d = 1; d < 256; d <<= 1) {
        Fp28 o = sh[lane + d < 256 ? lane + d : lane];
        __syncthreads();
        if (lane + d < 256) suf = mul(suf, o);
        sh[lane] = suf;
        __syncthreads();
    }
    const Fp28 F = lane + 1 < 256 ? sh[lane + 1] : one();
    __syncthreads();
    if (lane == 0) sh[0] = FREE_INV ? T : inv_gcd(T);
    __syncthreads();
    Fp28 inv_run = mul(mul(sh[0], E), F);  // 1 / (product of this lane's d)
    // phase 3: back-substitution and the additions
#pragma unroll 1
    for (int g = G - 1; g >= 0; --g) {
        const uint64_t q = blk_base + (uint64_t)g * 256 + lane;
        const G1A28 a = load_pt(pairs, 2 * q), b = load_pt(pairs, 2 * q + 1);
        const Fp28 d = norm(sub<3>(b.x, a.x));
        const Fp28 inv_d = mul(inv_run, pref[q]);
        inv_run = mul(inv_run, d);
        const Fp28 lam = mul(norm(sub<3>(b.y, a.y)), inv_d);
        const Fp28 x3 = norm(sub<3>(sub<3>(sqr(lam), a.x), b.x));
        const Fp28 y3 = norm(sub<3>(mul(lam, sub<12>(a.x, x3)), a.y));
        out[q] = {x3, y3};
    }
}

// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern"): the access
// pattern of msm_accumulate and nothing else - every lane gathers `run` records of 112 B (7 x 16-byte loads) at pseudo-random record
// indices of an array of n_pts records.  Requested bytes = lanes * run * 112, printed; rocprofv3 --pmc FETCH_SIZE of this kernel gives
// the counter's reading for it (tools/pmc_traffic.py --calib).
__global__ void __launch_bounds__(256) k_gather_only(const G1A28* __restrict__ pts, uint64_t n_pts, uint32_t run, uint32_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t z = t * 0x9E3779B97F4A7C15ull + 7;
    uint32_t acc = 0;
    for (uint32_t j = 0; j < run; ++j) {
        z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        const G1A28 p = load_pt(pts, z % n_pts);
#pragma unroll
        for (int k = 0; k < 14; ++k) acc += p.x.l[k] ^ p.y.l[k];
    }
    out[t] = acc;
}

template <class Launch>
static float time_ms(Launch&& launch, int reps = 3) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a, 0);
        launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

template <int G, bool FREE>
static int run_ba(const G1A28* pairs, Fp28* pref, G1A28* out, uint64_t n_pairs) {
    const uint64_t per_blk = 256ull * G;
    const unsigned blocks = (unsigned)(n_pairs / per_blk);
    const float ms = time_ms([&] { hipLaunchKernelGGL((k_ba<G, FREE>), dim3(blocks), dim3(256), 0, 0, pairs, pref, out); });
    const double adds = (double)blocks * per_blk;
    printf("ba%s G=%-3d  %8.3f ms  %7.3f G additions/s   (%.1f products per addition incl. the shared part)\n", FREE ? "_free" : "     ", G, ms,
           adds / ms / 1e6, 6.0 + (FREE ? 19.0 : 94.0) / G);
    return 0;
}

static int calib() {
    const uint64_t n_big = 1ull << 24;  // 1.9 GB of records: beyond the 256 MB Infinity Cache
    G1A28* pts;
    uint32_t* out;
    const uint64_t lanes = 1ull << 20;
    CK(hipMalloc(&pts, n_big * sizeof(G1A28)));
    CK(hipMalloc(&out, lanes * 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n_big + 255) / 256)), dim3(256), 0, 0, pts, n_big);
    CK(hipDeviceSynchronize());
    for (int pass = 0; pass < 2; ++pass) {
        const uint64_t n_pts = pass == 0 ? (1ull << 20) : n_big;
        const uint32_t run = 16;
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_gather_only, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, pts, n_pts, run, out); }, 2);
        printf("calib gather: array %llu records (%.0f MB), %llu lanes x %u records: requested %llu bytes per launch, %.3f ms (%.0f GB/s)\n",
               (unsigned long long)n_pts, n_pts * 112.0 / 1e6, (unsigned long long)lanes, run, (unsigned long long)(lanes * run * 112ull), ms,
               lanes * run * 112.0 / ms / 1e6);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'c') return calib();
    const uint64_t n_pairs = 1ull << 23;  // the first halving round of a 2^20-point MSM adds 8.4 M pairs
    G1A28 *pairs, *out;
    Fp28* pref;
    G1X28* xo;
    CK(hipMalloc(&pairs, 2 * n_pairs * sizeof(G1A28)));
    CK(hipMalloc(&out, n_pairs * sizeof(G1A28)));
    CK(hipMalloc(&pref, n_pairs * sizeof(Fp28)));
    const uint64_t lanes = 1ull << 19;
    CK(hipMalloc(&xo, lanes * sizeof(G1X28)));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((2 * n_pairs + 255) / 256)), dim3(256), 0, 0, pairs, 2 * n_pairs);
    CK(hipDeviceSynchronize());
    {
        const uint32_t run = 32;
        const uint64_t n_pts = 1ull << 20;  // 117 MB of bases: Infinity-Cache resident, as in the MSM
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_xyzz, dim3((unsigned)(lanes / 128)), dim3(128), 0, 0, pairs, n_pts, run, xo); });
        printf("xyzz mixed add, runs of %u, %llu lanes: %8.3f ms  %7.3f G additions/s   (10 products per addition)\n", run,
               (unsigned long long)lanes, ms, (double)lanes * run / ms / 1e6);
    }
    run_ba<8, false>(pairs, pref, out, n_pairs);
    run_ba<16, false>(pairs, pref, out, n_pairs);
    run_ba<32, false>(pairs, pref, out, n_pairs);
    run_ba<64, false>(pairs, pref, out, n_pairs);
    run_ba<128, false>(pairs, pref, out, n_pairs);
    run_ba<16, true>(pairs, pref, out, n_pairs);
    run_ba<64, true>(pairs, pref, out, n_pairs);
    CK(hipDeviceSynchronize());
    return 0;
}
