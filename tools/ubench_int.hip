// Integer-ALU micro-benchmark for gfx950: measures the issue rate of the instructions a 381-bit
// Montgomery product is made of, and the product itself.  This is the "binding roofline" of the MSM /
// Poseidon kernels (not in the local guides - measure it).  Standalone tool:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_int tools/ubench_int.hip && ./tools/ubench_int
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../bazuka_amd/csrc/bzk_field.cuh"
#include "../bazuka_amd/csrc/bzk_fp28.cuh"
#include "../bazuka_amd/csrc/bzk_fr29.cuh"
using namespace bzk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[ILP];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = (uint64_t)x * (uint32_t)(y + i) + acc[i];   // v_mad_u64_u32
        x = (uint32_t)acc[0];
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_mullo(uint32_t* out, uint32_t a, int iters) {
    uint32_t acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = acc[i] * (acc[i] | 1);   // v_mul_lo_u32
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_mulhi(uint32_t* out, uint32_t a, int iters) {
    uint32_t acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = __umulhi(acc[i], acc[i] | 0x80000001u);   // v_mul_hi_u32
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_add64(uint64_t* out, uint64_t a, int iters) {
    uint64_t acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] += (acc[i] >> 7) ^ a;   // shift + xor + 64-bit add (2 adds)
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_mad24(uint32_t* out, uint32_t a, int iters) {
    uint32_t acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = (acc[i] & 0xffffffu) * (a & 0xffffffu) + acc[i];  // v_mad_u32_u24
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_fma64(double* out, double a, int iters) {
    double acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = __builtin_fma(acc[i], a, 1.0);   // v_fma_f64
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class P, int ILP>
__global__ void k_femul(uint32_t* out, int iters) {
    Fe<P> x[ILP], y;
#pragma unroll
    for (int i = 0; i < P::N; ++i) y.l[i] = P::R2[i] ^ threadIdx.x;
    y.l[P::N - 1] &= 0x0fffffff;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
#pragma unroll
        for (int i = 0; i < P::N; ++i) x[k].l[i] = P::ONE[i] + k;
        x[k].l[P::N - 1] &= 0x0fffffff;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) x[k] = fe_mul<P>(x[k], y);
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s ^= x[k].l[0] ^ x[k].l[P::N - 1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same product through a real function call (by-value args travel in VGPRs)
__device__ __noinline__ Fp fp_mul_call(Fp a, Fp b) { return fe_mul<FpParams>(a, b); }
template <int ILP>
__global__ void k_femul_call(uint32_t* out, int iters) {
    Fp x[ILP], y;
#pragma unroll
    for (int i = 0; i < 12; ++i) y.l[i] = FpParams::R2[i] ^ threadIdx.x;
    y.l[11] &= 0x0fffffff;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
#pragma unroll
        for (int i = 0; i < 12; ++i) x[k].l[i] = FpParams::ONE[i] + k;
        x[k].l[11] &= 0x0fffffff;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) x[k] = fp_mul_call(x[k], y);
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s ^= x[k].l[0] ^ x[k].l[11];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// reduced-radix prototype: 14 x 28-bit limbs, column accumulators in 64 bits, no carry chains
struct UFp28 { uint32_t l[14]; };
__device__ constexpr uint32_t P28[14] = {0xfffaaab, 0xffeffff, 0x53ffffb, 0x3fffeb1, 0xf6241ea, 0x0a0f6b0, 0x12bf673,
                                         0x084f385, 0x764774b, 0x034bacd, 0xba7b643, 0x069a4b1, 0xea397fe, 0x01a0111};
template <bool CALL>
__device__ __forceinline__ UFp28 mul28_body(const UFp28& a, const UFp28& b) {
    constexpr int N = 14, W = 28;
    constexpr uint32_t MASK = (1u << W) - 1, PINV = 0xffcfffd;
    uint64_t c[2 * N];
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a.l[i] * b.l[j];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = ((uint32_t)c[i] * PINV) & MASK;
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)m * P28[j];
        c[i + 1] += c[i] >> W;
    }
    UFp28 r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        c[k + 1] += c[k] >> W;
        r.l[k - N] = (uint32_t)c[k] & MASK;
    }
    r.l[N - 1] = (uint32_t)c[2 * N - 1] & MASK;
    return r;
}
__device__ __noinline__ UFp28 mul28_call(UFp28 a, UFp28 b) { return mul28_body<true>(a, b); }
template <int ILP, bool CALL>
__global__ void k_mul28(uint32_t* out, int iters) {
    UFp28 x[ILP], y;
#pragma unroll
    for (int i = 0; i < 14; ++i) y.l[i] = (P28[i] ^ threadIdx.x) & 0xfffffff;
#pragma unroll
    for (int k = 0; k < ILP; ++k)
#pragma unroll
        for (int i = 0; i < 14; ++i) x[k].l[i] = (P28[13 - i] + k) & 0xfffffff;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) x[k] = CALL ? mul28_call(x[k], y) : mul28_body<false>(x[k], y);
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s ^= x[k].l[0] ^ x[k].l[13];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the library's own products (vector-argument calls): chains of dependent products, the shape of an XYZZ add
template <int ILP>
__global__ void k_mul28_lib(uint32_t* out, int iters) {
    bzk::Fp28 x[ILP], y;
#pragma unroll
    for (int i = 0; i < 14; ++i) y.l[i] = (P28[i] ^ threadIdx.x) & 0xfffffff;
#pragma unroll
    for (int k = 0; k < ILP; ++k)
#pragma unroll
        for (int i = 0; i < 14; ++i) x[k].l[i] = (P28[13 - i] + k) & 0xfffffff;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) x[k] = bzk::fp28::mul(x[k], y);
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s ^= x[k].l[0] ^ x[k].l[13];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_mul29_lib(uint32_t* out, int iters) {
    bzk::Fr29 x[ILP], y;
#pragma unroll
    for (int i = 0; i < 9; ++i) y.l[i] = (bzk::fr29::R.v[i] ^ threadIdx.x) & 0x1fffffff;
#pragma unroll
    for (int k = 0; k < ILP; ++k)
#pragma unroll
        for (int i = 0; i < 9; ++i) x[k].l[i] = (bzk::fr29::R.v[8 - i] + k) & 0x1fffffff;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) x[k] = bzk::fr29::mul(x[k], y);
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s ^= x[k].l[0] ^ x[k].l[8];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_kernel(F launch, int reps = 3) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();  // warm
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("device: %s, CUs %d, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    void* buf; CK(hipMalloc(&buf, 256 << 20));
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 4096;
    const double lanes = (double)blocks * threads;
    {
        auto run = [&](const char* nm, auto kern, auto arg, int ilp) {
            double t = time_kernel([&] { kern(arg); });
            printf("%-30s ILP=%-2d %10.2f Gop/s   (%.3f ms)\n", nm, ilp, lanes * iters * ilp / t / 1e9, t * 1e3);
        };
        run("v_mad_u64_u32", [&](int) { hipLaunchKernelGGL(k_mad64<1>, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 3u, 5u, iters); }, 0, 1);
        run("v_mad_u64_u32", [&](int) { hipLaunchKernelGGL(k_mad64<4>, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 3u, 5u, iters); }, 0, 4);
        run("v_mad_u64_u32", [&](int) { hipLaunchKernelGGL(k_mad64<12>, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 3u, 5u, iters); }, 0, 12);
        run("v_mul_lo_u32", [&](int) { hipLaunchKernelGGL(k_mullo<8>, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 3u, iters); }, 0, 8);
        run("v_mul_hi_u32", [&](int) { hipLaunchKernelGGL(k_mulhi<8>, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 3u, iters); }, 0, 8);
        run("v_mad_u32_u24", [&](int) { hipLaunchKernelGGL(k_mad24<8>, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 3u, iters); }, 0, 8);
        run("add64+shift+xor (4 valu)", [&](int) { hipLaunchKernelGGL(k_add64<8>, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, (uint64_t)3, iters); }, 0, 8);
        run("v_fma_f64", [&](int) { hipLaunchKernelGGL(k_fma64<8>, dim3(blocks), dim3(threads), 0, 0, (double*)buf, 1.0000001, iters); }, 0, 8);
    }
    {
        const int it2 = 512;
        auto run = [&](const char* nm, auto kern, int ilp, int bl, int th) {
            double t = time_kernel([&] { kern(bl, th); });
            printf("%-30s ILP=%-2d blocks/CU=%-2d %10.3f Gmul/s (%.3f ms)\n", nm, ilp, bl / prop.multiProcessorCount, (double)bl * th * it2 * ilp / t / 1e9, t * 1e3);
        };
        for (int bpc : {1, 2, 4, 8}) {
            int bl = prop.multiProcessorCount * bpc;
            run("Fp mul (12x32 CIOS)", [&](int b, int t) { hipLaunchKernelGGL((k_femul<FpParams, 1>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 1, bl, 256);
            run("Fp mul (12x32 CIOS)", [&](int b, int t) { hipLaunchKernelGGL((k_femul<FpParams, 2>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 2, bl, 256);
            run("Fr mul (8x32 CIOS)", [&](int b, int t) { hipLaunchKernelGGL((k_femul<FrParams, 2>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 2, bl, 256);
            run("Fp mul 12x32 via CALL", [&](int b, int t) { hipLaunchKernelGGL((k_femul_call<1>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 1, bl, 256);
            run("Fp mul 14x28 inline", [&](int b, int t) { hipLaunchKernelGGL((k_mul28<1, false>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 1, bl, 256);
            run("Fp mul 14x28 inline", [&](int b, int t) { hipLaunchKernelGGL((k_mul28<2, false>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 2, bl, 256);
            run("Fp mul 14x28 via CALL", [&](int b, int t) { hipLaunchKernelGGL((k_mul28<1, true>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 1, bl, 256);
            run("Fp28 lib mul (vector-arg CALL)", [&](int b, int t) { hipLaunchKernelGGL((k_mul28_lib<1>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 1, bl, 256);
            run("Fp28 lib mul (vector-arg CALL)", [&](int b, int t) { hipLaunchKernelGGL((k_mul28_lib<2>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 2, bl, 256);
            run("Fr29 lib mul (vector-arg CALL)", [&](int b, int t) { hipLaunchKernelGGL((k_mul29_lib<2>), dim3(b), dim3(t), 0, 0, (uint32_t*)buf, it2); }, 2, bl, 256);
        }
    }
    CK(hipFree(buf));
    return 0;
}
