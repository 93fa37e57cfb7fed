#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __noinline__ unsigned f(unsigned x) { return x * 3u + 1u; }
template <int SRC> __device__ unsigned q(unsigned v) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false);
}
__global__ void k(unsigned* out, int mode) {
    unsigned t = threadIdx.x;
    unsigned quad = t >> 2;
    unsigned r = 0;
    // only some quads active (like the tree levels)
    if (quad < (unsigned)mode) {
        unsigned v = f(t);
        unsigned a = q<0>(v), b = q<1>(v), c = q<2>(v), d = q<3>(v);
        unsigned a2 = __shfl(v, 0, 4), b2 = __shfl(v, 1, 4), c2 = __shfl(v, 2, 4), d2 = __shfl(v, 3, 4);
        r = (a == a2) | ((b == b2) << 1) | ((c == c2) << 2) | ((d == d2) << 3);
    } else r = 15;
    out[t] = r;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    for (int mode : {64, 32, 7, 1}) {
        k<<<1, 256>>>(d, mode);
        unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 256; ++i) bad += h[i] != 15;
        printf("mode %d: bad lanes %d (first: ", mode, bad);
        for (int i = 0, c = 0; i < 256 && c < 6; ++i) if (h[i] != 15) { printf("%d:%x ", i, h[i]); ++c; }
        printf(")\n");
    }
    return 0;
}
