// Curve policies for the Pippenger pipeline (msm_impl.cuh): what a bucket is made of and how points are
// loaded, added and handed back to the host.
//   G1Fast  - Fp in reduced radix (bzk_fp28.cuh): buckets are G1X28 (224 B), bases are converted once per
//             call to the internal 112-byte form (x, y in 14 x 28-bit limbs, Montgomery 2^392)
//   G2Fast  - the same over Fp2 = Fp28[u]/(u^2+1) with the generic XYZZ formulas (buckets 448 B, bases 224 B)
#pragma once
#include "bzk_endo.cuh"
#include "bzk_fp28.cuh"
#include "bzk_g2pair.cuh"

namespace bzk {

#ifndef BZK_G1_ACC_INLINE
#define BZK_G1_ACC_INLINE 1  // 1: the accumulate kernel's mixed addition with its eight products inlined: 4 750 instead of 5 245 instructions per addition (no
                             //    call-ABI moves; the multiply-adds are the same 3 545), 170 registers, 42 KB loop.  Same box, alternating: msm_accumulate 2.67 ->
                             //    2.55 ms at 2^20, 3.33 -> 3.03 ms over a static table (profiles/r04_run30_inlined_products_ab.txt).  0: calls (A/B builds)
#endif
#ifndef BZK_G2_ACC_OCC
#define BZK_G2_ACC_OCC 1
#endif
struct alignas(16) U128 {
    uint32_t x, y, z, w;
};

struct G1Fast {
    static constexpr bool PAIR_ACC = false;  // one lane per task
    typedef FpOps HostF;
    typedef G1X28 Pt;
    typedef G1A28 DevAff;
    static constexpr int RAW = 96, PACKED = 97;
    static constexpr bool CONVERT_BASES = true;
    static constexpr int WSUM_THREADS = 256;  // 256 x 224 B = 56 KiB LDS
#ifndef BZK_G1_ACC_OCC
#define BZK_G1_ACC_OCC 2  // waves per SIMD the accumulate kernel is compiled for (A/B builds)
#endif
    static constexpr int ACC_OCC = BZK_G1_ACC_OCC;
#ifndef BZK_G1_ACC_LDS
#define BZK_G1_ACC_LDS 1  // the accumulation's next base global memory -> LDS by direct loads, requested at the TOP of the current addition (msm_accumulate_kernel); 0: into registers under the fused-Y tail
#endif
    static constexpr bool ACC_LDS = BZK_G1_ACC_LDS != 0;
    // endomorphism form (bzk_endo.cuh): scalars split into ENDO signed sub-scalars of ENDO_BITS bits over the images X^(2 m) P
    static constexpr int ENDO = 2, ENDO_BITS = 128;
    static constexpr int ENDO_DEFAULT = 2;
    // 0: plain form for this curve, 1: every whole-MSM call over a set with images, 2 (default): only calls with BZK_F_THROUGHPUT - the
    // prover's overlapping MSMs, where fewer buckets mean less WORK for the reduction.  A stand-alone call gains nothing: its bucket
    // reduction is as long as one lane's chain whatever the bucket count, and fuller buckets balance worse (profiles/r04_run4_5_endo_ab.txt)
    static constexpr const char* ENDO_ENV = "BZK_MSM_ENDO_G1";
    __device__ static __forceinline__ void endo_images(const DevAff& p, DevAff* out, size_t stride) { out[stride] = endo::g1_image(p); }
#ifndef BZK_G1_PARK_REDUCE
#define BZK_G1_PARK_REDUCE 0
#endif
    // 1: msm_reduce / folds / window sums keep ONE point in registers and read the second operand from memory (as G2 always does):
    // msm_reduce drops from 354 registers (one wave per SIMD) to 256 (two per SIMD; it could share a SIMD with an accumulate wave of
    // another MSM).  Built and MEASURED in round 2, no gain: msm_reduce 0.65 -> 0.68 ms at 2^20 and 0.63 -> 0.97 ms
    // at 2^22 / 2^24, inside a proof 0.66 - 0.71 -> 0.70 - 1.05 ms, pipelined proofs/s unchanged (profiles/r02_run16_g1_park_reduce.txt):
    // the kernel is bound by its products, and the register-resident form issues fewer loads.  0 (default): both points resident.
    static constexpr bool PARK_REDUCE = BZK_G1_PARK_REDUCE != 0;
    __device__ static __forceinline__ void add_mem(Pt& acc, const Pt* q) { g1x28::add_mem(acc, q); }
    __device__ static __forceinline__ Pt identity() { return g1x28::identity(); }
    __device__ static __forceinline__ void add_mixed(Pt& acc, const DevAff& p, bool neg) { g1x28::add_mixed(acc, p, neg); }
    template <class Pre>
    __device__ static __forceinline__ void add_mixed_pre(Pt& acc, const DevAff& p, bool neg, Pre&& pre) {
#if BZK_G1_ACC_INLINE
        g1x28::add_mixed(acc, p, neg, pre, g1x28::MulInline());
#else
        g1x28::add_mixed(acc, p, neg, pre);
#endif
    }
    __device__ static __forceinline__ void add(Pt& acc, const Pt& q) { g1x28::add_full(acc, q); }
    __device__ static __forceinline__ Pt mul_u32(const Pt& p, uint32_t k) { return g1x28::mul_u32(p, k); }
    __device__ static __forceinline__ Pt dbl(const Pt& p) { return g1x28::dbl(p); }
    __device__ static __forceinline__ DevAff to_dev_affine(const Pt& p) {  // p must not be the identity
        using namespace fp28;
        Fp28 i3 = fp28::inv(p.ZZZ);                       // 1/ZZZ
        Fp28 i2 = fp28::mul(fp28::sqr(p.ZZ), fp28::sqr(i3));  // 1/ZZ = ZZ^2 / ZZZ^2
        return {fp28::mul(p.X, i2), fp28::mul(p.Y, i3)};
    }
    // batched to-affine of the de-duplication stage (msm_impl.cuh section 8)
    typedef Fp28 Fld;
    __device__ static __forceinline__ bool is_identity(const Pt& p) { return g1x28::is_identity(p); }
    __device__ static __forceinline__ Fld f_one() { return fp28::one(); }
    __device__ static __forceinline__ Fld f_mul(const Fld& a, const Fld& b) { return fp28::mul(a, b); }
    __device__ static __forceinline__ Fld f_inv(const Fld& a) { return fp28::inv(a); }
    __device__ static __forceinline__ DevAff affine_with_inv(const Pt& p, const Fld& i3) {  // i3 = 1/ZZZ
        Fp28 t = fp28::mul(p.ZZ, i3);  // ZZ/ZZZ ; its square is 1/ZZ (ZZ^3 = ZZZ^2)
        return {fp28::mul(p.X, fp28::sqr(t)), fp28::mul(p.Y, i3)};
    }
    __device__ static __forceinline__ XyzzT<FpOps> to_std(const Pt& p) { return g1x28::to_std(p); }
    // raw 96-byte affine (12 x 32-bit Montgomery-384) -> internal
    __device__ static __forceinline__ DevAff convert(const void* raw, uint64_t i) {
        const U128* p = (const U128*)raw + i * 6;
        G1Affine a;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            U128 v = p[k], w = p[k + 3];
            a.x.l[4 * k] = v.x; a.x.l[4 * k + 1] = v.y; a.x.l[4 * k + 2] = v.z; a.x.l[4 * k + 3] = v.w;
            a.y.l[4 * k] = w.x; a.y.l[4 * k + 1] = w.y; a.y.l[4 * k + 2] = w.z; a.y.l[4 * k + 3] = w.w;
        }
        return g1x28::affine_to28(a);
    }
    // gather of one internal base: 112 B = 7 x 16 B
    __device__ static __forceinline__ DevAff load(const void* internal, uint32_t idx) {
        const U128* p = (const U128*)internal + (size_t)idx * 7;
        U128 v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = p[k];
        DevAff a;
        uint32_t* dst = a.x.l;  // x.l[0..13] then y.l[0..13] are contiguous (2 x 56 B)
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            dst[4 * k] = v[k].x; dst[4 * k + 1] = v[k].y; dst[4 * k + 2] = v[k].z; dst[4 * k + 3] = v[k].w;
        }
        return a;
    }
};
static_assert(sizeof(G1A28) == 112 && sizeof(G1X28) == 224, "internal G1 layouts");

// G2 over Fp2x28: generic XYZZ code on the reduced-radix field; bases converted per call (224 B each)
struct G2Fast {
    // the accumulation on PAIRS of lanes (bzk_g2pair.cuh): half a point per lane, two waves per SIMD, every product inlined.
    // env BZK_G2_PAIR=0 keeps the one-lane kernel (same-box A/B; both are compiled in)
    static constexpr bool PAIR_ACC = true;
    static constexpr bool ACC_LDS = false;  // (the one-lane G2 accumulation, kept for A/B, gathers into registers)
    typedef Fp2Ops HostF;
    typedef G2X28 Pt;
    typedef G2A28 DevAff;
    static constexpr int RAW = 192, PACKED = 193;
    static constexpr bool CONVERT_BASES = true;
    static constexpr int WSUM_THREADS = 128;  // 128 x 448 B = 56 KiB LDS
    // waves per SIMD the accumulate kernel is compiled for: 2 -> 256 VGPRs, 402 spilled to scratch; 1 -> 512 (VGPR + AGPR),
    // 35 spilled (profiles/r01_run15: 12.7 ms vs 13.0 ms at 2^20)
    static constexpr int ACC_OCC = BZK_G2_ACC_OCC;
    // endomorphism form (bzk_endo.cuh): four signed 64-bit sub-scalars over the images X^m P, m = 0 .. 3
    static constexpr int ENDO = 4, ENDO_BITS = 64;
    static constexpr int ENDO_DEFAULT = 2;
    static constexpr const char* ENDO_ENV = "BZK_MSM_ENDO_G2";
    __device__ static __forceinline__ void endo_images(const DevAff& p, DevAff* out, size_t stride) {
        out[stride] = endo::g2_image<1>(p);
        out[2 * stride] = endo::g2_image<2>(p);
        out[3 * stride] = endo::g2_image<3>(p);
    }
    __device__ static __forceinline__ Pt identity() { return xyzz_identity<Fp2x28Ops>(); }
    template <class Pre>
    __device__ static __forceinline__ void add_mixed_pre(Pt& acc, const DevAff& p_in, bool neg, Pre&& pre) {
#if BZK_G2_FAST_MIXED
        g2x28::add_mixed(acc, p_in, neg, pre);
#else
        pre();
        add_mixed(acc, p_in, neg);
#endif
    }
    __device__ static __forceinline__ void add_mixed(Pt& acc, const DevAff& p_in, bool neg) {
#if BZK_G2_FAST_MIXED
        g2x28::add_mixed(acc, p_in, neg);
#else
        DevAff p = p_in;
        if (neg) p.y = Fp2x28Ops::neg(p.y);
        xyzz_add_mixed<Fp2x28Ops>(acc, p);
#endif
    }
    // second operand left in memory (LDS / global): see xyzz_add_mem
    static constexpr bool PARK_REDUCE = true;
    __device__ static __forceinline__ void add(Pt& acc, const Pt& q) { xyzz_add<Fp2x28Ops>(acc, q); }  // by-value form: unused by the G2 kernels
#if BZK_G2_FAST_TAILS
    __device__ static __forceinline__ void add_mem(Pt& acc, const Pt* q) { g2x28::add_mem(acc, q); }
    __device__ static __forceinline__ Pt mul_u32(const Pt& p, uint32_t k) { return g2x28::mul_u32(p, k); }
    __device__ static __forceinline__ Pt dbl(const Pt& p) { return g2x28::dbl(p); }
#else
    __device__ static __forceinline__ void add_mem(Pt& acc, const Pt* q) { xyzz_add_mem<Fp2x28Ops>(acc, q); }
    __device__ static __forceinline__ Pt mul_u32(const Pt& p, uint32_t k) { return xyzz_mul_u32<Fp2x28Ops>(p, k); }
    __device__ static __forceinline__ Pt dbl(const Pt& p) { return xyzz_dbl<Fp2x28Ops>(p); }
#endif
    __device__ static __forceinline__ DevAff to_dev_affine(const Pt& p) {
        DevAff a;
        xyzz_to_affine<Fp2x28Ops>(p, a);
        return a;
    }
    typedef Fp2x28 Fld;
    __device__ static __forceinline__ bool is_identity(const Pt& p) { return xyzz_is_identity<Fp2x28Ops>(p); }
    __device__ static __forceinline__ Fld f_one() { return Fp2x28Ops::one(); }
    __device__ static __forceinline__ Fld f_mul(const Fld& a, const Fld& b) { return Fp2x28Ops::mul(a, b); }
    __device__ static __forceinline__ Fld f_inv(const Fld& a) { return Fp2x28Ops::inv(a); }
    __device__ static __forceinline__ DevAff affine_with_inv(const Pt& p, const Fld& i3) {
        Fld t = Fp2x28Ops::mul(p.ZZ, i3);
        DevAff a;
        a.x = Fp2x28Ops::mul(p.X, Fp2x28Ops::sqr(t));
        a.y = Fp2x28Ops::mul(p.Y, i3);
        return a;
    }
    __device__ static __forceinline__ XyzzT<Fp2Ops> to_std(const Pt& p) { return g2x28::to_std(p); }
    __device__ static __forceinline__ DevAff convert(const void* raw, uint64_t i) {
        const U128* p = (const U128*)raw + i * 12;
        G2Affine a;
        Fp* f[4] = {&a.x.c0, &a.x.c1, &a.y.c0, &a.y.c1};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                U128 v = p[3 * e + k];
                f[e]->l[4 * k] = v.x; f[e]->l[4 * k + 1] = v.y; f[e]->l[4 * k + 2] = v.z; f[e]->l[4 * k + 3] = v.w;
            }
        }
        return g2x28::affine_to28(a);
    }
    // gather of one internal base: 224 B = 14 x 16 B
    __device__ static __forceinline__ DevAff load(const void* internal, uint32_t idx) {
        const U128* p = (const U128*)internal + (size_t)idx * 14;
        DevAff a;
        uint32_t* dst = a.x.c0.l;  // x.c0, x.c1, y.c0, y.c1: 4 x 56 B contiguous
#pragma unroll
        for (int k = 0; k < 14; ++k) {
            U128 v = p[k];
            dst[4 * k] = v.x; dst[4 * k + 1] = v.y; dst[4 * k + 2] = v.z; dst[4 * k + 3] = v.w;
        }
        return a;
    }
};
static_assert(sizeof(G2A28) == 224 && sizeof(G2X28) == 448, "internal G2 layouts");

}  // namespace bzk
