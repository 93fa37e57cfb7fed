// K5: G2 instantiation of the Pippenger pipeline (see msm_impl.cuh) on the reduced-radix field
// (Fp2x28Ops, bzk_fp28.cuh): an Fp2 product is three 14 x 28-bit base-field products, each a call to the
// one resident copy of fp28::mul, so the kernels stay inside the instruction cache.
#define BZK_FP_NOINLINE 1
#include "msm_impl.cuh"
using namespace bzk;

extern "C" {

int32_t bzk_msm_g2_dev(bzk_ctx* ctx, const void* bases, const void* scalars, uint64_t n, uint32_t flags, uint8_t out[193]) {
    return msm_entry_dev<G2Fast>(ctx, bases, scalars, n, flags, 0, -1, out);
}
int32_t bzk_msm_g2_windows_dev(bzk_ctx* ctx, const void* bases, const void* scalars, uint64_t n, uint32_t flags,
                               uint32_t w_begin, uint32_t w_end, uint8_t out[193]) {
    return msm_entry_dev<G2Fast>(ctx, bases, scalars, n, flags, (int)w_begin, (int)w_end, out);
}
int32_t bzk_msm_g2(bzk_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[193]) {
    return msm_entry_host<G2Fast>(ctx, bases, scalars, n, flags, out);
}
int32_t bzk_g2_sum(const uint8_t* pts, uint32_t count, uint8_t out[193]) { return sum_packed<Fp2Ops>(pts, count, out); }
int32_t bzk_g2_synth_bases_dev(bzk_ctx* ctx, uint64_t seed, uint64_t start, uint64_t n, void* out_dev) {
    if (!ctx || (n && !out_dev)) return BZK_E_ARG;
    if (!n) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    auto k = synth_bases_kernel<Fp2Ops>;
    BZK_LAUNCH(ctx, "synth_bases_g2", k, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g2_generator_host(), seed, start, n,
               (G2Affine*)out_dev);
    return BZK_OK;
}

// static-base tables (see msm_impl.cuh 4c)
int32_t bzk_msm_g2_table_build(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_table** out) {
    MsmTable* t = nullptr;
    int32_t st = msm_table_build<G2Fast>(ctx, bases_dev, n, &t);
    if (out) *out = (bzk_msm_table*)t;
    return st;
}
int32_t bzk_msm_g2_table_build_levels(bzk_ctx* ctx, const void* bases_dev, uint64_t n, uint32_t levels, bzk_msm_table** out) {
    MsmTable* t = nullptr;
    int32_t st = msm_table_build<G2Fast>(ctx, bases_dev, n, &t, (int)levels);
    if (out) *out = (bzk_msm_table*)t;
    return st;
}
int32_t bzk_msm_g2_table_run_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags,
                                 uint8_t out[193]) {
    return msm_table_entry<G2Fast>(ctx, (const MsmTable*)table, scalars_dev, n, flags, 0, -1, out);
}
int32_t bzk_msm_g2_table_windows_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[193]) {
    return msm_table_entry<G2Fast>(ctx, (const MsmTable*)table, scalars_dev, n, flags, (int)w_begin, (int)w_end, out);
}

// resident base sets (see msm_impl.cuh MsmBases): a static point set converted once to the internal form
int32_t bzk_msm_g2_bases_load_dev(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_bases** out) {
    MsmBases* b = nullptr;
    int32_t st = msm_bases_load<G2Fast>(ctx, bases_dev, n, &b);
    if (out) *out = (bzk_msm_bases*)b;
    return st;
}
int32_t bzk_msm_g2_bases_run_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags,
                                 uint8_t out[193]) {
    return msm_bases_entry<G2Fast>(ctx, (const MsmBases*)bases, scalars_dev, n, flags, 0, -1, out);
}
int32_t bzk_msm_g2_bases_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[193]) {
    return msm_bases_entry<G2Fast>(ctx, (const MsmBases*)bases, scalars_dev, n, flags, (int)w_begin, (int)w_end, out);
}

}  // extern "C"

// hooks for mg.hip / groth16.hip (not part of the C ABI)
namespace bzk {
int32_t msm_g2_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* bases_raw, const void* scalars, uint64_t n, uint32_t flags,
                           int w_begin, int w_end, void* d_win, int32_t info[5]) {
    return msm_windows_dev<G2Fast>(ctx, (const MsmBases*)bases, bases_raw, scalars, n, flags, w_begin, w_end, d_win, info);
}
int32_t g2_horner_packed(const void* S, int count, int c, int w0, uint8_t* out) { return horner_packed<Fp2Ops>(S, count, c, w0, out); }
}  // namespace bzk

