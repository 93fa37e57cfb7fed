// K4: G1 instantiation of the Pippenger pipeline (see msm_impl.cuh).  The Fp product is a real
// function (BZK_FP_NOINLINE) for every kernel but one: with the round-1 product (~11 KB of code) an inlined
// XYZZ addition overflowed the 64 KB instruction cache and the kernels were fetch-bound.  Since the
// reduced-radix product (bzk_fp28.cuh, ~3.4 KB) the accumulation's mixed addition fits: its hot loop is
// 42 KB with all eight products inlined (g1x28::MulInline, BZK_G1_ACC_INLINE) and runs without the call
// ABI's ~500 argument / result moves per addition; the general additions of the tail kernels (14 products,
// two of them per step) stay calls.
#define BZK_FP_NOINLINE 1
#include "msm_impl.cuh"
using namespace bzk;

extern "C" {

uint32_t bzk_msm_window_count(uint64_t n) {
    int c = msm_pick_c(n ? n : 1);
    if (const char* e = getenv("BZK_MSM_C")) {
        int v = atoi(e);
        if (v >= 2 && v <= 20) c = v;
    }
    return (uint32_t)msm_windows_for(c);
}
int32_t bzk_msm_g1_dev(bzk_ctx* ctx, const void* bases, const void* scalars, uint64_t n, uint32_t flags, uint8_t out[97]) {
    return msm_entry_dev<G1Fast>(ctx, bases, scalars, n, flags, 0, -1, out);
}
int32_t bzk_msm_g1_windows_dev(bzk_ctx* ctx, const void* bases, const void* scalars, uint64_t n, uint32_t flags,
                               uint32_t w_begin, uint32_t w_end, uint8_t out[97]) {
    return msm_entry_dev<G1Fast>(ctx, bases, scalars, n, flags, (int)w_begin, (int)w_end, out);
}
int32_t bzk_msm_g1(bzk_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[97]) {
    return msm_entry_host<G1Fast>(ctx, bases, scalars, n, flags, out);
}
int32_t bzk_g1_sum(const uint8_t* pts, uint32_t count, uint8_t out[97]) { return sum_packed<FpOps>(pts, count, out); }
int32_t bzk_g1_synth_bases_dev(bzk_ctx* ctx, uint64_t seed, uint64_t start, uint64_t n, void* out_dev) {
    if (!ctx || (n && !out_dev)) return BZK_E_ARG;
    if (!n) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    auto k = synth_bases_kernel<FpOps>;
    BZK_LAUNCH(ctx, "synth_bases_g1", k, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g1_generator_host(), seed, start, n,
               (G1Affine*)out_dev);
    return BZK_OK;
}

// static-base tables (see msm_impl.cuh 4c)
int32_t bzk_msm_g1_table_build(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_table** out) {
    MsmTable* t = nullptr;
    int32_t st = msm_table_build<G1Fast>(ctx, bases_dev, n, &t);
    if (out) *out = (bzk_msm_table*)t;
    return st;
}
int32_t bzk_msm_g1_table_build_levels(bzk_ctx* ctx, const void* bases_dev, uint64_t n, uint32_t levels, bzk_msm_table** out) {
    MsmTable* t = nullptr;
    int32_t st = msm_table_build<G1Fast>(ctx, bases_dev, n, &t, (int)levels);
    if (out) *out = (bzk_msm_table*)t;
    return st;
}
int32_t bzk_msm_g1_table_build_c(bzk_ctx* ctx, const void* bases_dev, uint64_t n, uint32_t c, bzk_msm_table** out) {
    MsmTable* t = nullptr;
    int32_t st = msm_table_build<G1Fast>(ctx, bases_dev, n, &t, 0, (int)c);
    if (out) *out = (bzk_msm_table*)t;
    return st;
}
int32_t bzk_msm_g1_table_run_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags,
                                 uint8_t out[97]) {
    return msm_table_entry<G1Fast>(ctx, (const MsmTable*)table, scalars_dev, n, flags, 0, -1, out);
}
int32_t bzk_msm_g1_table_windows_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[97]) {
    return msm_table_entry<G1Fast>(ctx, (const MsmTable*)table, scalars_dev, n, flags, (int)w_begin, (int)w_end, out);
}

void bzk_msm_table_free(bzk_ctx* ctx, bzk_msm_table* table) { msm_table_free(ctx, (MsmTable*)table); }
uint32_t bzk_msm_table_window_count(const bzk_msm_table* table) {
    const MsmTable* t = (const MsmTable*)table;
    return !t ? 0 : (uint32_t)(t->wpl > 1 ? t->wpl : t->w_total);
}
uint32_t bzk_msm_table_levels(const bzk_msm_table* table) { return table ? (uint32_t)((const MsmTable*)table)->levels : 0; }

// resident base sets (see msm_impl.cuh MsmBases): a static point set converted once to the internal form
int32_t bzk_msm_g1_bases_load_dev(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_bases** out) {
    MsmBases* b = nullptr;
    int32_t st = msm_bases_load<G1Fast>(ctx, bases_dev, n, &b);
    if (out) *out = (bzk_msm_bases*)b;
    return st;
}
int32_t bzk_msm_g1_bases_run_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags,
                                 uint8_t out[97]) {
    return msm_bases_entry<G1Fast>(ctx, (const MsmBases*)bases, scalars_dev, n, flags, 0, -1, out);
}
int32_t bzk_msm_g1_bases_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[97]) {
    return msm_bases_entry<G1Fast>(ctx, (const MsmBases*)bases, scalars_dev, n, flags, (int)w_begin, (int)w_end, out);
}
void bzk_msm_bases_free(bzk_ctx* ctx, bzk_msm_bases* bases) { msm_bases_free(ctx, (MsmBases*)bases); }
uint64_t bzk_msm_bases_size(const bzk_msm_bases* bases) { return bases ? ((const MsmBases*)bases)->n : 0; }
int32_t bzk_msm_bases_info(const bzk_msm_bases* bases, uint64_t* n, int32_t* forms, uint64_t* device_bytes) {
    const MsmBases* b = (const MsmBases*)bases;
    if (!b) return BZK_E_ARG;
    if (n) *n = b->n;
    if (forms) *forms = b->endo;
    if (device_bytes) *device_bytes = b->bytes;
    return BZK_OK;
}

}  // extern "C"

// hooks for mg.hip / groth16.hip (not part of the C ABI)
namespace bzk {
int32_t msm_g1_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* bases_raw, const void* scalars, uint64_t n, uint32_t flags,
                           int w_begin, int w_end, void* d_win, int32_t info[5]) {
    return msm_windows_dev<G1Fast>(ctx, (const MsmBases*)bases, bases_raw, scalars, n, flags, w_begin, w_end, d_win, info);
}
// what a window-range call over n points leaves per bucket set at d_win: 0 = one window sum, k > 0 = the k terms of the multiplication-free reduction
// (msm_impl.cuh section 6b) - a function of n and the process's environment alone, so every rank of a device group sizes its exchange alike
int msm_g1_window_terms(uint64_t n) { return msm_terms_per_set<G1Fast>(msm_window_bits(n)); }
int32_t g1_horner_terms_packed(const void* T, int count, int c, int w0, uint8_t* out) { return horner_terms_packed<FpOps>(T, count, c, w0, out); }
int msm_window_bits(uint64_t n) {  // the window size behind bzk_msm_window_count(n): what a call that names a window range runs with
    int c = msm_pick_c(n ? n : 1);
    if (const char* e = getenv("BZK_MSM_C")) {
        int v = atoi(e);
        if (v >= 2 && v <= 20) c = v;
    }
    return c;
}
int32_t g1_horner_packed(const void* S, int count, int c, int w0, uint8_t* out) { return horner_packed<FpOps>(S, count, c, w0, out); }
}  // namespace bzk

