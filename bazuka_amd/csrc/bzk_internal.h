// Internal plumbing of libbzk: context, error handling, workspace, per-kernel event timing.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string.h>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bzk.h"

struct bzk_prof_rec {
    const char* name;
    hipEvent_t a, b;
};

// An instance's assignment resident in HBM (bzk_r1cs_stage, groth16.hip): z | A.z | B.z | C.z as the prover reads them, complete - the deferred-value
// program, if the instance had one, has run behind the uploads on the staging context's stream; `ready` marks the end of that work
struct bzk_staged {
    bzk_ctx* owner = nullptr;
    void* buf = nullptr;          // one allocation: z (n_vars) | az | bz | cz (n_rows each), 32-byte scalars
    size_t cap = 0;               // bytes behind buf
    uint64_t n_vars = 0, n_rows = 0;
    hipEvent_t ready = nullptr;
    uint32_t* flags_host = nullptr;  // pinned: wf::FLAG_* of the program (0 without one), valid once `ready` has completed
    void* z() const { return buf; }
    void* ev(int k) const { return (char*)buf + (n_vars + (uint64_t)k * n_rows) * 32; }
};

struct bzk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    int32_t last_refusal = 0;  // BZK_REFUSE_*: what the last BZK_E_ARG of a bzk_state_* call stands for (bzk_last_refusal)
    bool prof = false;
    std::string prof_only;  // non-empty: only launches whose label contains it get event pairs (bzk_prof_filter)
    std::vector<bzk_prof_rec> recs;
    // grow-only scratch buffer reused across calls
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // pinned host staging (small results)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // device-resident Poseidon constants per width t (index t), Montgomery, rc then mds
    void* poseidon_dev[18] = {};
    // MSM tuning overrides (0 = automatic); settable through env BZK_MSM_C / BZK_MSM_CHUNK
    int msm_c_override = 0;
    int msm_chunk_override = 0;
    bool msm_no_endo = false;  // env BZK_MSM_NO_ENDO=1 (and the device groups' contexts): resident base sets without endomorphism images, plain windows
    int msm_reduce2 = 0;  // env BZK_MSM_REDUCE2: 1 forces the two-level bucket reduction, -1 forbids it, 0 = per call (BZK_F_THROUGHPUT)
    bool debug = false;  // env BZK_DEBUG=1: synchronise + log after every launch (hang localisation)
    // NTT twiddle cache: per log_n, forward and inverse tables
    void* ntt_tw[33][2] = {};
    // lanes: child contexts (own stream, workspace, pinned staging) for independent sub-jobs of one call that
    // should overlap on the device - the five MSMs of a Groth16 proof.  Created on first use, owned by the parent.
    std::vector<bzk_ctx*> lanes;
    std::vector<struct bzk_lane_thread*> lane_threads;  // one persistent host thread per lane (ctx.hip: lane_post / lane_wait)
    // round 6 (run 18): child contexts of ONE stand-alone MSM call whose windows run as several ranges in flight (msm_impl.cuh msm_run_split): the
    // front chain of one range hides under the accumulation of another, the tails of one under the accumulation of the next.  Own stream + workspace
    // each, created on first use, owned by the parent; `is_part` marks such a child (it never splits again).  split_terms: device staging of the terms
    // the ranges leave (W x terms per set x 192 B), split_ev: the fork event recorded on the parent's stream
    // msm_split: ranges per call (0 = the library default, 1 = off), msm_split_min_log: smallest log2(n) that splits, msm_split_prio: children at the highest
    // stream priority - read from env BZK_MSM_SPLIT / BZK_MSM_SPLIT_MIN_LOG / BZK_MSM_SPLIT_PRIO when the context is created (tests and A/B runs)
    std::vector<bzk_ctx*> parts;
    int msm_split = 0, msm_split_min_log = 0, msm_split_prio = -1;
    int msm_split_cuts[4] = {0, 0, 0, 0};  // env BZK_MSM_SPLIT_CUTS="a,b[,c[,d]]": windows per range, highest range first (A/B runs; used when they add up to W)
    bool is_part = false, split_active = false;
    void* split_terms = nullptr;
    size_t split_terms_bytes = 0;
    void* split_conv = nullptr;  // internal form of the raw bases of a split call (msm_entry_dev), grow-only; released by bzk_ctx_trim
    size_t split_conv_bytes = 0;
    hipEvent_t split_ev = nullptr;
    bool no_coop = false;  // env BZK_NO_COOP=1: never use the cooperative (8 lanes per node) Poseidon kernel (A/B runs)
    bool timing = false;  // env BZK_TIMING=1: host-side phase timings of bzk_groth16_prove on stderr
    hipEvent_t ev_z = nullptr;  // "assignment staged" event of bzk_groth16_prove: created on first use, destroyed with the ctx
    // side stream of one MSM call: work that the bucket pipeline does not depend on until the accumulation (the per-call base
    // conversion) runs here beside digits / sort; fork / join through the two events.  Created on first use.
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // env BZK_PROVE_H_PRIO=1 (A/B runs, measured neutral): highest-priority side stream of bzk_groth16_prove for the h polynomial
    // (evaluations staged + seven transforms).  Created on first use.
    hipStream_t hprio = nullptr;
    hipEvent_t ev_h = nullptr;
    // round 6: the SATURATING kernels of an MSM (accumulation, folds, bucket reduction) on a lowest-priority side stream, so that the short kernels of
    // OTHER streams' MSMs (digits, sorts, boundaries - the chain that gates their next accumulation) are not starved beside them (msm_impl.cuh HeavyScope)
    hipStream_t heavy = nullptr;
    hipEvent_t ev_heavy_in = nullptr, ev_heavy_out = nullptr;
    bool heavy_tried = false;
    bool heavy_force = false;  // a later window range of a split call (msm_run_split, priority mode 2): HeavyScope applies whatever BZK_MSM_HEAVY_PRIO says
    void* wf_state = nullptr;  // witfill.hip: device copies of the deferred-witness programs, dense Poseidon constants, scratch (witfill_free)
    // bzk_r1cs_stage: staged assignments handed back by bzk_staged_free (possibly from another thread: the prover's), kept for the next call
    std::mutex staged_mu;
    std::vector<bzk_staged*> staged_pool;
};

#define BZK_HIP(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e__);         \
            return BZK_E_DEVICE;                                                            \
        }                                                                                   \
    } while (0)

#define BZK_TRY(expr)                 \
    do {                              \
        int32_t s__ = (expr);         \
        if (s__ != BZK_OK) return s__; \
    } while (0)

// groth16.hip hooks used by setup.hip
int32_t bzk_params_alloc_internal(bzk_ctx* ctx, const bzk_params_desc* d, bzk_params** out);
void bzk_params_buffers_internal(bzk_params* p, void** h, void** l, void** a, void** b_g1, void** b_g2);
void bzk_params_set_vk_internal(bzk_params* p, const uint8_t vk[870]);

namespace bzk {

int32_t ws_reserve(bzk_ctx* ctx, size_t bytes);           // ensures ctx->ws has >= bytes
bzk_ctx* ctx_lane(bzk_ctx* ctx, size_t i);                // i-th child context (nullptr on failure)
bzk_ctx* ctx_part(bzk_ctx* ctx, size_t i, bool high_prio); // i-th window-range child of a split MSM call (nullptr on failure)
int32_t pinned_reserve(bzk_ctx* ctx, size_t bytes);
// runs `job` on the persistent host thread of lane i (created on first use, bound to the ctx's device); lane_wait blocks until that
// job has returned.  One job per lane at a time.
void lane_post(bzk_ctx* ctx, size_t i, std::function<void()> job);
void lane_wait(bzk_ctx* ctx, size_t i);

// bump allocator over ctx->ws
struct WsCursor {
    char* base;
    size_t off = 0;
    explicit WsCursor(void* b) : base((char*)b) {}
    template <class T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = (T*)(base + off);
        off += count * sizeof(T);
        return p;
    }
};
void witfill_free(bzk_ctx* ctx);  // witfill.hip
int32_t ntt_run(bzk_ctx* ctx, void* data_dev, uint32_t log_n, int inverse, int coset);  // ntt.hip
int32_t ntt_h_chain(bzk_ctx* ctx, void* a, void* b, void* c, uint32_t log_m);              // ntt.hip: the h polynomial's 7 transforms, fused
// msm_g1.hip / msm_g2.hip: windows [w_begin, w_end) (w_end < 0: all) of an MSM over a resident base set (or raw bases when `bases` is
// null); the window sums stay in DEVICE memory at d_win as standard-limb XYZZ points (G1 192 B, G2 384 B each), in stream order,
// nothing is read back.  info = {c, w_total, w_begin, w_end, terms per bucket set (0: window sums; G1 since round 6: msm_g1_window_terms)}.  *_horner_packed: host Horner sum_k 2^(c (w0 + k)) S[k] -> packed point
int32_t msm_g1_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* bases_raw, const void* scalars, uint64_t n, uint32_t flags,
                           int w_begin, int w_end, void* d_win, int32_t info[5]);
int32_t msm_g2_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* bases_raw, const void* scalars, uint64_t n, uint32_t flags,
                           int w_begin, int w_end, void* d_win, int32_t info[5]);
int msm_window_bits(uint64_t n);
int msm_g1_window_terms(uint64_t n);  // 0: a window-range G1 call leaves window sums at d_win; k > 0: the k terms of every bucket set (info[4] of the call agrees)
int32_t g1_horner_terms_packed(const void* T, int count, int c, int w0, uint8_t* out);
int32_t g1_horner_packed(const void* S, int count, int c, int w0, uint8_t* out);
int32_t g2_horner_packed(const void* S, int count, int c, int w0, uint8_t* out);
static inline size_t ws_pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
// an environment switch: set and not "0" (one getenv: the value is read from the pointer it was tested on)
static inline bool env_on(const char* name) {
    const char* e = getenv(name);
    return e && atoi(e) != 0;
}

// A schedule of Poseidon hashes over one value array (state.hip): values [0, n_up) are uploaded 32-byte scalars, group g's outputs
// follow in group order.  Groups run one batched launch each, in order, so a group may consume anything uploaded or produced by an
// earlier group.  Used by the general state compress and by the device path of the MPN witness builders (mpn.hip).
struct HashGroup {
    uint32_t arity = 0;
    std::vector<uint32_t> in;  // count * arity value ids (already resolved: < n_up + outputs of earlier groups)
    uint32_t count() const { return arity ? (uint32_t)(in.size() / arity) : 0; }
};
// uploaded: n_up x 32 B Montgomery scalars; hashed_out receives sum(count) x 32 B in group order
int32_t hash_plan_run(bzk_ctx* ctx, const uint8_t* uploaded, uint64_t n_up, const std::vector<HashGroup>& groups, std::vector<uint8_t>& hashed_out);

struct ProfScope {
    bzk_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    ProfScope(bzk_ctx* c, const char* n) : ctx(c), name(n) {
        if (ctx->prof && (ctx->prof_only.empty() || strstr(n, ctx->prof_only.c_str()))) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~ProfScope() {
        if (a) {
            (void)hipEventRecord(b, ctx->stream);
            ctx->recs.push_back({name, a, b});
        }
    }
};

// launch + error check; `name` must be a string literal (kept by pointer)
#define BZK_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                                      \
    do {                                                                                            \
        {                                                                                           \
            bzk::ProfScope ps__(ctx, name);                                                         \
            hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);              \
        }                                                                                           \
        hipError_t e__ = hipGetLastError();                                                         \
        if (e__ == hipSuccess && (ctx)->debug) {                                                    \
            fprintf(stderr, "[bzk] launched %s ...", name);                                         \
            fflush(stderr);                                                                         \
            e__ = hipStreamSynchronize((ctx)->stream);                                              \
            fprintf(stderr, " %s\n", e__ == hipSuccess ? "done" : hipGetErrorString(e__));          \
            fflush(stderr);                                                                         \
        }                                                                                           \
        if (e__ != hipSuccess) {                                                                    \
            (ctx)->last_error = std::string("launch ") + name + ": " + hipGetErrorString(e__);      \
            return BZK_E_DEVICE;                                                                    \
        }                                                                                           \
    } while (0)

}  // namespace bzk
