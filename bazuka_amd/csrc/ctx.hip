// Context, memory plumbing and event-based per-kernel timing for libbzk.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>

#include "bzk_internal.h"

namespace bzk {

void ntt_free_tables(bzk_ctx* ctx);  // ntt.hip

int32_t ws_reserve(bzk_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return BZK_OK;
    if (ctx->ws) {
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BZK_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    size_t want = bytes + (bytes >> 3);
    hipError_t e = hipMalloc(&ctx->ws, want);
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipMalloc workspace: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return BZK_E_ALLOC;
    }
    ctx->ws_bytes = want;
    return BZK_OK;
}

extern "C" int32_t bzk_ctx_trim(bzk_ctx* ctx, uint64_t* released) {
    if (!ctx) return BZK_E_ARG;
    if (released) *released = 0;
    (void)hipSetDevice(ctx->device);
    uint64_t kids = 0;
    for (bzk_ctx* c : ctx->parts) {  // the window-range children of split MSM calls hold workspaces of their own
        uint64_t r = 0;
        BZK_TRY(bzk_ctx_trim(c, &r));
        kids += r;
    }
    if (ctx->split_conv) {
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BZK_HIP(ctx, hipFree(ctx->split_conv));
        kids += ctx->split_conv_bytes;
        ctx->split_conv = nullptr;
        ctx->split_conv_bytes = 0;
    }
    if (released) *released = kids;
    if (!ctx->ws) return BZK_OK;
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    BZK_HIP(ctx, hipFree(ctx->ws));
    if (released) *released = kids + ctx->ws_bytes;
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    return BZK_OK;
}

int32_t pinned_reserve(bzk_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) return BZK_OK;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    hipError_t e = hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return BZK_E_ALLOC;
    }
    ctx->pinned_bytes = bytes;
    return BZK_OK;
}

bzk_ctx* ctx_lane(bzk_ctx* ctx, size_t i) {
    while (ctx->lanes.size() <= i) {
        bzk_ctx* c = nullptr;
        // lane 0 carries the job with the longest latency-bound tail (the G2 MSM of a proof): its stream gets the
        // highest priority so that this tail is reached early and hides under the other lanes' accumulation
        hipStream_t s = nullptr;
        // env BZK_PRIO (A/B runs): "g2" (default) = lane 0 high; "main" = the ctx's own stream high (bzk_ctx_create), lanes normal; "none"
        static const bool lane0_high = [] { const char* e = getenv("BZK_PRIO"); return !e || !strcmp(e, "g2"); }();
        if (ctx->lanes.empty() && lane0_high) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                s = nullptr;
            }
        }
        if (bzk_ctx_create(ctx->device, s, &c) != BZK_OK) {
            if (s) (void)hipStreamDestroy(s);
            return nullptr;
        }
        if (s) c->own_stream = true;  // created here, destroyed with the lane
        ctx->lanes.push_back(c);
    }
    bzk_ctx* c = ctx->lanes[i];
    c->prof = ctx->prof;
    c->prof_only = ctx->prof_only;
    c->debug = ctx->debug;
    c->timing = ctx->timing;
    c->msm_c_override = ctx->msm_c_override;
    c->msm_chunk_override = ctx->msm_chunk_override;
    c->msm_reduce2 = ctx->msm_reduce2;
    c->msm_no_endo = ctx->msm_no_endo;
    return c;
}

// window-range children of a split MSM call (bzk_ctx::parts): own non-blocking stream (highest priority on request: the front chain of a later
// range - digits, sorts, boundaries - then outranks the accumulation it runs beside), own workspace and pinned staging
bzk_ctx* ctx_part(bzk_ctx* ctx, size_t i, bool high_prio) {
    while (ctx->parts.size() <= i) {
        bzk_ctx* c = nullptr;
        hipStream_t s = nullptr;
        if (high_prio) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                s = nullptr;
            }
        }
        if (bzk_ctx_create(ctx->device, s, &c) != BZK_OK) {
            if (s) (void)hipStreamDestroy(s);
            return nullptr;
        }
        if (s) c->own_stream = true;
        c->is_part = true;
        ctx->parts.push_back(c);
    }
    bzk_ctx* c = ctx->parts[i];
    c->prof = ctx->prof;
    c->prof_only = ctx->prof_only;
    c->debug = ctx->debug;
    c->timing = ctx->timing;
    c->msm_c_override = ctx->msm_c_override;
    c->msm_chunk_override = ctx->msm_chunk_override;
    c->msm_reduce2 = ctx->msm_reduce2;
    c->msm_no_endo = ctx->msm_no_endo;
    return c;
}

}  // namespace bzk

struct bzk_lane_thread {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, busy = false, quit = false;
};

namespace bzk {

static void lane_thread_main(bzk_lane_thread* t, int device) {
    (void)pthread_setname_np(pthread_self(), "bzk-lane");  // per-thread CPU accounting by name (tools/host_cpu_probe.py reads /proc/self/task/*/comm)
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(t->m);
    for (;;) {
        t->cv.wait(lk, [&] { return t->has_job || t->quit; });
        if (t->quit) return;
        auto job = std::move(t->job);
        t->has_job = false;
        lk.unlock();
        job();
        lk.lock();
        t->busy = false;
        t->cv.notify_all();
    }
}

void lane_post(bzk_ctx* ctx, size_t i, std::function<void()> job) {
    while (ctx->lane_threads.size() <= i) {
        bzk_lane_thread* t = new bzk_lane_thread();
        t->th = std::thread(lane_thread_main, t, ctx->device);
        ctx->lane_threads.push_back(t);
    }
    bzk_lane_thread* t = ctx->lane_threads[i];
    std::unique_lock<std::mutex> lk(t->m);
    t->cv.wait(lk, [&] { return !t->busy; });
    t->job = std::move(job);
    t->has_job = true;
    t->busy = true;
    t->cv.notify_all();
}

void lane_wait(bzk_ctx* ctx, size_t i) {
    if (i >= ctx->lane_threads.size()) return;
    bzk_lane_thread* t = ctx->lane_threads[i];
    std::unique_lock<std::mutex> lk(t->m);
    t->cv.wait(lk, [&] { return !t->busy; });
}

}  // namespace bzk

extern "C" {

uint32_t bzk_abi_version(void) { return 1; }

const char* bzk_strerror(int32_t s) {
    switch (s) {
        case BZK_OK: return "ok";
        case BZK_E_ARG: return "bad argument";
        case BZK_E_ALLOC: return "allocation failed";
        case BZK_E_DEVICE: return "device error";
        case BZK_E_UNSAT: return "constraint system not satisfied";
        case BZK_E_INTERNAL: return "internal error";
        default: return "unknown status";
    }
}

int32_t bzk_ctx_create(int32_t device_id, void* stream, bzk_ctx** out) {
    if (!out) return BZK_E_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return BZK_E_DEVICE;  // no CPU fallback by design
    }
    if (device_id < 0 || device_id >= count) return BZK_E_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return BZK_E_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "libbzk: device %d is %s, this library is built for gfx950 only\n", device_id, prop.gcnArchName);
        return BZK_E_DEVICE;
    }
    if (hipSetDevice(device_id) != hipSuccess) return BZK_E_DEVICE;
    // env BZK_SYNC_BLOCKING=1: host threads that wait for the GPU sleep on an interrupt instead of spinning.  A prover keeps ~4 host
    // threads per slot waiting most of the time; inside a CPU-quota'd container (cgroup cpu.max) their spinning is charged against the
    // same budget as the witness producers' work (profiles/r02_run37_45_host_interference.txt)
    // The variable is read ONCE, by the first context of the process: switching the device to blocking waits after contexts and streams
    // exist hung the process (profiles/r03_run23_27...: waits on streams created in the other mode never return).
    static const bool blocking_waits = [] { const char* e = getenv("BZK_SYNC_BLOCKING"); return e && atoi(e) != 0; }();
    if (blocking_waits && hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
    bzk_ctx* ctx = new (std::nothrow) bzk_ctx();
    if (!ctx) return BZK_E_ALLOC;
    ctx->device = device_id;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        static const bool main_high = [] { const char* e = getenv("BZK_PRIO"); return e && !strcmp(e, "main"); }();
        int lo = 0, hi = 0;
        hipError_t e = hipErrorUnknown;
        if (main_high && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, hi);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        }
        if (e != hipSuccess) {
            delete ctx;
            return BZK_E_DEVICE;
        }
        ctx->own_stream = true;
    }
    if (const char* e = getenv("BZK_MSM_C")) ctx->msm_c_override = atoi(e);
    if (const char* e = getenv("BZK_MSM_CHUNK")) ctx->msm_chunk_override = atoi(e);
    if (const char* e = getenv("BZK_MSM_REDUCE2")) ctx->msm_reduce2 = atoi(e);
    if (const char* e = getenv("BZK_MSM_NO_ENDO")) ctx->msm_no_endo = atoi(e) != 0;
    if (const char* e = getenv("BZK_MSM_SPLIT")) ctx->msm_split = atoi(e);
    if (const char* e = getenv("BZK_MSM_SPLIT_MIN_LOG")) ctx->msm_split_min_log = atoi(e);
    if (const char* e = getenv("BZK_MSM_SPLIT_PRIO")) ctx->msm_split_prio = atoi(e);
    if (const char* e = getenv("BZK_MSM_SPLIT_CUTS")) {
        int k = 0;
        for (const char* q = e; *q && k < 4; ++k) {
            ctx->msm_split_cuts[k] = atoi(q);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    if (const char* e = getenv("BZK_DEBUG")) ctx->debug = atoi(e) != 0;
    if (const char* e = getenv("BZK_TIMING")) ctx->timing = atoi(e) != 0;
    if (const char* e = getenv("BZK_NO_COOP")) ctx->no_coop = atoi(e) != 0;
    *out = ctx;
    return BZK_OK;
}

void bzk_ctx_destroy(bzk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (bzk_lane_thread* t : ctx->lane_threads) {
        {
            std::unique_lock<std::mutex> lk(t->m);
            t->cv.wait(lk, [&] { return !t->busy; });
            t->quit = true;
            t->cv.notify_all();
        }
        if (t->th.joinable()) t->th.join();
        delete t;
    }
    for (bzk_ctx* c : ctx->lanes) bzk_ctx_destroy(c);
    for (bzk_ctx* c : ctx->parts) bzk_ctx_destroy(c);
    if (ctx->split_terms) (void)hipFree(ctx->split_terms);
    if (ctx->split_conv) (void)hipFree(ctx->split_conv);
    if (ctx->split_ev) (void)hipEventDestroy(ctx->split_ev);
    for (auto& r : ctx->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (auto& p : ctx->poseidon_dev)
        if (p) (void)hipFree(p);
    bzk::ntt_free_tables(ctx);
    bzk::witfill_free(ctx);
    for (bzk_staged* st : ctx->staged_pool) {  // (handles still out at this point are the caller's leak: bzk_staged_free before bzk_ctx_destroy)
        (void)hipFree(st->buf);
        (void)hipEventDestroy(st->ready);
        (void)hipHostFree(st->flags_host);
        delete st;
    }
    if (ctx->ev_z) (void)hipEventDestroy(ctx->ev_z);
    if (ctx->aux) {
        (void)hipStreamSynchronize(ctx->aux);
        (void)hipStreamDestroy(ctx->aux);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->hprio) {
        (void)hipStreamSynchronize(ctx->hprio);
        (void)hipStreamDestroy(ctx->hprio);
    }
    if (ctx->ev_h) (void)hipEventDestroy(ctx->ev_h);
    if (ctx->heavy) {
        (void)hipStreamSynchronize(ctx->heavy);
        (void)hipStreamDestroy(ctx->heavy);
    }
    if (ctx->ev_heavy_in) (void)hipEventDestroy(ctx->ev_heavy_in);
    if (ctx->ev_heavy_out) (void)hipEventDestroy(ctx->ev_heavy_out);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int32_t bzk_sync(bzk_ctx* ctx) {
    if (!ctx) return BZK_E_ARG;
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

const char* bzk_last_error(bzk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null ctx"; }
int32_t bzk_last_refusal(bzk_ctx* ctx) { return ctx ? ctx->last_refusal : BZK_REFUSE_NONE; }

int32_t bzk_dev_alloc(bzk_ctx* ctx, uint64_t bytes, void** dptr) {
    if (!ctx || !dptr) return BZK_E_ARG;
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
        (void)hipGetLastError();
        return BZK_E_ALLOC;
    }
    return BZK_OK;
}

int32_t bzk_dev_free(bzk_ctx* ctx, void* dptr) {
    if (!ctx) return BZK_E_ARG;
    if (!dptr) return BZK_OK;
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    BZK_HIP(ctx, hipFree(dptr));
    return BZK_OK;
}

int32_t bzk_h2d(bzk_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return BZK_E_ARG;
    BZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

int32_t bzk_d2h(bzk_ctx* ctx, void* dst, const void* src, uint64_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return BZK_E_ARG;
    BZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

int32_t bzk_prof_enable(bzk_ctx* ctx, int32_t on) {
    if (!ctx) return BZK_E_ARG;
    ctx->prof = on != 0;
    return BZK_OK;
}

int32_t bzk_prof_filter(bzk_ctx* ctx, const char* substr) {
    if (!ctx) return BZK_E_ARG;
    ctx->prof_only = substr ? substr : "";
    for (bzk_ctx* c : ctx->lanes) c->prof_only = ctx->prof_only;
    for (bzk_ctx* c : ctx->parts) c->prof_only = ctx->prof_only;
    return BZK_OK;
}

int32_t bzk_prof_reset(bzk_ctx* ctx) {
    if (!ctx) return BZK_E_ARG;
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (bzk_ctx* c : ctx->lanes) BZK_TRY(bzk_prof_reset(c));
    for (bzk_ctx* c : ctx->parts) BZK_TRY(bzk_prof_reset(c));
    for (auto& r : ctx->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    ctx->recs.clear();
    return BZK_OK;
}

int32_t bzk_prof_query(bzk_ctx* ctx, const char* name, uint64_t* launches, double* total_ms) {
    if (!ctx || !name) return BZK_E_ARG;
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t n = 0;
    double tot = 0;
    std::vector<bzk_ctx*> kids(ctx->lanes);
    kids.insert(kids.end(), ctx->parts.begin(), ctx->parts.end());
    for (bzk_ctx* c : kids) {  // kernels launched on the lanes / window-range children count towards the parent's totals
        uint64_t ln = 0;
        double lt = 0;
        BZK_TRY(bzk_prof_query(c, name, &ln, &lt));
        n += ln;
        tot += lt;
    }
    for (auto& r : ctx->recs) {
        if (strcmp(r.name, name) != 0) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            tot += ms;
            ++n;
        }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    return BZK_OK;
}

int32_t bzk_prof_dump(bzk_ctx* ctx, char* buf, uint64_t cap) {
    if (!ctx || !buf || !cap) return BZK_E_ARG;
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::map<std::string, std::pair<uint64_t, double>> agg;
    std::vector<bzk_ctx*> all(ctx->lanes);
    for (bzk_ctx* l : ctx->lanes) all.insert(all.end(), l->parts.begin(), l->parts.end());
    all.insert(all.end(), ctx->parts.begin(), ctx->parts.end());
    all.push_back(ctx);
    for (bzk_ctx* c : all) {
        (void)hipStreamSynchronize(c->stream);
        for (auto& r : c->recs) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                auto& e = agg[r.name];
                e.first++;
                e.second += ms;
            }
        }
    }
    std::string s;
    for (auto& kv : agg) {
        char line[256];
        snprintf(line, sizeof line, "%s %llu %.6f\n", kv.first.c_str(), (unsigned long long)kv.second.first, kv.second.second);
        s += line;
    }
    strncpy(buf, s.c_str(), cap - 1);
    buf[cap - 1] = 0;
    return BZK_OK;
}

}  // extern "C"
