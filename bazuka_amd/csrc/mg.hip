// K6 / row (e): multi-GPU entry points behind the C ABI (bzk_mg_*).
//
// What shards (reference analogue of the replica split: /root/reference/src/mpn/mod.rs:79-107):
//   * one MSM  - by SCALAR-WINDOW RANGE: rank r of `world` computes windows [W r / world, W (r + 1) / world) of the signed c-bit
//                recoding over ALL points, from a resident base set replicated on every device when the CRS is loaded (static,
//                converted once: no rank converts or re-uploads anything per call).  Each rank stops at its WINDOW SUMS (device
//                memory).  The only exchange of the call is one all-gather of those sums - W x 192 B (G1) / 384 B (G2) for the whole
//                group, 3 - 6 KB - after which the Horner combine over the W windows runs once, on the host, exactly as in the
//                single-GPU call (a point doubling is a ~10 us dependency chain on a GPU lane and ~0.3 us on a host core; there are
//                256 of them).  RCCL's reduction operators cannot add curve points, hence all-gather + combine instead of all-reduce.
//   * proofs   - do not shard (NTT / witness / assembly are per proof): replicas.  A pool of prover slots (context + streams + scratch,
//                CRS shared per device) over all devices takes proofs from one queue.
// Two deployments, one code path:
//   bzk_mg_create(device_ids, n)              one process drives n devices (one persistent host thread per device)
//   bzk_mg_create_rank(device, rank, world)   one process per GPU (torchrun style); the 128-byte id from bzk_mg_unique_id() is
//                                             handed to the other ranks by the host's own means (bench.py: torch.distributed)
// Exchange transports (BZK_MG_X_*):
//   RCCL  ncclAllGather on uint8 over xGMI (librccl is dlopen'ed: ncclCommInitAll in one process, ncclCommInitRank across processes)
//   HOST  one process: every device copies its sums into one pinned host array; across processes: a POSIX shared-memory segment
//         named after the group id, double-buffered, sequence-numbered (single node by contract) - also what ranks that SHARE a
//         device use (RCCL refuses two ranks on one GPU), i.e. the rehearsal mode on a one-GPU box
//   PEER  one process: hipMemcpyPeerAsync into device 0's gather buffer, one read-back from there
// AUTO = RCCL when the group spans distinct devices and librccl loads, else HOST.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <set>
#include <thread>

#include "bzk_internal.h"
#include "mg_exchange.h"

struct bzk_mg_params;

namespace {

// the window partition, the exchange record and the shared-memory transport live in mg_exchange.h (no HIP in there: the CPU suite runs that very code
// with the device stage replaced by oracle window sums, tests/host/mgx_check.cpp)
using bzk::mgx::slots_per_rank;
using bzk::mgx::window_range;
constexpr int MG_MAX_W = bzk::mgx::MAX_W;
constexpr size_t MG_SLOT_G2 = bzk::mgx::SLOT_G2;
constexpr size_t MG_SLOT_MAX = bzk::mgx::SLOT_MAX;  // per window: a G2 window sum, or the <= 11 terms of a G1 bucket set (round 6)
typedef bzk::mgx::Hdr MgHdr;
constexpr size_t MG_HDR = sizeof(MgHdr);
static_assert(BZK_MG_UID_BYTES == bzk::mgx::UID_BYTES, "group id");

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommInitAll && CommDestroy && AllGather; }
};
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a process that already carries an RCCL (torch's, when bench.py runs under the nccl backend) gets that very library
        for (const char* name : {"librccl.so.1", "librccl.so"})
            if ((api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (api.lib || (api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        }
        if (!api.lib) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    });
    return &api;
}

// persistent host thread bound to one local device
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int32_t()> job;
    bool has_job = false, done = false, quit = false;
    int32_t status = BZK_OK;
};

}  // namespace

struct bzk_mg {
    int world = 1, rank0 = 0, n_local = 1;
    bool multi_process = false;
    uint32_t exchange = BZK_MG_X_HOST;
    std::vector<int> devices;
    std::vector<bzk_ctx*> ctxs;
    std::vector<Worker*> workers;  // n_local > 1 only
    std::string last_error;
    std::mutex call_mutex;  // one collective call at a time per group
    // exchange buffers
    std::vector<void*> d_send, d_all;  // per local device: own window sums / every rank's (RCCL, PEER: d_all of local 0 only)
    std::vector<void*> d_stage;        // per local device: grow-only staging of host scalars
    std::vector<size_t> d_stage_bytes;
    uint8_t* h_win = nullptr;          // pinned: world x MG_MAX_W slots
    std::vector<ncclComm_t> comms;
    // multi-process HOST exchange
    bzk::mgx::ShmExchange shmx;
    bool shm_open_ = false;
    uint64_t seq = 0;
    // bzk_mg_stats: where a window-sharded call's time goes on THIS rank (local device 0), cumulative since creation / the last reset
    struct Stats {
        uint64_t calls = 0;
        double local_ms = 0, exchange_ms = 0, peer_wait_ms = 0, combine_ms = 0, create_s = 0, comm_init_s = 0;
    } stats;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // start of the call / local stage queued / exchange queued, on local device 0's stream
};
// proof pool: `slots` prover slots (context + lanes + scratch) per local device over one shared CRS per device; one host thread per
// slot takes proofs from a common queue - whichever slot is free next, on whichever device (replicas: proofs do not shard)
struct MgProofJob {
    bzk_assignment asg;  // by value: the caller's 48-byte struct may be gone by the time a slot takes the job (only the ARRAYS must live)
    uint8_t r[32], s[32];
    uint8_t* out;
    uint64_t ticket;
};
struct bzk_mg_params {
    bzk_mg* mg = nullptr;
    struct Slot { int dev_index; bzk_ctx* ctx; bzk_params* params; std::thread th; uint64_t proofs = 0; };
    std::vector<Slot*> slots;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<MgProofJob> queue;
    std::vector<std::pair<uint64_t, int32_t>> finished;  // (ticket, status)
    std::set<uint64_t> outstanding;                      // submitted and not yet handed back by bzk_mg_prove_wait
    std::vector<std::pair<uint64_t, std::string>> errors;
    uint64_t next_ticket = 1;
    bool quit = false;
};
struct bzk_mg_bases {
    std::vector<bzk_msm_bases*> per_dev;
    uint64_t n = 0;
    int g2 = 0;
};

namespace {

int32_t mg_fail(bzk_mg* mg, int32_t st, const std::string& why) {
    mg->last_error = why;
    return st;
}

void worker_loop(Worker* w, int device) {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        auto job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        const int32_t st = job();
        lk.lock();
        w->status = st;
        w->done = true;
        w->cv.notify_all();
    }
}

// runs fn(i) for every local device i - inline when the group has one, on the persistent threads otherwise; first failure wins
int32_t run_all(bzk_mg* mg, const std::function<int32_t(int)>& fn) {
    if (mg->n_local == 1) {
        (void)hipSetDevice(mg->devices[0]);
        const int32_t st = fn(0);
        if (st != BZK_OK && !mg->ctxs.empty()) mg->last_error = "device " + std::to_string(mg->devices[0]) + ": " + mg->ctxs[0]->last_error;
        return st;
    }
    for (int i = 0; i < mg->n_local; ++i) {
        Worker* w = mg->workers[i];
        std::lock_guard<std::mutex> lk(w->m);
        w->job = [&fn, i] { return fn(i); };
        w->done = false;
        w->has_job = true;
        w->cv.notify_all();
    }
    int32_t st = BZK_OK;
    for (int i = 0; i < mg->n_local; ++i) {
        Worker* w = mg->workers[i];
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return w->done; });
        if (st == BZK_OK && w->status != BZK_OK) {
            st = w->status;
            mg->last_error = "device " + std::to_string(mg->devices[i]) + ": " + mg->ctxs[i]->last_error;
        }
    }
    return st;
}

int32_t mg_alloc_buffers(bzk_mg* mg) {
    const size_t slot_all = (size_t)mg->world * (MG_MAX_W * MG_SLOT_MAX + MG_HDR);
    mg->d_send.assign(mg->n_local, nullptr);
    mg->d_all.assign(mg->n_local, nullptr);
    mg->d_stage.assign(mg->n_local, nullptr);
    mg->d_stage_bytes.assign(mg->n_local, 0);
    for (int i = 0; i < mg->n_local; ++i) {
        if (hipSetDevice(mg->devices[i]) != hipSuccess) return mg_fail(mg, BZK_E_DEVICE, "hipSetDevice");
        if (hipMalloc(&mg->d_send[i], (size_t)MG_MAX_W * MG_SLOT_MAX + MG_HDR) != hipSuccess) return mg_fail(mg, BZK_E_ALLOC, "exchange buffer");
        if (hipMalloc(&mg->d_all[i], slot_all) != hipSuccess) return mg_fail(mg, BZK_E_ALLOC, "exchange buffer");
    }
    if (hipHostMalloc((void**)&mg->h_win, slot_all, hipHostMallocPortable) != hipSuccess) return mg_fail(mg, BZK_E_ALLOC, "pinned exchange buffer");
    return BZK_OK;
}

int32_t mg_open_shm(bzk_mg* mg, const uint8_t uid[BZK_MG_UID_BYTES]) {
    std::string err;
    if (!mg->shmx.open(uid, mg->world, mg->rank0, err)) return mg_fail(mg, BZK_E_DEVICE, err);
    mg->shm_open_ = true;
    return BZK_OK;
}

int32_t shm_barrier(bzk_mg* mg, uint64_t seq) {
    std::string err;
    if (!mg->shmx.barrier(seq, err)) return mg_fail(mg, BZK_E_DEVICE, err);
    return BZK_OK;
}

static int32_t mg_finish_create_impl(bzk_mg* mg, uint32_t exchange, const uint8_t* uid);
int32_t mg_finish_create(bzk_mg* mg, uint32_t exchange, const uint8_t* uid) {
    const auto t0 = std::chrono::steady_clock::now();
    const int32_t st = mg_finish_create_impl(mg, exchange, uid);
    mg->stats.create_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st == BZK_OK && !mg->devices.empty()) {  // timing events of bzk_mg_stats; a failure here only leaves the spans at zero
        (void)hipSetDevice(mg->devices[0]);
        for (auto& e : mg->ev)
            if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); e = nullptr; }
    }
    return st;
}
static int32_t mg_finish_create_impl(bzk_mg* mg, uint32_t exchange, const uint8_t* uid) {
    // contexts (own non-blocking stream each)
    for (int i = 0; i < mg->n_local; ++i) {
        bzk_ctx* c = nullptr;
        const int32_t st = bzk_ctx_create(mg->devices[i], nullptr, &c);
        if (st != BZK_OK) return mg_fail(mg, st, "bzk_ctx_create(device " + std::to_string(mg->devices[i]) + ")");
        c->msm_no_endo = true;  // the group shards PLAIN windows: its resident sets carry no endomorphism images (E x the memory for nothing)
        mg->ctxs.push_back(c);
    }
    BZK_TRY(mg_alloc_buffers(mg));
    std::set<int> distinct(mg->devices.begin(), mg->devices.end());
    const bool shared_device = mg->multi_process ? false : (int)distinct.size() != mg->n_local;
    if (const char* e = getenv("BZK_MG_EXCHANGE")) {  // A/B runs and rehearsals: host | peer | rccl
        if (!strcmp(e, "host")) exchange = BZK_MG_X_HOST;
        else if (!strcmp(e, "peer")) exchange = BZK_MG_X_PEER;
        else if (!strcmp(e, "rccl")) exchange = BZK_MG_X_RCCL;
    }
    if (exchange == BZK_MG_X_AUTO) exchange = (mg->world > 1 && !shared_device && rccl_api()->ok()) ? BZK_MG_X_RCCL : BZK_MG_X_HOST;
    if (exchange == BZK_MG_X_PEER && mg->multi_process) return mg_fail(mg, BZK_E_ARG, "PEER exchange needs one process driving all devices");
    if (exchange == BZK_MG_X_RCCL) {
        RcclApi* R = rccl_api();
        if (!R->ok()) return mg_fail(mg, BZK_E_DEVICE, "librccl could not be loaded");
        if (shared_device) return mg_fail(mg, BZK_E_ARG, "RCCL refuses two ranks on one device: use the HOST exchange");
        mg->comms.assign(mg->n_local, nullptr);
        ncclResult_t r;
        if (mg->multi_process) {
            ncclUniqueId id;
            memcpy(id.internal, uid, NCCL_UNIQUE_ID_BYTES);
            (void)hipSetDevice(mg->devices[0]);
            const auto t0 = std::chrono::steady_clock::now();
            r = R->CommInitRank(&mg->comms[0], mg->world, id, mg->rank0);
            mg->stats.comm_init_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            r = R->CommInitAll(mg->comms.data(), mg->n_local, mg->devices.data());
            mg->stats.comm_init_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        if (r != ncclSuccess) return mg_fail(mg, BZK_E_DEVICE, std::string("RCCL communicator: ") + (R->GetErrorString ? R->GetErrorString(r) : "error"));
    } else if (exchange == BZK_MG_X_PEER) {
        for (int i = 1; i < mg->n_local; ++i) {
            if (mg->devices[i] == mg->devices[0]) continue;
            (void)hipSetDevice(mg->devices[i]);
            if (hipDeviceEnablePeerAccess(mg->devices[0], 0) != hipSuccess) (void)hipGetLastError();  // already enabled / staged copies still work
        }
    } else if (mg->multi_process) {
        BZK_TRY(mg_open_shm(mg, uid));
        BZK_TRY(shm_barrier(mg, ++mg->seq));  // everybody has mapped the segment ...
        if (mg->rank0 == 0) mg->shmx.unlink_name();  // ... so its name can go: nothing is left behind whatever happens later
    }
    mg->exchange = exchange;
    if (mg->n_local > 1) {
        for (int i = 0; i < mg->n_local; ++i) {
            Worker* w = new Worker();
            w->th = std::thread(worker_loop, w, mg->devices[i]);
            mg->workers.push_back(w);
        }
    }
    return BZK_OK;
}

int32_t stage_scalars(bzk_mg* mg, int i, const uint8_t* host, uint64_t n, const void** out) {
    bzk_ctx* c = mg->ctxs[i];
    const size_t bytes = (size_t)n * 32;
    if (bytes > mg->d_stage_bytes[i]) {
        if (mg->d_stage[i]) {
            BZK_HIP(c, hipStreamSynchronize(c->stream));
            (void)hipFree(mg->d_stage[i]);
            mg->d_stage[i] = nullptr;
            mg->d_stage_bytes[i] = 0;
        }
        if (hipMalloc(&mg->d_stage[i], bytes) != hipSuccess) {
            (void)hipGetLastError();
            c->last_error = "bzk_mg: scalar staging allocation";
            return BZK_E_ALLOC;
        }
        mg->d_stage_bytes[i] = bytes;
    }
    BZK_HIP(c, hipMemcpyAsync(mg->d_stage[i], host, bytes, hipMemcpyHostToDevice, c->stream));
    *out = mg->d_stage[i];
    return BZK_OK;
}

// one window-sharded MSM.  scalars_dev: n_local device pointers (or null with scalars_host set: staged per device)
int32_t mg_msm(bzk_mg* mg, const bzk_mg_bases* B, int g2, const void* const* scalars_dev, const uint8_t* scalars_host, uint64_t n, uint32_t flags,
               uint8_t* out) {
    if (!mg || !B || !out || (n && !scalars_dev && !scalars_host)) return BZK_E_ARG;
    if (B->g2 != g2 || n > B->n || (int)B->per_dev.size() != mg->n_local) return BZK_E_ARG;
    std::lock_guard<std::mutex> call(mg->call_mutex);
    // what a rank leaves per window: a window sum (G2: 384 bytes), or - G1 since round 6 - the TERMS of the window's bucket set (msm_impl.cuh section 6b:
    // c / 2 + 1 points of 192 bytes; a function of n and the environment alone, so every rank sizes its block alike): the ranks' chunked running sums with
    // their 41-link chains (0.6 + 0.2 ms per rank whatever the group's size) are gone from the window-sharded G1 MSM as they are from the single-GPU one
    const int n_terms = g2 ? 0 : bzk::msm_g1_window_terms(n ? n : 1);
    const size_t sz = g2 ? MG_SLOT_G2 : (size_t)(n_terms ? n_terms : 1) * (MG_SLOT_G2 / 2);
    if (sz > MG_SLOT_MAX) return mg_fail(mg, BZK_E_INTERNAL, "terms per window");
    auto windows = g2 ? bzk::msm_g2_windows_dev : bzk::msm_g1_windows_dev;
    auto horner = g2 ? bzk::g2_horner_packed : (n_terms ? bzk::g1_horner_terms_packed : bzk::g1_horner_packed);
    // the window count is a function of n alone (bzk_msm_window_count): every rank derives the same partition without talking
    const int W = (int)bzk_msm_window_count(n ? n : 1);
    if (W > MG_MAX_W) return mg_fail(mg, BZK_E_INTERNAL, "window count");
    const int spr = slots_per_rank(W, mg->world);
    const size_t blk = (size_t)spr * sz;  // a rank's sums in an exchange; the RCCL / shared-memory block carries an MgHdr after them
    const uint64_t seq = ++mg->seq;
    std::vector<int32_t> cs(mg->n_local, 0);
    std::vector<MgHdr> hdr(mg->n_local);
    const uint32_t x = mg->exchange;
    const bool shm = mg->multi_process && x == BZK_MG_X_HOST;
    int32_t st = run_all(mg, [&](int i) -> int32_t {
        bzk_ctx* c = mg->ctxs[i];
        const int rank = mg->rank0 + i;
        int lo, hi;
        window_range(W, rank, mg->world, &lo, &hi);
        const bool timed = i == 0 && mg->ev[0] && mg->ev[1] && mg->ev[2];
        if (timed) (void)hipEventRecord(mg->ev[0], c->stream);
        // local stage; its status travels with the exchange (RCCL, shared memory: the peers cannot see it otherwise)
        const int32_t lst = [&]() -> int32_t {
#ifdef BZK_TEST_HOOKS
            // fault injection, compiled into bazuka_amd/libbzk_testhooks.so only (a stray environment variable must not be able
            // to fail a production call): "rank:call" makes that rank's local stage of that call of the group fail (tests/test_gpu_mg.py)
            static const char* fault = getenv("BZK_MG_TEST_FAULT");
            if (fault) {
                int fr = -1;
                unsigned long long fs = 0;
                if (sscanf(fault, "%d:%llu", &fr, &fs) == 2 && fr == rank && fs == seq) { c->last_error = "bzk_mg: injected fault"; return BZK_E_ALLOC; }
            }
#endif
            const void* sc = scalars_dev ? scalars_dev[i] : nullptr;
            if (!scalars_dev && n) BZK_TRY(stage_scalars(mg, i, scalars_host, n, &sc));
            int32_t info[5] = {0, 0, 0, 0, 0};
            // a rank without windows (world > W) still takes part in the exchange
            if (hi > lo && n) {
                BZK_TRY(windows(c, B->per_dev[i], nullptr, sc, n, flags, lo, hi, mg->d_send[i], info));
                if (info[1] != W) { c->last_error = "bzk_mg: window count disagrees with bzk_msm_window_count"; return BZK_E_INTERNAL; }
                if (info[4] != n_terms) { c->last_error = "bzk_mg: terms per window disagree with msm_g1_window_terms"; return BZK_E_INTERNAL; }
                cs[i] = info[0];
            }
            return BZK_OK;
        }();
        hdr[i] = MgHdr{seq, lst, cs[i]};
        if (timed) (void)hipEventRecord(mg->ev[1], c->stream);
        const size_t mine = lst == BZK_OK ? (size_t)(hi - lo) * sz : 0;
        if (x == BZK_MG_X_RCCL) {
            RcclApi* R = rccl_api();
            if (hipMemcpyAsync((char*)mg->d_send[i] + blk, &hdr[i], MG_HDR, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                (void)hipGetLastError();  // the stream is gone: nothing can be told to the peers through it
                if (lst == BZK_OK) c->last_error = "bzk_mg: exchange record upload";
                return lst != BZK_OK ? lst : BZK_E_DEVICE;
            }
            const ncclResult_t r = R->AllGather(mg->d_send[i], mg->d_all[i], blk + MG_HDR, ncclUint8, mg->comms[i], c->stream);
            if (r != ncclSuccess) {
                if (lst == BZK_OK) c->last_error = std::string("ncclAllGather: ") + (R->GetErrorString ? R->GetErrorString(r) : "error");
                return lst != BZK_OK ? lst : BZK_E_DEVICE;
            }
            if (i == 0) BZK_HIP(c, hipMemcpyAsync(mg->h_win, mg->d_all[0], (size_t)mg->world * (blk + MG_HDR), hipMemcpyDeviceToHost, c->stream));
        } else if (x == BZK_MG_X_PEER) {
            if (mine)
                BZK_HIP(c, hipMemcpyPeerAsync((char*)mg->d_all[0] + (size_t)rank * blk, mg->devices[0], mg->d_send[i], mg->devices[i], mine, c->stream));
        } else if (mine) {  // HOST: straight into the shared pinned array (one process) / this rank's staging (shared memory below)
            BZK_HIP(c, hipMemcpyAsync(mg->h_win + (size_t)rank * blk, mg->d_send[i], mine, hipMemcpyDeviceToHost, c->stream));
        }
        if (timed) (void)hipEventRecord(mg->ev[2], c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) {
            (void)hipGetLastError();
            if (lst == BZK_OK) c->last_error = "bzk_mg: exchange synchronisation";
            return lst != BZK_OK ? lst : BZK_E_DEVICE;
        }
        if (timed) {  // device-side spans of this call: local windows, then the all-gather / peer copy / read-back behind them
            float a = 0, b = 0;
            if (hipEventElapsedTime(&a, mg->ev[0], mg->ev[1]) == hipSuccess && hipEventElapsedTime(&b, mg->ev[1], mg->ev[2]) == hipSuccess) {
                mg->stats.local_ms += a;
                mg->stats.exchange_ms += b;
            } else {
                (void)hipGetLastError();
            }
        }
        return lst;
    });
    ++mg->stats.calls;
    if (st != BZK_OK && !shm) return st;   // (RCCL: the peers read this rank's status from the gathered records below)
    if (x == BZK_MG_X_RCCL) {
        // compact the gathered blocks [sums | record] into the plain per-rank layout the combine below reads
        for (int r = 0; r < mg->world; ++r) {
            MgHdr h;
            memcpy(&h, mg->h_win + (size_t)r * (blk + MG_HDR) + blk, MG_HDR);
            const std::string why = bzk::mgx::judge(h, seq, r, false);
            if (!why.empty()) return mg_fail(mg, BZK_E_DEVICE, why);
            if (r) memmove(mg->h_win + (size_t)r * blk, mg->h_win + (size_t)r * (blk + MG_HDR), blk);
        }
    }
    if (x == BZK_MG_X_PEER) {
        bzk_ctx* c = mg->ctxs[0];
        (void)hipSetDevice(mg->devices[0]);
        if (hipMemcpyAsync(mg->h_win, mg->d_all[0], (size_t)mg->world * blk, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            return mg_fail(mg, BZK_E_DEVICE, "peer gather read-back");
    }
    // the window size, like the window count, is a function of n alone for calls that name a window range
    const int c_bits = bzk::msm_window_bits(n);
    if (st == BZK_OK)
        for (int v : cs)
            if (v && v != c_bits) { st = mg_fail(mg, BZK_E_INTERNAL, "window size disagrees with msm_window_bits"); break; }
    if (shm) {
        // shared-memory all-gather (mg_exchange.h): [record | own sums] into slot [parity][rank], sequence-numbered arrival.  A rank that failed
        // locally still arrives - with its status in the record - so every rank of the group returns an error for this call
        std::string err;
        const int32_t gst = mg->shmx.all_gather(seq, st, cs[0], mg->h_win + (size_t)mg->rank0 * blk, W, sz, blk, mg->h_win, &mg->stats.peer_wait_ms, err);
        if (st != BZK_OK) return st;
        if (gst != BZK_OK) return mg_fail(mg, gst, err);
    }
    // compact the per-rank slots into window order and combine
    std::vector<uint8_t> S((size_t)W * sz);
    if (n) bzk::mgx::compact_to_window_order(mg->h_win, W, mg->world, sz, blk, S.data());
    const auto tc0 = std::chrono::steady_clock::now();
    const int32_t hst = horner(S.data(), n ? W : 0, c_bits, 0, out);
    mg->stats.combine_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
    return hst;
}

}  // namespace

extern "C" {

int32_t bzk_mg_unique_id(uint8_t uid[BZK_MG_UID_BYTES]) {
    if (!uid) return BZK_E_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) { (void)hipGetLastError(); n_dev = 0; }
    RcclApi* R = n_dev > 0 ? rccl_api() : nullptr;  // without a device RCCL has nothing to offer (and says so loudly)
    if (R && R->ok()) {
        ncclUniqueId id;
        if (R->GetUniqueId(&id) == ncclSuccess) {
            memcpy(uid, id.internal, BZK_MG_UID_BYTES);
            return BZK_OK;
        }
    }
    // no RCCL in this process: any 128 random bytes name a HOST-exchange group
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0) return BZK_E_DEVICE;
    const ssize_t got = read(fd, uid, BZK_MG_UID_BYTES);
    close(fd);
    return got == BZK_MG_UID_BYTES ? BZK_OK : BZK_E_DEVICE;
}

int32_t bzk_mg_probe(int32_t device_id) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) { (void)hipGetLastError(); n_dev = 0; }
    int32_t mask = 0;
    if (device_id >= 0 && device_id < n_dev) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && strstr(prop.gcnArchName, "gfx950")) mask |= 1;
        else (void)hipGetLastError();
    }
    if (n_dev > 0 && rccl_api()->ok()) mask |= 2;
    return mask;
}

int32_t bzk_mg_create(const int32_t* device_ids, int32_t n_devices, uint32_t exchange, bzk_mg** out) {
    if (!out) return BZK_E_ARG;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64 || exchange > BZK_MG_X_RCCL) return BZK_E_ARG;
    bzk_mg* mg = new (std::nothrow) bzk_mg();
    if (!mg) return BZK_E_ALLOC;
    mg->world = mg->n_local = n_devices;
    mg->devices.assign(device_ids, device_ids + n_devices);
    const int32_t st = mg_finish_create(mg, exchange, nullptr);
    if (st != BZK_OK) {
        fprintf(stderr, "libbzk: bzk_mg_create: %s\n", mg->last_error.c_str());
        bzk_mg_destroy(mg);
        return st;
    }
    *out = mg;
    return BZK_OK;
}

int32_t bzk_mg_create_rank(int32_t device_id, int32_t rank, int32_t world, const uint8_t uid[BZK_MG_UID_BYTES], uint32_t exchange, bzk_mg** out) {
    if (!out) return BZK_E_ARG;
    *out = nullptr;
    if (!uid || world < 1 || world > 64 || rank < 0 || rank >= world || exchange > BZK_MG_X_RCCL) return BZK_E_ARG;
    bzk_mg* mg = new (std::nothrow) bzk_mg();
    if (!mg) return BZK_E_ALLOC;
    mg->world = world;
    mg->rank0 = rank;
    mg->n_local = 1;
    mg->multi_process = true;
    mg->devices.assign(1, device_id);
    const int32_t st = mg_finish_create(mg, exchange, uid);
    if (st != BZK_OK) {
        fprintf(stderr, "libbzk: bzk_mg_create_rank(rank %d of %d): %s\n", rank, world, mg->last_error.c_str());
        bzk_mg_destroy(mg);
        return st;
    }
    *out = mg;
    return BZK_OK;
}

void bzk_mg_destroy(bzk_mg* mg) {
    if (!mg) return;
    for (Worker* w : mg->workers) {
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->quit = true;
            w->cv.notify_all();
        }
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    for (size_t i = 0; i < mg->comms.size(); ++i)
        if (mg->comms[i]) {
            (void)hipSetDevice(mg->devices[i]);
            (void)rccl_api()->CommDestroy(mg->comms[i]);
        }
    for (int i = 0; i < (int)mg->ctxs.size(); ++i) {
        (void)hipSetDevice(mg->devices[i]);
        if (mg->ctxs[i]) (void)hipStreamSynchronize(mg->ctxs[i]->stream);
        if (i < (int)mg->d_send.size() && mg->d_send[i]) (void)hipFree(mg->d_send[i]);
        if (i < (int)mg->d_all.size() && mg->d_all[i]) (void)hipFree(mg->d_all[i]);
        if (i < (int)mg->d_stage.size() && mg->d_stage[i]) (void)hipFree(mg->d_stage[i]);
        bzk_ctx_destroy(mg->ctxs[i]);
    }
    if (!mg->devices.empty()) (void)hipSetDevice(mg->devices[0]);
    for (auto& e : mg->ev)
        if (e) (void)hipEventDestroy(e);
    if (mg->h_win) (void)hipHostFree(mg->h_win);
    if (mg->shm_open_) mg->shmx.close();
    delete mg;
}

int32_t bzk_mg_stats(bzk_mg* mg, int32_t reset, double out[8]) {
    if (!mg) return BZK_E_ARG;
    std::lock_guard<std::mutex> call(mg->call_mutex);
    if (out) {
        const bzk_mg::Stats& t = mg->stats;
        const double v[8] = {(double)t.calls, t.local_ms, t.exchange_ms, t.peer_wait_ms, t.combine_ms, t.create_s, t.comm_init_s, 0.0};
        memcpy(out, v, sizeof v);
    }
    if (reset) {
        const double cs = mg->stats.create_s, ci = mg->stats.comm_init_s;
        mg->stats = bzk_mg::Stats();
        mg->stats.create_s = cs;
        mg->stats.comm_init_s = ci;
    }
    return BZK_OK;
}
int32_t bzk_mg_world(const bzk_mg* mg) { return mg ? mg->world : 0; }
int32_t bzk_mg_local(const bzk_mg* mg) { return mg ? mg->n_local : 0; }
int32_t bzk_mg_rank(const bzk_mg* mg) { return mg ? mg->rank0 : -1; }
uint32_t bzk_mg_exchange(const bzk_mg* mg) { return mg ? mg->exchange : 0; }
bzk_ctx* bzk_mg_ctx(bzk_mg* mg, int32_t i) { return (mg && i >= 0 && i < mg->n_local) ? mg->ctxs[i] : nullptr; }
const char* bzk_mg_last_error(bzk_mg* mg) { return mg ? mg->last_error.c_str() : "null group"; }

static int32_t mg_bases_load(bzk_mg* mg, int g2, const uint8_t* host, const void* const* dev, uint64_t n, bzk_mg_bases** out) {
    if (!mg || !out || (!host && !dev) || n == 0) return BZK_E_ARG;
    *out = nullptr;
    bzk_mg_bases* B = new (std::nothrow) bzk_mg_bases();
    if (!B) return BZK_E_ALLOC;
    B->n = n;
    B->g2 = g2;
    B->per_dev.assign(mg->n_local, nullptr);
    const size_t raw = g2 ? 192 : 96;
    std::lock_guard<std::mutex> call(mg->call_mutex);
    const int32_t st = run_all(mg, [&](int i) -> int32_t {
        bzk_ctx* c = mg->ctxs[i];
        // two local ranks on one device (rehearsal groups) share the first one's set: it is read-only
        for (int j = 0; j < i; ++j)
            if (mg->devices[j] == mg->devices[i]) return BZK_OK;
        const void* src = dev ? dev[i] : nullptr;
        void* tmp = nullptr;
        if (!dev) {
            if (hipMalloc(&tmp, n * raw) != hipSuccess) { (void)hipGetLastError(); c->last_error = "bzk_mg: raw base staging"; return BZK_E_ALLOC; }
            if (hipMemcpyAsync(tmp, host, n * raw, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipFree(tmp); return BZK_E_DEVICE; }
            src = tmp;
        }
        const int32_t s = g2 ? bzk_msm_g2_bases_load_dev(c, src, n, &B->per_dev[i]) : bzk_msm_g1_bases_load_dev(c, src, n, &B->per_dev[i]);
        if (tmp) (void)hipFree(tmp);
        return s;
    });
    if (st == BZK_OK)
        for (int i = 0; i < mg->n_local; ++i)
            for (int j = 0; j < i && !B->per_dev[i]; ++j)
                if (mg->devices[j] == mg->devices[i]) B->per_dev[i] = B->per_dev[j];
    if (st != BZK_OK) {
        bzk_mg_bases_free(mg, B);
        return st;
    }
    *out = B;
    return BZK_OK;
}
int32_t bzk_mg_bases_g1_load(bzk_mg* mg, const uint8_t* bases_host, uint64_t n, bzk_mg_bases** out) { return mg_bases_load(mg, 0, bases_host, nullptr, n, out); }
int32_t bzk_mg_bases_g2_load(bzk_mg* mg, const uint8_t* bases_host, uint64_t n, bzk_mg_bases** out) { return mg_bases_load(mg, 1, bases_host, nullptr, n, out); }
int32_t bzk_mg_bases_g1_load_dev(bzk_mg* mg, const void* const* bases_dev, uint64_t n, bzk_mg_bases** out) { return mg_bases_load(mg, 0, nullptr, bases_dev, n, out); }
int32_t bzk_mg_bases_g2_load_dev(bzk_mg* mg, const void* const* bases_dev, uint64_t n, bzk_mg_bases** out) { return mg_bases_load(mg, 1, nullptr, bases_dev, n, out); }
void bzk_mg_bases_free(bzk_mg* mg, bzk_mg_bases* B) {
    if (!B) return;
    std::set<bzk_msm_bases*> freed;
    for (size_t i = 0; i < B->per_dev.size(); ++i) {
        if (!B->per_dev[i] || freed.count(B->per_dev[i])) continue;
        freed.insert(B->per_dev[i]);
        bzk_msm_bases_free(mg && i < mg->ctxs.size() ? mg->ctxs[i] : nullptr, B->per_dev[i]);
    }
    delete B;
}

int32_t bzk_mg_msm_g1_dev(bzk_mg* mg, const bzk_mg_bases* bases, const void* const* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[97]) {
    return mg_msm(mg, bases, 0, scalars_dev, nullptr, n, flags, out);
}
int32_t bzk_mg_msm_g2_dev(bzk_mg* mg, const bzk_mg_bases* bases, const void* const* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[193]) {
    return mg_msm(mg, bases, 1, scalars_dev, nullptr, n, flags, out);
}
int32_t bzk_mg_msm_g1(bzk_mg* mg, const bzk_mg_bases* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[97]) {
    return mg_msm(mg, bases, 0, nullptr, scalars, n, flags, out);
}
int32_t bzk_mg_msm_g2(bzk_mg* mg, const bzk_mg_bases* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[193]) {
    return mg_msm(mg, bases, 1, nullptr, scalars, n, flags, out);
}

// ---- proof pool (replicas) -------------------------------------------------------------------------------------------------
static void mg_slot_main(bzk_mg_params* P, bzk_mg_params::Slot* sl) {
    (void)hipSetDevice(P->mg->devices[sl->dev_index]);
    for (;;) {
        MgProofJob job;
        {
            std::unique_lock<std::mutex> lk(P->m);
            P->cv_work.wait(lk, [&] { return P->quit || !P->queue.empty(); });
            if (P->queue.empty()) return;  // quit and drained
            job = P->queue.front();
            P->queue.pop_front();
        }
        const int32_t st = bzk_groth16_prove(sl->ctx, sl->params, &job.asg, job.r, job.s, job.out);
        {
            std::lock_guard<std::mutex> lk(P->m);
            ++sl->proofs;
            P->finished.push_back({job.ticket, st});
            if (st != BZK_OK) P->errors.push_back({job.ticket, "device " + std::to_string(P->mg->devices[sl->dev_index]) + ": " + sl->ctx->last_error});
            P->cv_done.notify_all();
        }
    }
}

int32_t bzk_mg_params_load(bzk_mg* mg, const bzk_params_desc* desc, uint32_t slots_per_device, bzk_mg_params** out) {
    if (!mg || !desc || !out || slots_per_device < 1 || slots_per_device > 16) return BZK_E_ARG;
    *out = nullptr;
    bzk_mg_params* P = new (std::nothrow) bzk_mg_params();
    if (!P) return BZK_E_ALLOC;
    P->mg = mg;
    int32_t st = BZK_OK;
    std::vector<bzk_params*> first(mg->n_local, nullptr);
    for (int i = 0; i < mg->n_local && st == BZK_OK; ++i) {
        (void)hipSetDevice(mg->devices[i]);
        // local entries that share a device share its CRS too
        for (int j = 0; j < i; ++j)
            if (mg->devices[j] == mg->devices[i]) first[i] = first[j];
        for (uint32_t k = 0; k < slots_per_device && st == BZK_OK; ++k) {
            bzk_mg_params::Slot* sl = new bzk_mg_params::Slot();
            sl->dev_index = i;
            sl->ctx = nullptr;
            sl->params = nullptr;
            P->slots.push_back(sl);
            st = bzk_ctx_create(mg->devices[i], nullptr, &sl->ctx);
            if (st != BZK_OK) break;
            st = first[i] ? bzk_params_slot(sl->ctx, first[i], &sl->params) : bzk_params_load(sl->ctx, desc, &sl->params);
            if (st == BZK_OK && !first[i]) first[i] = sl->params;
            if (st != BZK_OK) mg->last_error = "bzk_mg_params_load, device " + std::to_string(mg->devices[i]) + ": " + sl->ctx->last_error;
        }
    }
    if (st != BZK_OK) {
        bzk_mg_params_free(mg, P);
        return st;
    }
    for (auto* sl : P->slots) sl->th = std::thread(mg_slot_main, P, sl);
    *out = P;
    return BZK_OK;
}

void bzk_mg_params_free(bzk_mg* mg, bzk_mg_params* P) {
    (void)mg;
    if (!P) return;
    {
        std::lock_guard<std::mutex> lk(P->m);
        P->quit = true;
        P->cv_work.notify_all();
    }
    for (auto* sl : P->slots)
        if (sl->th.joinable()) sl->th.join();
    for (auto* sl : P->slots) {
        if (sl->ctx) {
            (void)hipSetDevice(P->mg->devices[sl->dev_index]);
            if (sl->params) bzk_params_free(sl->ctx, sl->params);
            bzk_ctx_destroy(sl->ctx);
        }
        delete sl;
    }
    delete P;
}

uint32_t bzk_mg_params_slots(const bzk_mg_params* P) { return P ? (uint32_t)P->slots.size() : 0; }

int32_t bzk_mg_prove_submit(bzk_mg* mg, bzk_mg_params* P, const bzk_assignment* asg, const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[387],
                            uint64_t* ticket) {
    if (!mg || !P || !asg || !r || !s || !proof_out || !ticket || P->mg != mg) return BZK_E_ARG;
    std::lock_guard<std::mutex> lk(P->m);
    MgProofJob j;
    j.asg = *asg;
    memcpy(j.r, r, 32);
    memcpy(j.s, s, 32);
    j.out = proof_out;
    j.ticket = *ticket = P->next_ticket++;
    P->outstanding.insert(j.ticket);
    P->queue.push_back(j);
    P->cv_work.notify_one();
    return BZK_OK;
}

int32_t bzk_mg_prove_wait(bzk_mg* mg, bzk_mg_params* P, uint64_t ticket) {
    if (!mg || !P || P->mg != mg) return BZK_E_ARG;
    std::unique_lock<std::mutex> lk(P->m);
    if (ticket == 0 || ticket >= P->next_ticket || !P->outstanding.count(ticket)) return BZK_E_ARG;  // unknown or already consumed
    for (;;) {
        for (size_t i = 0; i < P->finished.size(); ++i) {
            if (P->finished[i].first != ticket) continue;
            const int32_t st = P->finished[i].second;
            P->finished.erase(P->finished.begin() + (long)i);
            P->outstanding.erase(ticket);
            for (size_t e = 0; e < P->errors.size(); ++e)
                if (P->errors[e].first == ticket) {
                    mg->last_error = P->errors[e].second;
                    P->errors.erase(P->errors.begin() + (long)e);
                    break;
                }
            return st;
        }
        P->cv_done.wait(lk);
    }
}

int32_t bzk_mg_prove(bzk_mg* mg, bzk_mg_params* P, const bzk_assignment* asg, const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[387]) {
    uint64_t t = 0;
    BZK_TRY(bzk_mg_prove_submit(mg, P, asg, r, s, proof_out, &t));
    return bzk_mg_prove_wait(mg, P, t);
}

// proofs each slot has finished so far (load-balance evidence for tests / the bench): out[i] for slot i, i < bzk_mg_params_slots()
int32_t bzk_mg_params_stats(bzk_mg_params* P, uint64_t* out, uint32_t cap) {
    if (!P || !out) return BZK_E_ARG;
    std::lock_guard<std::mutex> lk(P->m);
    for (uint32_t i = 0; i < cap && i < P->slots.size(); ++i) out[i] = P->slots[i]->proofs;
    return BZK_OK;
}

}  // extern "C"
