// The host-side half of a device group's exchange (mg.hip): the window partition, the (sequence, status) record every rank's block carries, the
// POSIX shared-memory all-gather ranks of different processes use when they cannot (or must not) go through RCCL, and the compaction of the
// gathered blocks into window order.  NO device code and no HIP call in this file: tests/host/mgx_check.cpp runs exactly this code on a GPU-less
// box - several processes, the device stage replaced by window sums from the CPU oracle - so that the transport and its failure protocol are
// exercised by the CPU suite, not only rehearsed on the one-GPU box.
// Reference analogue of what is being distributed: the replica split of /root/reference/src/mpn/mod.rs:79-107; partition: SURVEY.md 8e.
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>

namespace bzk {
namespace mgx {

constexpr int MAX_W = 64;          // windows of a call (c >= 4 -> W <= 64)
constexpr size_t SLOT_G2 = 384;    // bytes of one standard-limb XYZZ window sum (G2; G1 = 192)
// bytes a rank may publish per window: G2 one window sum; G1 since round 6 the TERMS of the window's bucket set (multiplication-free reduction, msm_impl.cuh
// section 6b: c / 2 + 1 <= 11 points of 192 bytes) - the ranks stop before any per-window combination and the host's Horner takes the terms at their bit positions
constexpr size_t SLOT_MAX = 11 * 192;
constexpr int MAX_WORLD = 64;
constexpr int UID_BYTES = 128;

// every rank's block of an exchange ends (RCCL) / starts (shared memory) with this record, so that a rank whose local stage failed still TAKES
// PART and every rank returns an error - instead of the peers blocking in ncclAllGather / spinning on the shared-memory barrier, or (worse)
// combining the stale sums of an earlier call
struct Hdr {
    uint64_t seq;
    int32_t status, c_bits;
};
static_assert(sizeof(Hdr) == 16, "exchange record");

inline void window_range(int W, int rank, int world, int* lo, int* hi) {
    *lo = (int)((int64_t)W * rank / world);
    *hi = (int)((int64_t)W * (rank + 1) / world);
}
inline int slots_per_rank(int W, int world) { return (W + world - 1) / world; }

// per-rank blocks (rank r's sums at gathered + r * blk, its windows [lo_r, hi_r) in order) -> window order
inline void compact_to_window_order(const uint8_t* gathered, int W, int world, size_t sz, size_t blk, uint8_t* S) {
    for (int r = 0; r < world; ++r) {
        int lo, hi;
        window_range(W, r, world, &lo, &hi);
        memcpy(S + (size_t)lo * sz, gathered + (size_t)r * blk, (size_t)(hi - lo) * sz);
    }
}
// what a gathered record means for the caller: empty = fine
inline std::string judge(const Hdr& h, uint64_t seq, int r, bool stale_wording) {
    if (h.seq != seq) return "bzk_mg: rank " + std::to_string(r) + (stale_wording ? " is at another call of the group (stale slot)" : " is at another call of the group");
    if (h.status != 0) return "bzk_mg: rank " + std::to_string(r) + " failed its local stage (status " + std::to_string(h.status) + ")";
    return std::string();
}

struct ShmHeader {
    std::atomic<uint64_t> arrive[MAX_WORLD];
};

// single node by contract: a segment named after the group id, double-buffered (parity of the call's sequence number), one slot per rank and parity:
// [Hdr | the rank's window sums or terms]
struct ShmExchange {
    ShmHeader* shm = nullptr;
    size_t bytes = 0;
    int world = 0, rank = 0;
    std::string name;
    static constexpr size_t RANK_BYTES = (size_t)MAX_W * SLOT_MAX + sizeof(Hdr);

    static std::string name_of(const uint8_t uid[UID_BYTES]) {
        char nm[64];
        static const char* hx = "0123456789abcdef";
        int o = snprintf(nm, sizeof nm, "/bzk_mg_");
        // the id may be an RCCL unique id whose leading bytes are a magic / address: mix all 128 bytes into the name
        uint64_t h[2] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full};
        for (int i = 0; i < UID_BYTES; ++i) {
            h[i & 1] = (h[i & 1] ^ uid[i]) * 0x100000001B3ull;
            h[(i + 1) & 1] ^= h[i & 1] >> 29;
        }
        for (int k = 0; k < 2; ++k)
            for (int b = 0; b < 16; ++b) nm[o++] = hx[(h[k] >> (4 * b)) & 15];
        nm[o] = 0;
        return nm;
    }
    // maps (creating if need be) the group's segment; err = what failed
    bool open(const uint8_t uid[UID_BYTES], int world_, int rank_, std::string& err) {
        if (world_ < 1 || world_ > MAX_WORLD || rank_ < 0 || rank_ >= world_) { err = "bzk_mg: rank / world out of range"; return false; }
        world = world_;
        rank = rank_;
        name = name_of(uid);
        bytes = sizeof(ShmHeader) + 2 * (size_t)world * RANK_BYTES;
        const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0) { err = "shm_open " + name; return false; }
        if (ftruncate(fd, (off_t)bytes) != 0) { ::close(fd); err = "ftruncate shm"; return false; }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) { err = "mmap shm"; return false; }
        shm = (ShmHeader*)p;
        return true;
    }
    void unlink_name() const { (void)shm_unlink(name.c_str()); }
    void close() {
        if (shm) (void)munmap(shm, bytes);
        shm = nullptr;
    }
    // sequence-numbered arrival: returns once every rank has published `seq` (or later); false + err after the time limit (env BZK_MG_TIMEOUT_S, 120 s)
    bool barrier(uint64_t seq, std::string& err) {
        static const double limit_s = [] { const char* e = getenv("BZK_MG_TIMEOUT_S"); return e ? atof(e) : 120.0; }();
        shm->arrive[rank].store(seq, std::memory_order_release);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < world; ++r) {
            uint32_t spins = 0;
            while (shm->arrive[r].load(std::memory_order_acquire) < seq) {
                if (++spins > 2000) {
                    sched_yield();
                    if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
                        err = "bzk_mg: rank " + std::to_string(r) + " did not reach the exchange (timeout)";
                        return false;
                    }
                }
            }
        }
        return true;
    }
    uint8_t* slot(uint64_t seq, int r) const { return (uint8_t*)(shm + 1) + ((seq & 1) * (size_t)world + (size_t)r) * RANK_BYTES; }
    // One all-gather.  This rank's record + sums go into its slot of the call's parity - a rank that failed locally still arrives, with its status in
    // the record, so every rank of the group returns an error for this call -, then everybody's sums are copied into `gathered` (rank r at r * blk).
    // Returns 0, or -3 (BZK_E_DEVICE) with err set: a peer failed, is at another call, or never arrived.  local_status != 0: returned as is after arriving.
    int32_t all_gather(uint64_t seq, int32_t local_status, int32_t c_bits, const uint8_t* my_sums, int W, size_t sz, size_t blk, uint8_t* gathered,
                       double* wait_ms, std::string& err) {
        int lo, hi;
        window_range(W, rank, world, &lo, &hi);
        uint8_t* mine = slot(seq, rank);
        const Hdr h0{seq, local_status, c_bits};
        memcpy(mine, &h0, sizeof(Hdr));
        if (local_status == 0 && my_sums) memcpy(mine + sizeof(Hdr), my_sums, (size_t)(hi - lo) * sz);
        const auto tw0 = std::chrono::steady_clock::now();
        std::string berr;
        const bool arrived = barrier(seq, berr);
        if (wait_ms) *wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        if (local_status != 0) return local_status;
        if (!arrived) { err = berr; return -3; }
        for (int r = 0; r < world; ++r) {
            if (r == rank) {
                if (my_sums && gathered + (size_t)r * blk != my_sums) memcpy(gathered + (size_t)r * blk, my_sums, (size_t)(hi - lo) * sz);
                continue;
            }
            const uint8_t* theirs = slot(seq, r);
            Hdr h;
            memcpy(&h, theirs, sizeof(Hdr));
            // arrive[r] >= seq let us through; the record says whether what lies in the slot belongs to THIS call and is a result
            const std::string why = judge(h, seq, r, true);
            if (!why.empty()) { err = why; return -3; }
            int rlo, rhi;
            window_range(W, r, world, &rlo, &rhi);
            memcpy(gathered + (size_t)r * blk, theirs + sizeof(Hdr), (size_t)(rhi - rlo) * sz);
        }
        return 0;
    }
};

}  // namespace mgx
}  // namespace bzk
