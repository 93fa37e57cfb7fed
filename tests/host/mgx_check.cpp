// CPU harness of a device group's REAL exchange (bazuka_amd/csrc/mg_exchange.h: the shared-memory all-gather, the (sequence, status) records, the
// compaction into window order) with the device stage stubbed: the window sums a rank would have computed on its GPU are handed in by the test
// (made from the CPU oracle).  One call of mgx_rank_run = the life of one rank of a process-per-GPU group: open the segment named after the group
// id, arrive, `calls` window-sharded MSMs - each: own window range -> exchange -> window order -> the product's own host Horner and packing
// (libbzk's bzk::g1_horner_packed, the very function mg.hip calls) - and leave.  tests/test_mg_exchange_cpu.py runs 2 and 4 such ranks as processes.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../bazuka_amd/csrc/mg_exchange.h"
#include "../../include/bzk.h"

namespace bzk {  // libbzk.so (msm_g1.hip): host code, usable without a device
int32_t g1_horner_packed(const void* S, int count, int c, int w0, uint8_t* out);
int32_t g1_horner_terms_packed(const void* T, int count, int c, int w0, uint8_t* out);
int msm_window_bits(uint64_t n);
int msm_g1_window_terms(uint64_t n);
}  // namespace bzk

extern "C" {

// n_terms: 0 = the ranks exchange one window SUM per window (192 bytes; G2's way and G1's with BZK_MSM_BITSUM=0), k > 0 = the k TERMS of every window's bucket
// set (k x 192 bytes per window: what a rank of a G1 group leaves since round 6) combined by bzk::g1_horner_terms_packed.
// sums: calls x W x (192 max(1, n_terms)) bytes - ALL window sums / terms of each call (a rank publishes only its own range, as mg.hip does).  fault_rank / fault_call (1-based
// call number, 0 = none): that rank's local stage of that call "fails" (status BZK_E_ALLOC travels in its record).  out: calls x 97 packed results;
// status: calls statuses (what bzk_mg_msm_g1 would have returned on this rank); err: the last error text.
int32_t mgx_rank_run(const uint8_t uid[128], int32_t rank, int32_t world, uint64_t n, int32_t calls, const uint8_t* sums, int32_t n_terms, int32_t fault_rank,
                     int32_t fault_call, uint8_t* out, int32_t* status, char* err, int32_t errcap) {
    using namespace bzk::mgx;
    const int W = (int)bzk_msm_window_count(n ? n : 1);
    if (W > MAX_W) return BZK_E_INTERNAL;
    const size_t sz = (SLOT_G2 / 2) * (size_t)(n_terms > 0 ? n_terms : 1);
    if (sz > SLOT_MAX) return BZK_E_ARG;
    const int c_bits = bzk::msm_window_bits(n);
    const size_t blk = (size_t)slots_per_rank(W, world) * sz;
    ShmExchange x;
    std::string e;
    auto fail = [&](int32_t st) {
        if (err && errcap > 0) { strncpy(err, e.c_str(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        return st;
    };
    if (!x.open(uid, world, rank, e)) return fail(BZK_E_DEVICE);
    uint64_t seq = 0;
    if (!x.barrier(++seq, e)) { x.close(); return fail(BZK_E_DEVICE); }  // everybody has mapped the segment ...
    if (rank == 0) x.unlink_name();                                     // ... so its name can go (as bzk_mg_create_rank does)
    std::vector<uint8_t> gathered((size_t)world * blk), S((size_t)W * sz);
    for (int k = 0; k < calls; ++k) {
        ++seq;
        int lo, hi;
        window_range(W, rank, world, &lo, &hi);
        const int32_t lst = (fault_call == k + 1 && fault_rank == rank) ? BZK_E_ALLOC : BZK_OK;
        const uint8_t* mine = sums + ((size_t)k * W + lo) * sz;
        double wait_ms = 0;
        e.clear();
        int32_t st = x.all_gather(seq, lst, c_bits, lst == BZK_OK ? mine : nullptr, W, sz, blk, gathered.data(), &wait_ms, e);
        if (st == BZK_OK) {
            compact_to_window_order(gathered.data(), W, world, sz, blk, S.data());
            st = n_terms > 0 ? bzk::g1_horner_terms_packed(S.data(), W, c_bits, 0, out + (size_t)k * 97) : bzk::g1_horner_packed(S.data(), W, c_bits, 0, out + (size_t)k * 97);
        } else {
            memset(out + (size_t)k * 97, 0xEE, 97);
            fail(st);
        }
        status[k] = st;
    }
    x.close();
    return BZK_OK;
}

int32_t mgx_window_bits(uint64_t n) { return bzk::msm_window_bits(n); }
int32_t mgx_window_terms(uint64_t n) { return bzk::msm_g1_window_terms(n); }  // what a G1 rank of the shipped library would exchange for n points

}  // extern "C"
