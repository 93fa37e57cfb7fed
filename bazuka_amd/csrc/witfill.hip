// Witness value traces on the device: the executor of a DeferProgram (host_r1cs.h) over all transitions of one R1CS instance.
// The ops and what they restate of the reference's gadgets: bzk_witfill.cuh.  One lane per (op, transition); pass 1 level by level
// (a level's ops only read registers of lower levels), pass 2 in one sweep per op kind.  The same ops run on the host in
// witfill_run_host (bzk_r1cs_fill_host: CPU consumers of the complete arrays, and the CPU suite's comparison with the host generator).
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <thread>

#include "bzk_internal.h"
#include "bzk_poseidon_opt.h"
#include "bzk_witfill.cuh"
#include "bzk_poseidon29_coop.cuh"
#include "host_r1cs.h"

namespace bzk {

int32_t poseidon_consts_dev_shared(bzk_ctx* ctx, int t, const void** out, int* rf, int* rp);  // poseidon.hip

namespace {

// dense constants of width t in the 29-bit form: rc[(rf + rp) t] | mds[t t] (the sparse form of the hash kernels has no per-round state)
std::vector<Fr29> dense_consts_host(int t) {
    const PoseidonHostParams P = poseidon_host_params_cached(t);
    std::vector<Fr29> out;
    const size_t n_rc = (size_t)t * (P.rf + P.rp), n_mds = (size_t)t * t;
    out.reserve(n_rc + n_mds);
    for (size_t i = 0; i < n_rc; ++i) out.push_back(fr29::norm(fr29::to29(P.rc[i])));
    for (size_t i = 0; i < n_mds; ++i) out.push_back(fr29::norm(fr29::to29(P.mds[i])));
    return out;
}

struct DevTables {
    const Fr29* dense[9] = {};
    const Fr29* sparse[9] = {};
    int rf[9] = {}, rp[9] = {};
};

template <int T>
__global__ void __launch_bounds__(64) wf_hash_kernel(const wf::Op* __restrict__ ops, uint32_t count, uint32_t n_tx, const Fr* __restrict__ inputs,
                                                     uint32_t n_inputs, Fr* __restrict__ regs, const int32_t* __restrict__ sel, const Fr29* __restrict__ consts,
                                                     int rf, int rp) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count * n_tx) return;
    const uint32_t tx = g % n_tx;
    wf::TxView v{inputs + (size_t)tx * n_inputs, regs + tx, n_tx, 0, 0, sel};
    wf::v_hash<T>(ops[g / n_tx], v, consts, rf, rp);
}
template <int T>
__global__ void __launch_bounds__(64) wf_poseidon_kernel(const wf::Op* __restrict__ ops, uint32_t count, uint32_t n_tx, const Fr* __restrict__ inputs,
                                                         uint32_t n_inputs, Fr* __restrict__ regs, const int32_t* __restrict__ sel, wf::Arrays A, size_t base_aux,
                                                         size_t stride_aux, size_t base_con, size_t stride_con, const Fr29* __restrict__ dense, int rf, int rp) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count * n_tx) return;
    const uint32_t tx = g % n_tx;
    wf::TxView v{inputs + (size_t)tx * n_inputs, regs + tx, n_tx, base_aux + tx * stride_aux, base_con + tx * stride_con, sel};
    wf::f_poseidon<T>(ops[g / n_tx], v, A, dense, rf, rp);
}
__global__ void __launch_bounds__(256) wf_small_kernel(const wf::Op* __restrict__ ops, uint32_t count, uint32_t n_tx, const Fr* __restrict__ inputs,
                                                       uint32_t n_inputs, Fr* __restrict__ regs, const int32_t* __restrict__ sel, wf::Arrays A, size_t base_aux,
                                                       size_t stride_aux, size_t base_con, size_t stride_con, uint32_t* __restrict__ flags) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count * n_tx) return;
    const uint32_t tx = g % n_tx;
    wf::TxView v{inputs + (size_t)tx * n_inputs, regs + tx, n_tx, base_aux + tx * stride_aux, base_con + tx * stride_con, sel};
    const wf::Op op = ops[g / n_tx];
    uint32_t f = 0;
    switch (op.kind) {
        case wf::F_MUX: wf::f_mux(op, v, A); break;
        case wf::F_ASSERT_EQ_IF: f = wf::f_assert_eq_if(op, v, A); break;
        case wf::F_ENFORCE_EQ: f = wf::f_enforce_eq(op, v, A); break;
        case wf::F_CHECK_EQ: f = wf::f_check_eq(op, v); break;
        default: break;
    }
    if (f) atomicOr(flags, f);
}

// ---- the whole program of one transition in ONE launch (round 5, run 8): a workgroup of four waves per transition walks the dependency levels
// with a barrier between them; inside a level (and in the final stage, pass 2) every wave runs its own segments - runs of <= 64 ops of one kind and
// width, so a wave never diverges - assigned by cost on the host (Schedule below).  Why: as ~45 dependent launches the program waited for the device
// behind other prover slots' MSM grids at every step (the process's streams share a handful of hardware queues): with live producers on the deferred
// generator the pipelined rate fell from 62 - 64 to 35 - 50 proofs/s (profiles/r05_run7...).  One launch queues once.
struct Seg {
    uint32_t start;   // first op (index into v_ops / f_ops)
    uint16_t count;   // <= 64
    uint8_t kind, t;
};
constexpr int WF_WAVES = 4;
// the eight heavy bodies as CALLS: inlined into one kernel the compiler allocated registers for all of them at once (512 + 930 spilled, 1.7 KB of scratch per
// lane); as functions each keeps the allocation it has as a kernel of its own and the kernel proper is a dispatcher
// (allocating the kernel for three waves per SIMD - amdgpu_waves_per_eu(3, 3) on the kernel, which the callees inherit: 168 registers, so that a wave fits beside two
// waves of the G1 accumulation - was built and measured in run 10: the width-6 / 8 bodies then live in 9 KB of scratch per lane, the kernel alone takes 19.5 instead of
// 15 ms and the pipelined rate with staged deferred producers FALLS from 51.7 - 52.7 to 45.4 - 47.0 proofs/s, profiles/r05_run10...: not taken)
template <int T>
static __device__ __noinline__ void tx_hash(const wf::Op* op, const wf::TxView* v, const Fr29* consts, int rf, int rp) { wf::v_hash<T>(*op, *v, consts, rf, rp); }
template <int T>
static __device__ __noinline__ void tx_trace(const wf::Op* op, const wf::TxView* v, const wf::Arrays* A, const Fr29* dense, int rf, int rp) {
    wf::f_poseidon<T>(*op, *v, *A, dense, rf, rp);
}
__global__ void __launch_bounds__(64 * WF_WAVES) wf_tx_kernel(const wf::Op* __restrict__ v_ops, const wf::Op* __restrict__ f_ops, const Seg* __restrict__ segs,
                                                              const uint32_t* __restrict__ idx, uint32_t n_stages, uint32_t n_tx, const Fr* __restrict__ inputs,
                                                              uint32_t n_inputs, Fr* __restrict__ regs, const int32_t* __restrict__ sel, wf::Arrays A, size_t base_aux,
                                                              size_t stride_aux, size_t base_con, size_t stride_con, DevTables tab, uint32_t* __restrict__ flags) {
    const uint32_t tx = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    wf::TxView v{inputs + (size_t)tx * n_inputs, regs + tx, n_tx, base_aux + tx * stride_aux, base_con + tx * stride_con, sel};
    uint32_t f = 0;
    for (uint32_t stage = 0; stage < n_stages; ++stage) {
        for (uint32_t s = idx[stage * WF_WAVES + wave]; s < idx[stage * WF_WAVES + wave + 1]; ++s) {
            const Seg sg = segs[s];
            if (lane >= sg.count) continue;
            const int t = sg.t;
            if (sg.kind == wf::V_HASH) {
                const wf::Op op = v_ops[sg.start + lane];
                switch (t) {
                    case 3: tx_hash<3>(&op, &v, tab.sparse[3], tab.rf[3], tab.rp[3]); break;
                    case 5: tx_hash<5>(&op, &v, tab.sparse[5], tab.rf[5], tab.rp[5]); break;
                    case 6: tx_hash<6>(&op, &v, tab.sparse[6], tab.rf[6], tab.rp[6]); break;
                    default: tx_hash<8>(&op, &v, tab.sparse[8], tab.rf[8], tab.rp[8]); break;
                }
            } else {
                const wf::Op op = f_ops[sg.start + lane];
                switch (sg.kind) {
                    case wf::F_POSEIDON:
                        switch (t) {
                            case 3: tx_trace<3>(&op, &v, &A, tab.dense[3], tab.rf[3], tab.rp[3]); break;
                            case 5: tx_trace<5>(&op, &v, &A, tab.dense[5], tab.rf[5], tab.rp[5]); break;
                            case 6: tx_trace<6>(&op, &v, &A, tab.dense[6], tab.rf[6], tab.rp[6]); break;
                            default: tx_trace<8>(&op, &v, &A, tab.dense[8], tab.rf[8], tab.rp[8]); break;
                        }
                        break;
                    case wf::F_MUX: wf::f_mux(op, v, A); break;
                    case wf::F_ASSERT_EQ_IF: f |= wf::f_assert_eq_if(op, v, A); break;
                    case wf::F_ENFORCE_EQ: f |= wf::f_enforce_eq(op, v, A); break;
                    case wf::F_CHECK_EQ: f |= wf::f_check_eq(op, v); break;
                    default: break;
                }
            }
        }
        __threadfence_block();  // a level's registers (global memory, written by one wave) are read by the other waves of this workgroup next
        __syncthreads();
    }
    if (f) atomicOr(flags, f);
}
// stages = the hash levels 1 .. n_levels, then pass 2; per (stage, wave) a run of segments.  Greedy by cost: the most expensive segment first, to the
// least loaded wave (relative costs from the run-7 trace: a sparse hash of width 3 / 5 / 6 / 8 ~ 0.25 / 0.41 / 0.56 / 0.8 ms per lone wave, the dense trace
// ~ 0.5 / 1.14 / 1.54 / 2.07 ms)
struct Schedule {
    std::vector<Seg> segs;
    std::vector<uint32_t> idx;
    uint32_t n_stages = 0;
};
Schedule make_schedule(const DeferProgram& P) {
    auto cost = [](uint8_t kind, uint8_t t) {
        if (kind == wf::V_HASH) return t <= 3 ? 0.25 : t <= 5 ? 0.41 : t <= 6 ? 0.56 : 0.8;
        if (kind == wf::F_POSEIDON) return t <= 3 ? 0.5 : t <= 5 ? 1.14 : t <= 6 ? 1.54 : 2.07;
        return 0.01;
    };
    Schedule S;
    S.n_stages = P.n_levels + 1;
    S.idx.push_back(0);
    auto stage = [&](std::vector<Seg> chunks) {
        std::stable_sort(chunks.begin(), chunks.end(), [&](const Seg& a, const Seg& b) { return cost(a.kind, a.t) > cost(b.kind, b.t); });
        std::vector<Seg> per_wave[WF_WAVES];
        double load[WF_WAVES] = {0, 0, 0, 0};
        for (const Seg& c : chunks) {
            int w = 0;
            for (int i = 1; i < WF_WAVES; ++i)
                if (load[i] < load[w]) w = i;
            per_wave[w].push_back(c);
            load[w] += cost(c.kind, c.t);
        }
        for (int w = 0; w < WF_WAVES; ++w) {
            S.segs.insert(S.segs.end(), per_wave[w].begin(), per_wave[w].end());
            S.idx.push_back((uint32_t)S.segs.size());
        }
    };
    auto chunks_of = [](const DeferGroup& g, std::vector<Seg>& out) {
        for (uint32_t o = 0; o < g.count; o += 64) out.push_back({g.start + o, (uint16_t)std::min<uint32_t>(64, g.count - o), g.kind, g.t});
    };
    for (uint32_t lvl = 1; lvl <= P.n_levels; ++lvl) {
        std::vector<Seg> c;
        for (const DeferGroup& g : P.v_groups)
            if (g.kind == wf::V_HASH && g.level == lvl) chunks_of(g, c);
        stage(c);
    }
    std::vector<Seg> c;
    for (const DeferGroup& g : P.f_groups) chunks_of(g, c);
    stage(c);
    return S;
}

// ---- round 6: the program on COOPERATING lanes.  The one-launch form above spends a lane per hash: a transition's 20 dependency
// levels are 20 one-lane Poseidon hashes back to back (0.25 - 0.8 ms each) and pass 2 is one lane per dense trace (0.5 - 2 ms) - 15 ms per proof, in a
// 512-register body with 6.8 KB of scratch that needs a drained CU.  Here:
//   pass 1  wf_pass1_coop_kernel: a workgroup of two waves per TRANSITION, per level each wave takes segments of <= 8 hashes of one width - EIGHT LANES PER
//           HASH (bzk_poseidon29_coop.cuh: ~2x shorter chain), a workgroup barrier between levels; an Update transition has at most 8 hashes per level,
//           i.e. one wave-step per level
//   pass 2  wf_trace_all_kernel: eight lanes per dense trace (lane j holds state element j and writes its own S-box / idle-lane slots), all traces of all
//           transitions and widths in ONE launch; the small ops stay on wf_small_kernel (a lane per op and transition)
// No body above 256 registers, no scratch: the waves fit beside the accumulation's.  Same values: the slots are written by the same formulas as
// wf::f_poseidon (bzk_witfill.cuh), which the CPU suite pins on the independent restatement's fixtures.
constexpr int WFC_WAVES = 2;
struct CSeg {
    uint32_t start;  // first op (index into v_ops)
    uint8_t count;   // <= 8
    uint8_t t;
    uint16_t pad;
};
__global__ void __launch_bounds__(64 * WFC_WAVES) wf_pass1_coop_kernel(const wf::Op* __restrict__ v_ops, const CSeg* __restrict__ segs, const uint32_t* __restrict__ idx,
                                                                      uint32_t n_levels, uint32_t n_tx, const Fr* __restrict__ inputs, uint32_t n_inputs,
                                                                      Fr* __restrict__ regs, const int32_t* __restrict__ sel, DevTables tab) {
    const uint32_t tx = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, grp = lane >> 3, j = lane & 7u;
    const wf::TxView v{inputs + (size_t)tx * n_inputs, regs + tx, n_tx, 0, 0, sel};
    for (uint32_t level = 0; level < n_levels; ++level) {
        for (uint32_t s = idx[level * WFC_WAVES + wave]; s < idx[level * WFC_WAVES + wave + 1]; ++s) {
            const CSeg sg = segs[s];
            const bool live = grp < sg.count;
            const wf::Op& op = v_ops[sg.start + (live ? grp : sg.count - 1u)];  // idle groups mirror the last hash: every group stays convergent for the shuffles
            const uint32_t t = sg.t, jj = j < t ? j : t - 1u;
            const Fr mine = jj >= 1u ? wf::operand(v, op.in[jj - 1u]) : Fr::zero();
            Fr h;
            switch (t) {  // wave-uniform
                case 3: h = poseidon29_coop_val<3>(mine, tab.sparse[3], tab.rf[3], tab.rp[3]); break;
                case 5: h = poseidon29_coop_val<5>(mine, tab.sparse[5], tab.rf[5], tab.rp[5]); break;
                case 6: h = poseidon29_coop_val<6>(mine, tab.sparse[6], tab.rf[6], tab.rp[6]); break;
                default: h = poseidon29_coop_val<8>(mine, tab.sparse[8], tab.rf[8], tab.rp[8]); break;
            }
            if (live && j == 1u) v.regs[(size_t)op.out * v.reg_stride] = h;
        }
        __threadfence_block();  // a level's registers (global memory) are read by the other wave of this workgroup next
        __syncthreads();
    }
}
// the dense trace of one Poseidon gadget instance on eight lanes: wf::f_poseidon's slots, lane j writing what belongs to state element j
struct TraceArgs {  // everything but the per-width part
    uint32_t n_tx, n_inputs;
    const Fr* inputs;
    Fr* regs;
    const int32_t* sel;
    wf::Arrays A;
    size_t base_aux, stride_aux, base_con, stride_con;
};
template <int T>
static __device__ __forceinline__ void trace_coop(const wf::Op* __restrict__ ops, uint32_t count, uint32_t block, const TraceArgs& a, const Fr29* __restrict__ dense, int rf, int rp) {
    const uint32_t n_tx = a.n_tx, n_inputs = a.n_inputs;
    const Fr* const inputs = a.inputs;
    Fr* const regs = a.regs;
    const int32_t* const sel = a.sel;
    const wf::Arrays A = a.A;
    const size_t base_aux = a.base_aux, stride_aux = a.stride_aux, base_con = a.base_con, stride_con = a.stride_con;
    const uint32_t lane = threadIdx.x & 63u, j = lane & 7u, g0 = lane & ~7u;
    const uint64_t total = (uint64_t)count * n_tx;
    uint64_t g = (uint64_t)block * 32 + (threadIdx.x >> 3);
    const bool live = g < total;
    if (!live) g = total - 1;  // whole groups stay convergent for the shuffles
    const uint32_t tx = (uint32_t)(g % n_tx);
    const wf::TxView v{inputs + (size_t)tx * n_inputs, regs + tx, n_tx, base_aux + tx * stride_aux, base_con + tx * stride_con, sel};
    const wf::Op& op = ops[g / n_tx];
    const uint32_t jj = j < (uint32_t)T ? j : (uint32_t)T - 1u;
    Fr29 e = jj >= 1u ? fr29::to29(wf::operand(v, op.in[jj - 1u])) : fr29::zero();
    if (j == 0) e = fr29::zero();
    const Fr one = wf::fr_one_mont();
    const Fr29* mds = dense + (size_t)(rf + rp) * T;
    const bool mine = live && j < (uint32_t)T;
    size_t off = 0;  // slot offset of the round inside the gadget's window (the same for variables and constraints)
#pragma unroll 1
    for (int rnd = 0; rnd < rf + rp; ++rnd) {
        e = fr29::norm(fr29::add(e, dense[(size_t)rnd * T + jj]));  // k <= 3
        const bool full = rnd < rf / 2 || rnd >= rf / 2 + rp;
        // every lane runs the S-box (one instruction stream); in a partial round only lane 0's is the gadget's
        const Fr29 x2 = fr29::sqr(e), x4 = fr29::sqr(x2), x5 = fr29::mul(e, x4);
        const bool sbox_lane = full ? mine : (live && j == 0);
        if (sbox_lane) {
            const size_t a = v.aux_base + op.aux_off + off + (full ? 3u * j : 0u), c = v.con_base + op.con_off + off + (full ? 3u * j : 0u);
            const Fr X = fr29::from29(e), X2 = fr29::from29(x2), X4 = fr29::from29(x4), X5 = fr29::from29(x5);
            A.z_aux[a] = X2; A.z_aux[a + 1] = X4; A.z_aux[a + 2] = X5;
            A.az[c] = X;  A.bz[c] = X;  A.cz[c] = X2;
            A.az[c + 1] = X2; A.bz[c + 1] = X2; A.cz[c + 1] = X4;
            A.az[c + 2] = X;  A.bz[c + 2] = X4; A.cz[c + 2] = X5;
        } else if (!full && mine) {  // Number::compress of an idle lane: v * 1 = v
            const size_t a = v.aux_base + op.aux_off + off + 3u + (j - 1u), c = v.con_base + op.con_off + off + 3u + (j - 1u);
            const Fr V = fr29::from29(e);
            A.z_aux[a] = V;
            A.az[c] = V; A.bz[c] = one; A.cz[c] = V;
        }
        if (full || j == 0) e = x5;
        off += full ? 3u * T : 3u + (T - 1u);
        Fr29 st[T];
#pragma unroll
        for (int k = 0; k < T; ++k) st[k] = shfl29(e, (int)g0 + k);
        e = p29::row_dot<T>(mds + (size_t)jj * T, st);
    }
}
// all dense traces of a program in ONE launch: the blocks of the four widths side by side (a block is of one width: no divergence), so that a proof waits
// for the longest trace once instead of for four launches one behind the other (run 5 of round 6: 0.58 - 0.72 ms each)
struct TraceWidths {
    const wf::Op* ops[4];   // widths 3, 5, 6, 8
    uint32_t count[4];
    uint32_t first_block[5];
};
__global__ void __launch_bounds__(256) wf_trace_all_kernel(TraceWidths W, TraceArgs a, DevTables tab) {
    const uint32_t b = blockIdx.x;
    if (b < W.first_block[1]) trace_coop<3>(W.ops[0], W.count[0], b - W.first_block[0], a, tab.dense[3], tab.rf[3], tab.rp[3]);
    else if (b < W.first_block[2]) trace_coop<5>(W.ops[1], W.count[1], b - W.first_block[1], a, tab.dense[5], tab.rf[5], tab.rp[5]);
    else if (b < W.first_block[3]) trace_coop<6>(W.ops[2], W.count[2], b - W.first_block[2], a, tab.dense[6], tab.rf[6], tab.rp[6]);
    else trace_coop<8>(W.ops[3], W.count[3], b - W.first_block[3], a, tab.dense[8], tab.rf[8], tab.rp[8]);
}

// pass-1 schedule of the cooperative form: per level and wave a run of segments (<= 8 hashes of one width = one wave-step), most expensive first
struct CoopSchedule {
    std::vector<CSeg> segs;
    std::vector<uint32_t> idx;  // n_levels * WFC_WAVES + 1
};
CoopSchedule make_coop_schedule(const DeferProgram& P) {
    auto cost = [](uint8_t t) { return t <= 3 ? 0.25 : t <= 5 ? 0.41 : t <= 6 ? 0.56 : 0.8; };
    CoopSchedule S;
    S.idx.push_back(0);
    for (uint32_t lvl = 1; lvl <= P.n_levels; ++lvl) {
        std::vector<CSeg> c;
        for (const DeferGroup& g : P.v_groups)
            if (g.kind == wf::V_HASH && g.level == lvl)
                for (uint32_t o = 0; o < g.count; o += 8) c.push_back({g.start + o, (uint8_t)std::min<uint32_t>(8, g.count - o), g.t, 0});
        std::stable_sort(c.begin(), c.end(), [&](const CSeg& a, const CSeg& b) { return cost(a.t) > cost(b.t); });
        std::vector<CSeg> per_wave[WFC_WAVES];
        double load[WFC_WAVES] = {};
        for (const CSeg& x : c) {
            int w = 0;
            for (int i = 1; i < WFC_WAVES; ++i)
                if (load[i] < load[w]) w = i;
            per_wave[w].push_back(x);
            load[w] += cost(x.t);
        }
        for (int w = 0; w < WFC_WAVES; ++w) {
            S.segs.insert(S.segs.end(), per_wave[w].begin(), per_wave[w].end());
            S.idx.push_back((uint32_t)S.segs.size());
        }
    }
    return S;
}

// per (context, program): the ops in device memory; per context: the constant tables and the grow-only scratch (registers + inputs)
struct CtxState {
    struct DevProg {
        wf::Op *v = nullptr, *f = nullptr;  // v_ops, f_ops
        int32_t* sel = nullptr;             // per register: the V_SEL that defines it (bzk_witfill.cuh TxView::sel)
        Seg* segs = nullptr;                // the one-launch schedule (make_schedule)
        uint32_t* idx = nullptr;
        uint32_t n_stages = 0;
        CSeg* csegs = nullptr;              // pass 1 of the cooperative form (make_coop_schedule)
        uint32_t* cidx = nullptr;
    };
    std::map<const DeferProgram*, DevProg> progs;
    DevTables tab;
    void* dense_dev[9] = {};
    void* scratch = nullptr;   // [flags word, padded to 256 B][input records][registers]
    size_t scratch_bytes = 0;
    uint32_t* flags_host = nullptr;  // pinned: where the flags word of the last run lands (stream order)
    // the program is ~150 small DEPENDENT launches (a Merkle path is a chain): beside the big grids of other prover slots each of them would wait
    // for a free slot on the device.  They run on a highest-priority side stream, forked from / joined to the context's stream by events
    // (env BZK_WF_PRIO=0: on the context's own stream, A/B)
    hipStream_t prio = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    bool prio_tried = false;
};

int32_t tables(bzk_ctx* ctx, CtxState& S, int t) {
    if (S.tab.dense[t]) return BZK_OK;
    const void* sp;
    int rf, rp;
    BZK_TRY(poseidon_consts_dev_shared(ctx, t, &sp, &rf, &rp));
    const std::vector<Fr29> d = dense_consts_host(t);
    void* dev = nullptr;
    BZK_HIP(ctx, hipMalloc(&dev, d.size() * sizeof(Fr29)));
    BZK_HIP(ctx, hipMemcpyAsync(dev, d.data(), d.size() * sizeof(Fr29), hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S.dense_dev[t] = dev;
    S.tab.dense[t] = (const Fr29*)dev;
    S.tab.sparse[t] = (const Fr29*)sp;
    S.tab.rf[t] = rf;
    S.tab.rp[t] = rp;
    return BZK_OK;
}

}  // namespace

void witfill_quiesce(bzk_ctx* ctx) {  // a failed prove call: nothing of the program may still be reading the instance's input records
    CtxState* S = (CtxState*)ctx->wf_state;
    if (S && S->prio) (void)hipStreamSynchronize(S->prio);
}
uint32_t witfill_flags(bzk_ctx* ctx) {  // after the stream of the last witfill_run_dev has been synchronised
    CtxState* S = (CtxState*)ctx->wf_state;
    return S && S->flags_host ? *S->flags_host : 0u;
}

int32_t witfill_run_dev(bzk_ctx* ctx, const DeferData& dd, const wf::Arrays& A, uint32_t* flags_dst) {
    const DeferProgram& P = *dd.prog;
    if (!dd.n_tx || P.ops.empty()) return BZK_OK;
    if (P.max_sel_chain > (uint32_t)wf::MAX_SEL_CHAIN) {  // operand() would stop in the middle of the chain and read a register nobody wrote
        ctx->last_error = "witfill: a chain of " + std::to_string(P.max_sel_chain) + " selections exceeds the device executor's limit";
        return BZK_E_INTERNAL;
    }
    // (a context is driven by one host thread at a time - the library's rule for every entry point - so its state needs no lock)
    if (!ctx->wf_state) ctx->wf_state = new CtxState();
    CtxState* S = (CtxState*)ctx->wf_state;
    if (!S->flags_host) BZK_HIP(ctx, hipHostMalloc((void**)&S->flags_host, 64));
    // flags_dst: a pinned word of the caller's (staged instances: several runs may be in flight on this stream); default: the context's own
    uint32_t* const flags_out = flags_dst ? flags_dst : S->flags_host;
    *flags_out = 0;
    for (const DeferGroup& g : P.v_groups)
        if (g.kind == wf::V_HASH) BZK_TRY(tables(ctx, *S, g.t));
    for (const DeferGroup& g : P.f_groups)
        if (g.kind == wf::F_POSEIDON) BZK_TRY(tables(ctx, *S, g.t));
    auto it = S->progs.find(&P);
    if (it == S->progs.end()) {
        CtxState::DevProg dp;
        std::vector<int32_t> sel((size_t)std::max<uint32_t>(1, P.n_regs) * 4, -1);
        for (const wf::Op& o : P.v_ops)
            if (o.kind == wf::V_SEL) {
                int32_t* s = &sel[(size_t)o.out * 4];
                s[0] = o.in[0]; s[1] = o.in[1]; s[2] = o.in[2]; s[3] = o.t;
            }
        const Schedule sch = make_schedule(P);
        const CoopSchedule csch = make_coop_schedule(P);
        dp.n_stages = sch.n_stages;
        auto upload = [&]() -> int32_t {
            BZK_HIP(ctx, hipMalloc((void**)&dp.v, std::max<size_t>(1, P.v_ops.size()) * sizeof(wf::Op)));
            BZK_HIP(ctx, hipMalloc((void**)&dp.f, std::max<size_t>(1, P.f_ops.size()) * sizeof(wf::Op)));
            BZK_HIP(ctx, hipMalloc((void**)&dp.sel, sel.size() * sizeof(int32_t)));
            BZK_HIP(ctx, hipMalloc((void**)&dp.segs, std::max<size_t>(1, sch.segs.size()) * sizeof(Seg)));
            BZK_HIP(ctx, hipMalloc((void**)&dp.idx, sch.idx.size() * sizeof(uint32_t)));
            BZK_HIP(ctx, hipMemcpyAsync(dp.v, P.v_ops.data(), P.v_ops.size() * sizeof(wf::Op), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipMemcpyAsync(dp.f, P.f_ops.data(), P.f_ops.size() * sizeof(wf::Op), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipMemcpyAsync(dp.sel, sel.data(), sel.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipMemcpyAsync(dp.segs, sch.segs.data(), sch.segs.size() * sizeof(Seg), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipMemcpyAsync(dp.idx, sch.idx.data(), sch.idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipMalloc((void**)&dp.csegs, std::max<size_t>(1, csch.segs.size()) * sizeof(CSeg)));
            BZK_HIP(ctx, hipMalloc((void**)&dp.cidx, csch.idx.size() * sizeof(uint32_t)));
            BZK_HIP(ctx, hipMemcpyAsync(dp.csegs, csch.segs.data(), csch.segs.size() * sizeof(CSeg), hipMemcpyHostToDevice, ctx->stream));
            BZK_HIP(ctx, hipMemcpyAsync(dp.cidx, csch.idx.data(), csch.idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            return BZK_OK;
        };
        const int32_t up = upload();
        // `sel`, `sch` and `csch` are locals: no copy out of them may be in flight when this frame goes, whether the upload succeeded or not
        const hipError_t se = hipStreamSynchronize(ctx->stream);
        if (up != BZK_OK || se != hipSuccess) {
            (void)hipGetLastError();
            for (void* q : {(void*)dp.v, (void*)dp.f, (void*)dp.sel, (void*)dp.segs, (void*)dp.idx, (void*)dp.csegs, (void*)dp.cidx})
                if (q) (void)hipFree(q);
            if (up == BZK_OK) ctx->last_error = std::string("witfill: program upload: ") + hipGetErrorString(se);
            return up != BZK_OK ? up : BZK_E_DEVICE;
        }
        it = S->progs.emplace(&P, dp).first;
    }
    const size_t n_tx = dd.n_tx;
    const size_t in_bytes = ws_pad(n_tx * (size_t)P.n_inputs * 32), reg_bytes = ws_pad(n_tx * (size_t)P.n_regs * 32);
    if (S->scratch_bytes < 256 + in_bytes + reg_bytes) {
        // the previous buffer may still be read by launches of an earlier call on this stream
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (S->prio) BZK_HIP(ctx, hipStreamSynchronize(S->prio));
        if (S->scratch) (void)hipFree(S->scratch);
        S->scratch = nullptr;
        S->scratch_bytes = 0;
        BZK_HIP(ctx, hipMalloc(&S->scratch, 256 + in_bytes + reg_bytes));
        S->scratch_bytes = 256 + in_bytes + reg_bytes;
    }
    uint32_t* flags_dev = (uint32_t*)S->scratch;
    Fr* d_in = (Fr*)((char*)S->scratch + 256);
    Fr* d_regs = (Fr*)((char*)S->scratch + 256 + in_bytes);
    BZK_HIP(ctx, hipMemsetAsync(flags_dev, 0, 4, ctx->stream));
    BZK_HIP(ctx, hipMemcpyAsync(d_in, dd.inputs.data(), n_tx * (size_t)P.n_inputs * 32, hipMemcpyHostToDevice, ctx->stream));
    // BZK_WF_MODE=levels: one launch per (level, width) group instead of the one-launch form (A/B); BZK_WF_PRIO=1: on a highest-priority side stream
    // (measured worse under load with the launch-per-level form: 35 vs 50 proofs/s at 8 slots, profiles/r05_run7...; default off)
    // BZK_WF_MODE: coop (default since round 6: eight lanes per hash, pass 1 as one launch with a workgroup per transition, pass 2 one launch per kind and width) |
    // one (round 5: the whole program of a transition in one launch, a lane per hash) | levels (a launch per (level, width) group)
    static const int wf_mode = [] { const char* e = getenv("BZK_WF_MODE"); return !e || strcmp(e, "coop") == 0 ? 2 : strcmp(e, "levels") == 0 ? 0 : 1; }();
    const bool one_launch = wf_mode == 1;
    static const bool want_prio = [] { const char* e = getenv("BZK_WF_PRIO"); return e && atoi(e) != 0; }();
    if (want_prio && !S->prio_tried) {
        S->prio_tried = true;
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&S->prio, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&S->ev_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&S->ev_out, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (S->prio) (void)hipStreamDestroy(S->prio);
            S->prio = nullptr;  // no side stream: everything on the context's stream
        }
    }
    struct StreamSwap {  // BZK_LAUNCH works on ctx->stream: it points at the side stream while the program is enqueued
        bzk_ctx* c;
        hipStream_t saved;
        StreamSwap(bzk_ctx* c_, hipStream_t s) : c(c_), saved(c_->stream) { if (s) c->stream = s; }
        ~StreamSwap() { c->stream = saved; }
    };
    const hipStream_t home = ctx->stream;
    if (S->prio) {
        BZK_HIP(ctx, hipEventRecord(S->ev_in, home));
        BZK_HIP(ctx, hipStreamWaitEvent(S->prio, S->ev_in, 0));
    }
    StreamSwap swap(ctx, S->prio);
    const uint32_t ntx = (uint32_t)n_tx;
    const int32_t* dsel = it->second.sel;
    if (one_launch) {
        for (int t = 0; t < 9; ++t)
            if (t != 3 && t != 5 && t != 6 && t != 8 && S->tab.dense[t]) { ctx->last_error = "witfill: no device form for Poseidon width " + std::to_string(t); return BZK_E_INTERNAL; }
        BZK_LAUNCH(ctx, "wf_tx", wf_tx_kernel, dim3(ntx), dim3(64 * WF_WAVES), 0, (const wf::Op*)it->second.v, (const wf::Op*)it->second.f, (const Seg*)it->second.segs,
                   (const uint32_t*)it->second.idx, it->second.n_stages, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, A, dd.base_aux, dd.stride_aux, dd.base_con,
                   dd.stride_con, S->tab, flags_dev);
        BZK_HIP(ctx, hipMemcpyAsync(flags_out, flags_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
        if (S->prio) {
            BZK_HIP(ctx, hipEventRecord(S->ev_out, S->prio));
            BZK_HIP(ctx, hipStreamWaitEvent(home, S->ev_out, 0));
        }
        return BZK_OK;
    }
    auto blocks = [&](uint32_t count, uint32_t bs) { return dim3((unsigned)(((uint64_t)count * ntx + bs - 1) / bs)); };
    if (wf_mode == 2) {
        for (int t = 0; t < 9; ++t)
            if (t != 3 && t != 5 && t != 6 && t != 8 && S->tab.dense[t]) { ctx->last_error = "witfill: no device form for Poseidon width " + std::to_string(t); return BZK_E_INTERNAL; }
        if (P.n_levels)
            BZK_LAUNCH(ctx, "wf_pass1", wf_pass1_coop_kernel, dim3(ntx), dim3(64 * WFC_WAVES), 0, (const wf::Op*)it->second.v, (const CSeg*)it->second.csegs,
                       (const uint32_t*)it->second.cidx, P.n_levels, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, S->tab);
        TraceWidths W{};
        const int width_of[4] = {3, 5, 6, 8};
        for (const DeferGroup& g : P.f_groups) {
            const wf::Op* o = it->second.f + g.start;
            if (g.kind != wf::F_POSEIDON) {
                BZK_LAUNCH(ctx, "wf_small", wf_small_kernel, blocks(g.count, 256), dim3(256), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, A, dd.base_aux,
                           dd.stride_aux, dd.base_con, dd.stride_con, flags_dev);
                continue;
            }
            for (int k = 0; k < 4; ++k)
                if (width_of[k] == g.t) {
                    if (W.count[k]) { ctx->last_error = "witfill: two trace groups of one width"; return BZK_E_INTERNAL; }  // finalize() sorts the F ops by (kind, width)
                    W.ops[k] = o;
                    W.count[k] = g.count;
                }
        }
        uint32_t nblk = 0;
        for (int k = 0; k < 4; ++k) {
            W.first_block[k] = nblk;
            nblk += (uint32_t)(((uint64_t)W.count[k] * ntx + 31) / 32);
        }
        W.first_block[4] = nblk;
        if (nblk) {
            const TraceArgs ta{ntx, P.n_inputs, (const Fr*)d_in, d_regs, dsel, A, dd.base_aux, dd.stride_aux, dd.base_con, dd.stride_con};
            BZK_LAUNCH(ctx, "wf_trace", wf_trace_all_kernel, dim3(nblk), dim3(256), 0, W, ta, S->tab);
        }
        BZK_HIP(ctx, hipMemcpyAsync(flags_out, flags_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
        if (S->prio) {
            BZK_HIP(ctx, hipEventRecord(S->ev_out, S->prio));
            BZK_HIP(ctx, hipStreamWaitEvent(home, S->ev_out, 0));
        }
        return BZK_OK;
    }
    // pass 1
    for (const DeferGroup& g : P.v_groups) {
        const wf::Op* o = it->second.v + g.start;
        if (g.kind == wf::V_SEL) continue;  // resolved where they are read
        const Fr29* c = S->tab.sparse[g.t];
        const int rf = S->tab.rf[g.t], rp = S->tab.rp[g.t];
        switch (g.t) {
            case 3: BZK_LAUNCH(ctx, "wf_hash", wf_hash_kernel<3>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, c, rf, rp); break;
            case 5: BZK_LAUNCH(ctx, "wf_hash", wf_hash_kernel<5>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, c, rf, rp); break;
            case 6: BZK_LAUNCH(ctx, "wf_hash", wf_hash_kernel<6>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, c, rf, rp); break;
            case 8: BZK_LAUNCH(ctx, "wf_hash", wf_hash_kernel<8>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, c, rf, rp); break;
            default: ctx->last_error = "witfill: no device form for Poseidon width " + std::to_string(g.t); return BZK_E_INTERNAL;
        }
    }
    // pass 2
    for (const DeferGroup& g : P.f_groups) {
        const wf::Op* o = it->second.f + g.start;
        if (g.kind != wf::F_POSEIDON) {
            BZK_LAUNCH(ctx, "wf_small", wf_small_kernel, blocks(g.count, 256), dim3(256), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, A, dd.base_aux,
                       dd.stride_aux, dd.base_con, dd.stride_con, flags_dev);
            continue;
        }
        const Fr29* c = S->tab.dense[g.t];
        const int rf = S->tab.rf[g.t], rp = S->tab.rp[g.t];
        switch (g.t) {
            case 3: BZK_LAUNCH(ctx, "wf_poseidon", wf_poseidon_kernel<3>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, A, dd.base_aux, dd.stride_aux, dd.base_con, dd.stride_con, c, rf, rp); break;
            case 5: BZK_LAUNCH(ctx, "wf_poseidon", wf_poseidon_kernel<5>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, A, dd.base_aux, dd.stride_aux, dd.base_con, dd.stride_con, c, rf, rp); break;
            case 6: BZK_LAUNCH(ctx, "wf_poseidon", wf_poseidon_kernel<6>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, A, dd.base_aux, dd.stride_aux, dd.base_con, dd.stride_con, c, rf, rp); break;
            case 8: BZK_LAUNCH(ctx, "wf_poseidon", wf_poseidon_kernel<8>, blocks(g.count, 64), dim3(64), 0, o, g.count, ntx, (const Fr*)d_in, P.n_inputs, d_regs, dsel, A, dd.base_aux, dd.stride_aux, dd.base_con, dd.stride_con, c, rf, rp); break;
            default: ctx->last_error = "witfill: no device form for Poseidon width " + std::to_string(g.t); return BZK_E_INTERNAL;
        }
    }
    BZK_HIP(ctx, hipMemcpyAsync(flags_out, flags_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (S->prio) {  // join: everything the caller enqueues on its stream from here on is behind the program and its flags word
        BZK_HIP(ctx, hipEventRecord(S->ev_out, S->prio));
        BZK_HIP(ctx, hipStreamWaitEvent(home, S->ev_out, 0));
    }
    return BZK_OK;
}

// What the one-launch kernel will be handed for `P` (host-side check, no device needed): info = {stages, segments, V_HASH ops covered, F ops covered,
// largest segment, violations}.  A violation: an op covered twice or not at all, a segment of more than 64 ops or of mixed kind / width, a hash segment in a
// stage other than its level's, an F segment before the last stage.
void witfill_schedule_info(const DeferProgram& P, uint64_t info[6]) {
    const Schedule S = make_schedule(P);
    std::vector<uint8_t> seen_v(P.v_ops.size(), 0), seen_f(P.f_ops.size(), 0);
    uint64_t bad = 0, largest = 0, nv = 0, nf = 0;
    for (uint32_t st = 0; st < S.n_stages; ++st)
        for (int w = 0; w < WF_WAVES; ++w)
            for (uint32_t k = S.idx[st * WF_WAVES + w]; k < S.idx[st * WF_WAVES + w + 1]; ++k) {
                const Seg& g = S.segs[k];
                largest = std::max<uint64_t>(largest, g.count);
                if (g.count == 0 || g.count > 64) ++bad;
                const bool is_v = g.kind == wf::V_HASH;
                const std::vector<wf::Op>& ops = is_v ? P.v_ops : P.f_ops;
                std::vector<uint8_t>& seen = is_v ? seen_v : seen_f;
                if ((size_t)g.start + g.count > ops.size()) { ++bad; continue; }
                for (uint32_t i = 0; i < g.count; ++i) {
                    const wf::Op& o = ops[g.start + i];
                    if (o.kind != g.kind || (o.kind == wf::V_HASH || o.kind == wf::F_POSEIDON ? o.t != g.t : false)) ++bad;
                    if (is_v && o.level != st + 1) ++bad;
                    if (!is_v && st + 1 != S.n_stages) ++bad;
                    if (seen[g.start + i]++) ++bad;
                    (is_v ? nv : nf) += 1;
                }
            }
    for (size_t i = 0; i < P.v_ops.size(); ++i)
        if (P.v_ops[i].kind == wf::V_HASH && !seen_v[i]) ++bad;  // (selections are resolved where they are read: no segment of their own)
    for (size_t i = 0; i < P.f_ops.size(); ++i)
        if (!seen_f[i]) ++bad;
    if (P.max_sel_chain > (uint32_t)wf::MAX_SEL_CHAIN) ++bad;  // deeper than the device resolves where an operand is read
    if (env_on("BZK_WF_DUMP")) {  // the shape of the program on stderr: ops per (kind, width, level) - what a device schedule has to work with
        for (const DeferGroup& g : P.v_groups) fprintf(stderr, "[bzk] wf program: V kind %u width %u level %u: %u ops\n", g.kind, g.t, g.level, g.count);
        for (const DeferGroup& g : P.f_groups) fprintf(stderr, "[bzk] wf program: F kind %u width %u: %u ops\n", g.kind, g.t, g.count);
        fprintf(stderr, "[bzk] wf program: %u registers, %u inputs, %u levels, longest selection chain %u\n", P.n_regs, P.n_inputs, P.n_levels, P.max_sel_chain);
    }
    info[0] = S.n_stages; info[1] = S.segs.size(); info[2] = nv; info[3] = nf; info[4] = largest; info[5] = bad;
}

void witfill_free(bzk_ctx* ctx) {  // bzk_ctx_destroy
    CtxState* S = (CtxState*)ctx->wf_state;
    if (!S) return;
    for (auto& kv : S->progs) {
        (void)hipFree(kv.second.v);
        (void)hipFree(kv.second.f);
        (void)hipFree(kv.second.sel);
        (void)hipFree(kv.second.segs);
        (void)hipFree(kv.second.idx);
        (void)hipFree(kv.second.csegs);
        (void)hipFree(kv.second.cidx);
    }
    for (void* p : S->dense_dev)
        if (p) (void)hipFree(p);
    if (S->scratch) (void)hipFree(S->scratch);
    if (S->flags_host) (void)hipHostFree(S->flags_host);
    if (S->prio) {
        (void)hipStreamSynchronize(S->prio);
        (void)hipStreamDestroy(S->prio);
    }
    if (S->ev_in) (void)hipEventDestroy(S->ev_in);
    if (S->ev_out) (void)hipEventDestroy(S->ev_out);
    delete S;
    ctx->wf_state = nullptr;
}

// the same program on the host, one transition after the other (a few host threads): CPU consumers and the CPU suite
uint32_t witfill_run_host(const DeferData& dd, const wf::Arrays& A) {
    const DeferProgram& P = *dd.prog;
    if (!dd.n_tx || P.ops.empty()) return 0;
    struct Tab {
        std::vector<Fr29> dense, sparse;
        int rf = 0, rp = 0;
    };
    static std::mutex mu;
    static Tab tabs[9];
    auto tab = [&](int t) -> const Tab& {
        std::lock_guard<std::mutex> lk(mu);
        Tab& T = tabs[t];
        if (T.dense.empty()) {
            const PoseidonHostParams Pp = poseidon_host_params_cached(t);
            T.rf = Pp.rf;
            T.rp = Pp.rp;
            T.dense = dense_consts_host(t);
            std::vector<Fr> flat;
            const std::vector<Fr> rc(Pp.rc, Pp.rc + (size_t)t * (Pp.rf + Pp.rp)), mds(Pp.mds, Pp.mds + (size_t)t * t);
            if (!poseidon_optimize(t, Pp.rf, Pp.rp, rc, mds, flat)) throw std::logic_error("witfill: no sparse form");
            T.sparse.resize(flat.size());
            for (size_t i = 0; i < flat.size(); ++i) T.sparse[i] = fr29::norm(fr29::to29(flat[i]));
        }
        return T;
    };
    for (const DeferGroup& g : P.v_groups)
        if (g.kind == wf::V_HASH) (void)tab(g.t);
    for (const DeferGroup& g : P.f_groups)
        if (g.kind == wf::F_POSEIDON) (void)tab(g.t);
    std::atomic<size_t> next(0);
    std::atomic<uint32_t> flags(0);
    auto worker = [&] {
        std::vector<Fr> regs(P.n_regs);
        for (;;) {
            const size_t tx = next.fetch_add(1);
            if (tx >= dd.n_tx) break;
            wf::TxView v{dd.inputs.data() + tx * P.n_inputs, regs.data(), 1, dd.base_aux + tx * dd.stride_aux, dd.base_con + tx * dd.stride_con, nullptr};
            uint32_t f = 0;
            for (const wf::Op& op : P.v_ops) {
                if (op.kind == wf::V_SEL) { wf::v_sel(op, v); continue; }
                const Tab& T = tabs[op.t];
                switch (op.t) {
                    case 3: wf::v_hash<3>(op, v, T.sparse.data(), T.rf, T.rp); break;
                    case 5: wf::v_hash<5>(op, v, T.sparse.data(), T.rf, T.rp); break;
                    case 6: wf::v_hash<6>(op, v, T.sparse.data(), T.rf, T.rp); break;
                    case 8: wf::v_hash<8>(op, v, T.sparse.data(), T.rf, T.rp); break;
                    default: throw std::logic_error("witfill: width");
                }
            }
            for (const wf::Op& op : P.f_ops) {
                switch (op.kind) {
                    case wf::F_MUX: wf::f_mux(op, v, A); break;
                    case wf::F_ASSERT_EQ_IF: f |= wf::f_assert_eq_if(op, v, A); break;
                    case wf::F_ENFORCE_EQ: f |= wf::f_enforce_eq(op, v, A); break;
                    case wf::F_CHECK_EQ: f |= wf::f_check_eq(op, v); break;
                    case wf::F_POSEIDON: {
                        const Tab& T = tabs[op.t];
                        switch (op.t) {
                            case 3: wf::f_poseidon<3>(op, v, A, T.dense.data(), T.rf, T.rp); break;
                            case 5: wf::f_poseidon<5>(op, v, A, T.dense.data(), T.rf, T.rp); break;
                            case 6: wf::f_poseidon<6>(op, v, A, T.dense.data(), T.rf, T.rp); break;
                            case 8: wf::f_poseidon<8>(op, v, A, T.dense.data(), T.rf, T.rp); break;
                            default: throw std::logic_error("witfill: width");
                        }
                        break;
                    }
                    default: break;
                }
            }
            if (f) flags.fetch_or(f);
        }
    };
    const size_t nt = std::min<size_t>(dd.n_tx, (size_t)std::min(8, host_default_threads()));
    std::vector<std::thread> th;
    for (size_t i = 1; i < nt; ++i) th.emplace_back(worker);
    worker();
    for (auto& x : th) x.join();
    return flags.load();
}

}  // namespace bzk
