// Host-side R1CS builder + the reference's gadget library, restated in C++ (product code).
//
// Mirrors, in allocation and constraint ORDER, what the reference builds on bellman 0.14's
// `ConstraintSystem<Fr>`:
//   Number / UnsignedInteger / mux / boolean helpers   src/zk/groth16/gadgets/common/{number,uint,mux,boolean}.rs
//   Poseidon gadget                                   src/zk/groth16/gadgets/poseidon/mod.rs:8-95
//   4-ary Merkle gadget                               src/zk/groth16/gadgets/merkle/mod.rs:21-78
//   EdDSA-on-Jubjub gadget                            src/zk/groth16/gadgets/eddsa/mod.rs:14-280
// and the bellman primitives they call (third-party, restated from the crate's published behaviour,
// SURVEY.md Appendix B): AllocatedNum::{alloc, mul, inputize, to_bits_le_strict},
// AllocatedBit::{alloc, and, and_not, nor, alloc_conditionally}, Boolean::and.
//
// One object serves both bellman roles: with `record_matrices` it is the KeypairAssembly (CSR of A, B,
// C for CRS generation), and it always is the ProvingAssignment (z, A.z, B.z, C.z, densities).
// Linear combinations merge duplicate variables (bellman appends; the evaluations are identical).
#pragma once
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

#include <algorithm>
#include <tuple>

#include "bzk_witfill.cuh"
#include "host_fr64.h"
#include "host_fr_ifma.h"
#include "host_zk.h"

struct bzk_ctx;
namespace bzk {

typedef uint32_t Var;  // bit 31 set => aux variable, else input variable (0 = ONE)
static constexpr Var VAR_AUX = 0x80000000u;
static constexpr Var VAR_ONE = 0;

// Witness-only synthesis (the ProvingAssignment role with a loaded CRS) needs no linear combinations: the
// gadgets hand every constraint's three VALUES to enforce().  `lc_tracking()` switches term bookkeeping off
// for that mode (thread-local so that transactions can be synthesized on worker threads).
inline bool& lc_tracking() {
    static thread_local bool on = true;
    return on;
}

struct LC {
    std::vector<std::pair<Var, Fr>> t;
    LC() {}
    LC& add(Var v, const Fr& c) {
        if (!lc_tracking()) return *this;
        for (auto& e : t)
            if (e.first == v) {
                e.second = fe_add<FrParams>(e.second, c);
                return *this;
            }
        t.emplace_back(v, c);
        return *this;
    }
    LC& add_var(Var v) { return add(v, Fr::one()); }
    LC& sub_var(Var v) { return add(v, fe_neg<FrParams>(Fr::one())); }
    LC& add_scaled(const LC& o, const Fr& k) {
        if (!lc_tracking()) return *this;
        for (auto& e : o.t) add(e.first, hfr::mul(e.second, k));
        return *this;
    }
    LC& add_lc(const LC& o) {
        for (auto& e : o.t) add(e.first, e.second);
        return *this;
    }
    LC& sub_lc(const LC& o) {
        for (auto& e : o.t) add(e.first, fe_neg<FrParams>(e.second));
        return *this;
    }
    static LC of(Var v) { return LC().add_var(v); }
    static LC one() { return LC().add_var(VAR_ONE); }
};

// Allocator for the arrays that are handed to the GPU (z, A.z, B.z, C.z: 116 MB per 2^20-class proof): large
// blocks come from a process-wide pool of PINNED host memory, so that bzk_groth16_prove's hipMemcpyAsync is a
// true asynchronous DMA at PCIe rate instead of a staged pageable copy.  Blocks return to the pool on free (the
// next proof of the same shape reuses them; hipHostMalloc itself is slow).  No device / hipHostMalloc failure
// => plain malloc: this is memory plumbing only, it computes nothing.
void* pinned_pool_take(size_t bytes);   // host_pinned pool, mpn.hip
void pinned_pool_give(void* p, size_t bytes);
template <class T>
struct PinnedPoolAlloc {
    using value_type = T;
    PinnedPoolAlloc() = default;
    template <class U>
    PinnedPoolAlloc(const PinnedPoolAlloc<U>&) {}
    T* allocate(size_t n) {
        void* p = pinned_pool_take(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return (T*)p;
    }
    void deallocate(T* p, size_t n) { pinned_pool_give(p, n * sizeof(T)); }
    // resize() DEFAULT-initialises (no zero fill): the witness arrays of a production circuit are gigabytes that the worker
    // threads overwrite completely (every slot of a window is accounted for, ConstraintSystem::set_window), and a
    // single-threaded memset of them was a third of the generator's wall time at 2^24 - 2^26 constraints
    template <class U>
    void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new ((void*)p) U;
    }
    template <class U, class... Args>
    void construct(U* p, Args&&... args) {
        ::new ((void*)p) U(std::forward<Args>(args)...);
    }
    template <class U>
    bool operator==(const PinnedPoolAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const PinnedPoolAlloc<U>&) const { return false; }
};
using FrVec = std::vector<Fr, PinnedPoolAlloc<Fr>>;

struct CsrBuilder {
    std::vector<uint32_t> row_ptr{0}, col;
    std::vector<Fr> val;
};

// ------------------------------------------------------------------------------------------------
// Deferred witness values (bzk_witfill.cuh): what the gadgets emit instead of values when a transition is synthesized with
// `cs.defer` set.  The PROGRAM is a property of the circuit shape (every transition runs the same gadget calls): it is recorded
// once, on a plan pass over a null transition; a normal pass only appends the host-known operand values to the transition's
// input record - in the same order - and skips the slots.
// ------------------------------------------------------------------------------------------------
struct DeferGroup {  // a run of ops of one kind and width (pass 1: inside one level): one launch
    uint8_t kind, t;
    uint16_t level;
    uint32_t start, count;
};
struct DeferProgram {
    std::vector<wf::Op> ops;            // emission order (plan pass)
    uint32_t n_regs = 0, n_inputs = 0;
    size_t n_aux = 0, n_con = 0;        // the transition's window
    // finalize(): the V ops by (level, kind, t), the F ops by (kind, t), each list cut into launches; the constraint slots the device
    // fills, as sorted disjoint ranges (the host's a b = c scan skips them)
    std::vector<wf::Op> v_ops, f_ops;
    std::vector<DeferGroup> v_groups, f_groups;
    std::vector<std::pair<uint32_t, uint32_t>> con_holes;
    uint32_t n_levels = 0;
    uint32_t max_sel_chain = 0;         // finalize(): selections naming selections, longest chain (device limit: wf::MAX_SEL_CHAIN)
    size_t hole_aux = 0, hole_con = 0;  // slots per transition left to the device
    void finalize() {
        v_ops.clear(); f_ops.clear();
        for (const wf::Op& o : ops) (o.kind == wf::V_HASH || o.kind == wf::V_SEL ? v_ops : f_ops).push_back(o);
        // selections keep their EMISSION order inside a level (no key on Is / Not): the host executor runs them in list order, and a selection may read the
        // register of an earlier selection of the same level; hashes of a level only read lower levels, so their order is free
        std::stable_sort(v_ops.begin(), v_ops.end(), [](const wf::Op& a, const wf::Op& b) {
            return std::make_tuple(a.level, a.kind, a.kind == wf::V_SEL ? (uint8_t)0 : a.t) < std::make_tuple(b.level, b.kind, b.kind == wf::V_SEL ? (uint8_t)0 : b.t);
        });
        // longest chain of selections naming selections: the device resolves them where they are read, through at most wf::MAX_SEL_CHAIN links
        {
            std::vector<uint32_t> depth(std::max<uint32_t>(1, n_regs), 0);
            max_sel_chain = 0;
            for (const wf::Op& o : v_ops)
                if (o.kind == wf::V_SEL && o.out >= 0 && (size_t)o.out < depth.size()) {
                    uint32_t d = 0;
                    for (int k = 1; k <= 2; ++k)
                        if (o.in[k] >= 0 && (size_t)o.in[k] < depth.size()) d = std::max(d, depth[(size_t)o.in[k]]);
                    depth[(size_t)o.out] = d + 1;
                    max_sel_chain = std::max(max_sel_chain, d + 1);
                }
        }
        std::stable_sort(f_ops.begin(), f_ops.end(), [](const wf::Op& a, const wf::Op& b) { return std::make_tuple(a.kind, a.t) < std::make_tuple(b.kind, b.t); });
        auto cut = [](const std::vector<wf::Op>& v, std::vector<DeferGroup>& g, bool by_level) {
            g.clear();
            for (uint32_t i = 0; i < v.size(); ++i) {
                if (!g.empty() && g.back().kind == v[i].kind && g.back().t == v[i].t && (!by_level || g.back().level == v[i].level)) ++g.back().count;
                else g.push_back({v[i].kind, v[i].t, v[i].level, i, 1u});
            }
        };
        cut(v_ops, v_groups, true);
        cut(f_ops, f_groups, false);
        n_levels = v_ops.empty() ? 0 : v_ops.back().level;
        con_holes.clear();
        hole_aux = hole_con = 0;
        for (const wf::Op& o : f_ops) {
            uint32_t na = 0, nc = 0;
            if (o.kind == wf::F_POSEIDON) {
                const PoseidonHostParams P = poseidon_host_params_cached(o.t);
                na = nc = wf::poseidon_slots(o.t, P.rf, P.rp);
            } else if (o.kind == wf::F_MUX) { na = 1; nc = 1; }
            else if (o.kind == wf::F_ASSERT_EQ_IF) { na = 1; nc = 2; }
            else if (o.kind == wf::F_ENFORCE_EQ) { nc = 1; }
            hole_aux += na; hole_con += nc;
            if (nc) con_holes.push_back({o.con_off, nc});
        }
        std::sort(con_holes.begin(), con_holes.end());
    }
};
struct Defer {
    DeferProgram* prog;  // written on the plan pass only
    bool plan;
    Fr* inputs;          // this transition's input record
    uint32_t cap_inputs;
    uint32_t n_in = 0;
    int32_t n_regs = 0;
    bool overflow = false;
    std::vector<uint16_t> reg_level;  // plan pass: level of the op producing each register
    int32_t in_val(const Fr& v) {
        if (n_in < cap_inputs) inputs[n_in] = v;
        else overflow = true;
        return ~(int32_t)(n_in++);
    }
    int32_t new_reg(uint16_t level) {
        if (plan) reg_level.push_back(level);
        return n_regs++;
    }
    uint16_t level_of(int32_t operand) const { return plan && operand >= 0 ? reg_level[(size_t)operand] : (uint16_t)0; }
    void emit(const wf::Op& op) {
        if (plan) prog->ops.push_back(op);
    }
};
// what an R1CS instance with deferred values carries besides its (incomplete) arrays
struct DeferData {
    const DeferProgram* prog = nullptr;
    FrVec inputs;                       // n_tx records of prog->n_inputs scalars (pinned: uploaded by DMA)
    size_t n_tx = 0;
    size_t base_aux = 0, base_con = 0;  // aux / constraint index of transition 0's window; transition t at base + t * stride
    size_t stride_aux = 0, stride_con = 0;
    bool filled = false;                // the host arrays are complete (bzk_r1cs_fill_host ran)
    uint32_t flags = 0;                 // wf::FLAG_* of that fill
};
// witfill.hip: the program over all transitions on the device (arrays = device views) / on the host
// enqueues on ctx->stream; the flags word follows in stream order (into flags_dst, a pinned word, or the context's own: witfill_flags)
int32_t witfill_run_dev(bzk_ctx* ctx, const DeferData& dd, const wf::Arrays& A, uint32_t* flags_dst = nullptr);
uint32_t witfill_flags(bzk_ctx* ctx);                                                // ... and is read here once that stream has been synchronised
uint32_t witfill_run_host(const DeferData& dd, const wf::Arrays& A);
void witfill_quiesce(bzk_ctx* ctx);
void witfill_schedule_info(const DeferProgram& P, uint64_t info[6]);  // the one-launch schedule of a program, checked on the host

class ConstraintSystem {
   public:
    bool record_matrices;
    std::vector<Fr> inputs;        // assignment, inputs[0] = 1
    FrVec aux;                     // assignment, auxiliary variables
    FrVec az, bz, cz;              // per-constraint evaluations
    CsrBuilder A, B, C;            // only when record_matrices (columns still Var-encoded until finalize)
    std::vector<uint8_t> a_in_d, a_aux_d, b_in_d, b_aux_d;  // densities by appearance
    bool finalized = false;

    explicit ConstraintSystem(bool record) : record_matrices(record) { inputs.push_back(Fr::one()); }

    // Window mode (witness-only synthesis of one transition by a worker thread): values go straight into a slice of
    // the parent system's arrays, whose size is known in advance (every transition of a circuit allocates the same
    // number of variables and constraints).  Writing past the slice is refused and reported through `win_overflow`.
    Fr *win_aux = nullptr, *win_az = nullptr, *win_bz = nullptr, *win_cz = nullptr;
    size_t win_n_aux = 0, win_n_con = 0, win_cap_aux = 0, win_cap_con = 0;
    bool win_overflow = false;
    Defer* defer = nullptr;  // window mode only: hash-dependent values are left to the device (DeferProgram above)
    void skip(size_t n_aux_slots, size_t n_con_slots) {
        win_n_aux += n_aux_slots;
        win_n_con += n_con_slots;
        if (win_n_aux > win_cap_aux || win_n_con > win_cap_con) win_overflow = true;
    }
    void set_window(Fr* a, size_t cap_a, Fr* x, Fr* y, Fr* z, size_t cap_c) {
        win_aux = a; win_az = x; win_bz = y; win_cz = z;
        win_cap_aux = cap_a; win_cap_con = cap_c;
    }

    Var alloc(const Fr& v) {
        if (win_aux) {
            if (win_n_aux < win_cap_aux) win_aux[win_n_aux] = v;
            else win_overflow = true;
            return VAR_AUX | (Var)(win_n_aux++);
        }
        aux.push_back(v);
        return VAR_AUX | (Var)(aux.size() - 1);
    }
    Var alloc_input(const Fr& v) {
        inputs.push_back(v);
        return (Var)(inputs.size() - 1);
    }
    const Fr& value(Var v) const { return (v & VAR_AUX) ? aux[v & ~VAR_AUX] : inputs[v]; }
    Fr eval(const LC& lc) const {
        Fr acc = Fr::zero();
        for (auto& e : lc.t) acc = fe_add<FrParams>(acc, hfr::mul(e.second, value(e.first)));
        return acc;
    }
    size_t num_constraints() const { return win_aux ? win_n_con : az.size(); }

    // The caller supplies <A,z>, <B,z>, <C,z> (it computed them while building the witness).  In tracking mode
    // the LCs are recorded (densities, matrices) and, when self_check is set, re-evaluated against the values.
    bool self_check = false;
    long check_failed_at = -1;
    void enforce(const LC& a, const Fr& av, const LC& b, const Fr& bv, const LC& c, const Fr& cv) {
        if (win_aux) {
            if (win_n_con < win_cap_con) {
                win_az[win_n_con] = av;
                win_bz[win_n_con] = bv;
                win_cz[win_n_con] = cv;
            } else {
                win_overflow = true;
            }
            ++win_n_con;
            return;
        }
        if (lc_tracking()) {
            if (self_check && check_failed_at < 0 && (!eval(a).equals(av) || !eval(b).equals(bv) || !eval(c).equals(cv)))
                check_failed_at = (long)az.size();
            mark(a, a_in_d, a_aux_d);
            mark(b, b_in_d, b_aux_d);
            if (record_matrices) {
                push_row(A, a);
                push_row(B, b);
                push_row(C, c);
            }
        }
        az.push_back(av);
        bz.push_back(bv);
        cz.push_back(cv);
    }

    // bellman appends `input_i * 0 = 0` for every input after synthesis (makes every input's A
    // polynomial non-zero); call exactly once.
    void finalize() {
        if (finalized) return;
        for (size_t i = 0; i < inputs.size(); ++i) enforce(LC::of((Var)i), inputs[i], LC(), Fr::zero(), LC(), Fr::zero());
        finalized = true;
    }

    bool is_satisfied() const {
        for (size_t k = 0; k < az.size(); ++k)
            if (!hfr::mul(az[k], bz[k]).equals(cz[k])) return false;
        return true;
    }
    // first violated constraint or -1
    long first_unsatisfied() const {
        for (size_t k = 0; k < az.size(); ++k)
            if (!hfr::mul(az[k], bz[k]).equals(cz[k])) return (long)k;
        return -1;
    }
    uint32_t flat_index(Var v) const { return (v & VAR_AUX) ? (uint32_t)inputs.size() + (v & ~VAR_AUX) : v; }

   private:
    static void mark(const LC& lc, std::vector<uint8_t>& din, std::vector<uint8_t>& daux) {
        for (auto& e : lc.t) {
            if (e.second.is_zero()) continue;
            std::vector<uint8_t>& d = (e.first & VAR_AUX) ? daux : din;
            const uint32_t i = e.first & ~VAR_AUX;
            if (d.size() <= i) d.resize(i + 1, 0);
            d[i] = 1;
        }
    }
    static void push_row(CsrBuilder& m, const LC& lc) {
        for (auto& e : lc.t) {
            if (e.second.is_zero()) continue;
            m.col.push_back(e.first);
            m.val.push_back(e.second);
        }
        m.row_ptr.push_back((uint32_t)m.col.size());
    }
};

// ------------------------------------------------------------------------------------------------
// bellman primitives
// ------------------------------------------------------------------------------------------------
struct Num {  // AllocatedNum
    Var var;
    Fr val;
    int32_t ref = -1;  // >= 0: the value is deferred (register of the transition's DeferProgram), `val` is not meaningful
};
struct Bit {  // AllocatedBit
    Var var;
    bool val;
};
struct Bool {  // Boolean
    enum Kind { IS, NOT, CONST } kind;
    Bit bit;
    bool cval;
    static Bool is(const Bit& b) { return {IS, b, false}; }
    static Bool not_(const Bit& b) { return {NOT, b, false}; }
    static Bool constant(bool v) { return {CONST, {0, false}, v}; }
    bool value() const { return kind == IS ? bit.val : kind == NOT ? !bit.val : cval; }
    Bool negate() const {
        if (kind == IS) return not_(bit);
        if (kind == NOT) return is(bit);
        return constant(!cval);
    }
};

static inline Fr fr_from_bool(bool b) { return b ? Fr::one() : Fr::zero(); }
static inline Fr fr_from_u64(uint64_t x) { return ZkScalar::from_u64(x).v; }

static inline Num num_alloc(ConstraintSystem& cs, const Fr& v) { return {cs.alloc(v), v}; }

static inline void num_inputize(ConstraintSystem& cs, const Num& n) {
    Var in = cs.alloc_input(n.val);
    cs.enforce(LC::of(in), n.val, LC::one(), Fr::one(), LC::of(n.var), n.val);
}

static inline Num num_mul(ConstraintSystem& cs, const Num& a, const Num& b) {  // AllocatedNum::mul
    Num r = num_alloc(cs, hfr::mul(a.val, b.val));
    cs.enforce(LC::of(a.var), a.val, LC::of(b.var), b.val, LC::of(r.var), r.val);
    return r;
}

static inline Bit bit_alloc(ConstraintSystem& cs, bool v) {
    Bit b = {cs.alloc(fr_from_bool(v)), v};
    cs.enforce(LC::one().sub_var(b.var), fr_from_bool(!v), LC::of(b.var), fr_from_bool(v), LC(), Fr::zero());  // (1 - a) * a = 0
    return b;
}
static inline Bit bit_alloc_conditionally(ConstraintSystem& cs, bool v, const Bit& must_be_false) {
    Bit b = {cs.alloc(fr_from_bool(v)), v};
    cs.enforce(LC::one().sub_var(must_be_false.var).sub_var(b.var),
               fe_sub<FrParams>(fe_sub<FrParams>(Fr::one(), fr_from_bool(must_be_false.val)), fr_from_bool(v)), LC::of(b.var),
               fr_from_bool(v), LC(), Fr::zero());
    return b;
}
static inline Bit bit_and(ConstraintSystem& cs, const Bit& a, const Bit& b) {
    Bit r = {cs.alloc(fr_from_bool(a.val && b.val)), a.val && b.val};
    cs.enforce(LC::of(a.var), fr_from_bool(a.val), LC::of(b.var), fr_from_bool(b.val), LC::of(r.var), fr_from_bool(r.val));
    return r;
}
static inline Bit bit_and_not(ConstraintSystem& cs, const Bit& a, const Bit& b) {  // a AND (NOT b)
    Bit r = {cs.alloc(fr_from_bool(a.val && !b.val)), a.val && !b.val};
    cs.enforce(LC::of(a.var), fr_from_bool(a.val), LC::one().sub_var(b.var), fr_from_bool(!b.val), LC::of(r.var), fr_from_bool(r.val));
    return r;
}
static inline Bit bit_nor(ConstraintSystem& cs, const Bit& a, const Bit& b) {  // (NOT a) AND (NOT b)
    Bit r = {cs.alloc(fr_from_bool(!a.val && !b.val)), !a.val && !b.val};
    cs.enforce(LC::one().sub_var(a.var), fr_from_bool(!a.val), LC::one().sub_var(b.var), fr_from_bool(!b.val), LC::of(r.var),
               fr_from_bool(r.val));
    return r;
}
static inline Bool bool_and(ConstraintSystem& cs, const Bool& a, const Bool& b) {  // Boolean::and
    if (a.kind == Bool::CONST) return a.cval ? b : Bool::constant(false);
    if (b.kind == Bool::CONST) return b.cval ? a : Bool::constant(false);
    if (a.kind == Bool::IS && b.kind == Bool::IS) return Bool::is(bit_and(cs, a.bit, b.bit));
    if (a.kind == Bool::IS && b.kind == Bool::NOT) return Bool::is(bit_and_not(cs, a.bit, b.bit));
    if (a.kind == Bool::NOT && b.kind == Bool::IS) return Bool::is(bit_and_not(cs, b.bit, a.bit));
    return Bool::is(bit_nor(cs, a.bit, b.bit));
}

// AllocatedNum::to_bits_le_strict: bits of the canonical value, constrained to be <= r - 1.
// Returned little-endian (bit 0 first); every entry is an allocated bit.
static inline std::vector<Bit> num_to_bits_le_strict(ConstraintSystem& cs, const Num& n) {
    uint32_t a[8], rm1[8];
    ZkScalar(n.val).to_canonical(a);
    {
        uint64_t borrow = 1;
        for (int i = 0; i < 8; ++i) {
            uint64_t d = (uint64_t)FrParams::MOD[i] - borrow;
            rm1[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
    }
    std::vector<Bit> result;  // big-endian while building
    std::vector<Bit> current_run;
    bool have_last_run = false, found_one = false;
    Bit last_run = {0, false};
    for (int i = 255; i >= 0; --i) {
        const bool b = (rm1[i >> 5] >> (i & 31)) & 1;
        const bool abit = (a[i >> 5] >> (i & 31)) & 1;
        found_one |= b;
        if (!found_one) continue;  // leading zero bit of r - 1 (bit 255): a's bit is zero as well
        if (b) {
            Bit ab = bit_alloc(cs, abit);
            current_run.push_back(ab);
            result.push_back(ab);
        } else {
            if (!current_run.empty()) {
                if (have_last_run) current_run.push_back(last_run);
                Bit cur = current_run[0];  // k-ary AND, left to right
                for (size_t k = 1; k < current_run.size(); ++k) cur = bit_and(cs, cur, current_run[k]);
                last_run = cur;
                have_last_run = true;
                current_run.clear();
            }
            result.push_back(bit_alloc_conditionally(cs, abit, last_run));
        }
    }
    LC lc;
    Fr coeff = Fr::one();
    for (size_t k = result.size(); k-- > 0;) {
        lc.add(result[k].var, coeff);
        coeff = fe_dbl<FrParams>(coeff);
    }
    lc.sub_var(n.var);
    cs.enforce(LC(), Fr::zero(), LC(), Fr::zero(), lc, Fr::zero());  // unpacking constraint: 0 * 0 = sum(bits) - value
    std::vector<Bit> le(result.rbegin(), result.rend());
    return le;
}

// ------------------------------------------------------------------------------------------------
// common/number.rs
// ------------------------------------------------------------------------------------------------
struct Number {
    LC lc;
    Fr val;
    int32_t ref = -1;  // as Num::ref.  Arithmetic on a deferred Number is refused: only the gadgets with a device form may consume one
    void known() const {
        if (ref >= 0) throw std::logic_error("arithmetic on a deferred witness value");
    }
    static Number zero() { return {LC(), Fr::zero()}; }
    static Number one() { return {LC::one(), Fr::one()}; }
    static Number constant(const Fr& v) { return {LC().add(VAR_ONE, v), v}; }
    static Number from(const Num& n) { return {LC::of(n.var), n.val, n.ref}; }
    static Number from_scaled(const Fr& k, const Num& n) {
        if (n.ref >= 0) throw std::logic_error("arithmetic on a deferred witness value");
        return {LC().add(n.var, k), hfr::mul(n.val, k)};
    }
    static Number from(const Bit& b) { return {LC::of(b.var), fr_from_bool(b.val)}; }
    void add_constant(const Fr& c) {
        lc.add(VAR_ONE, c);
        val = fe_add<FrParams>(val, c);
    }
    void add_num(const Fr& coeff, const Num& n) {
        known();
        if (n.ref >= 0) throw std::logic_error("arithmetic on a deferred witness value");
        lc.add(n.var, coeff);
        val = fe_add<FrParams>(val, hfr::mul(n.val, coeff));
    }
    Number plus(const Number& o) const {
        known(); o.known();
        Number r = *this;
        r.lc.add_lc(o.lc);
        r.val = fe_add<FrParams>(val, o.val);
        return r;
    }
    Number plus_scaled(const Fr& k, const Number& o) const {
        known(); o.known();
        Number r = *this;
        r.lc.add_scaled(o.lc, k);
        r.val = fe_add<FrParams>(val, hfr::mul(k, o.val));
        return r;
    }
    Number minus(const Number& o) const {
        known(); o.known();
        Number r = *this;
        r.lc.sub_lc(o.lc);
        r.val = fe_sub<FrParams>(val, o.val);
        return r;
    }
    Num mul(ConstraintSystem& cs, const Number& o) const {  // number.rs:48-65
        known(); o.known();
        Num r = num_alloc(cs, hfr::mul(val, o.val));
        cs.enforce(lc, val, o.lc, o.val, LC::of(r.var), r.val);
        return r;
    }
    Num compress(ConstraintSystem& cs) const { return mul(cs, one()); }  // :66-72
    Bool is_zero(ConstraintSystem& cs) const {                            // :75-111
        known();
        const bool z = val.is_zero();
        Bit isz = bit_alloc(cs, z);
        Num inv = num_alloc(cs, z ? Fr::zero() : hfr::inv(val));
        cs.enforce(LC().sub_lc(lc), fe_neg<FrParams>(val), LC::of(inv.var), inv.val, LC::of(isz.var).sub_var(VAR_ONE),
                   fe_sub<FrParams>(fr_from_bool(z), Fr::one()));
        cs.enforce(LC::of(isz.var), fr_from_bool(z), lc, val, LC(), Fr::zero());
        return Bool::is(isz);
    }
    Bool is_equal(ConstraintSystem& cs, const Number& o) const { return minus(o).is_zero(cs); }  // :113-119
    void assert_equal(ConstraintSystem& cs, const Number& o) const {  // :121-128
        if (cs.defer && (ref >= 0 || o.ref >= 0)) {  // this * 1 = other, on the device
            Defer& D = *cs.defer;
            wf::Op f{};
            f.kind = wf::F_ENFORCE_EQ; f.out = -1; f.aux_off = (uint32_t)cs.win_n_aux; f.con_off = (uint32_t)cs.win_n_con;
            f.in[0] = ref >= 0 ? ref : D.in_val(val);
            f.in[1] = o.ref >= 0 ? o.ref : D.in_val(o.val);
            D.emit(f);
            cs.skip(0, 1);
            return;
        }
        cs.enforce(lc, val, LC::one(), Fr::one(), o.lc, o.val);
    }
    void assert_equal_if_enabled(ConstraintSystem& cs, const Bool& enabled, const Number& o) const {  // :130-177
        if (cs.defer && (ref >= 0 || o.ref >= 0) && enabled.kind == Bool::IS) {
            Defer& D = *cs.defer;
            wf::Op f{};
            f.kind = wf::F_ASSERT_EQ_IF; f.out = -1; f.aux_off = (uint32_t)cs.win_n_aux; f.con_off = (uint32_t)cs.win_n_con;
            f.in[0] = D.in_val(fr_from_bool(enabled.bit.val));
            f.in[1] = ref >= 0 ? ref : D.in_val(val);
            f.in[2] = o.ref >= 0 ? o.ref : D.in_val(o.val);
            D.emit(f);
            cs.skip(1, 2);
            return;
        }
        known(); o.known();
        if (enabled.kind == Bool::IS) {
            const Fr ev = enabled.bit.val ? val : Fr::zero(), en = fr_from_bool(enabled.bit.val);
            Var eis = cs.alloc(ev);
            cs.enforce(LC::of(enabled.bit.var), en, lc, val, LC::of(eis), ev);
            cs.enforce(LC::of(enabled.bit.var), en, o.lc, o.val, LC::of(eis), ev);
        } else if (enabled.kind == Bool::CONST) {
            if (enabled.cval) assert_equal(cs, o);
        } else {
            throw std::logic_error("assert_equal_if_enabled(Boolean::Not) is unimplemented in the reference");
        }
    }
};

// ------------------------------------------------------------------------------------------------
// common/boolean.rs, common/mux.rs
// ------------------------------------------------------------------------------------------------
static inline Number extract_bool(const Bool& b) {
    if (b.kind == Bool::IS) return Number::from(b.bit);
    if (b.kind == Bool::NOT) return Number::one().minus(Number::from(b.bit));
    return b.cval ? Number::one() : Number::zero();
}
static inline void assert_true(ConstraintSystem& cs, const Bool& b) { extract_bool(b).assert_equal(cs, Number::one()); }
static inline Bool boolean_or(ConstraintSystem& cs, const Bool& a, const Bool& b) {
    return bool_and(cs, a.negate(), b.negate()).negate();
}
// select ? b : a      (mux.rs:7-47)
static inline Num mux(ConstraintSystem& cs, const Bool& select, const Number& a, const Number& b) {
    if (cs.defer && (a.ref >= 0 || b.ref >= 0) && select.kind != Bool::CONST) {
        // one variable, one constraint, filled in by the device (F_MUX); the selected VALUE is a register of pass 1 (V_SEL)
        Defer& D = *cs.defer;
        wf::Op f{};
        f.kind = wf::F_MUX; f.t = select.kind == Bool::NOT ? 1 : 0; f.out = -1;
        f.aux_off = (uint32_t)cs.win_n_aux; f.con_off = (uint32_t)cs.win_n_con;
        f.in[0] = D.in_val(fr_from_bool(select.bit.val));
        f.in[1] = a.ref >= 0 ? a.ref : D.in_val(a.val);
        f.in[2] = b.ref >= 0 ? b.ref : D.in_val(b.val);
        wf::Op v = f;
        v.kind = wf::V_SEL; v.aux_off = v.con_off = 0;
        v.level = std::max(D.level_of(f.in[1]), D.level_of(f.in[2]));  // a selection costs no level: the device resolves it where it is read
        v.out = D.new_reg(v.level);
        D.emit(v);
        D.emit(f);
        cs.skip(1, 1);
        Num ret{VAR_AUX | (Var)(cs.win_n_aux - 1), Fr::zero()};
        ret.ref = v.out;
        return ret;
    }
    a.known(); b.known();
    if (select.kind == Bool::IS) {
        Num ret = num_alloc(cs, select.bit.val ? b.val : a.val);
        cs.enforce(LC().add_lc(a.lc).sub_lc(b.lc), fe_sub<FrParams>(a.val, b.val), LC::of(select.bit.var), fr_from_bool(select.bit.val),
                   LC().add_lc(a.lc).sub_var(ret.var), fe_sub<FrParams>(a.val, ret.val));
        return ret;
    }
    if (select.kind == Bool::NOT) {
        Num ret = num_alloc(cs, select.bit.val ? a.val : b.val);
        cs.enforce(LC().add_lc(b.lc).sub_lc(a.lc), fe_sub<FrParams>(b.val, a.val), LC::of(select.bit.var), fr_from_bool(select.bit.val),
                   LC().add_lc(b.lc).sub_var(ret.var), fe_sub<FrParams>(b.val, ret.val));
        return ret;
    }
    throw std::logic_error("mux(Boolean::Constant) is unimplemented in the reference");
}

// ------------------------------------------------------------------------------------------------
// common/uint.rs
// ------------------------------------------------------------------------------------------------
struct UInt {
    std::vector<Bit> bits;
    Number num;
    static UInt constrain(ConstraintSystem& cs, const Number& num, int num_bits) {  // :66-91
        UInt u;
        u.num = num;
        const ZkScalar v(num.val);
        LC all;
        Fr coeff = Fr::one();
        for (int i = 0; i < num_bits; ++i) {
            Bit b = bit_alloc(cs, v.bit(i));
            all.add(b.var, coeff);
            u.bits.push_back(b);
            coeff = fe_dbl<FrParams>(coeff);
        }
        {
            // sum(bit_i 2^i) recomputed from the allocated bits so that an out-of-range value is REPORTED as an
            // unsatisfied constraint (as bellman's prover would produce an invalid proof), not silently accepted
            Fr packed = Fr::zero(), cf = Fr::one();
            for (int i = 0; i < num_bits; ++i) {
                if (u.bits[i].val) packed = fe_add<FrParams>(packed, cf);
                cf = fe_dbl<FrParams>(cf);
            }
            cs.enforce(all, packed, LC::one(), Fr::one(), num.lc, num.val);
        }
        return u;
    }
    static UInt alloc(ConstraintSystem& cs, const Fr& val, int bits) {  // :34-41
        Num a = num_alloc(cs, val);
        return constrain(cs, Number::from(a), bits);
    }
    static UInt alloc_64(ConstraintSystem& cs, uint64_t v) { return alloc(cs, fr_from_u64(v), 64); }
    Bool lt(ConstraintSystem& cs, const UInt& other) const {  // :94-109
        const int nb = (int)bits.size();
        if ((int)other.bits.size() != nb) throw std::logic_error("UnsignedInteger::lt: width mismatch");
        Fr two_bits = Fr::one();
        for (int i = 0; i < nb + 1; ++i) two_bits = fe_dbl<FrParams>(two_bits);  // 2^(nb+1)
        Number sub = num.minus(other.num);
        sub.add_constant(two_bits);
        UInt sb = constrain(cs, sub, nb + 2);
        return Bool::is(sb.bits[nb]);
    }
    Bool gt(ConstraintSystem& cs, const UInt& other) const { return other.lt(cs, *this); }
    Bool lte(ConstraintSystem& cs, const UInt& other) const { return gt(cs, other).negate(); }
    Bool gte(ConstraintSystem& cs, const UInt& other) const { return lt(cs, other).negate(); }
};

// ------------------------------------------------------------------------------------------------
// poseidon/mod.rs
// ------------------------------------------------------------------------------------------------
static inline Num g_sbox(ConstraintSystem& cs, const Number& a) {
    Num a2 = a.mul(cs, a);
    Num a4 = num_mul(cs, a2, a2);
    return a.mul(cs, Number::from(a4));
}
static inline std::vector<Number> g_product_mds(const std::vector<Number>& vals, const PoseidonHostParams& P) {
    const int t = (int)vals.size();
    std::vector<Number> res(t, Number::zero());
    for (int j = 0; j < t; ++j)
        for (int k = 0; k < t; ++k) res[j] = res[j].plus_scaled(P.mds[j * t + k], vals[k]);
    return res;
}
// Witness-only synthesis (lc_tracking() off: every LC is empty, only values are recorded): the same allocations and constraints in the
// same order as the LC form below - per S-box  x2 = x x, x4 = x2 x2, x5 = x x4  (three variables, three constraints), per idle lane of a
// partial round  v = v * 1  - on plain field values, the MDS rows as dot products with one reduction each (host_fr64.h).  The Poseidon
// gadget is ~2/3 of a transaction's witness time; this form needs no vector<Number> traffic and ~0.6 of the word multiplications.
static inline bool defer_width_ok(int t) { return t == 3 || t == 5 || t == 6 || t == 8; }  // the widths the device fill is instantiated for (witfill.hip)
static inline Number g_poseidon_values(ConstraintSystem& cs, const std::vector<Number>& vals, bool need_value) {
    static const LC none;
    const int t = (int)vals.size() + 1;
    const PoseidonHostParams P = poseidon_host_params_cached(t);
    if (cs.defer && defer_width_ok(t)) {
        // the gadget's variables and constraints are written by the device (F_POSEIDON); the OUTPUT is a register of pass 1 (V_HASH: the
        // sparse-round hash) unless the caller needs it here (EdDSA's bit decomposition): then all inputs must be known and the host hashes
        Defer& D = *cs.defer;
        wf::Op f{};
        f.kind = wf::F_POSEIDON; f.t = (uint8_t)t; f.out = -1;
        f.aux_off = (uint32_t)cs.win_n_aux; f.con_off = (uint32_t)cs.win_n_con;
        bool any_ref = false;
        uint16_t lvl = 0;
        for (int i = 0; i < t - 1; ++i) {
            f.in[i] = vals[i].ref >= 0 ? vals[i].ref : D.in_val(vals[i].val);
            any_ref |= vals[i].ref >= 0;
            lvl = std::max(lvl, D.level_of(f.in[i]));
        }
        D.emit(f);
        const uint32_t slots = wf::poseidon_slots(t, P.rf, P.rp);
        cs.skip(slots, slots);
        Number out{LC(), Fr::zero()};
        if (need_value) {
            if (any_ref) throw std::logic_error("a Poseidon output is needed on the host but depends on deferred values");
            ZkScalar in[8];
            for (int i = 0; i < t - 1; ++i) in[i].v = vals[i].val;
            out.val = poseidon_hash(in, t - 1).v;
        } else {
            wf::Op v = f;
            v.kind = wf::V_HASH; v.aux_off = v.con_off = 0;
            v.level = (uint16_t)(lvl + 1);
            v.out = D.new_reg(v.level);
            D.emit(v);
            out.ref = v.out;
        }
        return out;
    }
    for (auto& x : vals) x.known();
    const hfr::MdsTable& mds_tab = poseidon_mds_table(t);
    Fr e[17], nw[17];
    e[0] = Fr::zero();
    for (int i = 1; i < t; ++i) e[i] = vals[i - 1].val;
    const Fr one = Fr::one();
    auto sbox = [&](const Fr& x) {
        const Fr x2 = hfr::mul(x, x);
        cs.alloc(x2);
        cs.enforce(none, x, none, x, none, x2);
        const Fr x4 = hfr::mul(x2, x2);
        cs.alloc(x4);
        cs.enforce(none, x2, none, x2, none, x4);
        const Fr x5 = hfr::mul(x, x4);
        cs.alloc(x5);
        cs.enforce(none, x, none, x4, none, x5);
        return x5;
    };
    int off = 0;
    for (int rnd = 0; rnd < P.rf + P.rp; ++rnd) {
        for (int i = 0; i < t; ++i) e[i] = fe_add<FrParams>(e[i], P.rc[off + i]);
        off += t;
        const bool full = rnd < P.rf / 2 || rnd >= P.rf / 2 + P.rp;
        if (full) {
            for (int i = 0; i < t; ++i) e[i] = sbox(e[i]);
        } else {
            e[0] = sbox(e[0]);
            for (int i = 1; i < t; ++i) {  // Number::compress: v * 1 = v
                cs.alloc(e[i]);
                cs.enforce(none, e[i], none, one, none, e[i]);
            }
        }
        hfr::mds_mul(mds_tab, e, nw);  // all t rows of the dense MDS at once (AVX-512 IFMA lanes; one dot product per row without it)
        for (int j = 0; j < t; ++j) e[j] = nw[j];
    }
    return {LC(), e[1]};
}
// need_value: the caller uses the output's VALUE on the host (only matters while values are deferred, see g_poseidon_values)
static inline Number g_poseidon(ConstraintSystem& cs, const std::vector<Number>& vals, bool need_value = false) {
    if (!lc_tracking()) return g_poseidon_values(cs, vals, need_value);
    std::vector<Number> e;
    e.push_back(Number::zero());
    for (auto& v : vals) e.push_back(v);
    const int t = (int)e.size();
    PoseidonHostParams P = poseidon_host_params(t);
    int off = 0;
    for (int rnd = 0; rnd < P.rf + P.rp; ++rnd) {
        for (int i = 0; i < t; ++i) e[i].add_constant(P.rc[off + i]);
        off += t;
        const bool full = rnd < P.rf / 2 || rnd >= P.rf / 2 + P.rp;
        if (full) {
            for (int i = 0; i < t; ++i) e[i] = Number::from(g_sbox(cs, e[i]));
        } else {
            e[0] = Number::from(g_sbox(cs, e[0]));
            for (int i = 1; i < t; ++i) e[i] = Number::from(e[i].compress(cs));
        }
        e = g_product_mds(e, P);
    }
    return e[1];
}

// ------------------------------------------------------------------------------------------------
// merkle/mod.rs
// ------------------------------------------------------------------------------------------------
typedef Num ProofTriple[3];
static inline Number g_merge_hash4(ConstraintSystem& cs, const Bit& s0b, const Bit& s1b, const Number& v, const Num* p) {
    Bool s0 = Bool::is(s0b), s1 = Bool::is(s1b);
    Bool andb = bool_and(cs, s0, s1);
    Bool orb = boolean_or(cs, s0, s1);
    Number p0 = Number::from(p[0]), p1 = Number::from(p[1]), p2 = Number::from(p[2]);
    Num v0 = mux(cs, orb, v, p0);
    Num v1p = mux(cs, s0, p0, v);
    Num v1 = mux(cs, s1, Number::from(v1p), p1);
    Num v2p = mux(cs, s0, v, p2);
    Num v2 = mux(cs, s1, p1, Number::from(v2p));
    Num v3 = mux(cs, andb, p2, v);
    return g_poseidon(cs, {Number::from(v0), Number::from(v1), Number::from(v2), Number::from(v3)});
}
struct MerkleProofWit {
    std::vector<Num> sib;  // 3 per level
};
static inline Number g_calc_root4(ConstraintSystem& cs, const UInt& index, const Number& val, const MerkleProofWit& proof) {
    if (index.bits.size() != proof.sib.size() / 3 * 2) throw std::logic_error("calc_root: index width != 2 * depth");
    Number cur = val;
    for (size_t lvl = 0; lvl < proof.sib.size() / 3; ++lvl)
        cur = g_merge_hash4(cs, index.bits[2 * lvl], index.bits[2 * lvl + 1], cur, &proof.sib[3 * lvl]);
    return cur;
}
static inline void g_check_proof4(ConstraintSystem& cs, const Bool& enabled, const UInt& index, const Number& val,
                                  const MerkleProofWit& proof, const Number& root) {
    Number nr = g_calc_root4(cs, index, val, proof);
    root.assert_equal_if_enabled(cs, enabled, nr);
}

// ------------------------------------------------------------------------------------------------
// eddsa/mod.rs
// ------------------------------------------------------------------------------------------------
struct APoint {
    Num x, y;
    PointAffine value() const { return {ZkScalar(x.val), ZkScalar(y.val)}; }
    static APoint alloc(ConstraintSystem& cs, const PointAffine& p) { return {num_alloc(cs, p.x.v), num_alloc(cs, p.y.v)}; }
    Bool is_null(ConstraintSystem& cs) const {
        Bool xz = Number::from(x).is_zero(cs);
        Bool yz = Number::from(y).is_zero(cs);
        return bool_and(cs, xz, yz);
    }
    Bool is_equal(ConstraintSystem& cs, const APoint& o) const {
        Bool xe = Number::from(x).is_equal(cs, Number::from(o.x));
        Bool ye = Number::from(y).is_equal(cs, Number::from(o.y));
        return bool_and(cs, xe, ye);
    }
    void assert_on_curve(ConstraintSystem& cs, const Bool& enabled) const {  // :64-75
        Num x2 = num_mul(cs, x, x), y2 = num_mul(cs, y, y), x2y2 = num_mul(cs, x2, y2);
        Number lhs = Number::from(y2).minus(Number::from(x2));
        Number rhs = Number::from_scaled(jubjub_d().v, x2y2).plus(Number::one());
        lhs.assert_equal_if_enabled(cs, enabled, rhs);
    }
    // `hint`: the sum, when the caller already knows it (ladder pre-pass); must equal what the formulas below give
    APoint add_const(ConstraintSystem& cs, const PointAffine& b, const PointAffine* hint = nullptr) const {  // :77-123
        PointAffine a = value(), sumv;  // default (0,0) when either operand is off-curve
        if (hint) {
            sumv = *hint;
        } else if (a.is_on_curve() && b.is_on_curve()) {
            sumv = a;
            sumv.add_assign(b);
        }
        APoint sum = alloc(cs, sumv);
        const Fr bx = b.x.v, by = b.y.v;
        const Fr dbb = hfr::mul(hfr::mul(jubjub_d().v, bx), by);
        Num common = num_mul(cs, x, y);
        const Fr kx = hfr::mul(dbb, common.val);
        cs.enforce(LC::one().add(common.var, dbb), fe_add<FrParams>(Fr::one(), kx), LC::of(sum.x.var), sum.x.val,
                   LC().add(x.var, by).add(y.var, bx), fe_add<FrParams>(hfr::mul(x.val, by), hfr::mul(y.val, bx)));
        // y_1 - y_2: by*y - (A*bx)*x with A = -1
        cs.enforce(LC::one().add(common.var, fe_neg<FrParams>(dbb)), fe_sub<FrParams>(Fr::one(), kx), LC::of(sum.y.var), sum.y.val,
                   LC().add(y.var, by).add(x.var, bx), fe_add<FrParams>(hfr::mul(y.val, by), hfr::mul(x.val, bx)));
        return sum;
    }
    APoint add(ConstraintSystem& cs, const APoint& o, const PointAffine* hint = nullptr) const {  // :125-172
        PointAffine a = value(), b = o.value(), sumv;
        if (hint) {
            sumv = *hint;
        } else if (a.is_on_curve() && b.is_on_curve()) {
            sumv = a;
            sumv.add_assign(b);
        }
        APoint sum = alloc(cs, sumv);
        const Fr d = jubjub_d().v;
        Num common = num_mul(cs, num_mul(cs, num_mul(cs, x, o.x), y), o.y);
        Num x1 = num_mul(cs, x, o.y), x2 = num_mul(cs, y, o.x);
        const Fr kd = hfr::mul(d, common.val);
        cs.enforce(LC::one().add(common.var, d), fe_add<FrParams>(Fr::one(), kd), LC::of(sum.x.var), sum.x.val,
                   LC::of(x1.var).add_var(x2.var), fe_add<FrParams>(x1.val, x2.val));
        Num y1 = num_mul(cs, y, o.y), y2 = num_mul(cs, x, o.x);
        // y_1 - A*y_2 with A = -1  =>  y_1 + y_2
        cs.enforce(LC::one().add(common.var, fe_neg<FrParams>(d)), fe_sub<FrParams>(Fr::one(), kd), LC::of(sum.y.var), sum.y.val,
                   LC::of(y1.var).add_var(y2.var), fe_add<FrParams>(y1.val, y2.val));
        return sum;
    }
    APoint mul(ConstraintSystem& cs, const Num& b) const {  // :174-202
        std::vector<Bit> le = num_to_bits_le_strict(cs, b);
        std::vector<Bit> bits(le.rbegin(), le.rend());  // MSB first
        APoint result = {mux(cs, Bool::is(bits[0]), Number::zero(), Number::from(x)),
                         mux(cs, Bool::is(bits[0]), Number::constant(Fr::one()), Number::from(y))};
        // witness values of the whole ladder with one inversion (only when the base is a curve point; otherwise
        // every sum is the off-curve default and the per-step path costs nothing)
        std::vector<PointAffine> hd, ha;
        const bool pre = value().is_on_curve();
        if (pre) {
            std::vector<bool> bv(bits.size());
            for (size_t i = 0; i < bits.size(); ++i) bv[i] = bits[i].val;
            jubjub_ladder(value(), bv, hd, ha);
        }
        for (size_t i = 1; i < bits.size(); ++i) {
            result = result.add(cs, result, pre ? &hd[i] : nullptr);
            APoint rpb = result.add(cs, *this, pre ? &ha[i] : nullptr);
            Num rx = mux(cs, Bool::is(bits[i]), Number::from(result.x), Number::from(rpb.x));
            Num ry = mux(cs, Bool::is(bits[i]), Number::from(result.y), Number::from(rpb.y));
            result = {rx, ry};
        }
        return result;
    }
};

static inline APoint g_base_mul(ConstraintSystem& cs, const PointAffine& base, const Num& b) {  // :205-236
    std::vector<Bit> le = num_to_bits_le_strict(cs, b);
    std::vector<Bit> bits(le.rbegin(), le.rend());
    APoint result = {mux(cs, Bool::is(bits[0]), Number::zero(), Number::constant(base.x.v)),
                     mux(cs, Bool::is(bits[0]), Number::constant(Fr::one()), Number::constant(base.y.v))};
    std::vector<PointAffine> hd, ha;
    const bool pre = base.is_on_curve();
    if (pre) {
        std::vector<bool> bv(bits.size());
        for (size_t i = 0; i < bits.size(); ++i) bv[i] = bits[i].val;
        jubjub_ladder(base, bv, hd, ha);
    }
    for (size_t i = 1; i < bits.size(); ++i) {
        result = result.add(cs, result, pre ? &hd[i] : nullptr);
        APoint rpb = result.add_const(cs, base, pre ? &ha[i] : nullptr);
        Num rx = mux(cs, Bool::is(bits[i]), Number::from(result.x), Number::from(rpb.x));
        Num ry = mux(cs, Bool::is(bits[i]), Number::from(result.y), Number::from(rpb.y));
        result = {rx, ry};
    }
    return result;
}

static inline APoint g_mul_cofactor(ConstraintSystem& cs, const APoint& p) {  // :238-247
    APoint q = p.add(cs, p);
    q = q.add(cs, q);
    q = q.add(cs, q);
    return q;
}

static inline void g_verify_eddsa(ConstraintSystem& cs, const Bool& enabled, const APoint& pk, const Number& msg,
                                  const APoint& sig_r, const Num& sig_s) {  // :249-280
    Num h = g_poseidon(cs, {Number::from(sig_r.x), Number::from(sig_r.y), Number::from(pk.x), Number::from(pk.y), msg}, true).compress(cs);
    APoint sb = g_base_mul(cs, jubjub_base_cofactor(), sig_s);
    APoint rpha = pk.mul(cs, h);
    rpha = rpha.add(cs, sig_r);
    rpha = g_mul_cofactor(cs, rpha);
    Number::from(rpha.x).assert_equal_if_enabled(cs, enabled, Number::from(sb.x));
    Number::from(rpha.y).assert_equal_if_enabled(cs, enabled, Number::from(sb.y));
}

}  // namespace bzk
