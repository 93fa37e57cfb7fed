// Witness value traces on the device (round 5): the part of an MPN circuit's assignment that hangs off Poseidon
// outputs - the Poseidon gadget's own S-box / idle-lane variables (2/3 of a transition's constraints), the Merkle gadget's muxes,
// the equality checks against computed roots - is not evaluated by the host generator but DEFERRED: the gadgets of host_r1cs.h emit
// a small program per transition instead (the same for every transition of a circuit shape) plus a record of the values the host does
// know, leave the slots of z / A.z / B.z / C.z untouched, and the device fills them in after the arrays have been uploaded.
//
//   reference gadgets whose VALUE semantics the ops restate (allocation order and the three values per constraint as the host forms do):
//     Poseidon gadget      /root/reference/src/zk/groth16/gadgets/poseidon/mod.rs:8-95     F_POSEIDON (trace), V_HASH (output only)
//     mux                  /root/reference/src/zk/groth16/gadgets/common/mux.rs:7-47       F_MUX, V_SEL
//     assert_equal(_if_enabled)  .../common/number.rs:121-177                               F_ENFORCE_EQ, F_ASSERT_EQ_IF
//
// Two passes.  Pass 1 (V ops, by dependency level: a Merkle path is a chain of hashes): registers = the deferred VALUES, one 32-byte
// Montgomery-256 scalar per (register, transition), computed with the sparse-partial-round hash of bzk_poseidon29.cuh.  Pass 2 (F ops, all
// independent once the registers are known: one lane per (op, transition)): the slots.  The dense intermediate states the gadget allocates are
// only produced here, in pass 2 - the sparse form skips them.  Everything is BZK_HD: the CPU harness and bzk_r1cs_fill_host run the same code.
#pragma once
#include "bzk_fr29.cuh"
#include "bzk_poseidon29.cuh"

namespace bzk {
namespace wf {

enum : uint8_t { V_HASH = 1, V_SEL = 2, F_POSEIDON = 3, F_MUX = 4, F_ASSERT_EQ_IF = 5, F_ENFORCE_EQ = 6, F_CHECK_EQ = 7 };
constexpr int MAX_SEL_CHAIN = 8;  // links operand() follows on the device; DeferProgram::finalize measures a program's, witfill_run_dev refuses deeper ones
enum : uint32_t { FLAG_UNSATISFIED = 1u, FLAG_CHAIN = 2u };  // a deferred constraint does not hold / a computed state differs from the builder's prediction

// operand: >= 0 a register, < 0 ~index into the transition's input record
struct Op {
    uint8_t kind;
    uint8_t t;        // V_HASH / F_POSEIDON: state width (arity + 1); V_SEL / F_MUX: 0 = Boolean::Is, 1 = Boolean::Not
    uint16_t level;   // V_HASH: 1 + the highest level among the producers of its register operands; V_SEL: that highest level itself
    int32_t out;      // V ops: the register written; F ops: unused (-1)
    uint32_t aux_off, con_off;  // F ops: first variable / constraint slot inside the transition's window
    int32_t in[7];
};

// counts of the Poseidon gadget for width t (arity t - 1): three variables and constraints per S-box, one per idle lane of a partial round
BZK_HD uint32_t poseidon_slots(int t, int rf, int rp) { return (uint32_t)(rf * t * 3 + rp * (3 + (t - 1))); }

struct Arrays {       // device (or host) views of the assignment; the aux part of z starts at z_aux
    Fr* z_aux;
    Fr* az;
    Fr* bz;
    Fr* cz;
};
struct TxView {       // where transition `tx` lives
    const Fr* inputs; // its input record
    Fr* regs;         // regs[r * reg_stride]
    size_t reg_stride;
    size_t aux_base, con_base;
    // device: per register {select input, a, b, Is / Not} of the V_SEL that defines it, or {.., .., .., -1}: selections are then never
    // materialised - an operand that names one is resolved here, through at most a few links (a Merkle level is two deep), which
    // takes the ~35 one-select launches out of pass 1.  nullptr (host executor): every V_SEL has been executed into its register
    const int32_t* sel;
};

BZK_HD Fr operand(const TxView& v, int32_t o) {
    if (v.sel) {
        for (int it = 0; it < MAX_SEL_CHAIN && o >= 0; ++it) {
            const int32_t* s = v.sel + 4 * (size_t)o;
            if (s[3] < 0) break;
            const bool bit = !v.inputs[~s[0]].is_zero();
            o = (s[3] == 0 ? bit : !bit) ? s[2] : s[1];
        }
    }
    return o >= 0 ? v.regs[(size_t)o * v.reg_stride] : v.inputs[~o];
}
BZK_HD Fr fr_one_mont() { return Fr::one(); }

// ---- pass 1 -------------------------------------------------------------------------------------------------------------------
template <int T>
BZK_HD void v_hash(const Op& op, const TxView& v, const Fr29* __restrict__ sparse_consts, int rf, int rp) {
    Fr in[T - 1];
#pragma unroll
    for (int i = 0; i < T - 1; ++i) in[i] = operand(v, op.in[i]);
    v.regs[(size_t)op.out * v.reg_stride] = poseidon29_hash<T>(in, sparse_consts, rf, rp);
}
// in[0] = the select bit's VALUE (an input: 0 / 1 in Montgomery form), in[1] = a, in[2] = b.   Is: bit ? b : a.   Not: bit ? a : b
BZK_HD void v_sel(const Op& op, const TxView& v) {
    const bool bit = !operand(v, op.in[0]).is_zero();
    const bool take_b = op.t == 0 ? bit : !bit;
    v.regs[(size_t)op.out * v.reg_stride] = operand(v, take_b ? op.in[2] : op.in[1]);
}

// ---- pass 2 -------------------------------------------------------------------------------------------------------------------
// dense constants of width T in the 29-bit form: rc[(rf + rp) * T] | mds[T * T]
template <int T>
BZK_HD void f_poseidon(const Op& op, const TxView& v, const Arrays& A, const Fr29* __restrict__ dense, int rf, int rp) {
    Fr29 e[T];
    e[0] = fr29::zero();
#pragma unroll
    for (int i = 1; i < T; ++i) e[i] = fr29::to29(operand(v, op.in[i - 1]));
    size_t a = v.aux_base + op.aux_off, c = v.con_base + op.con_off;
    const Fr one = fr_one_mont();
    const Fr29* mds = dense + (size_t)(rf + rp) * T;
    auto sbox = [&](const Fr29& x) {  // x2 = x x, x4 = x2 x2, x5 = x x4: three variables, three constraints
        const Fr29 x2 = fr29::sqr(x), x4 = fr29::sqr(x2), x5 = fr29::mul(x, x4);
        const Fr X = fr29::from29(x), X2 = fr29::from29(x2), X4 = fr29::from29(x4), X5 = fr29::from29(x5);
        A.z_aux[a] = X2; A.z_aux[a + 1] = X4; A.z_aux[a + 2] = X5;
        A.az[c] = X;  A.bz[c] = X;  A.cz[c] = X2;
        A.az[c + 1] = X2; A.bz[c + 1] = X2; A.cz[c + 1] = X4;
        A.az[c + 2] = X;  A.bz[c + 2] = X4; A.cz[c + 2] = X5;
        a += 3; c += 3;
        return x5;
    };
#pragma unroll 1
    for (int rnd = 0; rnd < rf + rp; ++rnd) {
        const Fr29* rc = dense + (size_t)rnd * T;
#pragma unroll
        for (int i = 0; i < T; ++i) e[i] = fr29::norm(fr29::add(e[i], rc[i]));  // k <= 3
        const bool full = rnd < rf / 2 || rnd >= rf / 2 + rp;
        if (full) {
#pragma unroll
            for (int i = 0; i < T; ++i) e[i] = sbox(e[i]);
        } else {
            e[0] = sbox(e[0]);
#pragma unroll
            for (int i = 1; i < T; ++i) {  // Number::compress: v * 1 = v
                const Fr V = fr29::from29(e[i]);
                A.z_aux[a] = V;
                A.az[c] = V; A.bz[c] = one; A.cz[c] = V;
                ++a; ++c;
            }
        }
        Fr29 nw[T];
#pragma unroll
        for (int j = 0; j < T; ++j) nw[j] = p29::row_dot<T>(mds + j * T, e);
#pragma unroll
        for (int j = 0; j < T; ++j) e[j] = nw[j];
    }
}
// select ? b : a, one variable, one constraint.  Is:  (a - b) * bit = a - ret.   Not: ret = bit ? a : b,  (b - a) * bit = b - ret
BZK_HD void f_mux(const Op& op, const TxView& v, const Arrays& A) {
    const Fr s = operand(v, op.in[0]), x = operand(v, op.in[1]), y = operand(v, op.in[2]);
    const bool bit = !s.is_zero();
    const size_t a = v.aux_base + op.aux_off, c = v.con_base + op.con_off;
    if (op.t == 0) {
        const Fr ret = bit ? y : x;
        A.z_aux[a] = ret;
        A.az[c] = fe_sub<FrParams>(x, y); A.bz[c] = s; A.cz[c] = fe_sub<FrParams>(x, ret);
    } else {
        const Fr ret = bit ? x : y;
        A.z_aux[a] = ret;
        A.az[c] = fe_sub<FrParams>(y, x); A.bz[c] = s; A.cz[c] = fe_sub<FrParams>(y, ret);
    }
}
// in[0] = enabled (bit value), in[1] = this, in[2] = other:  eis = enabled ? this : 0;  enabled * this = eis;  enabled * other = eis
BZK_HD uint32_t f_assert_eq_if(const Op& op, const TxView& v, const Arrays& A) {
    const Fr en = operand(v, op.in[0]), x = operand(v, op.in[1]), y = operand(v, op.in[2]);
    const bool on = !en.is_zero();
    const Fr ev = on ? x : Fr::zero();
    const size_t a = v.aux_base + op.aux_off, c = v.con_base + op.con_off;
    A.z_aux[a] = ev;
    A.az[c] = en; A.bz[c] = x; A.cz[c] = ev;
    A.az[c + 1] = en; A.bz[c + 1] = y; A.cz[c + 1] = ev;
    return on && !x.equals(y) ? FLAG_UNSATISFIED : 0u;
}
// this * 1 = other
BZK_HD uint32_t f_enforce_eq(const Op& op, const TxView& v, const Arrays& A) {
    const Fr x = operand(v, op.in[0]), y = operand(v, op.in[1]);
    const size_t c = v.con_base + op.con_off;
    A.az[c] = x; A.bz[c] = fr_one_mont(); A.cz[c] = y;
    return x.equals(y) ? 0u : FLAG_UNSATISFIED;
}
// no slot: the state a transition computes against the state the witness builder predicted for it
BZK_HD uint32_t f_check_eq(const Op& op, const TxView& v) { return operand(v, op.in[0]).equals(operand(v, op.in[1])) ? 0u : FLAG_CHAIN; }

}  // namespace wf
}  // namespace bzk
