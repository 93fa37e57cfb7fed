// `groth16_verify` on the host (row a8): the check the reference's nodes run on every `ZkProof::Groth16` -
// /root/reference/src/zk/groth16/mod.rs:67-121: the five public inputs [commitment, height, state, aux_data, next_state] are handed
// to bellman's `verify_proof(prepare_verifying_key(vk), proof, inputs)` (third-party crate, bellman 0.14 / bls12_381 0.8):
//     e(A, B) = e(alpha, beta) * e(sum_i x_i IC_i, gamma) * e(C, delta),   x_0 = 1
// Verification stays a CPU job (milliseconds, three pairings); it lives in the product so that a proving worker can check
// what it is about to post with the work's own verifying key, and a Rust host gets the whole `src/zk::groth16` surface from one
// library.  The pairing itself is host_pairing.h (64-bit-limb host field, shared Miller accumulator with batched slope inversions,
// cyclotomic final exponentiation): ~3 ms per verification on one core where the first version (generic 32-bit-limb field, one
// inversion per line, a 2030-bit plain exponentiation) took 25 - 35 ms - a proving worker that checks its own proofs at 60 proofs/s
// spent two cores on it.  The verdicts are the same: checked against the oracle's Python verifier and, piece by piece, against the
// first version's forms kept in the header (tests/test_groth16_verify_cpu.py, tests/test_pairing_cpu.py).
#include <string.h>

#include <vector>

#include "bzk_curve.cuh"
#include "bzk_internal.h"
#include "host_pairing.h"

using namespace bzk;

namespace {

using hp::E2;
using hp::F1;
using hp::F2;
using hp::G1A;
using hp::G2A;

// Montgomery limbs of an Fp element are below p: anything else is not a value `Fp([u64; 6])` can legitimately hold, and arithmetic on
// it would leave the verdict to the reduction details of whichever library runs it
bool fp_in_range(const HFp& a) { return !hfp::geq_p(a.l, hfp::consts().p); }
HFp fp_four() {  // Montgomery form of 4
    const HFp two = F1::dbl(F1::one());
    return F1::dbl(two);
}
bool g1_unpack(const uint8_t* in, G1A& o) {  // packed 97 bytes; range + on-curve check
    o.inf = in[96] != 0;
    memcpy(o.x.l, in, 48);
    memcpy(o.y.l, in + 48, 48);
    if (o.inf) return true;
    if (!fp_in_range(o.x) || !fp_in_range(o.y)) return false;
    return F1::eq(F1::sqr(o.y), F1::add(F1::mul(F1::sqr(o.x), o.x), fp_four()));
}
bool g2_unpack(const uint8_t* in, G2A& o) {
    o.inf = in[192] != 0;
    memcpy(o.x.c0.l, in, 48); memcpy(o.x.c1.l, in + 48, 48); memcpy(o.y.c0.l, in + 96, 48); memcpy(o.y.c1.l, in + 144, 48);
    if (o.inf) return true;
    if (!fp_in_range(o.x.c0) || !fp_in_range(o.x.c1) || !fp_in_range(o.y.c0) || !fp_in_range(o.y.c1)) return false;
    const E2 b = {fp_four(), fp_four()};
    return F2::eq(F2::sqr(o.y), F2::add(F2::mul(F2::sqr(o.x), o.x), b));
}

}  // namespace

extern "C" {

// 1 = the proof verifies, 0 = it does not (incl. malformed / off-curve points, as the reference returns `false` on any error),
// negative = bad arguments.  vk = bincode(Groth16VerifyingKey) (870 + 8 + 97 n bytes, n = n_inputs + 1); inputs Montgomery.
int32_t bzk_groth16_verify(const uint8_t* vk, uint64_t vk_len, const uint8_t* inputs, uint32_t n_inputs, const uint8_t proof[387]) {
    if (!vk || !proof || (n_inputs && !inputs)) return BZK_E_ARG;
    if (vk_len < 878) return BZK_E_ARG;
    uint64_t n_ic;
    memcpy(&n_ic, vk + 870, 8);
    if (n_ic != (uint64_t)n_inputs + 1 || vk_len != 878 + 97 * n_ic) return 0;  // verify_proof: VerificationError::InvalidVerifyingKey
    G1A alpha, a, c, ic;
    G2A beta, gamma, delta, b;
    if (!g1_unpack(vk, alpha) || !g2_unpack(vk + 194, beta) || !g2_unpack(vk + 387, gamma) || !g2_unpack(vk + 677, delta)) return 0;
    if (!g1_unpack(proof, a) || !g2_unpack(proof + 97, b) || !g1_unpack(proof + 290, c)) return 0;
    // X = IC_0 + sum x_i IC_i: the scalars share their doublings (4-bit windows, most significant first: 252 doublings in all instead of
    // 255 per input) - for the five inputs of an MPN proof ~8 k field products instead of ~18 k
    typedef XyzzT<HFpOps> Pt;
    Pt acc = xyzz_identity<HFpOps>();
    std::vector<Pt> table((size_t)n_inputs * 15);   // table[15 i + j - 1] = j IC_(i + 1)
    std::vector<Fr> scalars(n_inputs);
    for (uint64_t i = 0; i < n_ic; ++i) {
        if (!g1_unpack(vk + 878 + 97 * i, ic)) return 0;
        if (i == 0) {
            if (!ic.inf) acc = xyzz_from_affine<HFpOps>({ic.x, ic.y});   // added at the end (the windows start from the identity)
            continue;
        }
        Fr x;
        memcpy(x.l, inputs + 32 * (i - 1), 32);
        {   // as for Fp above: Montgomery limbs at or above r are not a value a ZkScalar can hold - refused, not computed with
            Fr t = x;
            fe_reduce_once<FrParams>(t);
            if (!t.equals(x)) return 0;
        }
        scalars[i - 1] = fe_from_mont<FrParams>(x);
        Pt* t = &table[15 * (i - 1)];
        t[0] = ic.inf ? xyzz_identity<HFpOps>() : xyzz_from_affine<HFpOps>({ic.x, ic.y});
        for (int jx = 1; jx < 15; ++jx) {
            t[jx] = t[jx - 1];
            xyzz_add<HFpOps>(t[jx], t[0]);
        }
    }
    if (n_inputs) {
        Pt r = xyzz_identity<HFpOps>();
        for (int nib = 63; nib >= 0; --nib) {
            if (nib != 63)
                for (int d = 0; d < 4; ++d) r = xyzz_dbl<HFpOps>(r);
            for (uint32_t i = 0; i < n_inputs; ++i) {
                const uint32_t w = (scalars[i].l[nib >> 3] >> ((nib & 7) * 4)) & 15u;
                if (w) xyzz_add<HFpOps>(r, table[15 * i + w - 1]);
            }
        }
        xyzz_add<HFpOps>(acc, r);
    }
    AffineT<HFpOps> xa;
    G1A X;
    X.inf = !xyzz_to_affine<HFpOps>(acc, xa);
    X.x = xa.x;
    X.y = xa.y;
    // e(A, B) e(X, -gamma) e(C, -delta) e(-alpha, beta) == 1
    G1A ps[4] = {a, X, c, alpha};
    G2A qs[4] = {b, gamma, delta, beta};
    qs[1].y = F2::neg(gamma.y);
    qs[2].y = F2::neg(delta.y);
    ps[3].y = F1::neg(alpha.y);
    bool degenerate = false;
    const hp::E12 f = hp::multi_miller(ps, qs, 4, &degenerate);
    if (degenerate) return 0;
    return hp::e12_is_one(hp::final_exp(f)) ? 1 : 0;
}

}  // extern "C"
