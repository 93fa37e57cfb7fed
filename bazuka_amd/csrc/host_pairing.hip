// `groth16_verify` on the host (row a8): the check the reference's nodes run on every `ZkProof::Groth16` -
// /root/reference/src/zk/groth16/mod.rs:67-121: the five public inputs [commitment, height, state, aux_data, next_state] are handed
// to bellman's `verify_proof(prepare_verifying_key(vk), proof, inputs)` (third-party crate, bellman 0.14 / bls12_381 0.8):
//     e(A, B) = e(alpha, beta) * e(sum_i x_i IC_i, gamma) * e(C, delta),   x_0 = 1
// Verification stays a CPU job (milliseconds, three pairings); it lives in the product so that a proving worker can check
// what it is about to post with the work's own verifying key, and a Rust host gets the whole `src/zk::groth16` surface from one
// library.  Host C++ over the library's own Fp / Fp2 (64-bit-limb Montgomery product), nothing GPU:
//   tower   Fp6 = Fp2[v] / (v^3 - (1 + u)),  Fp12 = Fp6[w] / (w^2 - v)
//   Miller  optimal ate over |x| = 0xd201000000010000 (x < 0: conjugate at the end), M-type twist, affine line functions; the
//           four pairs (A, B), (X, -gamma), (C, -delta), (-alpha, beta) share one accumulator (one squaring per step)
//   final   f^(p^6 - 1) by conjugate / inverse, then the rest as ONE plain exponentiation by (p^6 + 1) / r (2030 bits): ~11 ms - a
//           verifier that favours being obviously right over being fast (the whole check is ~20 ms on one core)
// Checked against the oracle's Python verifier (tests/test_groth16_verify_cpu.py) on oracle-made proofs: accepts, rejects a wrong
// input / a tampered proof / an off-curve point, same verdict as the oracle on every case.
#include <string.h>

#include "bzk_curve.cuh"
#include "bzk_internal.h"

using namespace bzk;

namespace {

typedef Fp2Ops F2;
typedef Fp2 E2;
struct E6 { E2 c0, c1, c2; };
struct E12 { E6 a0, a1; };

inline E2 e2_mul_xi(const E2& a) { return {fe_sub<FpParams>(a.c0, a.c1), fe_add<FpParams>(a.c0, a.c1)}; }  // * (1 + u)
inline E2 e2_scale(const E2& a, const Fp& k) { return {fe_mul<FpParams>(a.c0, k), fe_mul<FpParams>(a.c1, k)}; }
inline E6 e6_zero() { return {F2::zero(), F2::zero(), F2::zero()}; }
inline E6 e6_one() { return {F2::one(), F2::zero(), F2::zero()}; }
inline E6 e6_add(const E6& a, const E6& b) { return {F2::add(a.c0, b.c0), F2::add(a.c1, b.c1), F2::add(a.c2, b.c2)}; }
inline E6 e6_sub(const E6& a, const E6& b) { return {F2::sub(a.c0, b.c0), F2::sub(a.c1, b.c1), F2::sub(a.c2, b.c2)}; }
inline E6 e6_neg(const E6& a) { return {F2::neg(a.c0), F2::neg(a.c1), F2::neg(a.c2)}; }
E6 e6_mul(const E6& a, const E6& b) {
    const E2 t0 = F2::mul(a.c0, b.c0), t1 = F2::mul(a.c1, b.c1), t2 = F2::mul(a.c2, b.c2);
    E6 r;
    r.c0 = F2::add(t0, e2_mul_xi(F2::add(F2::mul(a.c1, b.c2), F2::mul(a.c2, b.c1))));
    r.c1 = F2::add(F2::add(F2::mul(a.c0, b.c1), F2::mul(a.c1, b.c0)), e2_mul_xi(t2));
    r.c2 = F2::add(F2::add(F2::mul(a.c0, b.c2), F2::mul(a.c2, b.c0)), t1);
    return r;
}
inline E6 e6_mul_v(const E6& a) { return {e2_mul_xi(a.c2), a.c0, a.c1}; }
E6 e6_inv(const E6& a) {
    const E2 c0 = F2::sub(F2::sqr(a.c0), e2_mul_xi(F2::mul(a.c1, a.c2)));
    const E2 c1 = F2::sub(e2_mul_xi(F2::sqr(a.c2)), F2::mul(a.c0, a.c1));
    const E2 c2 = F2::sub(F2::sqr(a.c1), F2::mul(a.c0, a.c2));
    const E2 t = F2::add(F2::mul(a.c0, c0), e2_mul_xi(F2::add(F2::mul(a.c2, c1), F2::mul(a.c1, c2))));
    const E2 ti = F2::inv(t);
    return {F2::mul(c0, ti), F2::mul(c1, ti), F2::mul(c2, ti)};
}
inline E12 e12_one() { return {e6_one(), e6_zero()}; }
E12 e12_mul(const E12& a, const E12& b) {
    const E6 t0 = e6_mul(a.a0, b.a0), t1 = e6_mul(a.a1, b.a1);
    E12 r;
    r.a0 = e6_add(t0, e6_mul_v(t1));
    r.a1 = e6_sub(e6_sub(e6_mul(e6_add(a.a0, a.a1), e6_add(b.a0, b.a1)), t0), t1);
    return r;
}
inline E12 e12_conj(const E12& a) { return {a.a0, e6_neg(a.a1)}; }
E12 e12_inv(const E12& a) {
    const E6 t = e6_inv(e6_sub(e6_mul(a.a0, a.a0), e6_mul_v(e6_mul(a.a1, a.a1))));
    return {e6_mul(a.a0, t), e6_neg(e6_mul(a.a1, t))};
}
bool e12_is_one(const E12& a) {
    return a.a0.c0.c0.equals(Fp::one()) && a.a0.c0.c1.is_zero() && F2::is_zero(a.a0.c1) && F2::is_zero(a.a0.c2) && F2::is_zero(a.a1.c0) &&
           F2::is_zero(a.a1.c1) && F2::is_zero(a.a1.c2);
}

// (p^6 + 1) / r, little-endian 32-bit words
const uint32_t FINAL_EXP[64] = {
    0xc0705d6au, 0x8739e1cdu, 0xe0381a16u, 0x09a5256du, 0x61c791e2u, 0x9cf0f70au, 0x7903f76eu, 0x3a09c449u, 0x3890f133u, 0x2d727156u,
    0x6fec7760u, 0x224741b3u, 0x2a12bd40u, 0x338259c2u, 0x778e0de7u, 0x38ee1cd4u, 0x188a20b0u, 0xc3b5ef4bu, 0xe2764d7bu, 0x1d615d49u,
    0xd076117du, 0x816101ddu, 0x7ebe3afcu, 0xf007c01eu, 0x935021c3u, 0x27d7bd90u, 0x57c0b15fu, 0xc3b5e2f5u, 0xc4f82384u, 0x5e886c94u,
    0x11e63f56u, 0xee6a95dbu, 0x4a9c4f6fu, 0x2b822f51u, 0xd21b73dau, 0x12d6a874u, 0xf499dffbu, 0x1304275eu, 0xbcb95d1fu, 0x967878feu,
    0x8b2f2922u, 0x4744497fu, 0xf0841855u, 0x85a2e707u, 0x6c802eecu, 0x9f0c5012u, 0xbd2fa489u, 0xfb46e197u, 0x9bc5f61au, 0x548ce080u,
    0x73beaa8cu, 0xcf56fb15u, 0x763bdf7cu, 0xad7375a3u, 0x179bdeccu, 0xe0ec9031u, 0x3c48c1dau, 0x6579aea8u, 0x64cf5bb3u, 0xdbf85ae6u,
    0x55ca7566u, 0x7b6f235cu, 0x14877503u, 0x000028b3u};

E12 final_exp(const E12& f) {
    const E12 g = e12_mul(e12_conj(f), e12_inv(f));  // f^(p^6 - 1)
    E12 r = e12_one();
    for (int i = 2029; i >= 0; --i) {
        r = e12_mul(r, r);
        if ((FINAL_EXP[i >> 5] >> (i & 31)) & 1) r = e12_mul(r, g);
    }
    return r;
}

struct G1A { Fp x, y; bool inf; };
struct G2A { E2 x, y; bool inf; };

// line through T (twist coordinates, slope lam) evaluated at P, up to a factor the final exponentiation kills:
// (lam xT - yT) + (-lam xP) w^2 + yP w^3
E12 line(const E2& lam, const G2A& t, const G1A& p) {
    E12 l;
    l.a0 = {F2::sub(F2::mul(lam, t.x), t.y), e2_scale(F2::neg(lam), p.x), F2::zero()};
    l.a1 = {F2::zero(), {p.y, Fp::zero()}, F2::zero()};
    return l;
}
G2A g2_dbl(const G2A& t, const E2& lam) {
    const E2 x3 = F2::sub(F2::sqr(lam), F2::add(t.x, t.x));
    return {x3, F2::sub(F2::mul(lam, F2::sub(t.x, x3)), t.y), false};
}
G2A g2_add(const G2A& t, const G2A& q, const E2& lam) {
    const E2 x3 = F2::sub(F2::sub(F2::sqr(lam), t.x), q.x);
    return {x3, F2::sub(F2::mul(lam, F2::sub(t.x, x3)), t.y), false};
}

// product of the Miller functions of n pairs (pairs with an identity member contribute 1).  *degenerate is set when a line's slope has
// a zero denominator (T of order 2, or T = +-Q): impossible for points of the prime-order subgroup, reachable only with low-order G2
// points, which the callers only check to be on the curve (as the reference does - it transmutes unchecked points into bellman's
// projective Miller loop).  Such a proof cannot satisfy the pairing equation; the verdict is then "does not verify" instead of a value
// computed from 1 / 0.
E12 multi_miller(const G1A* p, const G2A* q, int n, bool* degenerate) {
    const uint64_t X = 0xd201000000010000ull;
    G2A t[4];
    bool live[4];
    for (int k = 0; k < n; ++k) { t[k] = q[k]; live[k] = !p[k].inf && !q[k].inf; }
    E12 f = e12_one();
    for (int i = 62; i >= 0; --i) {  // bit 63 is the leading one
        f = e12_mul(f, f);
        for (int k = 0; k < n; ++k) {
            if (!live[k]) continue;
            const E2 xx = F2::sqr(t[k].x);
            const E2 den = F2::add(t[k].y, t[k].y);
            if (F2::is_zero(den)) { *degenerate = true; return e12_one(); }
            const E2 lam = F2::mul(F2::add(F2::add(xx, xx), xx), F2::inv(den));
            f = e12_mul(f, line(lam, t[k], p[k]));
            t[k] = g2_dbl(t[k], lam);
        }
        if ((X >> i) & 1) {
            for (int k = 0; k < n; ++k) {
                if (!live[k]) continue;
                const E2 den = F2::sub(q[k].x, t[k].x);
                if (F2::is_zero(den)) { *degenerate = true; return e12_one(); }
                const E2 lam = F2::mul(F2::sub(q[k].y, t[k].y), F2::inv(den));
                f = e12_mul(f, line(lam, t[k], p[k]));
                t[k] = g2_add(t[k], q[k], lam);
            }
        }
    }
    return e12_conj(f);  // the curve parameter is -|x|
}

// Montgomery limbs of an Fp element are below p: anything else is not a value `Fp([u64; 6])` can legitimately hold, and arithmetic on
// it would leave the verdict to the reduction details of whichever library runs it
bool fp_in_range(const Fp& a) {
    Fp t = a;
    fe_reduce_once<FpParams>(t);
    return t.equals(a);
}
bool g1_unpack(const uint8_t* in, G1A& o) {  // packed 97 bytes; range + on-curve check
    o.inf = in[96] != 0;
    memcpy(o.x.l, in, 48);
    memcpy(o.y.l, in + 48, 48);
    if (o.inf) return true;
    if (!fp_in_range(o.x) || !fp_in_range(o.y)) return false;
    Fp four = Fp::zero();
    four.l[0] = 4;
    four = fe_to_mont<FpParams>(four);
    return fe_sqr<FpParams>(o.y).equals(fe_add<FpParams>(fe_mul<FpParams>(fe_sqr<FpParams>(o.x), o.x), four));
}
bool g2_unpack(const uint8_t* in, G2A& o) {
    o.inf = in[192] != 0;
    memcpy(o.x.c0.l, in, 48); memcpy(o.x.c1.l, in + 48, 48); memcpy(o.y.c0.l, in + 96, 48); memcpy(o.y.c1.l, in + 144, 48);
    if (o.inf) return true;
    if (!fp_in_range(o.x.c0) || !fp_in_range(o.x.c1) || !fp_in_range(o.y.c0) || !fp_in_range(o.y.c1)) return false;
    Fp four = Fp::zero();
    four.l[0] = 4;
    four = fe_to_mont<FpParams>(four);
    const E2 b = {four, four};
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
    // X = IC_0 + sum x_i IC_i
    G1Xyzz acc = xyzz_identity<FpOps>();
    for (uint64_t i = 0; i < n_ic; ++i) {
        if (!g1_unpack(vk + 878 + 97 * i, ic)) return 0;
        if (ic.inf) continue;
        G1Xyzz term = xyzz_from_affine<FpOps>({ic.x, ic.y});
        if (i > 0) {
            Fr x;
            memcpy(x.l, inputs + 32 * (i - 1), 32);
            const Fr xc = fe_from_mont<FrParams>(x);
            G1Xyzz r = xyzz_identity<FpOps>();
            for (int bit = 254; bit >= 0; --bit) {
                r = xyzz_dbl<FpOps>(r);
                if ((xc.l[bit >> 5] >> (bit & 31)) & 1) xyzz_add<FpOps>(r, term);
            }
            term = r;
        }
        xyzz_add<FpOps>(acc, term);
    }
    G1Affine xa;
    G1A X;
    X.inf = !xyzz_to_affine<FpOps>(acc, xa);
    X.x = xa.x;
    X.y = xa.y;
    // e(A, B) e(X, -gamma) e(C, -delta) e(-alpha, beta) == 1
    G1A ps[4] = {a, X, c, alpha};
    G2A qs[4] = {b, gamma, delta, beta};
    qs[1].y = F2::neg(gamma.y);
    qs[2].y = F2::neg(delta.y);
    ps[3].y = fe_neg<FpParams>(alpha.y);
    bool degenerate = false;
    const E12 f = multi_miller(ps, qs, 4, &degenerate);
    if (degenerate) return 0;
    return e12_is_one(final_exp(f)) ? 1 : 0;
}

}  // extern "C"
