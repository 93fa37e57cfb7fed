// G2 point arithmetic on PAIRS of lanes (round 5): lane 2k holds the c0 component of every Fp2 value of a point, lane 2k + 1 the c1
// component; the partner's component crosses with one DPP quad_perm move per limb (VALU, no LDS).
//
// Why: a G2 point is 8 base-field elements = 112 registers, a mixed addition's working set ~ 400: the one-lane accumulation runs at ONE
// wave per SIMD (512 registers, still spilling), where a lone wave can neither hide the 224-byte gather of the next base (17 % of its
// cycles wait, more once the base set outgrows the 256 MB Infinity Cache: endomorphism form 9.0 -> 11.0 ms, profiles/r05_run1...) nor reach
// the multiplier's rate at two waves (60.1 vs 70.2 G products/s, profiles/r04_ubench_int.txt).  Split by component the state halves
// (a point = 56 registers per lane), the kernel fits two waves per SIMD with every product inlined - no call ABI, no wait-for-all-loads
// at a callee's entry, so the next base's loads fly during the whole addition - and the multiply-add count does not change:
//     Fp2 product   even lane  a0 b0 + a1 (K p - b1)       odd lane  a1 b0 + a0 b1       2 x 196 + 196 each = the 1176 of Karatsuba's 3 x 392
//     Fp2 square    even lane  (a0 + a1)(a0 + K p - a1)    odd lane  (2 a1) a0            392 each           = the 784 of the one-lane form
//     Y3 = R T - Y PPP: one four-product sum with one reduction per lane (980 each = the 1960 of the two mul4_body of g2x28::add_mixed)
// and every product output is a plain Montgomery output (normalised, < 2 p) - tighter than the Karatsuba form's (5 p, 8 p).  Extra per
// addition and lane: ~14 x 14 DPP moves and ~12 x 14 selects beside 5 292 multiply-adds.
//
// Discipline of a stored point (per component): X normalised and < 12 p, Y normalised and < 3 p, ZZ / ZZZ product outputs (< 2 p);
// identity <=> every ZZ limb of BOTH lanes zero.  Affine bases: components normalised and < 8 p (converted bases are product outputs,
// group sums and table entries come out of the one-lane Fp2 products: c0 < 5 p, c1 < 8 p).  Branches that contain an exchange are taken by
// both lanes of a pair together: their conditions are computed from exchanged values.
//
// The same source runs on the CPU (tests/host/hostcheck.hip: two threads per pair, the exchange a rendezvous) with the field's bound
// assertions on, against the oracle - the limb logic is checked in the GPU-less container, the DPP moves on the GPU box.
#pragma once
#include "bzk_fp28.cuh"

namespace bzk {
namespace g2p {
using namespace fp28;

// ---- the exchange: the partner lane's value (quad_perm [1, 0, 3, 2]).  Device-only code; with BZK_G2P_HOST_EMU the same functions are
// host functions whose exchange is provided by the test harness (tests/host/hostcheck.hip: two threads per pair, a rendezvous)
#if defined(BZK_G2P_HOST_EMU)
#define G2P_FN inline
uint32_t bzk_g2p_host_swp32(uint32_t v);
bool bzk_g2p_host_lane_odd();
inline uint32_t swp32(uint32_t v) { return bzk_g2p_host_swp32(v); }
inline bool lane_odd() { return bzk_g2p_host_lane_odd(); }
#else
#define G2P_FN __device__ __forceinline__
__device__ __forceinline__ uint32_t swp32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ bool lane_odd() { return (threadIdx.x & 1u) != 0; }
#endif
G2P_FN Fp28 swp(const Fp28& v) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = swp32(v.l[i]);
    return r;
}
BZK_HD Fp28 sel(bool odd, const Fp28& if_even, const Fp28& if_odd) {
    Fp28 r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.l[i] = odd ? if_odd.l[i] : if_even.l[i];
    return r;
}

// ---- a0 b0 + a1 b1 with ONE Montgomery reduction (the columns only grow).  Needs 14 (2^(La0+Lb0) + 2^(La1+Lb1) + 2^56) < 2^64 and
// ka0 kb0 + ka1 kb1 <= 2048; gives a product output (normalised, < 2 p)
BZK_HD Fp28 mul2_body(const Fp28& a0, const Fp28& b0, const Fp28& a1, const Fp28& b1) {
#if defined(BZK_FP28_CHECK) && !defined(__HIP_DEVICE_COMPILE__)
    {
        const Fp28* o[4] = {&a0, &b0, &a1, &b1};
        unsigned __int128 worst = (unsigned __int128)14 * MASK * MASK + ((unsigned __int128)1 << 40);
        for (int t = 0; t < 2; ++t) {
            uint32_t ma = 0, mb = 0;
            for (int i = 0; i < N; ++i) {
                if (o[2 * t]->l[i] > ma) ma = o[2 * t]->l[i];
                if (o[2 * t + 1]->l[i] > mb) mb = o[2 * t + 1]->l[i];
            }
            worst += (unsigned __int128)14 * ma * mb;
        }
        assert(worst < ((unsigned __int128)1 << 64));
    }
#endif
    uint64_t c[2 * N];
#pragma unroll
    for (int k = 0; k < 2 * N; ++k) c[k] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a0.l[i] * b0.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)a1.l[i] * b1.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t m = mont_m((uint32_t)c[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) c[i + j] += (uint64_t)m * P.v[j];
        c[i + 1] += c[i] >> W;
    }
    Fp28 r;
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
        c[k + 1] += c[k] >> W;
        r.l[k - N] = (uint32_t)c[k] & MASK;
    }
    r.l[N - 1] = (uint32_t)c[2 * N - 1];
    return r;
}

// ---- this lane's component of a b.  b_ot normalised with value < K p; needs ka (kb + K) <= 2048
template <int K>
G2P_FN Fp28 mul(const Fp28& a_me, const Fp28& a_ot, const Fp28& b_me, const Fp28& b_ot) {
    const bool odd = lane_odd();
    const Fp28 nb = sub<K>(zero(), b_ot);  // K p - b_ot
    return mul2_body(a_me, sel(odd, b_me, b_ot), a_ot, sel(odd, nb, b_me));
}
// ---- this lane's component of a^2.  a normalised with value < K p; needs 2 ka (ka + K) <= 2048
template <int K>
G2P_FN Fp28 sqr(const Fp28& a_me, const Fp28& a_ot) {
    const bool odd = lane_odd();
    const Fp28 x = add(a_me, sel(odd, a_ot, a_me));          // a0 + a1 | 2 a1
    const Fp28 y = sel(odd, sub<K>(a_me, a_ot), a_ot);       // a0 + K p - a1 | a0
    return mul_body(x, y);
}
// is this pair's Fp2 value (product outputs in both lanes) zero ?
G2P_FN bool pair_mulout_is_zero(const Fp28& me) {
    const uint32_t z = mulout_is_zero(me) ? 1u : 0u;
    return (z & swp32(z)) != 0;
}

struct Aff {
    Fp28 x, y;  // this lane's component of the affine coordinates
};
struct Pt {
    Fp28 X, Y, ZZ, ZZZ;  // this lane's component of each coordinate
};

G2P_FN Fp28 one_me() { return lane_odd() ? zero() : one(); }  // Fp2 one = (1, 0)
G2P_FN Pt identity() { return {zero(), one_me(), zero(), zero()}; }
G2P_FN bool is_identity(const Pt& p) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) z |= p.ZZ.l[i];
    return (z | swp32(z)) == 0;
}

// 2 q for an affine q (mdbl-2008-s-1); the rare branch of add_mixed
G2P_FN Pt dbl_affine(const Aff& q) {
    const Fp28 x = reduce(q.x), y = reduce(q.y);  // < 3 p
    const Fp28 xo = swp(x), yo = swp(y);
    const Fp28 Un = norm(add(y, y)), Uon = norm(add(yo, yo));  // k 6
    const Fp28 V = sqr<12>(Un, Uon);               // (12)(6 + 12)
    const Fp28 Vo = swp(V);
    const Fp28 Wv = mul<3>(Un, Uon, V, Vo);        // 6 (2 + 3)
    const Fp28 S = mul<3>(x, xo, V, Vo);           // 3 (2 + 3)
    const Fp28 xx = sqr<6>(x, xo);                 // (6)(3 + 6)
    const Fp28 M = norm(add(add(xx, xx), xx));     // k 6
    const Fp28 Mo = swp(M);
    const Fp28 MM = sqr<12>(M, Mo);                // (12)(6 + 12)
    Pt r;
    r.X = norm(sub<3>(sub<3>(MM, S), S));          // k 8
    const Fp28 T = norm(sub<12>(S, r.X));          // k 14
    const Fp28 To = swp(T);
    const Fp28 m1 = mul<24>(M, Mo, T, To);         // 6 (14 + 24)
    const Fp28 Wo = swp(Wv);
    const Fp28 m2 = mul<6>(Wv, Wo, y, yo);         // 2 (3 + 6)
    r.Y = reduce(sub<3>(m1, m2));                  // < 3 p
    r.ZZ = V;
    r.ZZZ = Wv;
    return r;
}

struct NoPre {
    BZK_HD void operator()() const {}
};
// acc += q (q affine, never the identity; neg_q: add -q).  `pre` is called exactly once, before the formula's tail: the accumulation
// issues the loads of the next base there (nothing in this function waits for them: every product is inlined).
template <class Pre = NoPre>
G2P_FN void add_mixed(Pt& acc, const Aff& q_in, bool neg_q, Pre&& pre = Pre()) {
    const bool odd = lane_odd();
    Aff q = q_in;
    if (neg_q) q.y = norm(sub<12>(zero(), q.y));  // 12 p - y
    if (is_identity(acc)) {
        pre();
        acc.X = reduce(q.x);  // < 3 p: once per run
        acc.Y = reduce(q.y);
        acc.ZZ = acc.ZZZ = one_me();
        return;
    }
    const Fp28 ZZo = swp(acc.ZZ), ZZZo = swp(acc.ZZZ);
    const Fp28 U2 = mul<3>(q.x, swp(q.x), acc.ZZ, ZZo);    // 8 (2 + 3)
    const Fp28 S2 = mul<3>(q.y, swp(q.y), acc.ZZZ, ZZZo);  // 12 (2 + 3)
    const Fp28 Pp = norm(sub<12>(U2, acc.X));              // k 14
    const Fp28 R = norm(sub<6>(S2, acc.Y));                // k 8
    const Fp28 Ppo = swp(Pp), Ro = swp(R);
    const Fp28 PP = sqr<24>(Pp, Ppo);                      // (28)(14 + 24)
    if (pair_mulout_is_zero(PP)) {  // same x: doubling or cancellation (rare).  PP = Pp^2 = 0 in Fp2 <=> Pp = 0
        const Fp28 RR = sqr<12>(R, Ro);
        if (pair_mulout_is_zero(RR)) acc = dbl_affine(q);
        else acc = identity();
        pre();
        return;
    }
    const Fp28 PPo = swp(PP);
    const Fp28 PPP = mul<3>(Pp, Ppo, PP, PPo);             // 14 (2 + 3)
    const Fp28 Q = mul<3>(acc.X, swp(acc.X), PP, PPo);     // 12 (2 + 3)
    const Fp28 RR = sqr<12>(R, Ro);                        // (16)(8 + 12)
    const Fp28 X3 = norm(sub<3>(sub<3>(sub<3>(RR, PPP), Q), Q));  // k 11
    const Fp28 PPPo = swp(PPP);
    acc.ZZ = mul<3>(acc.ZZ, ZZo, PP, PPo);
    acc.ZZZ = mul<3>(acc.ZZZ, ZZZo, PPP, PPPo);
    pre();
    // Y3 = R T - Y PPP, T = Q - X3:   even  R0 T0 + R1 (24 p - T1) + (3 p - Y0) PPP0 + Y1 PPP1
    //                                 odd   R1 T0 + R0 T1 + (3 p - Y1) PPP0 + (3 p - Y0) PPP1
    const Fp28 T = norm(sub<12>(Q, X3));                   // k 14
    const Fp28 To = swp(T);
    const Fp28 nY = sub<3>(zero(), acc.Y);                 // Y < 3 p
    const Fp28 Yo = swp(acc.Y), nYo = swp(nY);
    const Fp28 nTo = sub<24>(zero(), To);
    acc.Y = mul4_body(R, sel(odd, T, To), Ro, sel(odd, nTo, T), nY, sel(odd, PPP, PPPo), sel(odd, Yo, nYo), sel(odd, PPPo, PPP));
    acc.X = X3;
}

// 2 p (dbl-2008-s-1).  Input / output: the stored-point discipline of the header comment
G2P_FN Pt dbl(const Pt& p) {
    if (is_identity(p)) return p;
    const Fp28 U = norm(add(p.Y, p.Y));            // k 6
    const Fp28 Uo = swp(U);
    const Fp28 V = sqr<12>(U, Uo);
    const Fp28 Vo = swp(V);
    const Fp28 Wv = mul<3>(U, Uo, V, Vo);
    const Fp28 Xo = swp(p.X);
    const Fp28 S = mul<3>(p.X, Xo, V, Vo);         // 12 (2 + 3)
    const Fp28 xx = sqr<12>(p.X, Xo);              // (24)(12 + 12)
    const Fp28 M = norm(add(add(xx, xx), xx));     // k 6
    const Fp28 Mo = swp(M);
    const Fp28 MM = sqr<12>(M, Mo);
    Pt r;
    r.X = norm(sub<3>(sub<3>(MM, S), S));          // k 8
    const Fp28 T = norm(sub<12>(S, r.X));          // k 14
    const Fp28 To = swp(T);
    const Fp28 m1 = mul<24>(M, Mo, T, To);
    const Fp28 Wo = swp(Wv);
    const Fp28 m2 = mul<3>(Wv, Wo, p.Y, swp(p.Y));  // Y < 3 p
    r.Y = reduce(sub<3>(m1, m2));
    r.ZZ = mul<3>(V, Vo, p.ZZ, swp(p.ZZ));
    r.ZZZ = mul<3>(Wv, Wo, p.ZZZ, swp(p.ZZZ));
    return r;
}

// acc += q (add-2008-s), both in the stored-point discipline.  `dbl_fn(acc)` doubles acc in place in the (rare) P + P branch: the tail
// kernels pass a CALL of their no-inline doubling there, which keeps their no-inline addition below the 128 KB reach of a conditional
// branch (with the doubling inlined the body is 137 KB, and the compiler's long-branch expansion of its early exits took s[30:31] - the
// live return address - as the scratch pair: a wave whose lanes ALL left early then returned to the function's own epilogue for ever;
// first GPU run of the pair tails, round 5 run 4; tests/test_code_objects_cpu.py now refuses such a function)
struct DblInline {
    template <class P>
    G2P_FN void operator()(P& p) const;
};
template <class DblFn = DblInline>
G2P_FN void add(Pt& acc, const Pt& q, DblFn&& dbl_fn = DblFn()) {
    if (is_identity(q)) return;
    if (is_identity(acc)) {
        acc = q;
        return;
    }
    const bool odd = lane_odd();
    const Fp28 aZZo = swp(acc.ZZ), aZZZo = swp(acc.ZZZ), qZZo = swp(q.ZZ), qZZZo = swp(q.ZZZ);
    const Fp28 U1 = mul<3>(acc.X, swp(acc.X), q.ZZ, qZZo);     // 12 (2 + 3)
    const Fp28 U2 = mul<3>(q.X, swp(q.X), acc.ZZ, aZZo);
    const Fp28 S1 = mul<3>(acc.Y, swp(acc.Y), q.ZZZ, qZZZo);   // 3 (2 + 3)
    const Fp28 S2 = mul<3>(q.Y, swp(q.Y), acc.ZZZ, aZZZo);
    const Fp28 Pp = norm(sub<3>(U2, U1)), R = norm(sub<3>(S2, S1));  // k 5
    const Fp28 Ppo = swp(Pp), Ro = swp(R);
    const Fp28 PP = sqr<6>(Pp, Ppo);                           // (10)(5 + 6)
    if (pair_mulout_is_zero(PP)) {
        const Fp28 RR = sqr<6>(R, Ro);
        if (pair_mulout_is_zero(RR)) dbl_fn(acc);
        else acc = identity();
        return;
    }
    const Fp28 PPo = swp(PP);
    const Fp28 PPP = mul<3>(Pp, Ppo, PP, PPo);
    const Fp28 U1o = swp(U1);
    const Fp28 Q = mul<3>(U1, U1o, PP, PPo);
    const Fp28 RR = sqr<6>(R, Ro);
    const Fp28 X3 = norm(sub<3>(sub<3>(sub<3>(RR, PPP), Q), Q));  // k 11
    const Fp28 PPPo = swp(PPP);
    const Fp28 zz = mul<3>(acc.ZZ, aZZo, q.ZZ, qZZo);
    const Fp28 zzz = mul<3>(acc.ZZZ, aZZZo, q.ZZZ, qZZZo);
    acc.ZZ = mul<3>(zz, swp(zz), PP, PPo);
    acc.ZZZ = mul<3>(zzz, swp(zzz), PPP, PPPo);
    const Fp28 T = norm(sub<12>(Q, X3));  // k 14
    const Fp28 To = swp(T);
    const Fp28 nS = sub<3>(zero(), S1);   // S1 a product output
    const Fp28 S1o = swp(S1), nSo = swp(nS);
    const Fp28 nTo = sub<24>(zero(), To);
    acc.Y = mul4_body(R, sel(odd, T, To), Ro, sel(odd, nTo, T), nS, sel(odd, PPP, PPPo), sel(odd, S1o, nSo), sel(odd, PPPo, PPP));
    acc.X = X3;
}

template <class P>
G2P_FN void DblInline::operator()(P& p) const { p = dbl(p); }

#undef G2P_FN
}  // namespace g2p
}  // namespace bzk
