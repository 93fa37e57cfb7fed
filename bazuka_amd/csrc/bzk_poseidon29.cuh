// Poseidon permutation on the 9 x 29-bit field (bzk_fr29.cuh), shared by the device kernel (poseidon.hip) and the
// CPU harness that checks its bounds (tests/host/hostcheck.hip).  Same function as the reference's
// `PoseidonState::hash` (/root/reference/src/zk/poseidon/mod.rs:24-84): state = [0, inputs...], R_F/2 full rounds,
// R_P partial rounds (S-box on element 0), R_F/2 full rounds, result = state[1] - evaluated in the equivalent
// "sparse partial rounds" form derived by bzk_poseidon_opt.h: 2T - 1 products per partial round instead of T^2.
//
// consts (internal 9 x 29-bit form), see poseidon_optimize():
//   rc_first (rf/2 * T) | pre (T) | rp x { s_i, row0[T], what[T-1] } | D ((T-1)^2) | rc_second (rf/2 * T) | mds (T*T)
//
// Bounds (k: value < k * r, see bzk_fr29.cuh; a group of products sharing one reduction needs sum ka*kb <= 70):
//   full round:    st (k <= 4) + rc -> k <= 5 <= 7 into the S-box ; MDS rows in groups of <= 6 products of k 2 operands
//   partial round: st[0] = sbox (k 2) + s_i -> k 3 ;  row 0 = one group: 3 + sum_j k_j  (j >= 1, constants k 1)
//                  st[j] += what[j] * st[0]  (k_j += 2, carries normalised) ; every RENORM rounds st[j] *= 1 -> k 2
#pragma once
#include "bzk_fr29.cuh"

namespace bzk {

namespace p29 {
#ifndef BZK_POSEIDON_PARTIAL_INLINE
#define BZK_POSEIDON_PARTIAL_INLINE 6  // widths up to this value run the products of a PARTIAL round inlined into the round loop (7 products + one row, ~17 KB of
#endif                                 // code for width 5) instead of calling the resident product functions: no argument / result moves, - 7 % instructions per
                                       // partial round.  Widths 7 and 8 keep the calls: inlined they need 410 registers and scratch (one wave per SIMD).  0: calls everywhere
// products of the partial-round loop: calls (as everywhere else) or inlined bodies
template <int T>
struct PartialOps {
    static constexpr bool INL = T <= BZK_POSEIDON_PARTIAL_INLINE;
    BZK_HD static Fr29 mul(const Fr29& a, const Fr29& b) { return INL ? fr29::mul_body(a, b) : fr29::mul(a, b); }
    BZK_HD static Fr29 sqr(const Fr29& a) { return INL ? fr29::sqr_body(a) : fr29::sqr(a); }
    BZK_HD static Fr29 sbox5(const Fr29& x) { return mul(sqr(sqr(x)), x); }
};
// dense row product: out = sum_k m[k] * v[k], n <= 8 operands, groups of <= 6 per reduction; operands k <= 5
template <int N>
BZK_HD Fr29 row_dot(const Fr29* __restrict__ m, const Fr29* v) {
    fr29::Wide w;
    fr29::wide_zero(w);
    constexpr int G0 = N < 6 ? N : 6;
#pragma unroll
    for (int k = 0; k < G0; ++k) fr29::wide_mac(w, m[k], v[k]);
    Fr29 acc = fr29::wide_reduce(w);
    if (N > 6) {
        fr29::wide_zero(w);
#pragma unroll
        for (int k = 6; k < N; ++k) fr29::wide_mac(w, m[k], v[k]);
        acc = fr29::norm(fr29::add(acc, fr29::wide_reduce(w)));  // k 4
    }
    return acc;
}
#ifndef BZK_POSEIDON_ROWS_STRAIGHT
#define BZK_POSEIDON_ROWS_STRAIGHT 1  // 0: the MDS rows of a full round as a loop the compiler may keep rolled (A/B builds: it did, and wrote the
#endif                                //    row results through scratch with a dynamic index - 192 B per lane, 2.2 GB of HBM writes per 2^24-leaf tree)
// rows J - 1 .. 0 as straight-line code: every nw[] index is a compile-time constant, the results stay in registers
template <int T, int J>
struct MdsRows {  // T = row length = row stride
    BZK_HD static void run(Fr29* nw, const Fr29* __restrict__ mds, const Fr29* st) {
        MdsRows<T, J - 1>::run(nw, mds, st);
        nw[J - 1] = row_dot<T>(mds + (J - 1) * T, st);
    }
};
template <int T>
struct MdsRows<T, 0> {
    BZK_HD static void run(Fr29*, const Fr29* __restrict__, const Fr29*) {}
};
#ifndef BZK_POSEIDON_MDS_RELOAD
#define BZK_POSEIDON_MDS_RELOAD 0  // 1: the MDS matrix is re-read (scalar loads) in every full round instead of being hoisted out of the round loop:
#endif                             //    hoisted it is T*T*9 scalars - more than the SGPR file - and the compiler parks them in VGPR lanes (v_readlane per use)
template <int T>
BZK_HD void full_round(Fr29* st, const Fr29* __restrict__ rc, const Fr29* __restrict__ mds_in) {
    const Fr29* mds = mds_in;
#if BZK_POSEIDON_MDS_RELOAD && defined(__HIP_DEVICE_COMPILE__)
    __asm__ volatile("" : "+s"(mds));  // the address is opaque per round: no loop-invariant hoisting of the loads
#endif
#pragma unroll
    for (int k = 0; k < T; ++k) st[k] = fr29::sbox5(fr29::norm(fr29::add(st[k], rc[k])));
    Fr29 nw[T];
#if BZK_POSEIDON_ROWS_STRAIGHT
    MdsRows<T, T>::run(nw, mds, st);
#else
#pragma unroll
    for (int j = 0; j < T; ++j) nw[j] = row_dot<T>(mds + j * T, st);
#endif
#pragma unroll
    for (int k = 0; k < T; ++k) st[k] = nw[k];
}
// rounds between renormalisations of st[1..]: row 0 needs 3 + (T-1) * k <= 70 (first group: at most 5 of them)
template <int T>
struct Renorm {
    static constexpr int OTHERS = T - 1 < 5 ? T - 1 : 5;
    static constexpr int KMAX_RAW = (70 - 3) / OTHERS;
    static constexpr int KMAX = KMAX_RAW > 34 ? 34 : KMAX_RAW;  // from29 / the 9-limb capacity want k <= 35
    static constexpr int PERIOD = (KMAX - 4) / 2;               // entering k <= 4, +2 per round
};
}  // namespace p29

template <int T>
BZK_HD Fr poseidon29_hash(const Fr* __restrict__ in, const Fr29* __restrict__ consts, int rf, int rp) {
    static_assert(T >= 2 && T <= 8, "widths above 8 use the generic kernel");
    static_assert(p29::Renorm<T>::PERIOD >= 1, "renormalisation period");
    Fr29 st[T];
    st[0] = fr29::zero();
#pragma unroll
    for (int k = 1; k < T; ++k) st[k] = fr29::to29(in[k - 1]);
    const int half_f = rf / 2;
    const Fr29* rc1 = consts;
    const Fr29* pre = rc1 + (size_t)half_f * T;
    const Fr29* part = pre + T;
    const Fr29* dmat = part + (size_t)rp * 2 * T;
    const Fr29* rc2 = dmat + (T - 1) * (T - 1);
    const Fr29* mds = rc2 + (size_t)half_f * T;
#pragma unroll 1
    for (int r = 0; r < half_f; ++r) p29::full_round<T>(st, rc1 + r * T, mds);
    // ---- partial block
#pragma unroll
    for (int k = 0; k < T; ++k) st[k] = fr29::norm(fr29::add(st[k], pre[k]));  // k <= 5
    const Fr29 one = fr29::from_consts(fr29::ONE);
    int since = 0;
#pragma unroll 1
    for (int i = 0; i < rp; ++i) {
        const Fr29* c = part + (size_t)i * 2 * T;  // s_i, row0[T], what[T-1]
        st[0] = fr29::norm(fr29::add(p29::PartialOps<T>::sbox5(st[0]), c[0]));  // k 3
        const Fr29 n0 = p29::row_dot<T>(c + 1, st);
#pragma unroll
        for (int j = 1; j < T; ++j) st[j] = fr29::norm(fr29::add(st[j], p29::PartialOps<T>::mul(c[T + j], st[0])));
        st[0] = n0;
        if (++since == p29::Renorm<T>::PERIOD) {
            since = 0;
#pragma unroll
            for (int j = 1; j < T; ++j) st[j] = p29::PartialOps<T>::mul(st[j], one);  // same value, k 2
        }
    }
    {
        Fr29 v[T > 2 ? T - 1 : 1], nw[T > 2 ? T - 1 : 1];
#pragma unroll
        for (int j = 1; j < T; ++j) v[j - 1] = since ? fr29::mul(st[j], one) : st[j];
#if BZK_POSEIDON_ROWS_STRAIGHT
        p29::MdsRows<T - 1, T - 1>::run(nw, dmat, v);
#else
#pragma unroll
        for (int j = 0; j < T - 1; ++j) nw[j] = p29::row_dot<T - 1>(dmat + j * (T - 1), v);
#endif
#pragma unroll
        for (int j = 1; j < T; ++j) st[j] = nw[j - 1];
    }
#pragma unroll 1
    for (int r = 0; r < half_f; ++r) p29::full_round<T>(st, rc2 + r * T, mds);
    return fr29::from29(st[1]);
}

}  // namespace bzk
