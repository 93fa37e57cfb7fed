// Poseidon permutation on the 9 x 29-bit field (bzk_fr29.cuh), shared by the device kernel (poseidon.hip) and the
// CPU harness that checks its bounds (tests/host/hostcheck.hip).  Same function as the reference's
// `PoseidonState::hash` (/root/reference/src/zk/poseidon/mod.rs:24-84): state = [0, inputs...], R_F/2 full rounds,
// R_P partial rounds (S-box on element 0), R_F/2 full rounds, dense MDS every round, result = state[1].
// consts = round constants (T * (rf + rp)) then the MDS matrix (T * T, row-major), internal form.
#pragma once
#include "bzk_fr29.cuh"

namespace bzk {

template <int T>
BZK_HD Fr poseidon29_hash(const Fr* __restrict__ in, const Fr29* __restrict__ consts, int rf, int rp) {
    static_assert(T >= 2 && T <= 8, "widths above 8 use the generic kernel");
    Fr29 st[T];
    st[0] = fr29::zero();
#pragma unroll
    for (int k = 1; k < T; ++k) st[k] = fr29::to29(in[k - 1]);
    const Fr29* rc = consts;
    const Fr29* mds = consts + (size_t)T * (rf + rp);
    const int half_f = rf / 2;
#pragma unroll 1
    for (int rnd = 0; rnd < rf + rp; ++rnd) {
#pragma unroll
        for (int k = 0; k < T; ++k) st[k] = fr29::norm(fr29::add(st[k], rc[rnd * T + k]));  // k <= 7, L 29
        const bool full = rnd < half_f || rnd >= half_f + rp;
        if (full) {
#pragma unroll
            for (int k = 0; k < T; ++k) st[k] = fr29::sbox5(st[k]);
        } else {
            st[0] = fr29::sbox5(st[0]);
        }
        Fr29 nw[T];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            // row j of the MDS product: <= 6 products per 64-bit column set, one Montgomery reduction per set
            fr29::Wide w;
            fr29::wide_zero(w);
            constexpr int G0 = T < 6 ? T : 6;
#pragma unroll
            for (int k = 0; k < G0; ++k) fr29::wide_mac(w, mds[j * T + k], st[k]);
            Fr29 acc = fr29::wide_reduce(w);
            if (T > 6) {
                fr29::wide_zero(w);
#pragma unroll
                for (int k = 6; k < T; ++k) fr29::wide_mac(w, mds[j * T + k], st[k]);
                acc = fr29::norm(fr29::add(acc, fr29::wide_reduce(w)));  // k 4
            }
            nw[j] = acc;
        }
#pragma unroll
        for (int k = 0; k < T; ++k) st[k] = nw[k];
    }
    return fr29::from29(st[1]);
}

}  // namespace bzk
