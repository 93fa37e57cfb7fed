// Cooperative Poseidon: EIGHT LANES PER HASH (lanes 0 .. T-1 of a group of eight hold one state element each, eight hashes per wavefront) for
// LATENCY-bound uses - the top of a tree, small batched tree updates (poseidon.hip), and since round 6 the dependency chain of the deferred
// witness values (witfill.hip: a Merkle path is a chain of hashes).  A round function is a serial chain (S-box -> row product -> next S-box);
// spreading the elements of a round over lanes shortens it from ~10 to ~6 product-times per partial round and from ~30 to ~6 per full round
// (about 2x per hash) at an eighth of the throughput.  All lanes execute the same instruction stream (values of the idle lanes are discarded);
// cross-lane traffic is ds_bpermute through __shfl.  Same constants block as poseidon29_hash<T> (bzk_poseidon29.cuh), same result as the
// reference's `PoseidonState::hash` (/root/reference/src/zk/poseidon/mod.rs:24-84).  Device only.
#pragma once
#include "bzk_poseidon29.cuh"

namespace bzk {

__device__ __forceinline__ Fr29 shfl29(const Fr29& v, int src_lane) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__shfl((int)v.l[i], src_lane, 64);
    return r;
}
__device__ __forceinline__ Fr29 shfl29_xor(const Fr29& v, int mask) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__shfl_xor((int)v.l[i], mask, 64);
    return r;
}
__device__ __forceinline__ Fr29 sel29(bool c, const Fr29& a, const Fr29& b) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

// mine: this lane's input - input j - 1 of the hash for lanes 1 <= j < T of the group (the other lanes' values are ignored); returns the hash in lane 1
// of the group (other lanes: garbage).  Any width 2 <= T <= 8 (eight lanes per node): T = 5 is the tree node, the others are struct hashes
template <int T>
__device__ Fr poseidon29_coop_val(const Fr& mine, const Fr29* __restrict__ consts, int rf, int rp) {
    static_assert(T >= 2 && T <= 8, "eight lanes per node");
    const int lane = threadIdx.x & 63, j = lane & 7, g0 = lane & ~7;
    const int jj = j < T ? j : T - 1;  // idle lanes mirror the last state lane (their values are never used)
    const int half_f = rf / 2;
    const Fr29* rc1 = consts;
    const Fr29* pre = rc1 + (size_t)half_f * T;
    const Fr29* part = pre + T;
    const Fr29* dmat = part + (size_t)rp * 2 * T;
    const Fr29* rc2 = dmat + (T - 1) * (T - 1);
    const Fr29* mds = rc2 + (size_t)half_f * T;
    const Fr29 one = fr29::from_consts(fr29::ONE);
    Fr29 st = fr29::to29(mine);
    if (j == 0) st = fr29::zero();
    auto full_round = [&](const Fr29* rc) {
        st = fr29::sbox5(fr29::norm(fr29::add(st, rc[jj])));
        Fr29 v[T];
#pragma unroll
        for (int k = 0; k < T; ++k) v[k] = shfl29(st, g0 + k);
        st = p29::row_dot<T>(mds + jj * T, v);
    };
#pragma unroll 1
    for (int r = 0; r < half_f; ++r) full_round(rc1 + r * T);
    st = fr29::norm(fr29::add(st, pre[jj]));
#pragma unroll 1
    for (int i = 0; i < rp; ++i) {
        const Fr29* c = part + (size_t)i * 2 * T;  // s_i, row0[T], what[T - 1]
        // lane 0: S-box and the scalar constant (the other lanes run the same instructions on values that are dropped)
        const Fr29 sb = fr29::norm(fr29::add(fr29::sbox5(j == 0 ? st : one), c[0]));
        const Fr29 s0 = shfl29(sb, g0);
        // row 0: one product per lane, butterfly sum over the group (lanes T..7 contribute zero)
        Fr29 prod = fr29::mul(c[1 + jj], j == 0 ? s0 : st);
        if (j >= T) prod = fr29::zero();
        prod = fr29::norm(fr29::add(prod, shfl29_xor(prod, 1)));
        prod = fr29::norm(fr29::add(prod, shfl29_xor(prod, 2)));
        prod = fr29::norm(fr29::add(prod, shfl29_xor(prod, 4)));  // k <= 10, in every lane
        // the other coordinates: x_j += what_j * s0
        const Fr29 u = fr29::mul(c[T + (jj >= 1 ? jj : 1)], s0);
        const Fr29 nx = fr29::norm(fr29::add(st, u));
        // one product by 1 brings every lane back to k 2 (lane 0 needs it for the next S-box; the others ride along)
        st = fr29::mul(j == 0 ? prod : nx, one);
    }
    {
        Fr29 v[T - 1];
#pragma unroll
        for (int k = 0; k < T - 1; ++k) v[k] = shfl29(st, g0 + 1 + k);
        const Fr29 d = p29::row_dot<T - 1>(dmat + (jj >= 1 ? jj - 1 : 0) * (T - 1), v);
        st = sel29(j == 0, st, d);
    }
#pragma unroll 1
    for (int r = 0; r < half_f; ++r) full_round(rc2 + r * T);
    return fr29::from29(st);
}

// in: the node's T - 1 inputs in memory (read by lanes 1..T-1)
template <int T>
__device__ Fr poseidon29_coop(const Fr* __restrict__ in, const Fr29* __restrict__ consts, int rf, int rp) {
    const int j = threadIdx.x & 7, jj = j < T ? j : T - 1;
    return poseidon29_coop_val<T>(in[jj >= 1 ? jj - 1 : 0], consts, rf, rp);
}

}  // namespace bzk
