"""CPU suite: the ARITHMETIC behind the multiplication-free bucket reduction of the G1 MSM (bazuka_amd/csrc/msm_impl.cuh section 6b), on integers.

    sum_b (b + 1) B_b  =  S_tot + sum_k 2^k S_k,    S_k = sum of the buckets whose index has bit k set

with b = L h + l: S_k from COLUMN sums (k < log2 L) or ROW sums (k >= log2 L); adjacent bits are combined on the device (D_j = S_2j + 2 S_(2j+1)) and the
remaining weights applied by the host's Horner at bit positions 2 j (msm_horner_terms_host).  This file restates the kernels' index maps (msm_rowcol_plan, the lane -> bucket maps of msm_rowcol_quad_kernel, leaf_index of
msm_bitsum_quad_kernel, the term positions of the Horner) in Python over the group Z - a MODEL of the device code, kept beside it so that the identity and
the maps are checked on the GPU-less box for every window size the path takes (c = 11 .. 16 and the static table's c = 20); the device code itself is pinned
by the GPU parity tests (every MSM result of tests/test_gpu_msm.py, test_gpu_fullsize.py, test_gpu_endo.py goes through it).
Replaces bellman 0.14 `multiexp`'s bucket summation (call site /root/reference/src/mpn/circuits/test.rs:135)."""
import random

import pytest

T = 256  # threads of a workgroup


def plan(cm1):
    lbits = (cm1 + 1) // 2
    hbits = cm1 - lbits
    L, H = 1 << lbits, 1 << hbits
    leaf_r, leaf_c = max(min(8, L), L // T), max(min(8, H), H // T)
    seg_r, seg_c = L // leaf_r, H // leaf_c
    return dict(lbits=lbits, hbits=hbits, leaf_r=leaf_r, leaf_c=leaf_c, seg_r=seg_r, seg_c=seg_c,
                wgs_r=(H + T // seg_r - 1) // (T // seg_r), wgs_c=(L + T // seg_c - 1) // (T // seg_c))


def rowcol(b, P):
    """msm_rowcol_quad_kernel: one workgroup per (pass, block of rows / columns); a lane sums `leaf` buckets, a tree per segment of lanes"""
    L, H = 1 << P["lbits"], 1 << P["hbits"]
    rows, cols = [None] * H, [None] * L
    for wg in range(P["wgs_r"] + P["wgs_c"]):
        col_pass = wg >= P["wgs_r"]
        local = wg - (P["wgs_r"] if col_pass else 0)
        sh, meta = [0] * T, {}
        seg = P["seg_c"] if col_pass else P["seg_r"]
        for tid in range(T):
            if not col_pass:
                row, part = local * (T // seg) + tid // seg, tid % seg
                valid, first, stride, leaf, slot, out = row < H, row * L + part * P["leaf_r"], 1, P["leaf_r"], tid, row
            else:
                cpw = T // seg
                col, part = local * cpw + tid % cpw, tid // cpw
                valid, first, stride, leaf, slot, out = col < L, part * P["leaf_c"] * L + col, L, P["leaf_c"], (tid % cpw) * seg + part, col
            assert slot not in meta
            sh[slot] = sum(b[first + j * stride] for j in range(leaf)) if valid else 0
            meta[slot] = (valid, out)
        s = seg // 2
        while s > 0:  # g1_quad_tree_seg
            for g in range(T // seg):
                for i in range(s):
                    sh[g * seg + i] += sh[g * seg + i + s]
            s //= 2
        for slot, (valid, out) in meta.items():
            if valid and slot % seg == 0:
                tgt = cols if col_pass else rows
                assert tgt[out] is None
                tgt[out] = sh[slot]
    return rows, cols


def bitsums(rows, cols, P):
    """msm_bitsum_quad_kernel: term t < n_pairs of a set is D_t = S_2t + 2 S_(2t+1) (two adjacent bits, one workgroup half each); term n_pairs is the plain sum"""
    n_bits = P["lbits"] + P["hbits"]
    n_pairs = (n_bits + 1) // 2

    def bit_sum(k):
        from_cols = k < P["lbits"]
        n_src = 1 << (P["lbits"] if from_cols else P["hbits"])
        bit = k if from_cols else k - P["lbits"]
        src = cols if from_cols else rows
        idx = [((((j >> bit) << 1) | 1) << bit) | (j & ((1 << bit) - 1)) for j in range(n_src // 2)]
        assert len(set(idx)) == n_src // 2 and all((i >> bit) & 1 for i in idx) and max(idx) < n_src
        return sum(src[i] for i in idx)

    out = []
    for t in range(n_pairs):
        d = bit_sum(2 * t)
        if 2 * t + 1 < n_bits:          # an odd bit count leaves the last pair's upper half empty
            d += 2 * bit_sum(2 * t + 1)
        out.append(d)
    out.append(sum(rows))               # the plain sum over the ROW sums
    return out


def horner_terms(terms, count, c, w0):
    """msm_horner_terms_host"""
    n_pairs = c // 2
    n_terms = n_pairs + 1
    acc = 0
    for k in range(count - 1, -1, -1):
        t = terms[k * n_terms:(k + 1) * n_terms]
        for bit in range(c - 1, -1, -1):
            acc *= 2
            if bit % 2 == 0 and bit // 2 < n_pairs:
                acc += t[bit // 2]
            if bit == 0:
                acc += t[n_pairs]
    return acc << (c * w0)


@pytest.mark.parametrize("c", [11, 12, 13, 14, 15, 16, 20])
def test_row_column_bit_sums_and_the_horner_positions_give_the_weighted_bucket_sum(c):
    rnd = random.Random(c)
    half, P = 1 << (c - 1), plan(c - 1)
    assert (1 << P["lbits"]) * (1 << P["hbits"]) == half and P["seg_r"] <= T and P["seg_c"] <= T
    W, w0 = 3, 2
    terms, want = [], 0
    for w in range(W):
        b = [rnd.randrange(1 << 20) if rnd.random() < 0.7 else 0 for _ in range(half)]   # empty buckets are identities
        rows, cols = rowcol(b, P)
        assert None not in rows and None not in cols
        assert sum(rows) == sum(cols) == sum(b)
        terms += bitsums(rows, cols, P)
        want += sum((i + 1) * x for i, x in enumerate(b)) << (c * (w + w0))
    assert len(terms) == W * (c // 2 + 1)
    assert horner_terms(terms, W, c, w0) == want


def test_general_additions_per_bucket():
    """what the form saves: 2 additions per bucket (+ trees) against the chunked running sum's 16 + ~23.5 per chunk of 8"""
    P = plan(15)
    leaf_adds = 2 * ((1 << 15) - (1 << 15) // 8)                 # both passes, 7 of 8 buckets per lane
    tree_adds = (1 << P["hbits"]) * (P["seg_r"] - 1) + (1 << P["lbits"]) * (P["seg_c"] - 1)
    bit_adds = sum((1 << (P["lbits"] - 1)) - 1 for _ in range(P["lbits"])) + sum((1 << (P["hbits"] - 1)) - 1 for _ in range(P["hbits"])) + (1 << P["hbits"]) - 1
    per_bucket = (leaf_adds + tree_adds + bit_adds) / (1 << 15)
    assert 1.9 < per_bucket < 2.1, per_bucket
    old = (16 + 15 * 0.65 + 7.5 + 1) / 8                         # running sums + 15-bit double-and-add (a doubling ~ 0.65 of an addition) + the final addition
    assert old / per_bucket > 2.0
