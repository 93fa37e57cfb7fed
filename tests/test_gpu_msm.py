"""GPU parity: libbzk Pippenger (K4/K5) vs the CPU oracle, through the C ABI.  Bit-exact (integer)."""
import os

import pytest
import torch

from util import dev_bytes, fr_bytes, fr_list, rand_scalars_bytes, to_dev

pytestmark = pytest.mark.gpu


def _neg_y(pr, raw96):
    return raw96[:48] + pr.fp_to_mont_bytes(-pr.fp_from_mont_bytes(raw96[48:96]))


@pytest.mark.parametrize("n", [1, 2, 3, 17, 100, 1000, 5000])
def test_msm_g1_small_vs_oracle(bzk, co, n):
    bases = co.g1_bases(21, 0, n, nthreads=co.ncpu())
    sc = rand_scalars_bytes(n, n)
    assert bzk.msm_g1(bases, sc) == co.msm_g1(bases, sc, nthreads=co.ncpu())


def test_msm_g1_empty_and_zero(bzk, co, pr):
    assert bzk.msm_g1(b"", b"") == pr.g1_to_bytes(None)
    bases = co.g1_bases(1, 0, 8)
    assert bzk.msm_g1(bases, fr_bytes([0] * 8)) == pr.g1_to_bytes(None)


def test_msm_g1_edge_scalars_and_repeats(bzk, co, pr):
    n = 64
    bases = bytearray(co.g1_bases(31, 0, n))
    # repeated points, P and -P with equal scalars (cancellation inside a bucket)
    bases[96:192] = bases[0:96]
    bases[192:288] = _neg_y(pr, bytes(bases[0:96]))
    sc = fr_list(n, 5)
    sc[0] = sc[1] = sc[2] = 12345          # same bucket: P + P + (-P)
    sc[3], sc[4], sc[5] = 0, 1, pr.R_MOD - 1
    sc[6] = 1 << 15                          # digit exactly half -> positive top bucket
    sc[7] = (1 << 16) - 1                    # carries into the next window
    sc[8] = pr.R_MOD - 2
    for mont in (True, False):
        scb = fr_bytes(sc, mont=mont)
        assert bzk.msm_g1(bytes(bases), scb, canonical=not mont) == co.msm_g1(bytes(bases), scb, mont=mont)


def test_msm_g1_all_same_scalar(bzk, co, pr):
    """every point lands in the same buckets (one bucket per window): worst-case skew"""
    n = 3000
    bases = co.g1_bases(41, 0, n, nthreads=co.ncpu())
    scb = fr_bytes([1] * n)
    assert bzk.msm_g1(bases, scb) == co.msm_g1(bases, scb, nthreads=co.ncpu())
    scb = fr_bytes([0x123456789ABCDEF0123456789ABCDEF] * n)
    assert bzk.msm_g1(bases, scb) == co.msm_g1(bases, scb, nthreads=co.ncpu())


def test_synth_bases_match_oracle(bzk, co):
    n = 200
    d = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(0x42415A554B41, 5, n, d)
    torch.cuda.synchronize()
    assert dev_bytes(d) == co.g1_bases(0x42415A554B41, 5, n, nthreads=co.ncpu())
    d2 = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    bzk.g2_synth_bases_dev(0x42415A554B41, 5, n, d2)
    torch.cuda.synchronize()
    assert dev_bytes(d2) == co.g2_bases(0x42415A554B41, 5, n, nthreads=co.ncpu())


def test_msm_g1_windows_partition(bzk, co):
    """window-range shards fold (bzk_g1_sum) to the full MSM - the multi-GPU combine on one GPU"""
    n = 4096
    bases, sc = to_dev(co.g1_bases(51, 0, n, nthreads=co.ncpu())), to_dev(rand_scalars_bytes(n, 3))
    full = bzk.msm_g1_dev(bases, sc, n)
    W = bzk.msm_window_count(n)
    for parts in (2, 3, W):
        cuts = [W * i // parts for i in range(parts + 1)]
        shards = b"".join(bzk.msm_g1_windows_dev(bases, sc, n, cuts[i], cuts[i + 1]) for i in range(parts))
        assert bzk.g1_sum(shards) == full
    assert full == co.msm_g1(dev_bytes(bases), dev_bytes(sc), nthreads=co.ncpu())


def test_msm_g1_full_size_2p20_vs_oracle(bzk, co):
    """BASELINE config[1]: 2^20 points, uniform scalars; oracle = 8-thread CPU Pippenger (seconds)"""
    n = 1 << 20
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(0x42415A554B41, 0, n, bases)
    scb = rand_scalars_bytes(n, 2020)
    sc = to_dev(scb)
    got = bzk.msm_g1_dev(bases, sc, n)
    assert got == co.msm_g1(dev_bytes(bases), scb, nthreads=co.ncpu())
    # witness-like scalars: 10 % ones, 5 % zeros, small values (skewed buckets)
    import numpy as np
    a = np.frombuffer(scb, dtype=np.uint8).reshape(n, 32).copy()
    rs = np.random.RandomState(7)
    sel = rs.rand(n)
    one = np.frombuffer(fr_bytes([1]), dtype=np.uint8)
    a[sel < 0.10] = one
    a[(sel >= 0.10) & (sel < 0.15)] = 0
    wb = a.tobytes()
    got = bzk.msm_g1_dev(bases, to_dev(wb), n)
    assert got == co.msm_g1(dev_bytes(bases), wb, nthreads=co.ncpu())


def test_msm_g1_linearity_2p20(bzk, pr):
    """size-independent property at full size: MSM(s) + MSM(t) == MSM(s + t)"""
    import numpy as np
    n = 1 << 20
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(99, 0, n, bases)
    rs = np.random.RandomState(5)
    s = rs.randint(0, 2 ** 62, size=(n, 4), dtype=np.uint64)
    t = rs.randint(0, 2 ** 62, size=(n, 4), dtype=np.uint64)
    u = s + t  # limb-wise, no carries (each limb < 2^63), value < 2^255 < ... canonical form
    s[:, 3] &= (1 << 60) - 1
    t[:, 3] &= (1 << 60) - 1
    u = s + t
    ms = bzk.msm_g1_dev(bases, to_dev(s.tobytes()), n, canonical=True)
    mt = bzk.msm_g1_dev(bases, to_dev(t.tobytes()), n, canonical=True)
    mu = bzk.msm_g1_dev(bases, to_dev(u.tobytes()), n, canonical=True)
    assert bzk.g1_sum(ms + mt) == mu


@pytest.mark.parametrize("n", [1, 2, 50, 1500])
def test_msm_g2_small_vs_oracle(bzk, co, n):
    bases = co.g2_bases(23, 0, n, nthreads=co.ncpu())
    sc = rand_scalars_bytes(n, 100 + n)
    assert bzk.msm_g2(bases, sc) == co.msm_g2(bases, sc, nthreads=co.ncpu())


def test_msm_g2_edges(bzk, co, pr):
    n = 16
    bases = co.g2_bases(29, 0, n)
    sc = fr_list(n, 9)
    sc[0], sc[1], sc[2] = 0, 1, pr.R_MOD - 1
    scb = fr_bytes(sc)
    assert bzk.msm_g2(bases, scb) == co.msm_g2(bases, scb)
    assert bzk.msm_g2(b"", b"") == pr.g2_to_bytes(None)


def test_msm_g2_2p16_vs_oracle(bzk, co):
    n = 1 << 16
    bases = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    bzk.g2_synth_bases_dev(77, 0, n, bases)
    scb = rand_scalars_bytes(n, 16)
    assert bzk.msm_g2_dev(bases, to_dev(scb), n) == co.msm_g2(dev_bytes(bases), scb, nthreads=co.ncpu())


@pytest.mark.parametrize("g2", [False, True])
def test_msm_giant_buckets_two_level_fold_vs_oracle(bzk, co, g2):
    """buckets of more than 640 tasks take the two-level fold (msm_fold_wide_kernel / msm_fold_wide_g2pair_kernel): one scalar repeated for every point (ONE
    giant bucket per window, 40 000 entries = 1 250 tasks of 32), three values (several giants per window, dealt round-robin over the grid), and a vector in which
    the giants share their windows with ordinary buckets - all equal to the oracle's bytes"""
    n = 40000
    bases = torch.empty(n * (192 if g2 else 96), dtype=torch.uint8, device="cuda")
    (bzk.g2_synth_bases_dev if g2 else bzk.g1_synth_bases_dev)(91, 0, n, bases)
    hb = dev_bytes(bases)
    run = bzk.msm_g2_dev if g2 else bzk.msm_g1_dev
    want = co.msm_g2 if g2 else co.msm_g1
    vals = [0x1F3A5C7E9B2D4F60718293A4B5C6D7E8F9012345678, 0x2B1, 0x7FFF0001FFFE0003FFFC0007]
    mixed = fr_list(n, 31)
    for i in range(0, n, 3):
        mixed[i] = vals[0]
    for scalars in ([vals[0]] * n, [vals[i % 3] for i in range(n)], mixed):
        scb = fr_bytes(scalars)
        assert run(bases, to_dev(scb), n) == want(hb, scb, nthreads=co.ncpu())


def test_msm_g1_static_table_matches_plain_and_oracle(bzk, co):
    """static-base tables (bzk_msm_g1_table_*): same bytes as the per-call pipeline and the oracle; prefix use
    (fewer scalars than table entries) and window-range shards"""
    n = 6000
    hb = co.g1_bases(71, 0, n, nthreads=co.ncpu())
    bases = to_dev(hb)
    tab = bzk.msm_table_build(bases, n)
    for m, seed in ((n, 1), (n - 1234, 2), (1, 3)):
        scb = rand_scalars_bytes(m, seed)
        sc = to_dev(scb)
        got = bzk.msm_table_run_dev(tab, sc, m)
        assert got == bzk.msm_g1_dev(bases, sc, m)
        assert got == co.msm_g1(hb[: 96 * m], scb, nthreads=co.ncpu())
    W = bzk.msm_table_window_count(tab)
    sc = to_dev(rand_scalars_bytes(n, 9))
    full = bzk.msm_table_run_dev(tab, sc, n)
    cuts = [0, W // 3, W // 2, W]
    shards = b"".join(bzk.msm_table_windows_dev(tab, sc, n, cuts[i], cuts[i + 1]) for i in range(3))
    assert bzk.g1_sum(shards) == full
    # skewed scalars (many ones / zeros) through the shared bucket set
    import numpy as np
    a = np.frombuffer(rand_scalars_bytes(n, 4), dtype=np.uint8).reshape(n, 32).copy()
    a[: n // 3] = np.frombuffer(fr_bytes([1]), dtype=np.uint8)
    a[n // 3: n // 2] = 0
    wb = a.tobytes()
    assert bzk.msm_table_run_dev(tab, to_dev(wb), n) == co.msm_g1(hb, wb, nthreads=co.ncpu())
    bzk.msm_table_free(tab)


def test_msm_g2_static_table_matches_oracle(bzk, co):
    n = 1500
    hb = co.g2_bases(73, 0, n, nthreads=co.ncpu())
    bases = to_dev(hb)
    tab = bzk.msm_table_build(bases, n, g2=True)
    scb = rand_scalars_bytes(n, 5)
    assert bzk.msm_table_run_dev(tab, to_dev(scb), n, g2=True) == co.msm_g2(hb, scb, nthreads=co.ncpu())
    bzk.msm_table_free(tab)


def test_msm_g1_static_table_2p20(bzk, co):
    n = 1 << 20
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(0x42415A554B41, 0, n, bases)
    tab = bzk.msm_table_build(bases, n)
    scb = rand_scalars_bytes(n, 2020)
    sc = to_dev(scb)
    assert bzk.msm_table_run_dev(tab, sc, n) == bzk.msm_g1_dev(bases, sc, n)
    bzk.msm_table_free(tab)


@pytest.mark.parametrize("levels", [2, 3, 4, 5])
def test_msm_g1_folded_table_matches_plain_and_oracle(bzk, co, pr, levels):
    """folded tables (bzk_msm_g1_table_build_levels): L levels 2^(c wpl j) P_i, windows j * wpl + w' share bucket set w';
    same bytes as the per-call pipeline and the oracle for level counts that do and do not divide the window count,
    prefix use, skewed scalars, bucket-set shards"""
    n = 6000
    hb = co.g1_bases(71, 0, n, nthreads=co.ncpu())
    bases = to_dev(hb)
    tab = bzk.msm_table_build(bases, n, levels=levels)
    assert 1 < bzk.msm_table_levels(tab) <= levels
    for m, seed in ((n, 1), (n - 1234, 2), (1, 3)):
        scb = rand_scalars_bytes(m, seed)
        sc = to_dev(scb)
        got = bzk.msm_table_run_dev(tab, sc, m)
        assert got == bzk.msm_g1_dev(bases, sc, m)
        assert got == co.msm_g1(hb[: 96 * m], scb, nthreads=co.ncpu())
    # extreme scalars: 0, 1, r - 1 (top window + carries), 2^254
    ext = [0, 1, pr.R_MOD - 1, 1 << 254, (1 << 255) % pr.R_MOD] + fr_list(59, 7)
    eb = fr_bytes(ext)
    assert bzk.msm_table_run_dev(tab, to_dev(eb), len(ext)) == co.msm_g1(hb[: 96 * len(ext)], eb, nthreads=co.ncpu())
    S = bzk.msm_table_window_count(tab)  # bucket sets of the folded table
    sc = to_dev(rand_scalars_bytes(n, 9))
    full = bzk.msm_table_run_dev(tab, sc, n)
    assert full == bzk.msm_g1_dev(bases, sc, n)
    cuts = [0, S // 3, S // 2, S]
    shards = b"".join(bzk.msm_table_windows_dev(tab, sc, n, cuts[i], cuts[i + 1]) for i in range(3))
    assert bzk.g1_sum(shards) == full
    import numpy as np
    a = np.frombuffer(rand_scalars_bytes(n, 4), dtype=np.uint8).reshape(n, 32).copy()
    a[: n // 3] = np.frombuffer(fr_bytes([1]), dtype=np.uint8)
    a[n // 3: n // 2] = 0
    wb = a.tobytes()
    assert bzk.msm_table_run_dev(tab, to_dev(wb), n) == co.msm_g1(hb, wb, nthreads=co.ncpu())
    bzk.msm_table_free(tab)


def test_msm_g2_folded_table_matches_oracle(bzk, co):
    n = 1500
    hb = co.g2_bases(73, 0, n, nthreads=co.ncpu())
    bases = to_dev(hb)
    tab = bzk.msm_table_build(bases, n, g2=True, levels=2)
    scb = rand_scalars_bytes(n, 5)
    assert bzk.msm_table_run_dev(tab, to_dev(scb), n, g2=True) == co.msm_g2(hb, scb, nthreads=co.ncpu())
    bzk.msm_table_free(tab)


def test_msm_g1_folded_table_2p20(bzk, co):
    n = 1 << 20
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(0x42415A554B41, 0, n, bases)
    sc = to_dev(rand_scalars_bytes(n, 2020))
    want = bzk.msm_g1_dev(bases, sc, n)
    for levels in (2, 4):
        tab = bzk.msm_table_build(bases, n, levels=levels)
        assert bzk.msm_table_run_dev(tab, sc, n) == want
        bzk.msm_table_free(tab)


# ---- scalar de-duplication (BZK_F_DEDUP): what bzk_groth16_prove uses for the witness MSMs

def _witness_like_scalars(n, seed):
    """the shape of a Groth16 assignment: most values occur twice, bits by the thousands, zeros, a few heavy values"""
    import random
    rnd = random.Random(seed)
    distinct = fr_list(n // 2, seed)
    sc = [distinct[rnd.randrange(len(distinct))] for _ in range(n)]
    for i in range(0, n, 9):
        sc[i] = 1
    for i in range(4, n, 11):
        sc[i] = 0
    heavy = distinct[0]
    for i in range(2, n, 5):
        sc[i] = heavy
    return sc


@pytest.mark.parametrize("n", [4096, 5000, 70000])
def test_msm_g1_dedup_equals_plain_and_oracle(bzk, co, pr, n):
    bases = bytearray(co.g1_bases(77, 0, n, nthreads=co.ncpu()))
    sc = _witness_like_scalars(n, n)
    # equal scalars on equal points (doubling inside a group sum) and on P, -P (a group sum that is the identity)
    bases[96:192] = bases[0:96]
    bases[3 * 96:4 * 96] = _neg_y(pr, bytes(bases[2 * 96:3 * 96]))
    sc[0] = sc[1] = 0x1234567
    sc[2] = sc[3] = 0x7654321
    for mont in (True, False):
        scb = fr_bytes(sc, mont=mont)
        want = co.msm_g1(bytes(bases), scb, mont=mont, nthreads=co.ncpu())
        assert bzk.msm_g1(bytes(bases), scb, canonical=not mont, dedup=True) == want
        assert bzk.msm_g1(bytes(bases), scb, canonical=not mont) == want


def test_msm_g1_dedup_degenerate_inputs(bzk, co, pr):
    n = 6000
    bases = co.g1_bases(78, 0, n, nthreads=co.ncpu())
    assert bzk.msm_g1(bases, fr_bytes([0] * n), dedup=True) == pr.g1_to_bytes(None)          # nothing left
    one = fr_bytes([5] * n)
    assert bzk.msm_g1(bases, one, dedup=True) == co.msm_g1(bases, one, nthreads=co.ncpu())    # one group of n members
    uniq = rand_scalars_bytes(n, 99)
    assert bzk.msm_g1(bases, uniq, dedup=True) == co.msm_g1(bases, uniq, nthreads=co.ncpu())  # no duplicates at all


def test_msm_g2_dedup_equals_oracle(bzk, co):
    n = 9000
    bases = co.g2_bases(79, 0, n, nthreads=co.ncpu())
    scb = fr_bytes(_witness_like_scalars(n, 3))
    want = co.msm_g2(bases, scb, nthreads=co.ncpu())
    assert bzk.msm_g2(bases, scb, dedup=True) == want
    assert bzk.msm_g2(bases, scb) == want


def test_msm_g1_dedup_2p20_witness_like_property(bzk, co):
    """BASELINE size: de-duplicated == plain on 2^20 witness-like scalars (both bit-exact vs the oracle elsewhere)"""
    n = 1 << 20
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(5, 0, n, bases)
    sc = to_dev(fr_bytes(_witness_like_scalars(n, 20)))
    a = bzk.msm_g1_dev(bases, sc, n, dedup=True)
    assert a == bzk.msm_g1_dev(bases, sc, n)
    assert a == co.msm_g1(dev_bytes(bases), dev_bytes(sc), nthreads=co.ncpu())


def test_msm_g1_2p22_vs_oracle(bzk, co):
    """four times the BASELINE size, bit-exact against the oracle (and de-duplicated == plain)"""
    n = 1 << 22
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(9, 0, n, bases)
    sc = to_dev(rand_scalars_bytes(n, 22))
    got = bzk.msm_g1_dev(bases, sc, n)
    assert got == co.msm_g1(dev_bytes(bases), dev_bytes(sc), nthreads=co.ncpu())
    assert bzk.msm_g1_dev(bases, sc, n, dedup=True) == got


def _host_mem_budget_gb():
    """what this process may allocate on the host: the cgroup limit if there is one, else MemAvailable"""
    try:
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v != "max":
            return int(v) / 2 ** 30
    except OSError:
        pass
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 2 ** 20
    return 0.0


def _uniform_scalars_dev(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    sc[:, 31] &= 0x3F
    sc = sc.contiguous().view(-1)
    torch.cuda.synchronize()  # the scalars were produced on torch's stream; libbzk works on its own
    return sc


def _large_g1_msm_vs_oracle(bzk, co, log_n, seed, parts):
    """one MSM over 2^log_n points: the per-call pipeline, the resident-set form and the `parts`-way window-sharded fold
    (what `parts` ranks of a device group compute) all equal the oracle's bytes"""
    n = 1 << log_n
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(seed, 0, n, bases)
    sc = _uniform_scalars_dev(n, seed + log_n)
    whole = bzk.msm_g1_dev(bases, sc, n)
    hb, hs = bases.cpu().numpy(), sc.cpu().numpy()
    want = co.msm_g1_np(hb, hs, nthreads=co.ncpu())
    del hb, hs
    assert whole == want
    h = bzk.msm_bases_load_dev(bases, n)
    try:
        del bases
        assert bzk.msm_bases_run_dev(h, sc, n) == want
        W = bzk.msm_window_count(n)
        cuts = [W * i // parts for i in range(parts + 1)]
        shards = b"".join(bzk.msm_bases_windows_dev(h, sc, n, cuts[i], cuts[i + 1]) for i in range(parts))
        assert bzk.g1_sum(shards) == want
    finally:
        bzk.msm_bases_free(h)
        del sc
        bzk.trim()                 # the grow-only call workspace (24 GB after 2^26 points) goes back to the device for the tests that follow
        torch.cuda.empty_cache()


def test_msm_g1_2p24_vs_oracle(bzk, co):
    """the production circuit's size: the 2^24-point G1 MSM with uniform scalars - the bulk regime of the
    fold (`MSM_FOLD_BULK_FROM`) that only large n reaches and the bench times as `msm_g1_2p24` - bit-exact against the oracle,
    as are the resident-set form and the 8-way window-sharded fold (~25 s of the box's host cores)"""
    _large_g1_msm_vs_oracle(bzk, co, 24, 10, 8)


@pytest.mark.skipif(os.environ.get("BZK_TEST_2P26", "1") == "0", reason="BZK_TEST_2P26=0")
def test_msm_g1_2p26_vs_oracle_and_8way_window_shards(bzk, co):
    """BASELINE configs[3] (65536 tx = one 2^26-point G1 MSM, window-sharded across 8 ranks): whole MSM == oracle,
    and the fold of the eight window-range shares == the same bytes.  ~90 s of host cores, ~16 GB of host memory."""
    need = 24.0
    have = _host_mem_budget_gb()
    if have < need:
        pytest.skip(f"host memory budget {have:.0f} GB < {need:.0f} GB needed by the CPU oracle at 2^26 points")
    _large_g1_msm_vs_oracle(bzk, co, 26, 11, 8)


# ---- one stand-alone call as several window ranges in flight (msm_impl.cuh msm_run_split)

def _ctx_with_env(env):
    """a context of its own created under `env` (the split knobs are read when a context is created)"""
    from bazuka_amd import Bzk
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Bzk(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("n", [1 << 16, (1 << 16) + 4321, 1 << 18])
def test_msm_g1_split_ranges_vs_oracle(bzk, co, n):
    """a resident-set G1 call run as 1 / 2 / 3 / 4 window ranges in flight (children at normal and at the highest stream priority) returns the
    oracle's bytes, as does the raw-base call (its bases converted once for all ranges); back-to-back calls on one context re-use the children"""
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(77, 0, n, bases)
    sc = _uniform_scalars_dev(n, 9000 + n % 1000)
    want = co.msm_g1_np(bases.cpu().numpy(), sc.cpu().numpy(), nthreads=co.ncpu())
    assert bzk.msm_g1_dev(bases, sc, n) == want
    for parts, prio in ((1, 0), (2, 0), (2, 1), (3, 0), (4, 1)):
        ctx = _ctx_with_env({"BZK_MSM_SPLIT": str(parts), "BZK_MSM_SPLIT_PRIO": str(prio), "BZK_MSM_SPLIT_MIN_LOG": "12"})
        try:
            h = ctx.msm_bases_load_dev(bases, n)
            try:
                for _ in range(3):
                    assert ctx.msm_bases_run_dev(h, sc, n) == want, (parts, prio)
                # per-kernel events of a split call land on the children and are reported with the parent's
                ctx.prof_enable(True); ctx.prof_reset()
                assert ctx.msm_bases_run_dev(h, sc, n) == want
                launches = ctx.prof_dump().get("msm_accumulate", (0, 0.0))[0]
                ctx.prof_enable(False)
                assert launches == parts, (parts, launches)
                # fewer scalars than the set holds, zero scalars, and the flagged forms (never split) on the same context
                m = n - 1000
                assert ctx.msm_bases_run_dev(h, sc[:32 * m], m) == co.msm_g1_np(bases[:96 * m].cpu().numpy(), sc[:32 * m].cpu().numpy(), nthreads=co.ncpu())
                z = torch.zeros(n * 32, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
                assert ctx.msm_bases_run_dev(h, z, n) == bzk.msm_g1_dev(bases, z, n)
                assert ctx.msm_bases_run_dev(h, sc, n, throughput=True) == want
                # the raw-base entry in the same number of ranges, twice (the converted bases live in a buffer of the context's), then on a shorter prefix
                assert ctx.msm_g1_dev(bases, sc, n) == want and ctx.msm_g1_dev(bases, sc, n) == want
                assert ctx.msm_g1_dev(bases[:96 * m], sc[:32 * m], m) == ctx.msm_bases_run_dev(h, sc[:32 * m], m)
            finally:
                ctx.msm_bases_free(h)
        finally:
            ctx.close()


# ---- every window size, not only the ones the size-based pick happens to choose for the test sizes

@pytest.mark.parametrize("c", [5, 9, 11, 13, 14, 15, 17])
def test_msm_g1_every_window_size_vs_oracle(co, pr, c, monkeypatch):
    """BZK_MSM_C (read when a context is created) forces the window size: plain, de-duplicated and window-sharded MSMs
    equal the oracle for sizes whose automatic pick never reaches most of these c (11 and 13 - 15 were first exercised
    by the witness-window A/B of run 48, which found a window-count bug the size-based tests could not see)."""
    from bazuka_amd import Bzk
    monkeypatch.setenv("BZK_MSM_C", str(c))
    ctx = Bzk(0)
    try:
        n = 20000
        hb = bytearray(co.g1_bases(91, 0, n, nthreads=co.ncpu()))
        hb[96:192] = hb[0:96]
        sc = _witness_like_scalars(n, 100 + c)
        sc[0] = sc[1] = 0x1234567
        sc[2], sc[3], sc[4] = pr.R_MOD - 1, 1 << 254, (1 << c) - 1
        scb = fr_bytes(sc)
        want = co.msm_g1(bytes(hb), scb, nthreads=co.ncpu())
        assert ctx.msm_g1(bytes(hb), scb) == want
        assert ctx.msm_g1(bytes(hb), scb, dedup=True) == want
        W = ctx.msm_window_count(n)
        assert W == (256 + c - 1) // c
        bases, sd = to_dev(bytes(hb)), to_dev(scb)
        cuts = [0, 1, W // 2, W]
        parts = b"".join(ctx.msm_g1_windows_dev(bases, sd, n, cuts[i], cuts[i + 1]) for i in range(3))
        assert ctx.g1_sum(parts) == want
    finally:
        ctx.close()


@pytest.mark.parametrize("c", [7, 12])
def test_msm_g2_other_window_sizes_vs_oracle(co, pr, c, monkeypatch):
    from bazuka_amd import Bzk
    monkeypatch.setenv("BZK_MSM_C", str(c))
    ctx = Bzk(0)
    try:
        n = 4500
        hb = co.g2_bases(93, 0, n, nthreads=co.ncpu())
        scb = fr_bytes(_witness_like_scalars(n, 200 + c))
        want = co.msm_g2(hb, scb, nthreads=co.ncpu())
        assert ctx.msm_g2(hb, scb) == want
        assert ctx.msm_g2(hb, scb, dedup=True) == want
    finally:
        ctx.close()


def test_witness_window_knob_in_a_fresh_process(co):
    """BZK_MSM_C_WIT_G1 / _G2 (process-wide, read once): the unsharded de-duplicated MSMs run with their own window size and
    still equal the oracle; a named window range keeps the plain pick (its window count comes from bzk_msm_window_count)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from bazuka_amd import Bzk
from oracle import coracle as co
from util import fr_bytes, to_dev
import random
rnd = random.Random(5)
from oracle import pyref as pr
n = 9000
sc = [rnd.choice([0, 1, 1, 2, rnd.randrange(1 << 64), rnd.randrange(pr.R_MOD)]) for _ in range(n)]
scb = fr_bytes(sc)
ctx = Bzk(0)
b1 = co.g1_bases(5, 0, n, nthreads=co.ncpu()); b2 = co.g2_bases(5, 0, 3000, nthreads=co.ncpu())
assert ctx.msm_g1(b1, scb, dedup=True) == co.msm_g1(b1, scb, nthreads=co.ncpu())
assert ctx.msm_g2(b2, scb[:3000 * 32], dedup=True) == co.msm_g2(b2, scb[:3000 * 32], nthreads=co.ncpu())
W = ctx.msm_window_count(n)
d1, ds = to_dev(b1), to_dev(scb)
parts = ctx.msm_g1_windows_dev(d1, ds, n, 0, W // 2, dedup=True) + ctx.msm_g1_windows_dev(d1, ds, n, W // 2, W, dedup=True)
assert ctx.g1_sum(parts) == co.msm_g1(b1, scb, nthreads=co.ncpu())
print("ok")
''' % (root, os.path.join(root, "tests"))
    env = dict(os.environ, BZK_MSM_C_WIT_G1="11", BZK_MSM_C_WIT_G2="6")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.mark.parametrize("c", [0, 4, 8, 13, 16])
def test_msm_canonical_scalars_at_and_above_r(co, pr, c, monkeypatch):
    """BZK_F_CANONICAL input is not range-checked by callers; values in [r, 2^256) are taken mod r (r P = O) for
    every window size, including c | 256 (4, 8, 16) where the signed recoding has no spare top bit"""
    from bazuka_amd import Bzk
    if c:
        monkeypatch.setenv("BZK_MSM_C", str(c))
    ctx = Bzk(0)
    try:
        n = 64
        bases = co.g1_bases(91, 0, n)
        small = [5, 0, 1, pr.R_MOD - 1] + fr_list(n - 4, 17)
        big = [v + (pr.R_MOD if v + pr.R_MOD < (1 << 256) else 0) for v in small]
        big[0] = 5 + 2 * pr.R_MOD
        big[1] = (1 << 256) - 1                      # all ones
        small[1] = ((1 << 256) - 1) % pr.R_MOD
        assert all(b < (1 << 256) for b in big) and sum(b >= (1 << 255) for b in big) > 10
        want = co.msm_g1(bases, fr_bytes(small, mont=False), mont=False)
        raw = b"".join(b.to_bytes(32, "little") for b in big)
        assert ctx.msm_g1(bases, raw, canonical=True) == want
        assert ctx.msm_g1(bases, fr_bytes(small, mont=False), canonical=True) == want
    finally:
        ctx.close()


@pytest.mark.parametrize("log_n,c", [(14, 14), (16, 16), (16, 19)])
def test_msm_g1_full_table_with_explicit_window(bzk, co, log_n, c):
    """bzk_msm_g1_table_build_c: a full static-base table whose window size is chosen by the caller (all windows share one bucket
    set, so c ~ log2 n pays) - what bzk_groth16_prove builds for the h query: same bytes as the per-call pipeline and the oracle, for a
    prefix of the bases too; witness-like scalars (skew inside the single bucket set)"""
    n = 1 << log_n
    bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(500 + c, 0, n, bases)
    hb = dev_bytes(bases)
    tab = bzk.msm_table_build_c(bases, n, c)
    try:
        assert bzk.msm_table_window_count(tab) == (256 + c - 1) // c
        for m, seed in ((n, 1), (n - 777, 2)):
            scb = rand_scalars_bytes(m, seed) if seed == 1 else fr_bytes(_witness_like_scalars(m, 50 + c))
            want = co.msm_g1(hb[: 96 * m], scb, nthreads=co.ncpu())
            assert bzk.msm_table_run_dev(tab, to_dev(scb), m) == want
            assert bzk.msm_g1_dev(bases, to_dev(scb), m) == want
    finally:
        bzk.msm_table_free(tab)


@pytest.mark.parametrize("log_n", [13, 16])
def test_msm_two_level_reduce_same_bytes(bzk, co, log_n):
    """BZK_F_THROUGHPUT (what bzk_groth16_prove passes for its five overlapping MSMs): the bucket reduction in its two-level
    form - chunk sums B_k and totals T_k first, then the chunk offsets sum_k k T_k as a second, 8x smaller reduction - is the
    same group element as the one-level form and the oracle: G1 and G2, uniform and witness-like (de-duplicated) scalars, a
    window range, and the shared bucket set of a full table (per_win = 2^(c-1) / 8 chunks in one 'window')"""
    n = (1 << log_n) - 3
    g1 = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    g2 = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(900 + log_n, 0, n, g1)
    bzk.g2_synth_bases_dev(901 + log_n, 0, n, g2)
    h1, h2 = dev_bytes(g1), dev_bytes(g2)
    for seed, dedup in ((1, False), (2, True)):
        scb = rand_scalars_bytes(n, seed) if not dedup else fr_bytes(_witness_like_scalars(n, 70 + log_n))
        sc = to_dev(scb)
        want1 = co.msm_g1(h1, scb, nthreads=co.ncpu())
        want2 = co.msm_g2(h2, scb, nthreads=co.ncpu())
        assert bzk.msm_g1_dev(g1, sc, n, dedup=dedup, throughput=True) == want1
        assert bzk.msm_g1_dev(g1, sc, n, dedup=dedup) == want1
        assert bzk.msm_g2_dev(g2, sc, n, dedup=dedup, throughput=True) == want2
    scb = rand_scalars_bytes(n, 5)
    sc = to_dev(scb)
    W = bzk.msm_window_count(n)
    parts = [bzk.msm_g1_windows_dev(g1, sc, n, 0, W // 2, throughput=True), bzk.msm_g1_windows_dev(g1, sc, n, W // 2, W, throughput=True)]
    assert bzk.g1_sum(b"".join(parts)) == co.msm_g1(h1, scb, nthreads=co.ncpu())
    tab = bzk.msm_table_build_c(g1, n, log_n)
    try:
        assert bzk.msm_table_run_dev(tab, sc, n, throughput=True) == co.msm_g1(h1, scb, nthreads=co.ncpu())
    finally:
        bzk.msm_table_free(tab)
