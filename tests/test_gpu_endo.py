"""GPU parity of the endomorphism form of the MSM (round 4; bazuka_amd/csrc/bzk_endo.cuh, msm_impl.cuh section 1b): a resident base
set carries its images X^m P and whole-MSM calls split every scalar into 2 (G1) / 4 (G2) signed sub-scalars whose windows share
8 / 4 bucket sets.  The result must be the same group element - the same 97 / 193 bytes - as
    the CPU oracle's Pippenger,   the plain form (a context created under BZK_MSM_NO_ENDO=1),   the per-call pipeline on raw bases
for uniform scalars, witness-like vectors under de-duplication (group sums get their images per call), scalars that sit on the
split's edges (digits at X/2, X/2 + 1, X - 1 in every position, 0, 1, r - 1, values >= r in canonical input), every window size
the planner can pick (small n: c < 16, ragged top windows), and the 2^20-point BASELINE size."""
import pytest
import torch

from oracle import pyref as pr
from util import dev_bytes, fr_bytes, fr_list, rand_scalars_bytes, to_dev

pytestmark = pytest.mark.gpu
X, R = pr.BLS_X, pr.R_MOD


@pytest.fixture(scope="module")
def plain(bzk):
    """a second context on the same device whose resident sets carry no images (plain 16-window form)"""
    import os
    from bazuka_amd import Bzk
    old = os.environ.get("BZK_MSM_NO_ENDO")
    os.environ["BZK_MSM_NO_ENDO"] = "1"
    try:
        ctx = Bzk(0)
    finally:
        if old is None:
            del os.environ["BZK_MSM_NO_ENDO"]
        else:
            os.environ["BZK_MSM_NO_ENDO"] = old
    yield ctx
    ctx.close()


def _edge_scalars(n):
    vals = [0, 1, 2, R - 1, R - 2, X, X - 1, X + 1, X // 2, X // 2 + 1, X * X, X * X - 1, X * X // 2, X * X // 2 + 1, X ** 3, X ** 3 - 1]
    for d in (X // 2, X // 2 + 1, X - 1):
        vals += [d * X ** m % R for m in range(4)]
        vals.append((d + d * X + d * X ** 2 + d * X ** 3) % R)
    vals += [(1 << k) % R for k in (15, 16, 17, 63, 64, 127, 128, 254)]
    rest = fr_list(n - len(vals), 4242)
    return (vals + rest)[:n]


def _witness_like(n, seed):
    vals = fr_list(n, seed)
    for i in range(0, n, 3):
        vals[i] = vals[(i * 7 + 1) % n]
    for i in range(1, n, 5):
        vals[i] = i % 2
    return fr_bytes(vals)


@pytest.mark.parametrize("g2", [False, True])
@pytest.mark.parametrize("n", [1, 3, 100, 3000, 40000])
def test_endo_form_equals_plain_form_and_oracle(bzk, plain, co, n, g2):
    bases = (co.g2_bases if g2 else co.g1_bases)(81, 0, n, nthreads=co.ncpu())
    msm = co.msm_g2 if g2 else co.msm_g1
    db = to_dev(bases)
    torch.cuda.synchronize()
    h, hp = bzk.msm_bases_load_dev(db, n, g2=g2), plain.msm_bases_load_dev(db, n, g2=g2)
    # the endomorphism form is taken by whole-MSM calls with the throughput hint (the prover's; msm_policy.cuh ENDO_DEFAULT = 2)
    for name, scb, kw in (("uniform", rand_scalars_bytes(n, n + 7), {"throughput": True}),
                          ("edges", fr_bytes(_edge_scalars(n)), {"throughput": True}),
                          ("witness-like + dedup", _witness_like(n, 13), {"dedup": True, "throughput": True}),
                          ("witness-like, latency form (plain windows on both contexts)", _witness_like(n, 14), {"dedup": True})):
        sc = to_dev(scb)
        torch.cuda.synchronize()
        want = msm(bases, scb, nthreads=co.ncpu())
        assert bzk.msm_bases_run_dev(h, sc, n, g2=g2, **kw) == want, (name, "endo")
        assert plain.msm_bases_run_dev(hp, sc, n, g2=g2, **kw) == want, (name, "plain")
    # canonical (non-Montgomery) input incl. values >= r: the split reduces them first
    can = [R + 5, 2 * R + 1, (1 << 256) - 1, R, 7][:n] + fr_list(max(0, n - 5), 5)
    scb = b"".join((v % (1 << 256)).to_bytes(32, "little") for v in can[:n])
    sc = to_dev(scb)
    torch.cuda.synchronize()
    want = msm(bases, fr_bytes([v % R for v in can[:n]]), nthreads=co.ncpu())
    assert bzk.msm_bases_run_dev(h, sc, n, g2=g2, canonical=True, throughput=True) == want
    # a prefix of the set (the images of the set are strided by the SET's size, not by the call's n)
    m = max(1, n // 3)
    scm = to_dev(rand_scalars_bytes(m, 99))
    torch.cuda.synchronize()
    assert bzk.msm_bases_run_dev(h, scm, m, g2=g2, throughput=True) == msm(bases[:m * (192 if g2 else 96)], dev_bytes(scm), nthreads=co.ncpu())
    bzk.msm_bases_free(h)
    plain.msm_bases_free(hp)


def test_endo_form_2p20_g1_and_g2(bzk, plain, co):
    n = 1 << 20
    for g2 in (False, True):
        d = torch.empty(n * (192 if g2 else 96), dtype=torch.uint8, device="cuda")
        (bzk.g2_synth_bases_dev if g2 else bzk.g1_synth_bases_dev)(2024, 0, n, d)
        bzk.sync()
        scb = rand_scalars_bytes(n, 2024)
        sc = to_dev(scb)
        torch.cuda.synchronize()
        want = (co.msm_g2 if g2 else co.msm_g1)(dev_bytes(d), scb, nthreads=co.ncpu())
        h = bzk.msm_bases_load_dev(d, n, g2=g2)
        assert bzk.msm_bases_run_dev(h, sc, n, g2=g2, throughput=True) == want
        assert bzk.msm_bases_run_dev(h, sc, n, g2=g2, dedup=True, throughput=True) == want
        assert bzk.msm_bases_run_dev(h, sc, n, g2=g2) == want      # latency form: plain windows over the same set
        bzk.msm_bases_free(h)
        del d


def test_the_endomorphism_form_is_the_path_taken(bzk, plain, co):
    """no silent fall-back: a throughput call over a set with images runs the split digits kernel, a latency-form call and a set
    without images do not (the launch labels of the per-kernel event table say which)"""
    n = 30000
    db = to_dev(co.g1_bases(5, 0, n, nthreads=co.ncpu()))
    sc = to_dev(rand_scalars_bytes(n, 5))
    torch.cuda.synchronize()

    scw = to_dev(_witness_like(n, 3))   # repeats: the de-duplication has group sums to make images of
    torch.cuda.synchronize()

    def labels(ctx, h, **kw):
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.msm_bases_run_dev(h, scw if kw.get("dedup") else sc, n, **kw)
        out = set(ctx.prof_dump())
        ctx.prof_enable(False)
        return out

    h, hp = bzk.msm_bases_load_dev(db, n), plain.msm_bases_load_dev(db, n)
    assert "msm_digits_endo" in labels(bzk, h, throughput=True)
    assert "dedup_images" in labels(bzk, h, throughput=True, dedup=True)
    assert "msm_digits_endo" not in labels(bzk, h) and "msm_digits" in labels(bzk, h)
    assert "msm_digits_endo" not in labels(plain, hp, throughput=True)
    bzk.msm_bases_free(h)
    plain.msm_bases_free(hp)
