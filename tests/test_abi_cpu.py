"""CPU suite: the C-ABI library loads and exports every symbol include/bzk.h declares; the host-run of
the __host__ __device__ field/curve headers (tests/host harness) matches the oracle limb for limb."""
import ctypes as C
import os
import random
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libbzk_exports_every_header_symbol():
    from bazuka_amd import lib as L
    hdr = open(os.path.join(ROOT, "include", "bzk.h")).read()
    declared = set(re.findall(r"\b(bzk_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"bzk_ctx"}
    assert len(declared) >= 30
    so = C.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(so, name), f"{name} declared in bzk.h but not exported by libbzk.so"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    L.load_library()


def test_flag_and_status_constants_match_the_header():
    """the ctypes driver's flag / status constants are the header's #defines (a flag added on one side only would be silently ignored)"""
    from bazuka_amd import lib as L
    hdr = open(os.path.join(ROOT, "include", "bzk.h")).read()
    defs = {k: int(v.rstrip("u")) for k, v in re.findall(r"#define\s+(BZK_[A-Z_]+)\s+\(?(-?\d+u?)\)?", hdr)}
    for name in ("BZK_F_CANONICAL", "BZK_F_DEDUP", "BZK_F_THROUGHPUT"):
        assert getattr(L, name) == defs[name], name
    flags = [defs[k] for k in defs if k.startswith("BZK_F_")]
    assert len(set(flags)) == len(flags) and all(f & (f - 1) == 0 for f in flags)  # distinct single bits
    assert L._flags(True, True, True) == defs["BZK_F_CANONICAL"] | defs["BZK_F_DEDUP"] | defs["BZK_F_THROUGHPUT"]
    assert defs["BZK_OK"] == 0 and defs["BZK_E_DEVICE"] == -3


def test_no_cpu_fallback_without_gpu():
    import torch
    from bazuka_amd import Bzk, BzkError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(BzkError):
        Bzk(0)


def test_status_strings():
    from bazuka_amd import load_library
    lib = load_library()
    assert lib.bzk_strerror(0) == b"ok"
    assert lib.bzk_strerror(-3) == b"device error"
    assert lib.bzk_abi_version() == 1


def test_staging_entries_refuse_bad_arguments_without_touching_a_device():
    """argument checks of the round-5 / 6 staging entries (no GPU needed: they return before any device call); synthesize refuses a record_matrices
    value outside {0, 1, BZK_SYNTH_DEFER}"""
    import ctypes as C
    from bazuka_amd import load_library
    lib = load_library()
    n = C.c_uint64(7)
    assert lib.bzk_staged_read(None, 0, None, 0, C.byref(n)) == -1 and n.value == 7     # BZK_E_ARG, nothing written
    assert lib.bzk_staged_wait(None) == -1
    lib.bzk_staged_free(None)                                                           # a no-op
    assert lib.bzk_r1cs_stage(None, None, None) == -1
    out = C.c_void_p()
    assert lib.bzk_mpn_work_synthesize(None, None, None, 0, 3, C.byref(out)) == -1


def test_host_g1_sum_matches_oracle(co):
    """bzk_g1_sum / bzk_g2_sum are host-side (no GPU): fold packed points like the multi-GPU combine."""
    from bazuka_amd import load_library
    lib = load_library()
    n = 5
    b = co.g1_bases(11, 0, n)
    packed = b"".join(b[96 * i:96 * i + 96] + b"\0" for i in range(n))
    packed += b"\0" * 48 + co.g1_generator()[48:96] + b"\1"  # an identity entry (y ignored)
    out = C.create_string_buffer(97)
    assert lib.bzk_g1_sum(packed, n + 1, out) == 0
    want = b[:96] + b"\0"
    for i in range(1, n):
        want = co.g1_add(want, b[96 * i:96 * i + 96] + b"\0")
    assert out.raw == want
    b2 = co.g2_bases(11, 0, n)
    packed2 = b"".join(b2[192 * i:192 * i + 192] + b"\0" for i in range(n))
    out2 = C.create_string_buffer(193)
    assert lib.bzk_g2_sum(packed2, n, out2) == 0
    want = b2[:192] + b"\0"
    for i in range(1, n):
        want = co.g2_add(want, b2[192 * i:192 * i + 192] + b"\0")
    assert out2.raw == want
    # P + (-P) = identity, packed as (0, R, inf=1)
    P = b[:96]
    from oracle import pyref as pr
    negP = P[:48] + pr.fp_to_mont_bytes(-pr.fp_from_mont_bytes(P[48:96]))
    assert lib.bzk_g1_sum(P + b"\0" + negP + b"\0", 2, out) == 0
    assert out.raw == pr.g1_to_bytes(None)


@pytest.fixture(scope="module")
def hc():
    so = os.path.join(ROOT, "tests", "host", "_hostcheck.so")
    if not os.path.exists(so):
        pytest.skip("tests/host/_hostcheck.so not built (build() compiles it)")
    return C.CDLL(so)


def _op(fn, o, a, b, n):
    out = C.create_string_buffer(n)
    assert fn(o, a, b, out) == 0
    return out.raw


def test_device_field_code_on_host_matches_oracle(hc, co, pr):
    rnd = random.Random(2)
    cases_r = [(rnd.randrange(pr.R_MOD), rnd.randrange(pr.R_MOD)) for _ in range(100)]
    cases_r += [(a, b) for a in (0, 1, pr.R_MOD - 1) for b in (0, 1, pr.R_MOD - 1)]
    for a, b in cases_r:
        A, B = pr.fr_to_mont_bytes(a), pr.fr_to_mont_bytes(b)
        for o in (0, 1, 2, 3, 4, 5, 6):
            assert _op(hc.hc_fr_op, o, A, B, 32) == co.fr_op(o, A, B), ("fr", o, a, b)
    cases_p = [(rnd.randrange(pr.P_MOD), rnd.randrange(pr.P_MOD)) for _ in range(100)]
    cases_p += [(a, b) for a in (0, 1, pr.P_MOD - 1) for b in (0, 1, pr.P_MOD - 1)]
    for a, b in cases_p:
        A, B = pr.fp_to_mont_bytes(a), pr.fp_to_mont_bytes(b)
        for o in (0, 1, 2, 3, 4, 5, 6):
            assert _op(hc.hc_fp_op, o, A, B, 48) == co.fp_op(o, A, B), ("fp", o, a, b)


def test_device_curve_code_on_host_matches_oracle(hc, co, pr):
    rnd = random.Random(3)
    n = 10
    bases = co.g1_bases(7, 0, n)
    ks = [rnd.randrange(1, 2 ** 32) for _ in range(n)]
    ks[0], ks[1] = 1, 0
    sc = b"".join(k.to_bytes(32, "little") for k in ks)
    out = C.create_string_buffer(97)
    hc.hc_g1_lincomb(bases, (C.c_uint32 * n)(*ks), n, out)
    assert out.raw == co.msm_g1(bases, sc, mont=False, naive=True)
    b2 = co.g2_bases(7, 0, n)
    out2 = C.create_string_buffer(193)
    hc.hc_g2_lincomb(b2, (C.c_uint32 * n)(*ks), n, out2)
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    P, Q = bases[:96], bases[96:192]
    negP = P[:48] + pr.fp_to_mont_bytes(-pr.fp_from_mont_bytes(P[48:96]))
    hc.hc_g1_sum_mixed(P + P + Q + negP + negP + P, 6, out)  # doubling + cancellation paths
    assert out.raw == co.g1_add(P + b"\0", Q + b"\0")
    hc.hc_g1_sum_mixed(P + negP, 2, out)
    assert out.raw[96] == 1


def test_reduced_radix_fields_on_host_match_oracle(hc, co, pr):
    """bzk_fp28.cuh / bzk_fr29.cuh (the device arithmetic) executed on the CPU with their bound assertions on"""
    rnd = random.Random(5)
    pv = [0, 1, pr.P_MOD - 1, pr.P_MOD - 2, 2 ** 380] + [rnd.randrange(pr.P_MOD) for _ in range(100)]
    for a in pv:
        A = pr.fp_to_mont_bytes(a)
        out = C.create_string_buffer(48)
        hc.hc_fp28_roundtrip(A, out)
        assert out.raw == A
    for _ in range(200):
        a, b = rnd.choice(pv), rnd.choice(pv)
        out = C.create_string_buffer(48)
        hc.hc_fp28_mul(pr.fp_to_mont_bytes(a), pr.fp_to_mont_bytes(b), out)
        assert out.raw == pr.fp_to_mont_bytes(a * b % pr.P_MOD)
    # the dedicated square (105 + 196 mads) is limb-identical to mul(a, a) (392 mads), also on weakly reduced inputs
    for a in pv:
        for grow in (0, 1, 2):  # limbs up to ~2^30: beyond what the curve formulas feed it (L <= 29.6)
            out = C.create_string_buffer(48)
            assert hc.hc_fp28_sqr_equals_mul(pr.fp_to_mont_bytes(a), grow, out) == 0
            assert out.raw == pr.fp_to_mont_bytes(a * a * 4 ** grow % pr.P_MOD)
    rv = [0, 1, pr.R_MOD - 1, 2 ** 254] + [rnd.randrange(pr.R_MOD) for _ in range(100)]
    for a in rv:  # the Fr29 square (the S-box's x^2 and x^4)
        for grow in (0, 1):
            out = C.create_string_buffer(32)
            assert hc.hc_fr29_sqr_equals_mul(pr.fr_to_mont_bytes(a), grow, out) == 0
            assert out.raw == pr.fr_to_mont_bytes(a * a * 4 ** grow % pr.R_MOD)
    for _ in range(200):
        a, b = rnd.choice(rv), rnd.choice(rv)
        out = C.create_string_buffer(32)
        hc.hc_fr29_mul(pr.fr_to_mont_bytes(a), pr.fr_to_mont_bytes(b), out)
        assert out.raw == pr.fr_to_mont_bytes(a * b % pr.R_MOD)
    for _ in range(20):
        a, b, rounds = rnd.randrange(pr.P_MOD), rnd.randrange(pr.P_MOD), rnd.randrange(1, 30)
        out = C.create_string_buffer(48)
        assert hc.hc_fp28_reduce_chain(pr.fp_to_mont_bytes(a), pr.fp_to_mont_bytes(b), rounds, out) == 0
        x = a
        for _i in range(rounds):
            x = (6 * x + b) % pr.P_MOD
        assert out.raw == pr.fp_to_mont_bytes(x)


def test_reduced_radix_curves_on_host_match_oracle(hc, co, pr):
    rnd = random.Random(6)
    n = 10
    bases = co.g1_bases(7, 0, n)
    ks = [rnd.randrange(1, 2 ** 32) for _ in range(n)]
    ks[0], ks[1], ks[2], ks[3] = 1, 0, 2, 0xFFFFFFFF
    neg = bytes(rnd.randrange(2) for _ in range(n))
    sc = b"".join(((pr.R_MOD - k) % pr.R_MOD if neg[i] else k).to_bytes(32, "little") for i, k in enumerate(ks))
    out = C.create_string_buffer(97)
    hc.hc_g1x28_lincomb(bases, (C.c_uint32 * n)(*ks), neg, n, out)
    assert out.raw == co.msm_g1(bases, sc, mont=False, naive=True)
    b2 = co.g2_bases(7, 0, n)
    out2 = C.create_string_buffer(193)
    hc.hc_g2x28_lincomb(b2, (C.c_uint32 * n)(*ks), neg, n, out2)
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    # the memory-operand addition of the G2 tail kernels (xyzz_add_mem): same points as xyzz_add, incl. acc / q identity,
    # P + P (doubling) and P - P (cancellation), with the reduced-radix bound assertions on
    assert hc.hc_g2x28_lincomb_mem(b2, (C.c_uint32 * n)(*ks), neg, n, 0, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    assert hc.hc_g2x28_lincomb_mem(b2, (C.c_uint32 * n)(*ks), neg, n, 2, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    assert hc.hc_g2x28_lincomb_mem(b2, (C.c_uint32 * n)(*ks), neg, n, 1, out2) == 0
    sc3 = b"".join((3 * int.from_bytes(sc[32 * i:32 * i + 32], "little") % pr.R_MOD).to_bytes(32, "little") for i in range(n))
    assert out2.raw == co.msm_g2(b2, sc3, mont=False, naive=True)
    # the static-bound general addition / doubling of the G2 tail kernels (g2x28::add_mem, g2x28::dbl): same three modes
    assert hc.hc_g2x28_lincomb_mem_fast(b2, (C.c_uint32 * n)(*ks), neg, n, 0, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    assert hc.hc_g2x28_lincomb_mem_fast(b2, (C.c_uint32 * n)(*ks), neg, n, 2, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    assert hc.hc_g2x28_lincomb_mem_fast(b2, (C.c_uint32 * n)(*ks), neg, n, 1, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc3, mont=False, naive=True)
    # g2x28::add_mixed (what msm_accumulate<G2> runs): long signed chains, doubling and cancellation through the mixed add, and its
    # result handed to the generic general addition; bound assertions on, invariants (X, Y < 3p, normalised) checked by the harness
    m2 = 120
    b4 = co.g2_bases(11, 0, m2)
    negs2 = bytes(rnd.randrange(2) for _ in range(m2))
    sc4 = b"".join(((pr.R_MOD - 1) if negs2[i] else 1).to_bytes(32, "little") for i in range(m2))
    assert hc.hc_g2x28_sum_mixed_fast(b4, negs2, m2, 0, out2) == 0
    assert out2.raw == co.msm_g2(b4, sc4, mont=False, naive=True)
    assert hc.hc_g2x28_sum_mixed_fast(b4, negs2, m2, 2, out2) == 0  # bases that are Fp2 product outputs (table entries, group sums)
    assert out2.raw == co.msm_g2(b4, sc4, mont=False, naive=True)
    assert hc.hc_g2x28_sum_mixed_fast(b4, negs2, m2, 1, out2) == 0
    sc4x2 = b"".join((2 * int.from_bytes(sc4[32 * i:32 * i + 32], "little") % pr.R_MOD).to_bytes(32, "little") for i in range(m2))
    assert out2.raw == co.msm_g2(b4, sc4x2, mont=False, naive=True)
    Q2 = b4[:192]
    assert hc.hc_g2x28_sum_mixed_fast(Q2 + Q2, bytes([0, 0]), 2, 0, out2) == 0
    assert out2.raw == co.msm_g2(Q2, (2).to_bytes(32, "little"), mont=False, naive=True)  # doubling through the mixed add
    assert hc.hc_g2x28_sum_mixed_fast(Q2 + Q2, bytes([1, 1]), 2, 0, out2) == 0
    assert out2.raw == co.msm_g2(Q2, (pr.R_MOD - 2).to_bytes(32, "little"), mont=False, naive=True)
    assert hc.hc_g2x28_sum_mixed_fast(Q2 + Q2, bytes([0, 1]), 2, 0, out2) == 0
    assert out2.raw[192] == 1  # cancellation -> identity
    assert hc.hc_g2x28_sum_mixed_fast(Q2 + Q2 + Q2, bytes([0, 1, 0]), 3, 0, out2) == 0
    assert out2.raw == co.msm_g2(Q2, (1).to_bytes(32, "little"), mont=False, naive=True)  # identity accumulator takes a point again
    # ---- the pair-lane G2 arithmetic of round 5 (bzk_g2pair.cuh: what msm_accumulate_g2pair_kernel runs): the SAME source on two host
    # threads per pair (the DPP exchange a rendezvous), bound assertions on, the stored-point discipline checked after every addition
    assert hc.hc_g2p_sum_mixed(b4, negs2, m2, 0, out2) == 0
    assert out2.raw == co.msm_g2(b4, sc4, mont=False, naive=True)
    assert hc.hc_g2p_sum_mixed(b4, negs2, m2, 2, out2) == 0   # bases that are one-lane Fp2 product outputs (group sums, table entries)
    assert out2.raw == co.msm_g2(b4, sc4, mont=False, naive=True)
    assert hc.hc_g2p_sum_mixed(b4, negs2, m2, 1, out2) == 0   # the pair's sum consumed by the one-lane generic addition
    assert out2.raw == co.msm_g2(b4, sc4x2, mont=False, naive=True)
    assert hc.hc_g2p_sum_mixed(Q2 + Q2, bytes([0, 0]), 2, 0, out2) == 0
    assert out2.raw == co.msm_g2(Q2, (2).to_bytes(32, "little"), mont=False, naive=True)  # doubling through the mixed add (dbl_affine)
    assert hc.hc_g2p_sum_mixed(Q2 + Q2, bytes([1, 1]), 2, 0, out2) == 0
    assert out2.raw == co.msm_g2(Q2, (pr.R_MOD - 2).to_bytes(32, "little"), mont=False, naive=True)
    assert hc.hc_g2p_sum_mixed(Q2 + Q2, bytes([0, 1]), 2, 0, out2) == 0
    assert out2.raw[192] == 1  # cancellation -> identity
    assert hc.hc_g2p_sum_mixed(Q2 + Q2 + Q2, bytes([0, 1, 0]), 3, 0, out2) == 0
    assert out2.raw == co.msm_g2(Q2, (1).to_bytes(32, "little"), mont=False, naive=True)  # identity accumulator takes a point again
    assert hc.hc_g2p_sum_mixed(b"", b"", 0, 0, out2) == 0 and out2.raw[192] == 1
    # g2p::add / g2p::dbl (general addition and doubling on a pair): scalar multiples by double-and-add, doubling and cancellation branches
    assert hc.hc_g2p_lincomb(b2, (C.c_uint32 * n)(*ks), neg, n, 0, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    assert hc.hc_g2p_lincomb(b2, (C.c_uint32 * n)(*ks), neg, n, 2, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc, mont=False, naive=True)
    assert hc.hc_g2p_lincomb(b2, (C.c_uint32 * n)(*ks), neg, n, 1, out2) == 0
    assert out2.raw == co.msm_g2(b2, sc3, mont=False, naive=True)
    m = 150  # long chains of mixed adds: the weak-reduction bounds must hold indefinitely
    b3 = co.g1_bases(9, 0, m)
    negs = bytes(rnd.randrange(2) for _ in range(m))
    hc.hc_g1x28_sum_mixed(b3, negs, m, out)
    sc = b"".join(((pr.R_MOD - 1) if negs[i] else 1).to_bytes(32, "little") for i in range(m))
    assert out.raw == co.msm_g1(b3, sc, mont=False, naive=True)
    P = bases[:96]
    hc.hc_g1x28_sum_mixed(P + P, bytes([0, 0]), 2, out)
    assert out.raw == co.g1_add(P + b"\0", P + b"\0")  # doubling through the mixed add
    hc.hc_g1x28_sum_mixed(P + P, bytes([0, 1]), 2, out)
    assert out.raw[96] == 1  # cancellation


def test_poseidon29_device_function_on_host_matches_oracle(hc, co, pr):
    rnd = random.Random(7)
    for arity in range(1, 8):
        consts = co.poseidon_params(arity + 1)
        rp = 56 if arity + 1 <= 5 else 57
        cases = [list(range(arity)), [pr.R_MOD - 1] * arity] + [[rnd.randrange(pr.R_MOD) for _ in range(arity)] for _ in range(3)]
        for inp in cases:
            ib = b"".join(pr.fr_to_mont_bytes(x) for x in inp)
            out = C.create_string_buffer(32)
            assert hc.hc_poseidon29(ib, arity, consts, len(consts) // 32, 8, rp, out) == 0
            assert out.raw == co.poseidon_batch(ib, arity), arity


def test_host_fp64_field_and_curve_match_oracle(hc, co, pr):
    """bazuka_amd/csrc/host_fp64.h (64-bit-limb host field of the window Horner, the to-affine inversion and the proof assembly, round 3)
    against the oracle: field operations on random and edge values, then 255-bit scalar multiples + one addition through the generic XYZZ
    formulas instantiated on it, G1 and G2 (Fp2 by Karatsuba), packed exactly as libbzk packs a result."""
    rnd = random.Random(64)
    cases = [(rnd.randrange(pr.P_MOD), rnd.randrange(pr.P_MOD)) for _ in range(200)]
    cases += [(a, b) for a in (0, 1, 2, pr.P_MOD - 1, pr.P_MOD - 2, (1 << 380) - 1) for b in (0, 1, pr.P_MOD - 1, (1 << 64) - 1, 1 << 64)]
    for a, b in cases:
        A, B = pr.fp_to_mont_bytes(a), pr.fp_to_mont_bytes(b)
        for o in (0, 1, 2, 3, 6):
            assert _op(hc.hc_hfp_op, o, A, B, 48) == co.fp_op(o, A, B), ("hfp", o, a, b)
        assert _op(hc.hc_hfp_op, 7, A, B, 48) == co.fp_op(0, A, A), ("hfp dbl", a)
        assert _op(hc.hc_hfp_op, 8, A, B, 48) == co.fp_op(2, A, A), ("hfp sqr", a)
    n = 4
    b1, b2 = co.g1_bases(11, 0, n + 1), co.g2_bases(11, 0, n + 1)
    ks = [rnd.randrange(pr.R_MOD) for _ in range(n - 2)] + [1, pr.R_MOD - 1]
    for i, k in enumerate(ks):
        kw = (C.c_uint32 * 8)(*[(k >> (32 * j)) & 0xffffffff for j in range(8)])
        sc = k.to_bytes(32, "little") + (1).to_bytes(32, "little")
        P, Q = b1[96 * i:96 * (i + 1)], b1[96 * (i + 1):96 * (i + 2)]
        out = C.create_string_buffer(97)
        assert hc.hc_hfp_g1_mul_add(P, kw, Q, out) == 0
        assert out.raw == co.msm_g1(P + Q, sc, mont=False, naive=True), ("g1", k)
        P2, Q2 = b2[192 * i:192 * (i + 1)], b2[192 * (i + 1):192 * (i + 2)]
        out2 = C.create_string_buffer(193)
        assert hc.hc_hfp_g2_mul_add(P2, kw, Q2, out2) == 0
        assert out2.raw == co.msm_g2(P2 + Q2, sc, mont=False, naive=True), ("g2", k)
    # k P - P... the identity packs as (0, 1, flag): (r - 1) P + P
    kw = (C.c_uint32 * 8)(*[((pr.R_MOD - 1) >> (32 * j)) & 0xffffffff for j in range(8)])
    out = C.create_string_buffer(97)
    assert hc.hc_hfp_g1_mul_add(b1[:96], kw, b1[:96], out) == 0
    assert out.raw == pr.g1_to_bytes(None)


def test_host_fr64_product_and_dot_match_oracle(hc, co, pr):
    """bazuka_amd/csrc/host_fr64.h (witness generator, round 3): the unrolled Fr product and the dot product with ONE Montgomery reduction
    (column-wise accumulation of up to 32 512-bit products) against the oracle's field arithmetic - random values and the extremes that
    stress the final reduction (every operand r - 1: the accumulated value is largest there)."""
    rnd = random.Random(464)
    M = pr.R_MOD
    for _ in range(300):
        a, b = rnd.randrange(M), rnd.randrange(M)
        A, B = pr.fr_to_mont_bytes(a), pr.fr_to_mont_bytes(b)
        out = C.create_string_buffer(32)
        assert hc.hc_hfr_mul(A, B, out) == 0
        assert out.raw == co.fr_op(2, A, B)
    for n in list(range(0, 33)):
        for mode in ("rand", "max", "mixed"):
            if mode == "rand":
                xs = [rnd.randrange(M) for _ in range(n)]; ys = [rnd.randrange(M) for _ in range(n)]
            elif mode == "max":
                xs = [M - 1] * n; ys = [M - 1] * n
            else:
                xs = [rnd.choice((0, 1, M - 1, rnd.randrange(M))) for _ in range(n)]; ys = [rnd.choice((0, 1, M - 1, rnd.randrange(M))) for _ in range(n)]
            A = b"".join(pr.fr_to_mont_bytes(x) for x in xs)
            B = b"".join(pr.fr_to_mont_bytes(y) for y in ys)
            out = C.create_string_buffer(32)
            assert hc.hc_hfr_dot(A, B, n, out) == 0
            want = sum(x * y for x, y in zip(xs, ys)) % M
            assert out.raw == pr.fr_to_mont_bytes(want), (n, mode)


def test_host_ifma_mds_product_and_satisfaction_scan(hc, pr):
    """bazuka_amd/csrc/host_fr_ifma.h (round 4): the dense MDS product of a Poseidon round with its rows in AVX-512 IFMA lanes (five 52-bit
    limbs, one Montgomery reduction by 2^260 on a table scaled by 16) and the eight-rows-at-a-time satisfaction scan a_k b_k = c_k: equal to
    the 64-bit scalar forms AND to plain integer arithmetic for every width 1..17, with the operands that stress the reduction (r - 1
    everywhere, zeros, small values).  On a CPU without the instructions both entries take their scalar form (the test then compares that
    with the integers); BZK_HOST_IFMA=0 forces it."""
    import ctypes
    hc.hc_products_first_mismatch.restype = ctypes.c_long
    rnd = random.Random(808)
    M, F = pr.R_MOD, pr.fr_to_mont_bytes
    took_ifma = set()
    for t in range(1, 18):
        for case in range(12):
            pick = [lambda: rnd.randrange(M), lambda: M - 1, lambda: rnd.choice((0, 1, 2, M - 1, M - 2)), lambda: rnd.randrange(1 << 64)][case % 4]
            mat, vec = [pick() for _ in range(t * t)], [pick() for _ in range(t)]
            mb, sb = b"".join(F(x) for x in mat), b"".join(F(x) for x in vec)
            o0, o1 = C.create_string_buffer(32 * t), C.create_string_buffer(32 * t)
            took_ifma.add(hc.hc_mds_mul(t, mb, sb, 0, o0))
            assert hc.hc_mds_mul(t, mb, sb, 1, o1) == 0 and o0.raw == o1.raw, (t, case)
            assert o0.raw == b"".join(F(sum(mat[j * t + k] * vec[k] for k in range(t)) % M) for j in range(t)), (t, case)
    assert took_ifma <= {0, 1} and (1 in took_ifma) == bool(hc.hc_host_ifma_available())
    for n in (0, 1, 7, 8, 9, 64, 1001):
        for bad_at in (None, 0, n // 2, n - 1):
            if bad_at is not None and n == 0:
                continue
            a = [rnd.choice((0, 1, M - 1, rnd.randrange(M), rnd.randrange(M))) for _ in range(n)]
            b = [rnd.choice((0, 1, M - 1, rnd.randrange(M), rnd.randrange(M))) for _ in range(n)]
            c = [x * y % M for x, y in zip(a, b)]
            if bad_at is not None:
                c[bad_at] = (c[bad_at] + rnd.choice((1, M - 1, 1 << 200))) % M
            got = hc.hc_products_first_mismatch(b"".join(F(x) for x in a), b"".join(F(x) for x in b), b"".join(F(x) for x in c), n)
            assert got == (-1 if bad_at is None else bad_at), (n, bad_at, got)
    for x in (0, 1, 2, M - 1, rnd.randrange(M), rnd.randrange(M)):
        out = C.create_string_buffer(32)
        assert hc.hc_hfr_inv(F(x), out) == 0 and out.raw == F(pow(x, M - 2, M)), x


def test_mg_probe_without_a_device_reports_nothing_usable():
    """bzk_mg_probe creates nothing and computes nothing: on a GPU-less host both capability bits are clear (the GPU suite sees 1 or 3)"""
    from bazuka_amd import mg_probe
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    assert mg_probe(0) == 0 and mg_probe(-1) == 0


def test_header_is_plain_c99_and_cxx11(tmp_path):
    """the boundary is a C ABI: include/bzk.h must compile as C99 (-pedantic) and as C++11 on its own - what a cgo / JNI / Rust-bindgen
    consumer feeds it to"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "bzk.h"\nint main(void) { return (int)sizeof(bzk_params_desc) == 0; }\n')
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror"], ["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-x", "c++"]):
        subprocess.check_call(cmd + ["-I", os.path.join(root, "include"), "-c", str(src), "-o", str(tmp_path / "hdr.o")])


def test_ctypes_signatures_agree_with_the_header_prototypes():
    """argument COUNT and argument CLASS (pointer / 64-bit / 32-bit / double) of every entry of bazuka_amd.lib.SIGNATURES against the prototype in
    include/bzk.h (parsed by the generator of the Rust raw layer): ctypes would silently pass a 32-bit value where the C side reads 64 bits, or
    drop / invent an argument, and only a GPU-side call would notice."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_sys as g
    from bazuka_amd import lib as L
    _, opaque, structs, funcs, _ = g.parse(open(os.path.join(ROOT, "include", "bzk.h")).read())
    protos = {name: (ret, params) for name, ret, params in funcs}
    assert set(protos) == set(L.SIGNATURES), set(protos) ^ set(L.SIGNATURES)

    def c_class(ty):
        ty = ty.strip()
        if "*" in ty:
            return "ptr"
        base = ty.replace("const", "").strip()
        return {"uint64_t": "i64", "int64_t": "i64", "size_t": "i64", "uint32_t": "i32", "int32_t": "i32", "int": "i32", "unsigned": "i32",
                "double": "f64", "void": "void", "uint8_t": "i8"}[base]

    def py_class(t):
        if t is None:
            return "void"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "_type_") and issubclass(t, (C._Pointer, C.Array)):
            return "ptr"
        if isinstance(t, type) and issubclass(t, C._Pointer):
            return "ptr"
        return {8: "i64", 4: "i32", 1: "i8"}[C.sizeof(t)] if t is not C.c_double else "f64"

    bad = []
    for name, (res, args) in L.SIGNATURES.items():
        ret, params = protos[name]
        want = [c_class(ty) for _, ty in params]
        got = [py_class(a) for a in args]
        if want != got:
            bad.append((name, want, got))
        rc, rp = c_class(ret), py_class(res)
        if rc != rp and not (rc == "ptr" and rp == "ptr"):
            bad.append((name, "returns " + rc, rp))
    assert bad == []


def test_a_plain_c_program_links_against_the_library_and_uses_it(tmp_path, co):
    """tests/host/abi_consumer.c: compiled as C99 -pedantic against include/bzk.h, linked against bazuka_amd/libbzk.so, run without a GPU - status
    strings, SHA3, the work decoder and the verifier called the way a cgo / Rust-FFI binding would call them (plain pointers and lengths, files in,
    verdicts out).  The proof and key are made by the CPU oracle, the work by the host builder."""
    import shutil
    import subprocess
    from bazuka_amd import lib as L
    from util import fr_bytes, fr_list, log2_ceil, r1cs_to_csr, synth_r1cs
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = tmp_path / "abi_consumer"
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "abi_consumer.c"), "-o", str(exe), "-L", libdir, "-lbzk",
                           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"])
    r1 = synth_r1cs(30, n_in=4, seed=4711)
    A, B, Cm = r1cs_to_csr(co, r1)
    params = co.groth16_setup(A, B, Cm, r1["n_in"], r1["n_aux"], log2_ceil(len(r1["rows"])), fr_bytes(fr_list(5, 79)))
    zb = fr_bytes(r1["z"])
    az, bz, cz = co.r1cs_eval(A, B, Cm, zb)
    rs = fr_bytes(fr_list(2, 12))
    proof = co.groth16_prove(params, zb, az, bz, cz, rs[:32], rs[32:])
    vkb = params["vk"] + (len(params["ic"]) // 97).to_bytes(8, "little") + params["ic"]
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        vks = [bytes.fromhex(v) for v in json.load(f)["verifying_keys_bincode_hex"]]
    Z = bytes.fromhex("01" + "00" * 31)
    w = L.MpnWorld(3, 3)
    for i in range(2):
        w.add_account(i, b"acct%d" % i, Z, 10 ** 9)
    w.set_height(5)
    w.push_deposit(0, Z, 1000)
    work = w.make_work(0, vks, 100).encode()
    files = {"vk.bin": vkb, "inputs.bin": fr_bytes(r1["z"][1:4]), "proof.bin": proof, "work.bin": work, "prover.bin": bytes(range(1, 33))}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    p = subprocess.run([str(exe)] + [str(tmp_path / n) for n in files], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all checks hold" in p.stdout and "groth16_verify(valid) = 1" in p.stdout and "kind 0" in p.stdout
