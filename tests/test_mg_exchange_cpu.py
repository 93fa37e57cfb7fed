"""CPU suite: the REAL exchange of a process-per-GPU device group (bazuka_amd/csrc/mg_exchange.h - the shared-memory all-gather with its
(sequence, status) records that mg.hip itself uses between processes, the compaction into window order, the product's own host Horner) run by 2, 3
and 4 PROCESSES on a GPU-less box, the device stage replaced by window sums made from the CPU oracle (until round 6 only
stub groups and the torch-side `allgather_fold` ran on the CPU).  Harness: tests/host/mgx_check.cpp.

  window w of scalar k: the signed c-bit digit d_w(k) of bazuka_amd/csrc/msm_impl.cuh msm_digits_kernel (c = msm_window_bits(n), W = bzk_msm_window_count(n));
  window sum S_w = sum_i d_w(k_i) P_i  - here: the ORACLE's MSM over the digits;   result = sum_w 2^(c w) S_w = the oracle's MSM over the scalars.
What must hold: every rank returns the same 97 bytes = the oracle's; consecutive calls do not see each other's sums (double-buffered slots); a rank
whose local stage fails makes EVERY rank return an error for THAT call only (no hang, no stale result).  Partition: SURVEY.md 8e,
/root/reference/src/mpn/mod.rs:79-107 is the reference's replica split."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "host", "_mgx_check.so")
SRC = os.path.join(ROOT, "tests", "host", "mgx_check.cpp")
N = 300


def _build():
    dep = [SRC, os.path.join(ROOT, "bazuka_amd", "csrc", "mg_exchange.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in dep):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC, "-L" + os.path.join(ROOT, "bazuka_amd"), "-lbzk",
                               "-Wl,-rpath," + os.path.join(ROOT, "bazuka_amd"), "-Wl,--allow-shlib-undefined", "-lrt", "-lpthread"])
    return SO


def _lib():
    lib = C.CDLL(_build())
    lib.mgx_rank_run.restype = C.c_int32
    lib.mgx_rank_run.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p,
                                 C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
    lib.mgx_window_terms.restype = C.c_int32
    lib.mgx_window_terms.argtypes = [C.c_uint64]
    lib.mgx_window_bits.restype = C.c_int32
    lib.mgx_window_bits.argtypes = [C.c_uint64]
    return lib


def _rank(rank, world, uid, calls, sums, n_terms, fault_rank, fault_call, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("BZK_MG_TIMEOUT_S", "60")
    lib = _lib()
    out = C.create_string_buffer(97 * calls)
    st = (C.c_int32 * calls)()
    err = C.create_string_buffer(256)
    rc = lib.mgx_rank_run(uid, rank, world, N, calls, sums, n_terms, fault_rank, fault_call, out, st, err, 256)
    q.put((rank, rc, list(st), out.raw, err.value.decode()))


def _digits(scalars_canon, c, W):
    half, rows = 1 << (c - 1), []
    for k in scalars_canon:
        carry, row = 0, []
        for w in range(W):
            d = ((k >> (c * w)) & ((1 << c) - 1)) + carry
            if d > half:
                d, carry = d - (1 << c), 1
            else:
                carry = 0
            row.append(d)
        assert carry == 0
        rows.append(row)
    return rows


def _xyzz(pr, p97):
    one = pr.fp_to_mont_bytes(1)
    return (p97[:96] + one + one) if p97[96] == 0 else (bytes(48) + one + bytes(96))


def _window_terms(co, pr, bases, scalars_canon, c, W):
    """the c / 2 + 1 TERMS of every window's bucket set (msm_impl.cuh section 6b) from the oracle: with b = |d| - 1 the bucket of a non-zero digit d,
    D_j = sum_i sign(d_i) (bit_2j(b_i) + 2 bit_(2j+1)(b_i)) P_i  and the plain sum  sum_i sign(d_i) P_i  - MSMs over tiny scalars"""
    digs = _digits(scalars_canon, c, W)
    n_pairs, out = c // 2, b""
    for w in range(W):
        for t in range(n_pairs + 1):
            sc = b""
            for i in range(len(scalars_canon)):
                d = digs[i][w]
                if d == 0:
                    v = 0
                else:
                    b = abs(d) - 1
                    v = 1 if t == n_pairs else ((b >> (2 * t)) & 1) + 2 * ((b >> (2 * t + 1)) & 1)
                    v = v if d > 0 else -v
                sc += pr.fr_to_mont_bytes(v % pr.R_MOD)
            out += _xyzz(pr, co.msm_g1(bases, sc))
    return out


def _window_sums(co, pr, bases, scalars_canon, c, W):
    """all W window sums of one call as 192-byte standard-limb XYZZ points (X | Y | ZZ | ZZZ, Montgomery 2^384; ZZ = 0 <=> identity)"""
    half, one = 1 << (c - 1), pr.fp_to_mont_bytes(1)
    digs = []
    for k in scalars_canon:
        carry, row = 0, []
        for w in range(W):
            d = ((k >> (c * w)) & ((1 << c) - 1)) + carry
            if d > half:
                d, carry = d - (1 << c), 1
            else:
                carry = 0
            row.append(d)
        assert carry == 0
        digs.append(row)
    out = b""
    for w in range(W):
        sc = b"".join(pr.fr_to_mont_bytes(digs[i][w] % pr.R_MOD) for i in range(len(scalars_canon)))
        p = co.msm_g1(bases, sc)
        out += (p[:96] + one + one) if p[96] == 0 else (bytes(48) + one + bytes(96))
    return out


@pytest.fixture(scope="module")
def calls_data(co, pr):
    if not os.path.exists(os.path.join(ROOT, "bazuka_amd", "libbzk.so")):
        pytest.skip("libbzk.so not built")
    from bazuka_amd import lib as L
    from util import fr_list
    lib = _lib()
    c, W = lib.mgx_window_bits(N), L.load_library().bzk_msm_window_count(N)
    assert 4 <= c <= 16 and W == (256 + c - 1) // c
    assert lib.mgx_window_terms(N) in (0, c // 2 + 1) and lib.mgx_window_terms(1 << 20) == 9      # 2^20 points: c = 16, 8 digit terms + the plain sum
    bases = co.g1_bases(71, 0, N)
    sums, terms, want = b"", b"", []
    n_terms = c // 2 + 1
    for call in range(3):
        ks = fr_list(N, 7100 + call)
        if call == 1:
            ks[:4] = [0, 1, pr.R_MOD - 1, (1 << 254) + 12345]      # edge digits: zero scalar, top-window carry
        sums += _window_sums(co, pr, bases, ks, c, W)
        terms += _window_terms(co, pr, bases, ks, c, W)
        want.append(co.msm_g1(bases, b"".join(pr.fr_to_mont_bytes(k) for k in ks)))
    assert len(sums) == 3 * W * 192 and len(terms) == 3 * W * n_terms * 192
    return {0: sums, n_terms: terms}, want, n_terms


def _run_group(world, sums, n_terms=0, fault_rank=-1, fault_call=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = os.urandom(128)
    ps = [ctx.Process(target=_rank, args=(r, world, uid, 3, sums, n_terms, fault_rank, fault_call, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,terms", [(2, False), (3, False), (4, False), (2, True), (3, True), (4, True)])
def test_ranks_as_processes_gather_and_combine_to_the_oracles_bytes(calls_data, world, terms):
    """terms = False: one window sum per window (G2's exchange, G1's with BZK_MSM_BITSUM=0); True: the terms of every window's bucket set - what the ranks of a
    G1 group exchange since round 6 (1.7 KB per window at c = 16 instead of 192 bytes: the per-rank bucket reduction chain is gone)"""
    data, want, n_terms = calls_data
    sums, nt = (data[n_terms], n_terms) if terms else (data[0], 0)
    for rank, rc, st, out, err in _run_group(world, sums, nt):
        assert rc == 0 and st == [0, 0, 0], (rank, rc, st, err)
        for k in range(3):
            assert out[97 * k:97 * k + 97] == want[k], (rank, k)


def test_a_failed_rank_fails_that_call_on_every_rank_and_only_that_call(calls_data):
    data, want, n_terms = calls_data
    for rank, rc, st, out, err in _run_group(4, data[n_terms], n_terms, fault_rank=2, fault_call=2):
        assert rc == 0, (rank, err)
        assert st[0] == 0 and st[2] == 0 and st[1] != 0, (rank, st)
        assert out[:97] == want[0] and out[194:291] == want[2], rank      # the calls around it: right, and not the failed call's leftovers
        if rank != 2:
            assert "rank 2 failed its local stage" in err, (rank, err)
