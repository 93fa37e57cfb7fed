"""GPU parity of row (e) behind the C ABI: resident base sets (bzk_msm_g*_bases_*) and device groups (bzk_mg_*).
One GPU is available to the tests, so groups list device 0 several times (one context + host thread per entry, HOST / PEER
exchange) or run as several processes sharing it (shared-memory exchange); the RCCL transport is exercised on a one-rank
communicator.  Everything must reproduce the single-GPU result bytes, which are checked against the CPU oracle."""
import os
import subprocess
import sys

import pytest
import torch

from util import dev_bytes, fr_bytes, fr_list, rand_scalars_bytes, to_dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _witness_like(n, seed):
    """a scalar vector that repeats itself like a Groth16 assignment: bits, small values, pairs"""
    vals = fr_list(n, seed)
    for i in range(0, n, 3):
        vals[i] = vals[(i * 7 + 1) % n]
    for i in range(1, n, 5):
        vals[i] = i % 2
    return fr_bytes(vals)


@pytest.mark.parametrize("n", [1, 100, 5000, 70000])
def test_bases_handle_equals_raw_call_and_oracle(bzk, co, n):
    bases = co.g1_bases(71, 0, n, nthreads=co.ncpu())
    db, sc = to_dev(bases), to_dev(rand_scalars_bytes(n, n + 1))
    torch.cuda.synchronize()
    h = bzk.msm_bases_load_dev(db, n)
    got = bzk.msm_bases_run_dev(h, sc, n)
    assert got == bzk.msm_g1_dev(db, sc, n)
    assert got == co.msm_g1(bases, dev_bytes(sc), nthreads=co.ncpu())
    # a prefix of the set
    m = max(1, n // 2)
    assert bzk.msm_bases_run_dev(h, sc, m) == bzk.msm_g1_dev(db, sc, m)
    bzk.msm_bases_free(h)


def test_bases_handle_dedup_and_windows(bzk, co):
    n = 20000
    bases = co.g1_bases(72, 0, n, nthreads=co.ncpu())
    scb = _witness_like(n, 9)
    db, sc = to_dev(bases), to_dev(scb)
    torch.cuda.synchronize()
    h = bzk.msm_bases_load_dev(db, n)
    want = co.msm_g1(bases, scb, nthreads=co.ncpu())
    assert bzk.msm_bases_run_dev(h, sc, n, dedup=True) == want       # group sums live beside the resident set
    assert bzk.msm_bases_run_dev(h, sc, n, dedup=True, throughput=True) == want
    assert bzk.msm_g1_dev(db, sc, n, dedup=True) == want
    W = bzk.msm_window_count(n)
    for parts in (2, 5):
        cuts = [W * i // parts for i in range(parts + 1)]
        shards = b"".join(bzk.msm_bases_windows_dev(h, sc, n, cuts[i], cuts[i + 1]) for i in range(parts))
        assert bzk.g1_sum(shards) == want
    # all-zero scalars under de-duplication: the early exit must leave a usable context behind
    z = to_dev(bytes(32 * n))
    torch.cuda.synchronize()
    ident = bzk.msm_bases_run_dev(h, z, n, dedup=True)
    assert ident[96] == 1
    assert bzk.msm_g1_dev(db, z, n, dedup=True) == ident
    assert bzk.msm_bases_run_dev(h, sc, n) == want
    bzk.msm_bases_free(h)


def test_bases_handle_g2(bzk, co):
    n = 6000
    bases = co.g2_bases(73, 0, n, nthreads=co.ncpu())
    scb = _witness_like(n, 11)
    db, sc = to_dev(bases), to_dev(scb)
    torch.cuda.synchronize()
    h = bzk.msm_bases_load_dev(db, n, g2=True)
    want = co.msm_g2(bases, scb, nthreads=co.ncpu())
    assert bzk.msm_bases_run_dev(h, sc, n, g2=True) == want
    assert bzk.msm_bases_run_dev(h, sc, n, g2=True, dedup=True) == want
    bzk.msm_bases_free(h)


@pytest.mark.parametrize("devices,exchange", [([0], 0), ([0, 0], 1), ([0, 0, 0, 0], 1), ([0, 0], 2), ([0, 0, 0], 2)])
def test_mg_single_process_group_reproduces_single_gpu_bytes(bzk, co, devices, exchange):
    """VERDICT r2 task 1: 2 and 4 contexts on the one available GPU reproduce the single-GPU 97 bytes through bzk_mg_*"""
    from bazuka_amd import Mg
    n = 1 << 16
    bases = co.g1_bases(81, 0, n, nthreads=co.ncpu())
    scb = rand_scalars_bytes(n, 5)
    db, sc = to_dev(bases), to_dev(scb)
    torch.cuda.synchronize()
    want = bzk.msm_g1_dev(db, sc, n)
    assert want == co.msm_g1(bases, scb, nthreads=co.ncpu())
    mg = Mg(devices=devices, exchange=exchange)
    assert mg.world == len(devices) and mg.local == len(devices)
    assert mg.exchange == {0: "host", 1: "host", 2: "peer"}[exchange]
    hb = mg.bases_load_dev([db] * len(devices), n)
    for _ in range(2):
        assert mg.msm_dev(hb, [sc] * len(devices), n) == want
    assert mg.msm(hb, scb, n) == want                       # host scalars, staged per device
    assert mg.msm_dev(hb, [sc] * len(devices), 1000) == bzk.msm_g1_dev(db, sc, 1000)  # prefix: another window size
    wl = _witness_like(n, 3)
    assert mg.msm(hb, wl, n, dedup=True) == co.msm_g1(bases, wl, nthreads=co.ncpu())
    mg.bases_free(hb)
    hb2 = mg.bases_load(bases, n)                            # host bases: uploaded + converted per device
    assert mg.msm(hb2, scb, n) == want
    mg.bases_free(hb2)
    mg.close()


def test_mg_group_g2_and_empty_call(bzk, co):
    from bazuka_amd import Mg
    n = 2000
    bases = co.g2_bases(82, 0, n, nthreads=co.ncpu())
    scb = rand_scalars_bytes(n, 8)
    mg = Mg(devices=[0, 0, 0], exchange=1)
    hb = mg.bases_load(bases, n, g2=True)
    assert mg.msm(hb, scb, n, g2=True) == co.msm_g2(bases, scb, nthreads=co.ncpu())
    assert mg.msm(hb, b"", 0, g2=True)[192] == 1
    mg.bases_free(hb)
    mg.close()


def test_mg_rccl_transport_one_rank(bzk, co):
    """the RCCL leg (dlopen, ncclCommInitRank / ncclCommInitAll, ncclAllGather on uint8) on a one-rank communicator - all a one-GPU
    box can host, since RCCL refuses two ranks on one device"""
    from bazuka_amd import Mg, mg_unique_id
    n = 1 << 14
    bases = co.g1_bases(83, 0, n, nthreads=co.ncpu())
    scb = rand_scalars_bytes(n, 6)
    db, sc = to_dev(bases), to_dev(scb)
    torch.cuda.synchronize()
    want = co.msm_g1(bases, scb, nthreads=co.ncpu())
    for mk in (lambda: Mg(devices=[0], exchange=3), lambda: Mg(device=0, rank=0, world=1, uid=mg_unique_id(), exchange=3)):
        mg = mk()
        assert mg.exchange == "rccl"
        hb = mg.bases_load_dev([db], n)
        assert mg.msm_dev(hb, [sc], n) == want
        mg.bases_free(hb)
        mg.close()
    with pytest.raises(Exception):
        Mg(devices=[0, 0], exchange=3)   # two ranks on one device: refused, never silently another transport


@pytest.mark.parametrize("world,g2", [(2, 0), (4, 0), (2, 1)])
def test_mg_process_per_gpu_shared_memory_exchange(bzk, co, world, g2):
    """process-per-GPU deployment (bzk_mg_create_rank): `world` processes share device 0 and exchange their window sums through
    the POSIX shared-memory transport; every rank must print the single-GPU result"""
    from bazuka_amd import mg_unique_id
    n, seed = 30000, 77
    uid = mg_unique_id().hex()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "mg_rank.py"), str(r), str(world), uid, str(n), str(seed),
                               str(g2), "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs.append([ln for ln in o.splitlines() if ln.startswith("RESULT")][0].split())
    d = torch.empty(n * (192 if g2 else 96), dtype=torch.uint8, device="cuda")
    (bzk.g2_synth_bases_dev if g2 else bzk.g1_synth_bases_dev)(seed, 0, n, d)
    bzk.sync()
    scb = rand_scalars_bytes(n, seed)
    want = (co.msm_g2 if g2 else co.msm_g1)(dev_bytes(d), scb, nthreads=co.ncpu())
    for r, o in enumerate(outs):
        assert o[1] == str(r) and o[2] == "host" and bytes.fromhex(o[3]) == want


def test_mg_local_failure_reaches_every_rank_and_the_group_survives(bzk, co):
    """a rank whose local stage fails (allocation, window error) must still take part in the exchange so that EVERY
    rank returns an error for that call - not block, and never combine the stale sums of an earlier call - and the next call of
    the group works again.  Fault injected into rank 1's local stage of the group's 3rd sequence number (1 = the creation barrier,
    so the 2nd MSM call): 4 calls per rank -> 3 results equal to the single-GPU bytes + 1 error, on both ranks."""
    from bazuka_amd import mg_unique_id
    n, seed, world = 20000, 78, 2
    uid = mg_unique_id().hex()
    # the fault hook is compiled into the -DBZK_TEST_HOOKS build only: the ranks load that library
    hooks = os.path.join(ROOT, "bazuka_amd", "libbzk_testhooks.so")
    assert os.path.exists(hooks), "build() makes bazuka_amd/libbzk_testhooks.so"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BZK_MG_TEST_FAULT="1:3", BZK_MG_TIMEOUT_S="60", BZK_LIBBZK=hooks)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "mg_rank.py"), str(r), str(world), uid, str(n), str(seed),
                               "0", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    d = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(seed, 0, n, d)
    bzk.sync()
    want = co.msm_g1(dev_bytes(d), rand_scalars_bytes(n, seed), nthreads=co.ncpu())
    for r, p in enumerate(procs):
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        res = [ln for ln in o.splitlines() if ln.startswith("RESULT")][0].split()
        err = [ln for ln in o.splitlines() if ln.startswith("ERROR")]
        assert bytes.fromhex(res[3]) == want and res[4:] == ["3", "1"], o
        assert len(err) == 1 and ("injected fault" in err[0] if r == 1 else "rank 1 failed its local stage" in err[0]), err


def test_mg_proof_pool_replicas(bzk, co, pr):
    """bzk_mg_params_load / bzk_mg_prove_submit / _wait: a pool of prover slots over the group's devices (one CRS upload per device)
    proves a queue of different (r, s) pairs; every proof equals the oracle's and more than one slot did work"""
    from bazuka_amd import Mg
    from test_gpu_groth16 import _setup
    r1, params, zb, az, bz, cz = _setup(co, pr, 2500, 777)
    rs = [(fr_bytes(fr_list(2, 90 + k))[:32], fr_bytes(fr_list(2, 90 + k))[32:]) for k in range(6)]
    want = [co.groth16_prove(params, zb, az, bz, cz, r, s, nthreads=co.ncpu()) for r, s in rs]
    mg = Mg(devices=[0, 0], exchange=1)
    ph = mg.params_load(params, slots_per_device=2)
    for _ in range(2):
        pend = [mg.prove_submit(ph, zb, az, bz, cz, r, s) for r, s in rs]
        got = [mg.prove_wait(p) for p in reversed(pend)][::-1]   # waited out of order
        assert got == want
    st = mg.params_stats(ph)
    assert len(st) == 4 and sum(st) == 12 and sum(1 for x in st if x) >= 2
    mg.params_free(ph)
    mg.close()
