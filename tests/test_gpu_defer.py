"""Deferred witness values ON THE DEVICE: bzk_groth16_prove_r1cs over an instance whose hash-dependent values were left
out by the host generator (bzk_witfill.cuh / witfill.hip run the instance's program before any MSM or transform reads the arrays) must give the
SAME 387 proof bytes as bzk_groth16_prove over the complete arrays of a plain synthesis - which tests/test_gpu_fullsize.py and
tests/test_gpu_mpn_prove.py pin on the oracle prover.  The CPU side of the same program (instance + host fill == the independent restatement's
fixtures): tests/test_defer_cpu.py."""
import hashlib
import json
import os

import pytest

import r1cs_scenarios as S
from bazuka_amd import lib as L
from util import fr_bytes, fr_list

pytestmark = pytest.mark.gpu
FIX = json.load(open(os.path.join(S.G, "r1cs_sha256.json")))


def oracle_prove(co, bzk, ph, r, rs, log_m=None):
    """the CPU ORACLE's prover (oracle/oracle.cpp) on this CRS read back from the device and the complete arrays `r` of a plain synthesis"""
    op = {"n_in": r.n_in, "n_aux": r.n_aux, "log_m": log_m or max(1, (r.n_constraints - 1).bit_length()), "a_density": r.view("a_density"),
          "b_density": r.view("b_density")}
    for which, key in ((0, "vk"), (1, "h"), (2, "l"), (3, "a"), (4, "b_g1"), (5, "b_g2")):
        op[key] = bzk.params_read(ph, which)
    op["n_a"], op["n_b"] = sum(op["a_density"]), sum(op["b_density"])
    return co.groth16_prove(op, *(bytes(r.raw(k)) for k in ("z", "az", "bz", "cz")), rs[:32], rs[32:], nthreads=co.ncpu())


# the DEVICE-filled arrays themselves, read back (bzk_staged_read) and hashed against the fixtures that the independent Python
# restatement made alone (tests/golden/r1cs_sha256.json, oracle/pycircuit.py; gadget value semantics:
# /root/reference/src/zk/groth16/gadgets/poseidon/mod.rs:8-95, merkle/mod.rs:21-78, common/mux.rs:7-47) - at the shapes `bzk-worker --defer` proves in
# production (Deposit / Withdraw (15,3,3), Update (15,3,4): src/config/blockchain.rs:22-26), not only at the test shapes
@pytest.mark.parametrize("name", ["update_3_3_1", "withdraw_3_3_1", "update_15_3_2", "deposit_15_3_3", "withdraw_15_3_3", "update_15_3_4"])
def test_device_filled_arrays_equal_the_independent_restatements_fixtures(bzk, name):
    if name == "update_15_3_4" and os.environ.get("BZK_TEST_PRODUCTION_BYTES", "1") == "0":
        pytest.skip("BZK_TEST_PRODUCTION_BYTES=0")
    from bazuka_amd import Bzk
    dec = L.MpnWork.decode(S.make_work(name))
    d = dec.synthesize(S.PROVER, defer=True)
    info = d.defer_info()
    assert info["deferred"] == 1 and info["filled"] == 0
    assert (d.n_in, d.n_aux, d.n_constraints) == (FIX[name]["n_in"], FIX[name]["n_aux"], FIX[name]["n_constraints"])
    for k in ("z", "az", "bz", "cz"):  # the host arrays have holes: whatever matches below was computed on the device
        assert hashlib.sha256(d.raw(k)).hexdigest() != FIX[name]["sha256"][k], k
    stager = Bzk(bzk.device)
    h = stager.r1cs_stage(d)
    stager.staged_wait(h)              # BZK_E_UNSAT would raise here: every deferred row holds
    for i, k in enumerate(("z", "az", "bz", "cz")):
        got = stager.staged_read(h, i)
        assert len(got) == len(d.raw(k)), k
        assert hashlib.sha256(got).hexdigest() == FIX[name]["sha256"][k], (name, k)
        del got
    assert d.defer_info()["filled"] == 0   # nothing was filled in on the CPU behind the scenes
    stager.staged_free(h)
    stager.close()
    d.free()


@pytest.mark.parametrize("name", ["update_3_3_1", "update_15_3_2", "withdraw_15_3_3"])
def test_device_fill_and_staging_give_the_ORACLE_provers_bytes(bzk, co, name):
    """prove_r1cs and prove_staged over a device-filled instance against co.groth16_prove - the oracle itself, not the product's plain path
    (update_15_3_4 likewise inside tests/test_gpu_production.py, which already holds the oracle's proof of that shape)"""
    from bazuka_amd import Bzk
    dec = L.MpnWork.decode(S.make_work(name))
    r, ph, vkb = _setup(bzk, dec)
    rs = fr_bytes(fr_list(2, 916))
    want = oracle_prove(co, bzk, ph, r, rs)
    d = dec.synthesize(S.PROVER, defer=True)
    assert d.defer_info()["deferred"] == 1
    assert bzk.groth16_prove_r1cs(ph, d, rs[:32], rs[32:]) == want
    stager = Bzk(bzk.device)
    h = stager.r1cs_stage(d)
    assert bzk.groth16_prove_staged(ph, h, rs[:32], rs[32:]) == want
    assert d.defer_info()["filled"] == 0
    assert L.groth16_verify(vkb, r.view("z")[32:32 * r.n_in], want)
    stager.staged_free(h)
    stager.close()
    bzk.params_free(ph)


def _setup(bzk, dec):
    r = dec.synthesize(S.PROVER, record_matrices=True)
    assert r.satisfied
    csr = [(r.n_constraints, r.view("rp" + w), r.view("col" + w), r.view("val" + w)) for w in "ABC"]
    ph, vkb = bzk.groth16_setup(csr, r.n_in, r.n_aux, fr_bytes(fr_list(5, 515)))
    return r, ph, vkb


@pytest.mark.parametrize("name", ["update_3_3_1", "update_15_3_2", "deposit_3_3_1", "withdraw_3_3_1", "deposit_15_3_3"])
def test_device_fill_gives_the_same_proof_bytes(bzk, name):
    dec = L.MpnWork.decode(S.make_work(name))
    r, ph, vkb = _setup(bzk, dec)
    rs = fr_bytes(fr_list(2, 516))
    want = bzk.groth16_prove(ph, *(r.view(k) for k in ("z", "az", "bz", "cz")), rs[:32], rs[32:])
    assert bzk.groth16_prove_r1cs(ph, r, rs[:32], rs[32:]) == want          # a complete instance through the new entry
    d = dec.synthesize(S.PROVER, threads=4, defer=True)
    assert d.defer_info()["deferred"] == 1
    for rep in range(3):                                                    # the program, its tables and scratch are cached per context
        assert bzk.groth16_prove_r1cs(ph, d, rs[:32], rs[32:]) == want, rep
    assert d.defer_info()["filled"] == 0                                    # the host arrays still have their holes: nothing was filled in on the CPU
    d.fill_host()
    assert bzk.groth16_prove_r1cs(ph, d, rs[:32], rs[32:]) == want          # ... and a host-filled instance is a complete one
    assert L.groth16_verify(vkb, d.view("z")[32:32 * d.n_in], want)
    bzk.params_free(ph)


def test_world_side_deferral_and_slots(bzk):
    """bzk_mpn_set_defer on the validator-side world + a second prover slot over the same CRS (the pipelined prover's arrangement)"""
    Z = S.ZIESHA

    def world(defer):
        w = L.MpnWorld(3, 3)
        w.set_defer(defer)
        for i in range(8):
            w.add_account(i, b"a%d" % i, Z, 10 ** 9)
        for i in range(4):
            w.push_tx(i, (i + 1) % 8, Z, 50 + i, Z, i)
        return w

    wa, wb = world(False), world(True)
    a = wa.update_synthesize(1, S.F(3), Z, record_matrices=True)
    csr = [(a.n_constraints, a.view("rp" + w), a.view("col" + w), a.view("val" + w)) for w in "ABC"]
    ph, _ = bzk.groth16_setup(csr, a.n_in, a.n_aux, fr_bytes(fr_list(5, 616)))
    slot = bzk.params_slot(ph)
    rs = fr_bytes(fr_list(2, 617))
    b = wb.update_synthesize(1, S.F(3), Z)
    assert b.defer_info()["deferred"] == 1
    want = bzk.groth16_prove(ph, *(a.view(k) for k in ("z", "az", "bz", "cz")), rs[:32], rs[32:])
    assert bzk.groth16_prove_r1cs(slot, b, rs[:32], rs[32:]) == want
    assert bzk.groth16_prove_r1cs(ph, b, rs[:32], rs[32:]) == want
    bzk.params_free(slot)
    bzk.params_free(ph)


def test_a_deferred_violation_is_refused_by_the_prove_call(bzk):
    blob = bytearray(S.make_work("update_3_3_1"))
    dec0 = L.MpnWork.decode(bytes(blob))
    r, ph, _ = _setup(bzk, dec0)
    rs = fr_bytes(fr_list(2, 716))
    found = False
    for off in range(len(blob) // 2, len(blob) // 2 + 4000, 97):
        mut = bytearray(blob)
        mut[off] ^= 1
        try:
            dec = L.MpnWork.decode(bytes(mut))
            if dec.synthesize(S.PROVER, threads=1).satisfied:
                continue
            d = dec.synthesize(S.PROVER, threads=2, defer=True)
        except Exception:
            continue
        if not d.defer_info()["deferred"] or not d.satisfied:
            continue
        with pytest.raises(L.BzkError):
            bzk.groth16_prove_r1cs(ph, d, rs[:32], rs[32:])
        found = True
        break
    assert found, "no sampled mutation produced a violation that only the deferred rows see"
    # the context is usable afterwards
    want = bzk.groth16_prove(ph, *(r.view(k) for k in ("z", "az", "bz", "cz")), rs[:32], rs[32:])
    assert bzk.groth16_prove_r1cs(ph, dec0.synthesize(S.PROVER, defer=True), rs[:32], rs[32:]) == want
    bzk.params_free(ph)


def test_staged_instances_prove_to_the_same_bytes(bzk):
    """bzk_r1cs_stage on a PRODUCER's context (uploads + the deferred-value program there), bzk_groth16_prove_staged on the prover's: plain and deferred
    instances, several staged at once on one staging stream, handles reused from the pool"""
    from bazuka_amd import Bzk
    dec = L.MpnWork.decode(S.make_work("update_15_3_1"))
    r, ph, vkb = _setup(bzk, dec)
    rs = [fr_bytes(fr_list(2, 800 + i)) for i in range(4)]
    want = [bzk.groth16_prove(ph, *(r.view(k) for k in ("z", "az", "bz", "cz")), x[:32], x[32:]) for x in rs]
    stager = Bzk(bzk.device)
    plain = dec.synthesize(S.PROVER)
    deferred = [dec.synthesize(S.PROVER, threads=2, defer=True) for _ in range(3)]
    handles = [stager.r1cs_stage(x) for x in [plain] + deferred]      # four instances in flight on the staging stream
    for i, h in enumerate(handles):
        assert bzk.groth16_prove_staged(ph, h, rs[i][:32], rs[i][32:]) == want[i], i
    stager.staged_wait(handles[0])
    for h in handles:
        stager.staged_free(h)
    h2 = stager.r1cs_stage(deferred[0])                               # a pooled buffer again
    assert bzk.groth16_prove_staged(ph, h2, rs[0][:32], rs[0][32:]) == want[0]
    assert deferred[0].defer_info()["filled"] == 0
    stager.staged_free(h2)
    # a violated deferred constraint surfaces from the wait and from the prove call
    blob = bytearray(S.make_work("update_15_3_1"))
    bad = None
    for off in range(len(blob) // 2, len(blob) // 2 + 6000, 97):
        mut = bytearray(blob)
        mut[off] ^= 1
        try:
            d2 = L.MpnWork.decode(bytes(mut))
            if d2.synthesize(S.PROVER, threads=1).satisfied:
                continue
            cand = d2.synthesize(S.PROVER, threads=2, defer=True)
        except Exception:
            continue
        if cand.defer_info()["deferred"] and cand.satisfied:
            bad = cand
            break
    if bad is not None:
        hb = stager.r1cs_stage(bad)
        with pytest.raises(L.BzkError):
            stager.staged_wait(hb)
        with pytest.raises(L.BzkError):
            bzk.groth16_prove_staged(ph, hb, rs[0][:32], rs[0][32:])
        stager.staged_free(hb)
    bzk.params_free(ph)
    stager.close()
