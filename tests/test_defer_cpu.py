"""Deferred witness values (bazuka_amd/csrc/bzk_witfill.cuh, host_r1cs.h DeferProgram) on the CPU.

With deferral the host generator leaves every value that hangs off a Poseidon output - the Poseidon gadget's own variables
(/root/reference/src/zk/groth16/gadgets/poseidon/mod.rs:8-95), the Merkle gadget's muxes (merkle/mod.rs:21-78), the checks against computed
roots - to a per-shape program the device runs.  The same ops run on the host in bzk_r1cs_fill_host; here: instance + host fill == the
fixtures of the INDEPENDENT Python restatement (tests/golden/r1cs_sha256.json), i.e. exactly what a synthesis without deferral gives.  The
device run of the same program is tests/test_gpu_defer.py."""
import hashlib
import json
import os

import pytest

import r1cs_scenarios as S
from bazuka_amd import lib as L

FIX = json.load(open(os.path.join(S.G, "r1cs_sha256.json")))
ARRAYS = ("z", "az", "bz", "cz", "a_density", "b_density")


@pytest.mark.parametrize("name,threads", [("update_3_3_1", 1), ("update_15_3_1", 3), ("update_15_3_2", 0), ("deposit_3_3_1", 1), ("withdraw_3_3_1", 2),
                                          ("deposit_15_3_3", 0), ("withdraw_15_3_3", 0)])
def test_deferred_instance_plus_host_fill_equals_the_independent_restatement(name, threads):
    dec = L.MpnWork.decode(S.make_work(name))
    r = dec.synthesize(S.PROVER, threads=threads, defer=True)
    d = r.defer_info()
    kind, l4, t4, b4 = S.SCENARIOS[name]
    assert d["deferred"] == 1 and d["n_tx"] == 4 ** b4 and d["filled"] == 0
    assert (r.n_in, r.n_aux, r.n_constraints) == (FIX[name]["n_in"], FIX[name]["n_aux"], FIX[name]["n_constraints"])
    # most of a transition is the device's: 3 * (l4 + t4) + ... Merkle levels of ~540 Poseidon constraints each
    # (deposit / withdraw circuits also hash every transaction into the batch root OUTSIDE the per-transition bodies: a smaller share there)
    assert d["hole_con"] * d["n_tx"] > (0.7 if l4 >= 15 and kind == "update" else 0.2) * r.n_constraints, d
    # the rows the host wrote hold; the others are not the host's to judge yet
    assert r.satisfied
    holes = sum(hashlib.sha256(r.view(k)).hexdigest() != FIX[name]["sha256"][k] for k in ("z", "az", "bz", "cz"))
    assert holes == 4, "nothing was deferred?"
    for k in ("a_density", "b_density"):   # densities are a property of the circuit, not of the values (witness-only mode: empty maps)
        assert hashlib.sha256(r.view(k)).hexdigest() == hashlib.sha256(dec.synthesize(S.PROVER, threads=threads).view(k)).hexdigest()
    # what the one-launch device kernel would be handed for this program: every hash op in the stage of its level, every fill op in the last stage, each once
    sch = r.defer_schedule_info()
    assert sch["violations"] == 0 and sch["stages"] == d["n_levels"] + 1 and 1 <= sch["largest_segment"] <= 64, sch
    assert sch["hash_ops"] + sch["fill_ops"] <= d["n_ops"] and sch["fill_ops"] > 0 and sch["hash_ops"] > 0, (sch, d)
    d = r.fill_host()
    assert d["filled"] == 1 and d["flags"] == 0
    for k in ("z", "az", "bz", "cz"):
        assert hashlib.sha256(r.view(k)).hexdigest() == FIX[name]["sha256"][k], (name, k)
    r2 = L.R1cs.__new__(L.R1cs)   # the satisfied-scan over the now complete arrays (a second info call on the same handle)
    r2.lib, r2.h = r.lib, r.h
    L.R1cs.__init__(r2, r.h)
    assert r2.satisfied
    r2.h = None


def test_world_side_entry_and_plain_instance_agree():
    """bzk_mpn_set_defer on the validator-side world: same transitions, one world with and one without deferral"""
    Z = S.ZIESHA

    def world(defer):
        w = L.MpnWorld(15, 3)
        w.set_threads(2)
        w.set_defer(defer)
        for i in range(8):
            w.add_account(i * 1001 + 5, b"a%d" % i, Z, 10 ** 9)
        for i in range(6):
            w.push_tx(i * 1001 + 5, ((i + 1) % 8) * 1001 + 5, Z, 50 + i, Z, i)
        w.push_tx(5, 5 + 1001, Z, 10 ** 10, Z, 1)     # more than the balance: rejected by the builder, not a transition
        return w.update_synthesize(2, S.F(77), Z)

    a, b = world(False), world(True)
    assert a.defer_info()["deferred"] == 0 and b.defer_info()["deferred"] == 1
    assert (a.accepted, a.rejected) == (b.accepted, b.rejected) == (6, 1)
    assert b.fill_host()["flags"] == 0
    for k in ARRAYS:
        assert a.view(k) == b.view(k), k


def test_a_wrong_sibling_is_reported_by_the_fill_not_by_the_host_scan():
    """an invalid work (one Merkle sibling of one transition changed): the plain generator reports the first unsatisfied row; the deferred
    one cannot see it on the host - the fill raises the flags (1: a deferred equality does not hold, 2: the state chain breaks)"""
    blob = bytearray(S.make_work("update_3_3_1"))
    dec0 = L.MpnWork.decode(bytes(blob))
    good = dec0.synthesize(S.PROVER, threads=1)
    assert good.satisfied
    # find a byte whose change keeps the work decodable but breaks satisfaction: walk from the end of the transition block backwards
    import itertools
    for off in itertools.chain(range(len(blob) // 2, len(blob) // 2 + 4000, 97)):
        mut = bytearray(blob)
        mut[off] ^= 1
        try:
            dec = L.MpnWork.decode(bytes(mut))
            plain = dec.synthesize(S.PROVER, threads=1)
        except Exception:
            continue
        if plain.satisfied:
            continue
        r = dec.synthesize(S.PROVER, threads=2, defer=True)
        if not r.defer_info()["deferred"]:
            continue   # the generator fell back to the full sequential walk (shape mismatch): nothing deferred
        host_sees = not r.satisfied
        flags = r.fill_host()["flags"]
        assert host_sees or flags != 0, (off, flags)
        if flags:
            return
    pytest.skip("no single-byte mutation in the sampled range produced a decodable work with a deferred violation")


def test_record_matrices_outside_its_three_values_is_refused():
    """2 became BZK_SYNTH_DEFER in round 5, so `non-zero = matrices` no longer holds - any other value is a caller's mistake, not a silent
    witness-only instance"""
    import ctypes as C
    dec = L.MpnWork.decode(S.make_work("update_3_3_1"))
    for bad in (3, -1, 7):
        h = C.c_void_p()
        assert dec.lib.bzk_mpn_work_synthesize(dec.h, S.PROVER, None, 1, bad, C.byref(h)) == -1 and not h.value
    for ok in (0, 1, 2):
        h = C.c_void_p()
        assert dec.lib.bzk_mpn_work_synthesize(dec.h, S.PROVER, None, 1, ok, C.byref(h)) == 0 and h.value
        L.R1cs(h).free()
