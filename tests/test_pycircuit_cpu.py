"""Row a3 / f-1 parity: the product's R1CS generator (bazuka_amd/csrc/{mpn.hip, host_r1cs.h} through the C ABI) against
oracle/pycircuit.py - a second restatement of the reference's circuits and gadgets written from the Rust sources
(src/mpn/circuits/{update,deposit,withdraw}_circuit.rs, src/zk/groth16/gadgets/**), not from the product.

What is compared, byte for byte, for Update / Deposit / Withdraw instances with a mix of enabled and `::null` slots: the
variable and constraint counts, the assignment z, <A,z>, <B,z>, <C,z>, both density maps, and the CSR form of A, B, C
(duplicate terms of a row summed, first-appearance order) - i.e. variable numbering and constraint order, the things that
decide whether a CRS made by the reference's `MpnCircuit::empty` setup (src/config/blockchain.rs:373-399) fits this prover.
Committed hashes of the same 15 arrays (tests/golden/r1cs_sha256.json, made by tests/golden/make_r1cs_fixtures.py from the
Python side only) pin them, including the 2^20-class (15,3,2) circuit; the GPU suite replays the hashes too.

The second half is the gadget suite the reference keeps for its own gadgets (accept / reject tables of
src/zk/groth16/gadgets/*/test.rs), run on the Python restatement with "proof verifies" read as "all constraints hold"."""
import hashlib
import json
import os

import pytest

import bincode_ref as B
import r1cs_scenarios as S
from bazuka_amd import lib as L
from oracle import pycircuit as pc
from oracle import pyref as pr

FIX = json.load(open(os.path.join(S.G, "r1cs_sha256.json")))
F, U = pr.fr_to_mont_bytes, pr.fr_from_mont_bytes


def _python_side(blob):
    work = B.decode(B.MpnWork, blob)
    pre = B.encode(B.Address, S.PROVER) + B.encode(B.U64, work["reward"])
    commitment = int.from_bytes(hashlib.sha3_256(pre).digest(), "little") % pr.R_MOD
    return pc.circuit_of_work(work, commitment, 1, lambda p: B.encode(B.ContractWithdraw, p)), commitment


@pytest.mark.parametrize("name", ["update_3_3_1", "deposit_3_3_1", "withdraw_3_3_1", "update_15_3_1"])
def test_product_r1cs_equals_independent_restatement(name):
    blob = S.make_work(name)
    assert hashlib.sha256(blob).hexdigest() == FIX[name]["work_sha256"]
    r, views, com = S.product_views(blob)
    cs, commitment = _python_side(blob)
    assert com == F(commitment)
    assert r.satisfied and pc.first_unsatisfied(cs) == -1
    assert (r.n_in, r.n_aux, r.n_constraints) == (cs.n_in, cs.n_aux, len(cs.A) + cs.n_in)
    assert (r.n_in, r.n_aux, r.n_constraints) == (FIX[name]["n_in"], FIX[name]["n_aux"], FIX[name]["n_constraints"])
    want = pc.all_views(cs)
    for key in L.R1cs.VIEWS:
        assert views[key] == want[key], (name, key)
        assert hashlib.sha256(want[key]).hexdigest() == FIX[name]["sha256"][key], (name, key)
    # bellman APPENDS linear-combination terms, the product merges duplicates of a row: identical evaluations always, and
    # identical densities as long as no row holds terms of one variable that cancel - none does in these circuits
    for which in "ABC":
        assert not pc.has_cancelling_duplicates(cs.rows(which))
    # some slots are enabled, some are `::null` (the `enabled` bits are the first allocation of every transition)
    kind = S.SCENARIOS[name][0]
    n_slots = 4 ** S.SCENARIOS[name][3]
    n_enabled = len(B.decode(B.MpnWork, blob)["data"][1])
    assert 0 < n_enabled < n_slots


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) >> 20
    except OSError:
        pass
    return 0


@pytest.mark.parametrize("name", list(S.PRODUCTION))
def test_production_shapes_match_committed_hashes(name):
    """the chain's real circuits (src/config/blockchain.rs:22-26): Deposit(15,3,3) 1 394 893, Withdraw(15,3,3) 2 346 957,
    Update(15,3,4) 14 443 117 constraints - the product's 15 arrays against hashes the Python restatement made alone
    (streaming form; the generator script took 34 s / 52 s / 5 min).  Hash only here; the GPU suite proves on them."""
    if name == "update_15_3_4" and _mem_available_gb() < 24:
        pytest.skip("the 14.4 M-constraint instance with its matrices needs ~14 GB")
    blob = S.make_work(name)
    fix = FIX[name]
    assert hashlib.sha256(blob).hexdigest() == fix["work_sha256"]
    dec, r, sha = S.product_hashes(blob)
    assert r.satisfied and r.accepted == fix["n_enabled"] and 0 < fix["n_enabled"] < 4 ** S.SCENARIOS[name][3]
    assert (r.n_in, r.n_aux, r.n_constraints) == (fix["n_in"], fix["n_aux"], fix["n_constraints"])
    assert sha == fix["sha256"]
    r.free()


@pytest.mark.parametrize("name", ["update_3_3_1", "deposit_3_3_1", "withdraw_3_3_1"])
def test_streaming_constraint_system_equals_the_stored_form(name):
    """the production fixtures come from pycircuit.StreamingConstraintSystem: same hashes as `all_views` of a stored instance"""
    blob = S.make_work(name)
    work = B.decode(B.MpnWork, blob)
    pre = B.encode(B.Address, S.PROVER) + B.encode(B.U64, work["reward"])
    commitment = int.from_bytes(hashlib.sha3_256(pre).digest(), "little") % pr.R_MOD
    cs = pc.StreamingConstraintSystem(6)
    pc.circuit_of_work(work, commitment, 1, lambda p: B.encode(B.ContractWithdraw, p), cs=cs)
    n_in, n_aux, n_rows, unsat, sha = cs.finish()
    assert unsat == -1 and (n_in, n_aux, n_rows) == (FIX[name]["n_in"], FIX[name]["n_aux"], FIX[name]["n_constraints"])
    assert sha == FIX[name]["sha256"]


@pytest.mark.parametrize("name", [n for n in S.SCENARIOS if n not in S.PRODUCTION])
def test_product_r1cs_matches_committed_hashes(name):
    """includes update_15_3_2, the 2^20-class circuit (903 037 constraints) - too slow for the Python side inside the suite,
    its hashes come from the fixture generator"""
    blob = S.make_work(name)
    assert hashlib.sha256(blob).hexdigest() == FIX[name]["work_sha256"]
    r, views, _ = S.product_views(blob)
    assert (r.n_in, r.n_aux, r.n_constraints) == (FIX[name]["n_in"], FIX[name]["n_aux"], FIX[name]["n_constraints"])
    for key in L.R1cs.VIEWS:
        assert hashlib.sha256(views[key]).hexdigest() == FIX[name]["sha256"][key], (name, key)
    # one worker thread (sequential walk) and the parallel per-transition workers emit the same instance
    if S.SCENARIOS[name][3] == 1:
        _, v1, _ = S.product_views(blob, threads=1)
        assert v1 == views


def test_empty_circuits_have_the_structure_of_the_restatement():
    """`MpnCircuit::empty(L, T, B)` is what the reference's setup synthesizes (src/config/blockchain.rs:373-399): the matrices
    must not depend on the witness.  Python: all-null transitions with the public inputs of src/mpn/circuits/test.rs:117-132."""
    com, st, aux = 456, 123, pr.poseidon([1, 0])
    args = (F(com), 0, F(st), F(aux), F(st))
    for kind, make in (("update", lambda: L.mpn_update_empty(3, 3, 1, *args, S.ZIESHA, record_matrices=True)),
                       ("deposit", lambda: L.mpn_circuit_empty(0, 3, 3, 1, *args, record_matrices=True)),
                       ("withdraw", lambda: L.mpn_circuit_empty(1, 3, 3, 1, *args, record_matrices=True))):
        r = make()
        if kind == "update":
            cs = pc.update_circuit(3, 3, com, 0, st, aux, st, 1, [pc.null_update_transition(3, 3)] * 4)
            assert r.satisfied and pc.first_unsatisfied(cs) == -1   # the reference's own test instance verifies
        elif kind == "deposit":
            cs = pc.deposit_circuit(3, 3, 1, com, 0, st, aux, st, [pc.null_deposit_transition(3, 3)] * 4)
        else:
            cs = pc.withdraw_circuit(3, 3, 1, com, 0, st, aux, st, [pc.null_withdraw_transition(3, 3)] * 4, None)
        full = FIX[kind + "_3_3_1"]
        assert (r.n_in, r.n_aux, r.n_constraints) == (full["n_in"], full["n_aux"], full["n_constraints"])
        want = pc.all_views(cs)
        for key in L.R1cs.VIEWS:
            assert r.view(key) == want[key], (kind, key)
        # structure (CSR + densities) is witness-independent: equal to the populated instance's
        for key in ("a_density", "b_density", "valA", "valB", "valC", "colA", "colB", "colC", "rpA", "rpB", "rpC"):
            assert hashlib.sha256(want[key]).hexdigest() == full["sha256"][key], (kind, key)


# ---- the reference's gadget tests on the restatement --------------------------------------------------------------------
def _ok(cs):
    return pc.first_unsatisfied(cs) == -1


def test_is_equal_table():  # gadgets/common/test.rs:38-65
    for a, b, eq, expected in [(123, 123, False, False), (123, 123, True, True), (123, 234, False, True), (123, 234, True, False)]:
        cs = pc.ConstraintSystem()
        an, bn = pc.AllocatedNum.alloc(cs, a), pc.AllocatedNum.alloc(cs, b)
        e = pc.AllocatedBit.alloc(cs, eq)
        res = pc.extract_bool(pc.Number.of(an).is_equal(cs, pc.Number.of(bn)))
        cs.enforce(list(res.lc), [(pc.ONE, 1)], [(e.var, 1)])
        assert _ok(cs) == expected, (a, b, eq)


LTE_ROWS = [(0, 0, True, True), (0, 0, False, False), (0, 123, True, True), (0, 123, False, False), (123, 0, True, False),
            (123, 0, False, True), (122, 123, True, True), (123, 123, True, True), (124, 123, False, True), (122, 123, False, False),
            (123, 123, False, False), (124, 123, True, False), (252, 253, True, True), (253, 253, True, True), (254, 253, False, True),
            (252, 253, False, False), (253, 253, False, False), (254, 253, True, False), (254, 255, True, True), (255, 256, False, False),
            (255, 256, True, False), (256, 255, False, False), (256, 255, True, False), (255, 257, False, False), (255, 257, True, False),
            (257, 255, False, False), (257, 255, True, False)]


def test_lte_table():  # gadgets/common/test.rs:113-141 (27 rows, 8-bit integers, out-of-range operands must fail both ways)
    for a, b, claim, expected in LTE_ROWS:
        cs = pc.ConstraintSystem()
        a8 = pc.UnsignedInteger.constrain(cs, pc.Number.of(pc.AllocatedNum.alloc(cs, a)), 8)
        b8 = pc.UnsignedInteger.constrain(cs, pc.Number.of(pc.AllocatedNum.alloc(cs, b)), 8)
        c = pc.AllocatedBit.alloc(cs, claim)
        res = pc.extract_bool(a8.lte(cs, b8))
        cs.enforce(list(res.lc), [(pc.ONE, 1)], [(c.var, 1)])
        assert _ok(cs) == expected, (a, b, claim)


def test_or_table():  # gadgets/common/test.rs:180-207
    for a in (False, True):
        for b in (False, True):
            for claim in (False, True):
                cs = pc.ConstraintSystem()
                ab, bb = pc.Boolean.is_(pc.AllocatedBit.alloc(cs, a)), pc.Boolean.is_(pc.AllocatedBit.alloc(cs, b))
                e = pc.AllocatedBit.alloc(cs, claim)
                pc.extract_bool(pc.boolean_or(cs, ab, bb)).assert_equal(cs, pc.Number.of(e))
                assert _ok(cs) == ((a or b) == claim)


def test_poseidon_gadget_equals_native():  # gadgets/poseidon/test.rs:121-151, KAT inputs of src/zk/poseidon/mod.rs:114-149
    for arity in (1, 2, 4, 5, 7, 16):
        cs = pc.ConstraintSystem()
        ins = [pc.AllocatedNum.alloc(cs, i) for i in range(arity)]
        out = pc.poseidon_gadget(cs, ins)
        assert out.value == pr.poseidon(list(range(arity))) and _ok(cs)
        t = arity + 1
        r_p = 56 if t <= 5 else 57
        assert len(cs.A) == 3 * 8 * t + r_p * (t + 2)  # SURVEY App. B cost model


def test_merkle_gadget_on_a_64_leaf_tree():  # gadgets/merkle/test.rs:60-103
    leaves = list(range(64))
    nodes = []
    root = pr.merkle4_root(leaves, 3, nodes)
    levels = [leaves, [pr.poseidon(leaves[4 * i:4 * i + 4]) for i in range(16)]]
    levels.append([pr.poseidon(levels[1][4 * i:4 * i + 4]) for i in range(4)])
    for i in (0, 1, 5, 27, 63):
        proof, idx = [], i
        for lvl in range(3):
            sib = [levels[lvl][(idx // 4) * 4 + k] for k in range(4) if k != idx % 4]
            proof.append(sib)
            idx //= 4
        for good in (True, False):
            cs = pc.ConstraintSystem()
            index = pc.UnsignedInteger.alloc(cs, i, 6)
            val = pc.AllocatedNum.alloc(cs, i if good else i + 1)
            pw = [[pc.AllocatedNum.alloc(cs, s) for s in p] for p in proof]
            rt = pc.AllocatedNum.alloc(cs, root)
            en = pc.Boolean.is_(pc.AllocatedBit.alloc(cs, True))
            pc.check_proof_poseidon4(cs, en, index, pc.Number.of(val), pw, pc.Number.of(rt))
            assert _ok(cs) == good


def test_eddsa_gadget_accept_reject_disabled():  # gadgets/eddsa/test.rs:46-95
    keys = pr.jj_generate_keys(b"salam")
    msg = 1234
    rr, s = pr.jj_sign(keys, msg)
    for enabled, m, expected in ((True, msg, True), (True, msg + 1, False), (False, msg + 1, True)):
        cs = pc.ConstraintSystem()
        en = pc.Boolean.is_(pc.AllocatedBit.alloc(cs, enabled))
        pk = pc.AllocatedPoint.alloc(cs, keys["pub"])
        mw = pc.AllocatedNum.alloc(cs, m)
        sr = pc.AllocatedPoint.alloc(cs, rr)
        ss = pc.AllocatedNum.alloc(cs, s)
        n0 = len(cs.A)
        pc.verify_eddsa(cs, en, pk, pc.Number.of(mw), sr, ss)
        assert _ok(cs) == expected
        assert (len(cs.A) - n0, len(cs.aux) - 7) == (10057, 10053)  # verify_eddsa = (10 053 vars, 10 057 constraints)


def test_reveal_gadget_equals_compress():  # gadgets/reveal/test.rs:97-140
    model = ("struct", ["scalar", ("list", 2, "scalar"), "scalar", "scalar"])
    lst = [0] * 16
    lst[2], lst[4], lst[10] = 10, 10, 15
    cs = pc.ConstraintSystem()
    state = [pc.Number.of(pc.AllocatedNum.alloc(cs, 123)), [pc.Number.of(pc.AllocatedNum.alloc(cs, v)) for v in lst],
             pc.Number.of(pc.AllocatedNum.alloc(cs, 0)), pc.Number.of(pc.AllocatedNum.alloc(cs, 0))]
    out = pc.reveal(cs, model, state)
    expected = pr.poseidon([123, pr.merkle4_root(lst, 2), 0, 0])
    assert out.value == expected and _ok(cs)


def test_decompress_restatement():
    """PointCompressed::decompress (src/crypto/jubjub/curve.rs:79-91): round trip on real keys; the default key (0, false)
    of a `::null` transaction decompresses to (0, -1), which IS on the curve - the value the circuits allocate there."""
    for seed in (b"a", b"acct0", b"salam"):
        pub = pr.jj_generate_keys(seed)["pub"]
        assert pc.pt_decompress(pub[0], pub[1] & 1) == pub
    assert pc.pt_decompress(0, False) == (0, pr.R_MOD - 1) and pc.pt_is_on_curve((0, pr.R_MOD - 1))
    assert pc.base_cofactor() == pr.jj_mul(pr.JJ_BASE, 8)


def test_recalled_bellman_behaviours():
    """every behaviour of bellman 0.14 that the R1CS parity rests on and the reference tree cannot confirm (un-vendored crate) is a
    named entry of oracle/pycircuit.py::RECALLED_BELLMAN with the bellman item it restates and an executable statement of the
    assumption: a maintainer with Rust checks the list; here each check pins the restatement to its statement"""
    from oracle import pycircuit as pc
    assert len(pc.RECALLED_BELLMAN) >= 12
    for what, item, check in pc.RECALLED_BELLMAN:
        assert item.startswith("bellman::") and what
        check()
