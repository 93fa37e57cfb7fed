"""GPU parity of the PERSISTENT device state (bzk_state_* = `KvStoreStateManager::{update_contract, get_data, prove, root}`,
/root/reference/src/zk/state/mod.rs:218-438) against tests/pystate.py::PyKvState, the pair-at-a-time restatement of the reference's
state manager: the reference's own state-test scripts (src/zk/test/mod.rs:43-287), random models under random delta batches
(root, size, get, prove, rollback after every batch), the bincode entry, the MPN model against the one-shot seam, and the error
cases with their all-or-nothing behaviour."""
import random

import pytest

import pystate as ps
from bazuka_amd import BzkError, DeviceState
from oracle import pyref as pr

pytestmark = pytest.mark.gpu
S = ("scalar",)
F = pr.fr_to_mont_bytes


def _pairs(delta):
    return [(k, F((v or 0) % pr.R_MOD)) for k, v in delta.items()]


class Both:
    """one delta applied to the device state and to the restatement; everything observable compared"""

    def __init__(self, bzk, model):
        self.dev, self.ref, self.model, self.h = DeviceState(bzk, ps.model_bincode(model)), ps.PyKvState(model), model, 0
        self.check_root()

    def check_root(self):
        h, n, height = self.dev.root()
        assert (h, n, height) == (F(self.ref.hash), self.ref.size, self.h)

    def update(self, delta, bincode=False):
        self.h += 1
        rb = self.ref.update_contract(delta, self.h)
        if bincode:
            assert self.dev.update_bincode(ps.delta_bincode(delta), self.h) == F(self.ref.hash) + self.ref.size.to_bytes(8, "little")
            prev = None
        else:
            ps_ = _pairs(delta)
            h, n, prev = self.dev.update(ps_, self.h, want_rollback=True)
            assert (h, n) == (F(self.ref.hash), self.ref.size)
            assert prev == [F(rb[k] or 0) for k, _ in ps_]
        self.check_root()
        return rb

    def check_get(self, locators):
        assert self.dev.get(locators) == [F(self.ref.get_data(l)) for l in locators]

    def check_prove(self, tree_loc, indices):
        got = self.dev.prove(tree_loc, indices)
        for i, g in zip(indices, got):
            assert g == [[F(x) for x in part] for part in self.ref.prove(tree_loc, i)]


def test_reference_state_manager_scripts(bzk):
    # test_state_manager_scalar (src/zk/test/mod.rs:64-86)
    b = Both(bzk, S)
    b.update({(): 0xF})
    b.check_get([()])
    # test_state_manager_struct (:88-165)
    b = Both(bzk, ("struct", [S, S]))
    roots = [b.dev.root()[0]]
    for d in ({(0,): 0xF}, {(1,): 0xF0}, {(0,): 0xF00}, {(0,): 0xF}, {(0,): 0, (1,): 0}):
        b.update(d)
        roots.append(b.dev.root()[0])
    assert roots[4] == roots[2] and roots[5] == roots[0] and b.dev.root()[1] == 0
    # test_state_manager_list (:167-287): List{3, Struct{Scalar, Scalar}}, items 62 and 33 written, overwritten, zeroed
    b = Both(bzk, ("list", 3, ("struct", [S, S])))
    roots = [b.dev.root()[0]]
    for d in ({(62, 0): 0xF00000}, {(33, 0): 0xF}, {(33, 1): 0xF0}, {(33, 0): 0xF00}, {(33, 0): 0xF}, {(33, 0): 0, (33, 1): 0}, {(62, 0): 0}):
        b.update(d)
        roots.append(b.dev.root()[0])
        b.check_prove((), [62, 33, 0])
        b.check_get([(33,), (62,), (33, 0), (5, 1), ()])
    assert roots[5] == roots[3] and roots[6] == roots[1] and roots[7] == roots[0] and b.dev.root()[1] == 0
    assert len(set(roots[:5])) == 5


def test_reference_membership_proof_script(bzk):
    """test_zk_list_membership_proof (:43-62): Struct{Scalar, List{4, Scalar}}, the 256 items written ONE delta at a time, then a
    proof for every index: folding a leaf with its proof must give the list's value (with Poseidon in place of the SumHasher)"""
    b = Both(bzk, ("struct", [S, ("list", 4, S)]))
    for i in range(256):
        b.h += 1
        b.dev.update([((1, i), F(i))], b.h)
    b.ref.update_contract({(1, i): i for i in range(256)}, b.h)
    b.check_root()
    (list_value,) = b.dev.get([(1,)])
    proofs = b.dev.prove((1,), range(256))
    for i in (0, 1, 77, 255):
        assert proofs[i] == [[F(x) for x in part] for part in b.ref.prove((1,), i)]
        cur, at = i, i
        for part in proofs[i]:
            kids = [pr.fr_from_mont_bytes(x) for x in part]
            kids.insert(at % 4, cur)
            cur, at = pr.poseidon(kids), at // 4
        assert F(cur) == list_value


def _random_model(rnd, depth):
    k = rnd.random()
    if depth == 0 or k < 0.25:
        return S
    if k < 0.6:
        return ("struct", [_random_model(rnd, depth - 1) for _ in range(rnd.randint(1, 4))])
    return ("list", rnd.randint(0, 3), _random_model(rnd, depth - 1))


def _random_locator(rnd, model, stop=0.0):
    loc = []
    while model[0] != "scalar" and rnd.random() >= stop:
        if model[0] == "struct":
            f = rnd.randrange(len(model[1]))
            loc.append(f)
            model = model[1][f]
        else:
            loc.append(rnd.randrange(min(4 ** model[1], 6)))     # a small index range: batches keep hitting the same neighbourhoods
            model = model[2]
    return tuple(loc), model


@pytest.mark.parametrize("seed", range(10))
def test_random_models_under_random_deltas(bzk, seed):
    rnd = random.Random(7000 + seed)
    model = _random_model(rnd, 4)
    b = Both(bzk, model)
    history = []
    for step in range(10):
        delta = {}
        for _ in range(rnd.choice([1, 1, 2, 5, 30])):
            delta[_random_locator(rnd, model)[0]] = rnd.choice([None, 0, 1, rnd.randrange(pr.R_MOD), pr.R_MOD - 1])
        history.append(b.update(delta, bincode=(step % 4 == 3)))
        locs = [_random_locator(rnd, model, stop=0.3)[0] for _ in range(12)]
        b.check_get(locs)
        for _ in range(3):
            loc, sub = _random_locator(rnd, model, stop=0.4)
            if sub[0] == "list":
                b.check_prove(loc, [rnd.randrange(4 ** sub[1]) for _ in range(3)])
    # the rollbacks, newest first, bring every earlier root back (`ZkState::rollbacks`, src/zk/mod.rs:514-530)
    for rb in reversed(history[-3:]):
        b.update(rb)
    assert b.dev.stats()["keys"] >= len(b.ref.db)


def test_mpn_model_in_batches_equals_the_one_shot_seam(bzk):
    L, T = 6, 3
    rnd = random.Random(44)
    model = ps.mpn_model(L, T)
    dev = DeviceState(bzk, ps.model_bincode(model))
    everything = {}
    for height in range(1, 4):
        delta = {}
        for idx in rnd.sample(range(4 ** L), 80):
            for j in range(4):
                delta[(idx, j)] = rnd.randrange(1, 1 << 60)
            for slot in rnd.sample(range(4 ** T), 2):
                delta[(idx, 4, slot, 0)] = rnd.randrange(1, 1 << 20)
                delta[(idx, 4, slot, 1)] = rnd.choice([0, rnd.randrange(1, 1 << 40)])
        everything.update(delta)
        h, n = dev.update(_pairs(delta), height)
        live = {k: v for k, v in everything.items()}
        want_h, want_n = bzk.state_compress(ps.model_bincode(model), _pairs(live))
        assert (h, n) == (want_h, want_n)
    # an account's Merkle proof from the persistent state == the dedicated restatement of the account tree
    st = ps.PyMpnState(L, T)
    by_acct = {}
    for k, v in everything.items():
        by_acct.setdefault(k[0], {})[k[1:]] = v
    for idx, cells in by_acct.items():
        toks = {}
        for k, v in cells.items():
            if k[0] == 4:
                t = toks.setdefault(k[1], [0, 0])
                t[k[2]] = v
        st.set_account(idx, [cells[(j,)] for j in range(4)], {s: tuple(t) for s, t in toks.items()})
    assert dev.root()[0] == F(st.root())
    some = list(by_acct)[:5]
    got = dev.prove((), some)
    for idx, g in zip(some, got):
        assert g == [[F(x) for x in part] for part in st.prove(idx)]


def test_errors_change_nothing(bzk):
    m = ("list", 2, ("struct", [S, ("list", 1, S)]))
    b = Both(bzk, m)
    b.update({(3, 0): 9, (3, 1, 2): 4})
    one = F(1)
    bad = [[((16, 0), one)], [((3,), one)], [((3, 2), one)], [((3, 0, 0), one)], [((3, 0), one), ((3, 0), one)], [((3, 0), b"\xff" * 32)],
           [((2, 0), one), ((3, 1, 9), one)]]      # the second pair is the bad one: the first must not land either
    # bzk_last_refusal: which of the reference's errors each refusal stands for (src/zk/state/mod.rs:12-27) - what the Rust shim maps
    # to StateManagerError::{LocatorError(InvalidLocator), NonScalarLocatorError, NonTreeLocatorError} instead of one catch-all
    INVALID, NON_SCALAR, NON_TREE, DUPLICATE, NON_CANONICAL = 1, 2, 3, 4, 5
    want = [INVALID, NON_SCALAR, INVALID, INVALID, DUPLICATE, NON_CANONICAL, INVALID]
    for pairs, code in zip(bad, want):
        with pytest.raises(BzkError):
            b.dev.update(pairs, 99)
        assert bzk.last_refusal() == code, (pairs, bzk.last_refusal())
        b.check_root()
    b.check_get([(2, 0), (3, 0), (3, 1, 2)])
    assert bzk.last_refusal() == 0            # a call that was not refused clears it
    with pytest.raises(BzkError):
        b.dev.get([(3, 0, 0)])
    assert bzk.last_refusal() == INVALID
    with pytest.raises(BzkError):
        b.dev.prove((3,), [0])                # a struct is not a tree: NonTreeLocatorError
    assert bzk.last_refusal() == NON_TREE
    with pytest.raises(BzkError):
        b.dev.prove((9, 9, 9), [0])           # names nothing in the model: InvalidLocator, not NonTreeLocatorError
    assert bzk.last_refusal() == INVALID
    with pytest.raises(BzkError):
        b.dev.prove((3, 1), [4])              # beyond the list
    assert bzk.last_refusal() == INVALID
    with pytest.raises(BzkError):
        DeviceState(bzk, ps.model_bincode(m) + b"\0")
    b.update({(2, 0): 1})                     # still usable
    b.check_prove((3, 1), [2, 0])
    b.check_prove((), [3, 2, 15])


def test_growth_past_the_first_allocation(bzk):
    """the value store starts at 4096 slots and doubles: 3000 accounts of the MPN model touch ~40 k nodes"""
    L, T = 8, 2
    model = ps.mpn_model(L, T)
    dev = DeviceState(bzk, ps.model_bincode(model))
    rnd = random.Random(5)
    allp = {}
    for height in range(1, 4):
        delta = {(idx, j): rnd.randrange(1, 1 << 50) for idx in rnd.sample(range(4 ** L), 1000) for j in range(4)}
        allp.update(delta)
        h, n = dev.update(_pairs(delta), height)
        assert (h, n) == bzk.state_compress(ps.model_bincode(model), _pairs(allp))
    assert dev.stats()["slots"] > 4096


def test_mutated_delta_blobs_are_applied_or_refused_never_half(bzk):
    """differential fuzzing of the bincode entry: byte flips, truncations and splices of valid `ZkDeltaPairs` blobs either decode to a delta
    the model accepts - then the device state moves exactly as the restatement does - or are refused with the state untouched"""
    rnd = random.Random(99)
    model = ("list", 2, ("struct", [S, ("list", 1, S), ("struct", [S, S])]))
    b = Both(bzk, model)
    applied = refused = 0
    for case in range(250):
        delta = {}
        for _ in range(rnd.randint(1, 4)):
            i, f = rnd.randrange(16), rnd.randrange(3)
            loc = {0: (i, 0), 1: (i, 1, rnd.randrange(4)), 2: (i, 2, rnd.randrange(2))}[f]
            delta[loc] = rnd.choice([None, 0, rnd.randrange(1, pr.R_MOD)])
        blob = bytearray(ps.delta_bincode(delta))
        mode = rnd.randrange(4)
        if mode == 1:
            for _ in range(rnd.randint(1, 3)):
                blob[rnd.randrange(len(blob))] ^= 1 << rnd.randrange(8)
        elif mode == 2:
            blob = blob[:rnd.randrange(len(blob))]
        elif mode == 3:
            at = rnd.randrange(len(blob))
            blob = blob[:at] + bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 9))) + blob[at:]
        try:
            want = ps.delta_decode(bytes(blob))
            for loc in want:
                if ps.model_locate(model, loc)[0] != "scalar":
                    raise ValueError("NonScalarLocatorError")
        except (ValueError, IndexError):
            want = None
        if want is None:
            with pytest.raises(BzkError):
                b.dev.update_bincode(bytes(blob), b.h + 1)
            refused += 1
        else:
            b.h += 1
            b.ref.update_contract(want, b.h)
            assert b.dev.update_bincode(bytes(blob), b.h) == F(b.ref.hash) + b.ref.size.to_bytes(8, "little"), (case, mode, want)
            applied += 1
        b.check_root()
    assert applied > 60 and refused > 60
    b.check_get([(i, 0) for i in range(16)] + [(i,) for i in range(16)] + [()])
