"""GPU parity of the general state-compress seam (bzk_state_compress = `ZkStateModel::compress`, src/zk/mod.rs:392-423 over
src/zk/state/mod.rs:66-90, 310-420) against tests/pystate.py's general restatement: the shapes of the reference's own state tests
(src/zk/test/mod.rs:43-287: scalar / struct / list of structs / struct holding a list), the MPN account model, random nestings
with random sparse pairs, the bincode entry, state_size, and the error cases."""
import random

import pytest

import pystate as ps
from oracle import pyref as pr

pytestmark = pytest.mark.gpu
S = ("scalar",)


def _run(bzk, model, pairs):
    got_h, got_n = bzk.state_compress(ps.model_bincode(model), [(k, pr.fr_to_mont_bytes(v % pr.R_MOD)) for k, v in pairs.items()])
    want_h, want_n = ps.compress(model, pairs)
    assert got_h == pr.fr_to_mont_bytes(want_h) and got_n == want_n
    # the bincode entry: what a Rust host holds already
    blob = bzk.state_compress_bincode(ps.model_bincode(model), ps.pairs_bincode(pairs))
    assert blob == pr.fr_to_mont_bytes(want_h) + want_n.to_bytes(8, "little")


def test_reference_test_shapes(bzk):
    # test_state_manager_scalar (src/zk/test/mod.rs:64-86)
    _run(bzk, S, {(): 0xF})
    _run(bzk, S, {})
    # test_state_manager_struct (:88-165): Struct{Scalar, Scalar}, fields set / overwritten / zeroed
    m = ("struct", [S, S])
    for pairs in ({(0,): 0xF}, {(0,): 0xF, (1,): 0xF0}, {(0,): 0xF00, (1,): 0xF0}, {(0,): 0, (1,): 0}, {}):
        _run(bzk, m, pairs)
    # test_state_manager_list (:167-287): List{3, Struct{Scalar, Scalar}} with items 62 and 33
    m = ("list", 3, ("struct", [S, S]))
    for pairs in ({(62, 0): 0xF00000}, {(62, 0): 0xF00000, (33, 0): 0xF}, {(62, 0): 0xF00000, (33, 0): 0xF, (33, 1): 0xF0},
                  {(62, 0): 0xF00000, (33, 0): 0, (33, 1): 0}, {(62, 0): 0}):
        _run(bzk, m, pairs)
    # test_zk_list_membership_proof (:43-62): Struct{Scalar, List{4, Scalar}} with all 256 items set
    m = ("struct", [S, ("list", 4, S)])
    _run(bzk, m, {(1, i): i for i in range(256)})
    _run(bzk, m, {(0,): 5, (1, 200): 7})


def test_mpn_model_equals_the_account_state_restatement(bzk):
    """`MpnConfig::state_model` through the GENERAL seam == the dedicated restatement of the account tree (PyMpnState)"""
    L, T = 5, 3
    rnd = random.Random(12)
    st = ps.PyMpnState(L, T)
    pairs = {}
    for idx in rnd.sample(range(4 ** L), 40):
        cells = [rnd.randrange(1, 1 << 30) for _ in range(4)]
        toks = {rnd.randrange(4 ** T): (rnd.randrange(1, 1 << 20), rnd.randrange(1, 1 << 40)) for _ in range(rnd.randint(0, 4))}
        st.set_account(idx, cells, toks)
        for j, c in enumerate(cells):
            pairs[(idx, j)] = c
        for slot, (tid, bal) in toks.items():
            pairs[(idx, 4, slot, 0)] = tid
            pairs[(idx, 4, slot, 1)] = bal
    model = ps.mpn_model(L, T)
    assert ps.compress(model, pairs)[0] == st.root()
    _run(bzk, model, pairs)


def _random_model(rnd, depth):
    k = rnd.random()
    if depth == 0 or k < 0.25:
        return S
    if k < 0.65:
        return ("struct", [_random_model(rnd, depth - 1) for _ in range(rnd.randint(1, 5))])
    return ("list", rnd.randint(0, 4), _random_model(rnd, depth - 1))


def _random_locator(rnd, model):
    loc = []
    while model[0] != "scalar":
        if model[0] == "struct":
            f = rnd.randrange(len(model[1]))
            loc.append(f)
            model = model[1][f]
        else:
            loc.append(rnd.randrange(4 ** model[1]))
            model = model[2]
    return tuple(loc)


@pytest.mark.parametrize("seed", range(12))
def test_random_nestings(bzk, seed):
    rnd = random.Random(1000 + seed)
    model = _random_model(rnd, 4)
    pairs = {}
    for _ in range(rnd.choice([0, 1, 3, 20, 200])):
        pairs[_random_locator(rnd, model)] = rnd.choice([0, 1, rnd.randrange(pr.R_MOD), pr.R_MOD - 1])
    _run(bzk, model, pairs)


def test_wide_struct_and_deep_lists(bzk):
    m = ("struct", [S] * 16)                       # MAX_ARITY fields
    _run(bzk, m, {(i,): i + 1 for i in range(0, 16, 3)})
    m = ("list", 12, ("list", 0, S))               # 2^24 slots, log4 = 0 inner list = the item itself
    rnd = random.Random(3)
    _run(bzk, m, {(rnd.randrange(4 ** 12), 0): rnd.randrange(pr.R_MOD) for _ in range(600)})


def test_errors_where_the_reference_errors(bzk):
    from bazuka_amd import BzkError
    m = ("list", 2, ("struct", [S, S]))
    one = pr.fr_to_mont_bytes(1)
    bad = [[((16, 0), one)],            # index >= 4^log4_size: ZkLocatorError::InvalidLocator
           [((3,), one)],               # ends at a struct: NonScalarLocatorError
           [((3, 2), one)],             # field index out of range
           [((3, 0, 0), one)],          # continues below a scalar
           [((3, 0), one), ((3, 0), one)],   # a HashMap holds a key once
           [((3, 0), b"\xff" * 32)]]    # not a field element
    for pairs in bad:
        with pytest.raises(BzkError):
            bzk.state_compress(ps.model_bincode(m), pairs)
    with pytest.raises(BzkError):
        bzk.state_compress((1).to_bytes(4, "little") + (17).to_bytes(8, "little") + ps.model_bincode(S) * 17, [])   # > MAX_ARITY fields
    with pytest.raises(BzkError):
        bzk.state_compress(ps.model_bincode(m) + b"\0", [])                                                             # trailing bytes
    # the context is still usable
    _run(bzk, m, {(3, 0): 9})
