"""CPU suite: `ZkStateModel::compress_default` behind the C ABI (bzk_state_model_default, host code) equals the general Python
restatement (tests/pystate.py), and the restatement itself reproduces the dedicated MPN account-state restatement."""
import random

import pystate as ps
from oracle import pyref as pr

S = ("scalar",)


def test_model_default_matches_restatement():
    from bazuka_amd import lib as L
    for model in (S, ("struct", [S, S]), ("list", 3, ("struct", [S, S])), ("struct", [S, ("list", 4, S)]), ps.mpn_model(15, 3),
                  ("list", 0, S), ("struct", [("list", 2, ("struct", [S, ("list", 1, S)]))] * 3)):
        assert L.state_model_default(ps.model_bincode(model)) == pr.fr_to_mont_bytes(ps.model_default(model))


def test_general_restatement_reproduces_mpn_restatement():
    L_, T = 4, 2
    rnd = random.Random(5)
    st = ps.PyMpnState(L_, T)
    pairs = {}
    for idx in rnd.sample(range(4 ** L_), 10):
        cells = [rnd.randrange(1 << 20) for _ in range(4)]
        toks = {rnd.randrange(4 ** T): (rnd.randrange(1, 99), rnd.randrange(1, 1 << 30)) for _ in range(2)}
        st.set_account(idx, cells, toks)
        for j, c in enumerate(cells):
            pairs[(idx, j)] = c
        for slot, (tid, bal) in toks.items():
            pairs[(idx, 4, slot, 0)], pairs[(idx, 4, slot, 1)] = tid, bal
    h, n = ps.compress(ps.mpn_model(L_, T), pairs)
    assert h == st.root() and n == sum(1 for v in pairs.values() if v)
    assert ps.compress(ps.mpn_model(L_, T), {})[0] == ps.PyMpnState(L_, T).root()


def test_malformed_models_are_refused():
    import ctypes as C
    from bazuka_amd import load_library
    lib = load_library()
    out = C.create_string_buffer(32)
    for blob in (b"", b"\x03\0\0\0", (1).to_bytes(4, "little") + (0).to_bytes(8, "little"), (2).to_bytes(4, "little") + bytes([40]) + (0).to_bytes(4, "little")):
        assert lib.bzk_state_model_default(blob, len(blob), out) == -1


def test_membership_proof_structure_against_the_reference_sum_hasher_constant():
    """The reference's own state-manager test (src/zk/test/mod.rs:7-18, 43-70) swaps Poseidon for an ADDITIVE hasher, fills a
    `List{log4_size: 4, Scalar}` with item i = i and asserts that every leaf plus all the values of its membership proof sums to
    32640 = sum(0..255): a reference-held constant for the SHAPE of a proof - log4_size levels, exactly the three siblings of the node
    on the path at each level.  The restatement the GPU trees are compared with (pystate `_levels` / `_prove`; tests/test_gpu_tree4.py,
    test_gpu_mpn_tree.py) run over that hasher must reproduce it."""
    add = lambda vals: sum(vals) % pr.R_MOD  # noqa: E731 - SumHasher::hash
    depth = 4
    defaults = [0] * (depth + 1)      # compress_default under the additive hasher: sums of zeros
    lv = ps.PyMpnState._levels({i: i for i in range(256)}, depth, defaults, hasher=add)
    assert lv[0][0] == 32640
    for i in range(256):
        proof = ps.PyMpnState._prove(lv, depth, defaults, i)
        assert len(proof) == depth and all(len(part) == 3 for part in proof)
        assert (i + sum(v for part in proof for v in part)) % pr.R_MOD == 32640
    # a sparse tree: missing nodes take the default of their level (zero here), the constant becomes the sum of what is populated
    some = {3: 30, 77: 700, 200: 9}
    lv = ps.PyMpnState._levels(some, depth, defaults, hasher=add)
    for i, v in some.items():
        assert (v + sum(x for part in ps.PyMpnState._prove(lv, depth, defaults, i) for x in part)) % pr.R_MOD == 739


def test_state_manager_restatement_replays_the_reference_membership_test():
    """`test_zk_list_membership_proof` (src/zk/test/mod.rs:43-62) replayed LITERALLY on the pair-at-a-time restatement of the state
    manager (pystate.PyKvState: `update_contract` -> `set_data`, then `prove`) over the additive hasher: Struct{Scalar, List{4, Scalar}},
    256 deltas of one pair, and leaf + proof = 32640 for every index - the constant the reference asserts."""
    add = lambda vals: sum(vals) % pr.R_MOD  # noqa: E731
    st = ps.PyKvState(("struct", [S, ("list", 4, S)]), hasher=add)
    for i in range(256):
        st.update_contract({(1, i): i}, i + 1)
    for i in range(256):
        proof = st.prove((1,), i)
        assert len(proof) == 4 and (i + sum(v for part in proof for v in part)) % pr.R_MOD == 32640
    assert st.root() == (32640, 255) and st.get_data((1,)) == 32640      # item 0 holds zero: 255 stored scalars


def test_state_manager_restatement_equals_one_shot_compress():
    """pair-at-a-time updates (with removals, overwrites and nodes falling back to their defaults) end at the value the one-shot
    `compress` restatement gives for the surviving pairs; the rollback deltas bring every earlier root back"""
    rnd = random.Random(9)
    m = ("list", 2, ("struct", [S, ("list", 1, S), ("struct", [S, S]), ("list", 0, S)]))
    st = ps.PyKvState(m)
    live, roots, rollbacks = {}, [st.root()], []
    for step in range(12):
        delta = {}
        for _ in range(rnd.randint(1, 4)):
            i, f = rnd.randrange(5), rnd.randrange(4)
            loc = {0: (i, 0), 1: (i, 1, rnd.randrange(4)), 2: (i, 2, rnd.randrange(2)), 3: (i, 3, 0)}[f]
            delta[loc] = rnd.choice([None, 0, rnd.randrange(1, pr.R_MOD)])
        rollbacks.append(st.update_contract(delta, step + 1))
        for k, v in delta.items():
            if v:
                live[k] = v
            else:
                live.pop(k, None)
        assert st.root() == ps.compress(m, live)
        roots.append(st.root())
    for rb in reversed(rollbacks):
        roots.pop()
        st.update_contract(rb, 0)
        assert st.root() == roots[-1]
    assert st.db == {} and st.root() == (ps.model_default(m), 0)


def test_delta_bincode_layout():
    blob = ps.delta_bincode({(3, 1): 5, (2,): None})
    assert blob[:8] == (2).to_bytes(8, "little")
    assert blob[8:16] == (2).to_bytes(8, "little") and blob[16:32] == (3).to_bytes(8, "little") + (1).to_bytes(8, "little")
    assert blob[32] == 1 and blob[33:65] == pr.fr_to_mont_bytes(5)
    assert blob[65:73] == (1).to_bytes(8, "little") and blob[73:81] == (2).to_bytes(8, "little") and blob[81:] == b"\x00"
