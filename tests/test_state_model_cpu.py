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
