"""SURVEY 8f-3, second half: the device-resident MPN account state (bzk_mpn_tree_*) - batched `set_mpn_account`,
`get_mpn_account` and `prove` for the production state model List{L, Struct{S, S, S, S, List{T, Struct{S, S}}}} - against
(1) a Python restatement of src/zk/state/mod.rs on random update sequences, (2) the host-side sparse state of the witness
builder (bzk_mpn_*): same roots, and the Merkle proofs its transitions carry."""
import random

import pytest

import bincode_ref as B
from bazuka_amd import lib as L
from bazuka_amd.lib import BzkError
from oracle import pyref as pr
from pystate import PyMpnState

pytestmark = pytest.mark.gpu
F, U = pr.fr_to_mont_bytes, pr.fr_from_mont_bytes


def _triples(raw, depth):
    return [[U(raw[96 * k + 32 * j:96 * k + 32 * j + 32]) for j in range(3)] for k in range(depth)]


@pytest.mark.parametrize("L4,T4", [(5, 2), (3, 3), (7, 1)])
def test_random_update_sequences_vs_state_manager_restatement(bzk, L4, T4):
    rnd = random.Random(1000 * L4 + T4)
    py = PyMpnState(L4, T4)
    tree = bzk.mpn_tree_create(L4, T4, 200)
    try:
        assert bzk.mpn_tree_root(tree) == F(py.root())                      # the empty state: compress_default
        n_acct, ts = 4 ** L4, 4 ** T4
        known = []
        for rnd_no in range(6):
            fresh = rnd.sample(range(n_acct), 12)
            touched = list(dict.fromkeys(fresh + rnd.sample(known, min(len(known), 8))))   # new accounts and earlier ones, distinct
            batch = []
            for idx in touched:
                cells = [rnd.randrange(1 << 32), rnd.randrange(1 << 32), rnd.randrange(pr.R_MOD), rnd.randrange(pr.R_MOD)]
                toks = {s: (rnd.randrange(1, pr.R_MOD), rnd.randrange(1 << 64)) for s in rnd.sample(range(ts), rnd.randint(0, min(3, ts)))}
                if rnd.random() < 0.2 and idx in py.accts and py.accts[idx]["tokens"]:
                    toks[next(iter(py.accts[idx]["tokens"]))] = (0, 0)      # a slot emptied again
                py.set_account(idx, cells, toks)
                batch.append((idx, [F(c) for c in cells], {s: (F(t), F(b)) for s, (t, b) in toks.items()}))
            known = list(dict.fromkeys(known + touched))
            bzk.mpn_tree_set_accounts(tree, batch)
            assert bzk.mpn_tree_root(tree) == F(py.root()), rnd_no
        # reads: populated accounts and one that was never set
        probe = rnd.sample(known, 10) + [next(i for i in range(n_acct) if i not in py.accts)]
        got = bzk.mpn_tree_get_accounts(tree, probe, T4)
        for idx, g in zip(probe, got):
            a = py.accts.get(idx, {"cells": [0, 0, 0, 0], "tokens": {}})
            assert [U(c) for c in g["cells"]] == a["cells"]
            assert U(g["tokens_root"]) == py.tokens_root(idx)
            assert {s: (U(t), U(b)) for s, (t, b) in g["tokens"].items()} == {s: v for s, v in a["tokens"].items() if v[0]}
        # proofs: account level and token level, incl. an unset account and an unset token slot
        raw = bzk.mpn_tree_prove(tree, probe, L4)
        for q, idx in enumerate(probe):
            assert _triples(raw[q * L4 * 96:(q + 1) * L4 * 96], L4) == py.prove(idx)
        slots = [rnd.randrange(ts) for _ in probe]
        raw = bzk.mpn_tree_prove_token(tree, probe, slots, T4)
        for q, (idx, s) in enumerate(zip(probe, slots)):
            assert _triples(raw[q * T4 * 96:(q + 1) * T4 * 96], T4) == py.prove_token(idx, s)
        # argument checking: duplicate account in one batch, token slot out of range, pool exhausted
        one = (probe[0], [F(1)] * 4, {})
        with pytest.raises(BzkError):
            bzk.mpn_tree_set_accounts(tree, [one, one])
        with pytest.raises(BzkError):
            bzk.mpn_tree_set_accounts(tree, [(probe[0], [F(1)] * 4, {ts: (F(1), F(1))})])
        assert bzk.mpn_tree_root(tree) == F(py.root())                      # refused calls change nothing
    finally:
        bzk.mpn_tree_free(tree)


def test_pool_exhaustion_is_an_error_not_a_crash(bzk):
    tree = bzk.mpn_tree_create(4, 1, 3)
    try:
        bzk.mpn_tree_set_accounts(tree, [(i, [F(i)] * 4, {}) for i in range(3)])
        root = bzk.mpn_tree_root(tree)
        with pytest.raises(BzkError, match="allocation"):
            bzk.mpn_tree_set_accounts(tree, [(7, [F(7)] * 4, {})])
        bzk.mpn_tree_set_accounts(tree, [(1, [F(9)] * 4, {})])              # existing accounts can still be written
        assert bzk.mpn_tree_root(tree) != root
    finally:
        bzk.mpn_tree_free(tree)


def test_production_depth_same_root_and_proofs_as_the_witness_builders_state(bzk):
    """L = 15, T = 3 (src/config/blockchain.rs:22-26): the device state and the host-side sparse state of the witness builder
    (bzk_mpn_*) hold the same accounts -> same root; the Merkle proofs inside the first transition of an update work (taken by
    the host builder against that state, src/mpn/update.rs:92-250) are the device's `prove` / `prove_token` answers."""
    L4, T4 = 15, 3
    ZIESHA = F(1)
    rnd = random.Random(15)
    w = L.MpnWorld(L4, T4)
    idxs = rnd.sample(range(4 ** L4), 40)
    accts = []
    for i, idx in enumerate(idxs):
        pub = w.add_account(idx, b"acct%d" % i, ZIESHA, 10 ** 9 + i)
        accts.append((idx, [F(0), F(0), pub[:32], pub[32:]], {0: (ZIESHA, F(10 ** 9 + i))}))
    tree = bzk.mpn_tree_create(L4, T4, 64)
    try:
        bzk.mpn_tree_set_accounts(tree, accts[:25])
        bzk.mpn_tree_set_accounts(tree, accts[25:])
        assert bzk.mpn_tree_root(tree) == w.root()
        w.push_tx(idxs[0], idxs[1], ZIESHA, 1000, ZIESHA, 7)
        import r1cs_scenarios as S
        work = B.decode(B.MpnWork, w.make_work(2, S.VKS, 1, log4_batches=(1, 1, 1)).encode())
        tr = work["data"][1][0]
        assert tr["src_index"] == idxs[0] and tr["dst_index"] == idxs[1]
        assert work["public_inputs"]["state"] == bzk.mpn_tree_root(tree)
        raw = bzk.mpn_tree_prove(tree, [idxs[0]], L4)
        assert [raw[96 * k:96 * k + 96] for k in range(L4)] == tr["src_proof"]
        raw = bzk.mpn_tree_prove_token(tree, [idxs[0], idxs[1]], [tr["src_token_index"], tr["dst_token_index"]], T4)
        assert [raw[96 * k:96 * k + 96] for k in range(T4)] == tr["src_balance_proof"]
        assert [raw[96 * (T4 + k):96 * (T4 + k) + 96] for k in range(T4)] == tr["dst_balance_proof"]
        got = bzk.mpn_tree_get_accounts(tree, [idxs[0]], T4)[0]
        assert got["tokens_root"] == tr["src_before_balances_hash"] and got["cells"][2:] == [accts[0][1][2], accts[0][1][3]]
        # the state after the work, replayed on the device: sender (nonce + 1, balance - amount - fee), receiver (+ amount)
        s_bal, d_bal = 10 ** 9 - 1000 - 7, 10 ** 9 + 1 + 1000
        bzk.mpn_tree_set_accounts(tree, [(idxs[0], [F(1), F(0)] + accts[0][1][2:], {0: (ZIESHA, F(s_bal))}),
                                         (idxs[1], accts[1][1], {0: (ZIESHA, F(d_bal))})])
        assert bzk.mpn_tree_root(tree) == work["public_inputs"]["next_state"] == w.root()
    finally:
        bzk.mpn_tree_free(tree)


@pytest.mark.parametrize("L4,T4", [(3, 2), (4, 1), (2, 3)])
def test_dense_state_compress_equals_restatement_and_device_tree(bzk, L4, T4):
    """bzk_mpn_state_compress_dev = `ZkStateModel::compress` of a fully populated MPN-shaped state (BASELINE configs[4], secondary
    instance): root == the state manager restatement == the device-resident tree after setting every account"""
    import torch
    from util import to_dev
    rnd = random.Random(100 * L4 + T4)
    n_acct, ts = 4 ** L4, 4 ** T4
    py = PyMpnState(L4, T4)
    cells, toks, batch = [], [], []
    for a in range(n_acct):
        c = [rnd.randrange(1 << 32), rnd.randrange(1 << 32), rnd.randrange(pr.R_MOD), rnd.randrange(pr.R_MOD)]
        t = {s: (rnd.randrange(pr.R_MOD), rnd.randrange(1 << 64)) if rnd.random() < 0.7 else (0, 0) for s in range(ts)}
        py.set_account(a, c, t)
        cells += [F(x) for x in c]
        for s in range(ts):
            toks += [F(t[s][0]), F(t[s][1])]
        batch.append((a, [F(x) for x in c], {s: (F(t[s][0]), F(t[s][1])) for s in range(ts)}))
    root = bzk.mpn_state_compress_dev(L4, T4, to_dev(b"".join(cells)), to_dev(b"".join(toks)))
    assert root == F(py.root())
    tree = bzk.mpn_tree_create(L4, T4, n_acct)
    try:
        bzk.mpn_tree_set_accounts(tree, batch)
        assert bzk.mpn_tree_root(tree) == root
    finally:
        bzk.mpn_tree_free(tree)
    # the empty dense state is the default chain
    z = torch.zeros(n_acct * 4 * 32, dtype=torch.uint8, device="cuda")
    zt = torch.zeros(n_acct * ts * 2 * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert bzk.mpn_state_compress_dev(L4, T4, z, zt) == F(PyMpnState(L4, T4).root())
