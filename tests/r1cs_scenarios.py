"""Shared by tests/test_pycircuit_cpu.py, tests/test_gpu_mpn_prove.py and tests/golden/make_r1cs_fixtures.py: the fixed MPN
scenarios whose circuit instances are pinned.  A scenario = a deterministic world (accounts from seeds, EdDSA is
deterministic), a queue of transactions that fills only part of the batch (so enabled and `::null` slots both occur),
turned into an `MpnWork` on the wire; the worker side decodes it and synthesizes."""
import json
import os

from bazuka_amd import lib as L
from oracle import pyref as pr

F = pr.fr_to_mont_bytes
ZIESHA = F(1)
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VKS = [bytes.fromhex(h) for h in json.load(open(os.path.join(G, "reference_vectors.json")))["verifying_keys_bincode_hex"]]
PROVER = bytes(range(1, 33))
KIND = {"deposit": 0, "withdraw": 1, "update": 2}

# name -> (kind, log4_tree, log4_token_tree, log4_batch)
SCENARIOS = {
    "update_3_3_1": ("update", 3, 3, 1),
    "deposit_3_3_1": ("deposit", 3, 3, 1),
    "withdraw_3_3_1": ("withdraw", 3, 3, 1),
    "update_15_3_1": ("update", 15, 3, 1),
    "update_15_3_2": ("update", 15, 3, 2),   # the 2^20-class circuit of BASELINE configs[1..2]: 903 037 constraints
}


def make_work(name, dev=None):
    """dev: a Bzk context - the validator-side builder then batches its Merkle hashing on the GPU (bzk_mpn_set_device)"""
    kind, L4, T4, B4 = SCENARIOS[name]
    w = L.MpnWorld(L4, T4)
    if dev is not None:
        w.set_device(dev)
    n_acct = 4 if B4 == 1 else 12
    for i in range(n_acct):
        w.add_account(i * 37 % (4 ** L4) if L4 > 3 else i, b"acct%d" % i, ZIESHA, 10 ** 9)
    idx = [(i * 37 % (4 ** L4) if L4 > 3 else i) for i in range(n_acct)]
    fresh = 4 ** L4 - 2
    w.add_key(fresh, b"fresh")
    w.set_height(11)
    if kind == "update":
        n_tx = 3 if B4 == 1 else 13      # of 4 / 16 slots: the rest are UpdateTransition::null
        for t in range(n_tx - 1):
            w.push_tx(idx[t % n_acct], idx[(t + 1) % n_acct], ZIESHA, 1000 + t, ZIESHA, 3 + t)
        w.push_tx(idx[2], fresh, ZIESHA, 10, ZIESHA, 1)   # to an account slot that does not exist yet
    elif kind == "deposit":
        w.push_deposit(idx[0], ZIESHA, 1000)
        w.push_deposit(fresh, F(777), 5)                  # new account, custom token
    else:
        w.push_withdraw(idx[0], ZIESHA, 400, ZIESHA, 2)
        w.push_withdraw(idx[1], ZIESHA, 9, ZIESHA, 0)
    lb = [1, 1, 1]
    lb[KIND[kind]] = B4
    work = w.make_work(KIND[kind], VKS, 5000, log4_batches=tuple(lb), num_batches=(1, 2, 3), state_size=42)
    return work.encode()


def product_views(blob, threads=0):
    """worker side through the C ABI: bzk_mpn_work_decode -> bzk_mpn_work_synthesize(record_matrices) -> 15 views"""
    dec = L.MpnWork.decode(blob)
    r = dec.synthesize(PROVER, threads=threads, record_matrices=True)
    views = {name: r.view(name) for name in L.R1cs.VIEWS}
    return r, views, dec.commitment(PROVER)
