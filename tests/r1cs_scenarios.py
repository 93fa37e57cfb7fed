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
    # the chain's real shapes (src/config/blockchain.rs:22-26: log4_tree 15, token tree 3, deposit / withdraw batches 4^3,
    # update batches 4^4; one of each per block, :326-328): 2^21 / 2^22 / 2^24 domains
    "deposit_15_3_3": ("deposit", 15, 3, 3),
    "withdraw_15_3_3": ("withdraw", 15, 3, 3),
    "update_15_3_4": ("update", 15, 3, 4),
}
PRODUCTION = ("deposit_15_3_3", "withdraw_15_3_3", "update_15_3_4")


def make_work(name, dev=None):
    """dev: a Bzk context - the validator-side builder then batches its Merkle hashing on the GPU (bzk_mpn_set_device)"""
    kind, L4, T4, B4 = SCENARIOS[name]
    w = L.MpnWorld(L4, T4)
    if dev is not None:
        w.set_device(dev)
    if name in PRODUCTION:
        return _make_production_work(kind, L4, T4, B4, dev)
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


def _make_production_work(kind, L4, T4, B4, dev):
    """about three quarters of the 4^B4 slots enabled, the rest `::null`; 48 funded accounts scattered over the 4^15 tree
    holding two tokens each; transfers to fresh slots, custom-token transfers with Ziesha fees, deposits that create accounts
    and token slots, several withdrawals per account"""
    w = L.MpnWorld(L4, T4)
    if dev is not None:
        w.set_device(dev)
    TOK = F(777)
    n_acct, size = 48, 4 ** L4
    idx = [(i * 22369621 + 5) % size for i in range(n_acct)]      # distinct: 22369621 is odd, size a power of two
    for i, a in enumerate(idx):
        w.add_account(a, b"acct%d" % i, ZIESHA, 10 ** 12)
    fresh = [size - 2 - 3 * j for j in range(24)]
    for j, a in enumerate(fresh):
        w.add_key(a, b"fresh%d" % j)
    w.set_height(11)
    n_slots = 4 ** B4
    n_tx = n_slots * 3 // 4 + 1
    if kind == "deposit":
        for t in range(n_tx):
            if t % 5 == 3:
                w.push_deposit(fresh[(t // 5) % len(fresh)], TOK if t % 2 else ZIESHA, 5 + t)     # new account (or its second token)
            else:
                w.push_deposit(idx[(7 * t) % n_acct], TOK if t % 3 == 1 else ZIESHA, 1000 + t)
    elif kind == "withdraw":
        for t in range(n_tx):
            w.push_withdraw(idx[(5 * t) % n_acct], ZIESHA, 400 + t, ZIESHA, t % 4)
    else:
        for t in range(n_tx):
            src = idx[(11 * t) % n_acct]
            if t % 9 == 4:
                w.push_tx(src, fresh[(t // 9) % len(fresh)], ZIESHA, 10 + t, ZIESHA, 1)          # to a slot that may not exist yet
            else:
                w.push_tx(src, idx[(11 * t + 17) % n_acct], ZIESHA, 1000 + t, ZIESHA, 3 + t % 5)
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


def product_hashes(blob, threads=0):
    """as product_views, for instances too large to copy around: sha256 straight off the generator's own arrays"""
    import hashlib
    dec = L.MpnWork.decode(blob)
    r = dec.synthesize(PROVER, threads=threads, record_matrices=True)
    return dec, r, {name: hashlib.sha256(r.raw(name)).hexdigest() if len(r.raw(name)) > 1 or name[0] not in "vc" else
                    hashlib.sha256(r.view(name)).hexdigest() for name in L.R1cs.VIEWS}
