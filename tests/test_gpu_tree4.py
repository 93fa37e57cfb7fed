"""GPU parity: device-resident 4-ary tree with batched updates and proofs (bzk_tree4_*) vs a plain Python restatement
of `KvStoreStateManager::{set_data, prove}` (src/zk/state/mod.rs:218-264, 310-420) on the oracle's Poseidon."""
import random

import pytest
import torch

from util import fr_bytes, fr_list, to_dev

pytestmark = pytest.mark.gpu


class PyTree:
    """dense / sparse-with-defaults 4-ary tree: level[0] = leaves ... level[depth] = root"""

    def __init__(self, pr, depth, leaves=None, default=0):
        self.pr, self.depth = pr, depth
        self.defaults = [default]
        for _ in range(depth):
            self.defaults.append(pr.poseidon([self.defaults[-1]] * 4))
        self.level = [dict() for _ in range(depth + 1)]
        if leaves is not None:
            for i, v in enumerate(leaves):
                self.set(i, v)

    def get(self, lv, i):
        return self.level[lv].get(i, self.defaults[lv])

    def set(self, i, v):
        self.level[0][i] = v
        for lv in range(self.depth):
            base = i & ~3
            i >>= 2
            self.level[lv + 1][i] = self.pr.poseidon([self.get(lv, base + j) for j in range(4)])

    def root(self):
        return self.get(self.depth, 0)

    def prove(self, i):
        out = []
        for lv in range(self.depth):
            base = i & ~3
            out.append([self.get(lv, base + j) for j in range(4) if base + j != i])
            i >>= 2
        return out


def _proof_bytes(pr, proofs):
    return b"".join(pr.fr_to_mont_bytes(x) for proof in proofs for triple in proof for x in triple)


def test_tree4_from_leaves_update_prove_vs_python(bzk, pr):
    log4 = 3
    n = 4 ** log4
    leaves = fr_list(n, 41)
    py = PyTree(pr, log4, leaves)
    tree = bzk.tree4_create(log4, to_dev(fr_bytes(leaves)))
    assert bzk.tree4_root(tree) == pr.fr_to_mont_bytes(py.root())
    assert bzk.tree4_root(tree) == bzk.merkle4_root(fr_bytes(leaves), log4)
    idx = [0, 5, 6, 7, 63, 21, 5]                       # siblings, both ends, a repeated index (the later entry wins)
    vals = fr_list(len(idx), 42)
    for i, v in zip(idx, vals):
        py.set(i, v)
    bzk.tree4_update(tree, idx, fr_bytes(vals))
    assert bzk.tree4_root(tree) == pr.fr_to_mont_bytes(py.root())
    assert bzk.tree4_node(tree, log4, 5) == pr.fr_to_mont_bytes(vals[6])
    assert bzk.tree4_node(tree, 1, 3) == pr.fr_to_mont_bytes(py.get(log4 - 1, 3))
    q = [0, 5, 22, 63]
    assert bzk.tree4_prove(tree, q, log4) == _proof_bytes(pr, [py.prove(i) for i in q])
    bzk.tree4_free(tree)


def test_tree4_empty_tree_and_sparse_updates_at_account_tree_depth(bzk, pr):
    """log4 = 12 (16.7 M leaves, 716 MB of nodes): empty tree of the default leaf, 40 scattered updates in two batches,
    proofs that cross default sub-trees; the Python side keeps only the touched paths."""
    log4 = 12
    rnd = random.Random(5)
    py = PyTree(pr, log4, None, default=0)
    tree = bzk.tree4_create(log4, None, bytes(32))
    assert bzk.tree4_root(tree) == pr.fr_to_mont_bytes(py.root())
    for batch in range(2):
        idx = [rnd.randrange(4 ** log4) for _ in range(18)] + [0, 4 ** log4 - 1]
        vals = fr_list(len(idx), 100 + batch)
        for i, v in zip(idx, vals):
            py.set(i, v)
        bzk.tree4_update(tree, idx, fr_bytes(vals))
        assert bzk.tree4_root(tree) == pr.fr_to_mont_bytes(py.root())
    q = idx[:5] + [12345]
    assert bzk.tree4_prove(tree, q, log4) == _proof_bytes(pr, [py.prove(i) for i in q])
    bzk.tree4_free(tree)


def test_tree4_bad_arguments(bzk):
    from bazuka_amd.lib import BzkError
    with pytest.raises(BzkError):
        bzk.tree4_create(16, None, bytes(32))
    tree = bzk.tree4_create(2, None, bytes(32))
    with pytest.raises(BzkError):
        bzk.tree4_update(tree, [16], bytes(32))
    with pytest.raises(BzkError):
        bzk.tree4_prove(tree, [99], 2)
    bzk.tree4_free(tree)


def test_tree4_large_batch_matches_full_rebuild(bzk):
    """size-independent property at 2^20 leaves: 50 000 updates through the batched path == rebuilding the whole tree"""
    log4 = 10
    n = 4 ** log4
    g = torch.Generator(device="cuda").manual_seed(3)
    leaves = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    leaves[:, 31] &= 0x3F
    leaves = leaves.contiguous()
    torch.cuda.synchronize()
    tree = bzk.tree4_create(log4, leaves)
    rnd = random.Random(9)
    idx = [rnd.randrange(n) for _ in range(50000)]
    vals = fr_bytes(fr_list(len(idx), 77))
    bzk.tree4_update(tree, idx, vals)
    host = bytearray(leaves.cpu().numpy().tobytes())
    for k, i in enumerate(idx):
        host[32 * i:32 * i + 32] = vals[32 * k:32 * k + 32]
    assert bzk.tree4_root(tree) == bzk.merkle4_root(bytes(host), log4)
    bzk.tree4_free(tree)
