"""Test infrastructure: a Python restatement of the MPN state model as the reference's state manager sees it - written from
/root/reference/src/zk/state/mod.rs:218-264 (`prove`), 310-420 (`set_data`: the level loop, defaults for missing nodes),
src/zk/mod.rs:401-423 (`compress_default`) and src/mpn/mod.rs:218-241 (`MpnConfig::state_model`):
    List{L, Struct{tx_nonce, withdraw_nonce, pub_x, pub_y, List{T, Struct{token_id, balance}}}}
Sparse: only populated accounts / token slots are stored; everything else is the default chain."""
from oracle import pyref as pr


class PyMpnState:
    def __init__(self, L, T):
        self.L, self.T = L, T
        self.accts = {}  # index -> {"cells": [4 ints], "tokens": {slot: (token_id, balance)}}
        # compress_default: Scalar -> 0; Struct -> H(field defaults); List -> log4_size times H4 of the item default
        self.tok_def = [0] * (T + 1)
        self.tok_def[T] = pr.poseidon([0, 0])
        for k in range(T - 1, -1, -1):
            self.tok_def[k] = pr.poseidon([self.tok_def[k + 1]] * 4)
        self.acct_def = [0] * (L + 1)
        self.acct_def[L] = pr.poseidon([0, 0, 0, 0, self.tok_def[0]])
        for k in range(L - 1, -1, -1):
            self.acct_def[k] = pr.poseidon([self.acct_def[k + 1]] * 4)

    def set_account(self, index, cells, tokens):
        """set_mpn_account (state/mod.rs:158-208): the four cells are written, the named token slots are written, the others stay"""
        a = self.accts.setdefault(index, {"cells": [0, 0, 0, 0], "tokens": {}})
        a["cells"] = list(cells)
        for slot, (tid, bal) in tokens.items():
            a["tokens"][slot] = (tid, bal)

    @staticmethod
    def _levels(leaves, depth, defaults):
        """maps {index: hash} for depth `depth` down to 0 from the populated leaves"""
        lv = [None] * (depth + 1)
        lv[depth] = dict(leaves)
        for k in range(depth - 1, -1, -1):
            cur = {}
            for p in {i >> 2 for i in lv[k + 1]}:
                cur[p] = pr.poseidon([lv[k + 1].get(4 * p + j, defaults[k + 1]) for j in range(4)])
            lv[k] = cur
        return lv

    def _token_levels(self, index):
        toks = self.accts.get(index, {"tokens": {}})["tokens"]
        return self._levels({s: pr.poseidon([t, b]) for s, (t, b) in toks.items()}, self.T, self.tok_def)

    def tokens_root(self, index):
        return self._token_levels(index)[0].get(0, self.tok_def[0])

    def leaf(self, index):
        if index not in self.accts:
            return self.acct_def[self.L]
        return pr.poseidon(self.accts[index]["cells"] + [self.tokens_root(index)])

    def _acct_levels(self):
        return self._levels({i: self.leaf(i) for i in self.accts}, self.L, self.acct_def)

    def root(self):
        return self._acct_levels()[0].get(0, self.acct_def[0])

    @staticmethod
    def _prove(lv, depth, defaults, index):
        out, cur = [], index
        for k in range(depth, 0, -1):
            base = cur & ~3
            out.append([lv[k].get(j, defaults[k]) for j in range(base, base + 4) if j != cur])
            cur >>= 2
        return out

    def prove(self, index):
        return self._prove(self._acct_levels(), self.L, self.acct_def, index)

    def prove_token(self, index, slot):
        return self._prove(self._token_levels(index), self.T, self.tok_def, slot)
