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
    def _levels(leaves, depth, defaults, hasher=None):
        """maps {index: hash} for depth `depth` down to 0 from the populated leaves.  hasher: the `ZkHasher` (default Poseidon; the
        reference's own membership-proof test runs the state manager over an additive SumHasher, src/zk/test/mod.rs:7-18)"""
        hasher = hasher or pr.poseidon
        lv = [None] * (depth + 1)
        lv[depth] = dict(leaves)
        for k in range(depth - 1, -1, -1):
            cur = {}
            for p in {i >> 2 for i in lv[k + 1]}:
                cur[p] = hasher([lv[k + 1].get(4 * p + j, defaults[k + 1]) for j in range(4)])
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


# ---------------------------------------------------------------------------------------------------------------------------
# General restatement of `ZkStateModel::compress` (test infrastructure for bzk_state_compress): ANY nesting of
#   ("scalar",) | ("struct", [field models]) | ("list", log4_size, item model)
# written from /root/reference/src/zk/mod.rs:392-423 (`compress`, `compress_default`) and src/zk/state/mod.rs:310-420 (`set_data`:
# struct = H(fields), list = 4-ary tree whose missing nodes take the default of their level, log4_size = 0 -> the item itself).
# Values are plain integers mod r.  `compress` returns (state_hash, state_size).
# ---------------------------------------------------------------------------------------------------------------------------
def model_default(model):
    if model[0] == "scalar":
        return 0
    if model[0] == "struct":
        return pr.poseidon([model_default(f) for f in model[1]])
    d = model_default(model[2])
    for _ in range(model[1]):
        d = pr.poseidon([d, d, d, d])
    return d


def model_locate(model, locator):
    """`ZkStateModel::locate`: the sub-model a locator names (ValueError where the reference errors or panics)"""
    cur = model
    for l in locator:
        if cur[0] == "struct":
            if l >= len(cur[1]):
                raise ValueError("field index out of range")
            cur = cur[1][l]
        elif cur[0] == "list":
            if l >= 1 << (2 * cur[1]):
                raise ValueError("InvalidLocator")
            cur = cur[2]
        else:
            raise ValueError("InvalidLocator")
    return cur


def _value(model, pairs, depth):
    """pairs: non-empty list of (locator, value) sharing their first `depth` indices"""
    if model[0] == "scalar":
        assert len(pairs) == 1 and len(pairs[0][0]) == depth
        return pairs[0][1]
    if model[0] == "struct":
        vals = []
        for f, fm in enumerate(model[1]):
            sub = [p for p in pairs if p[0][depth] == f]
            vals.append(_value(fm, sub, depth + 1) if sub else model_default(fm))
        return pr.poseidon(vals)
    log4, item = model[1], model[2]
    level = {}
    for idx in {p[0][depth] for p in pairs}:
        level[idx] = _value(item, [p for p in pairs if p[0][depth] == idx], depth + 1)
    d = model_default(item)
    for _ in range(log4):
        up = {}
        for parent in {i >> 2 for i in level}:
            up[parent] = pr.poseidon([level.get(4 * parent + j, d) for j in range(4)])
        level, d = up, pr.poseidon([d, d, d, d])
    return level[0]


def compress(model, pairs):
    """pairs: dict {locator tuple: int}.  -> (state_hash, state_size)"""
    for loc in pairs:
        if model_locate(model, loc)[0] != "scalar":
            raise ValueError("NonScalarLocatorError")
    items = [(tuple(k), v % pr.R_MOD) for k, v in pairs.items()]
    size = sum(1 for _, v in items if v)
    if not items:
        return model_default(model), 0
    return _value(model, items, 0), size


def model_bincode(model) -> bytes:
    """bincode 1.3 of `ZkStateModel` (src/zk/mod.rs:332-345): u32 tag, Struct: u64 count + fields, List: u8 log4_size + boxed item"""
    if model[0] == "scalar":
        return (0).to_bytes(4, "little")
    if model[0] == "struct":
        return (1).to_bytes(4, "little") + len(model[1]).to_bytes(8, "little") + b"".join(model_bincode(f) for f in model[1])
    return (2).to_bytes(4, "little") + bytes([model[1]]) + model_bincode(model[2])


def pairs_bincode(pairs) -> bytes:
    """bincode of `ZkDataPairs(HashMap<ZkDataLocator, ZkScalar>)`: u64 count; per entry Vec<u64> + the 4 Montgomery limbs"""
    out = len(pairs).to_bytes(8, "little")
    for loc, v in pairs.items():
        out += len(loc).to_bytes(8, "little") + b"".join(int(x).to_bytes(8, "little") for x in loc) + pr.fr_to_mont_bytes(v % pr.R_MOD)
    return out


def mpn_model(L, T):
    """`MpnConfig::state_model` (src/mpn/mod.rs:218-241)"""
    S = ("scalar",)
    return ("list", L, ("struct", [S, S, S, S, ("list", T, ("struct", [S, S]))]))


# ---------------------------------------------------------------------------------------------------------------------------
# `KvStoreStateManager` over a plain dict (test infrastructure for the persistent device state, bzk_state_*): a restatement of
# /root/reference/src/zk/state/mod.rs - `set_data` :310-420 (one pair at a time: the scalar, then every enclosing list / struct up to
# the root; nodes that equal their default are REMOVED from the store; size_diff bookkeeping on zero <-> non-zero), `get_data`
# :422-438, `prove` :218-264, `root` :274-284, `update_contract` :286-308, and of `ZkState::push_delta` (src/zk/mod.rs:521-530: the
# rollback of a delta = the previous value of every key it names, None where there was none).  Store keys:
#   ("v", locator)            scalar / struct / list value at a locator        (keys::local_value)
#   ("a", locator, heap idx)  inner node of the list at `locator`, heap index (4^k - 1) / 3 + i at depth k   (keys::local_tree_aux)
# ---------------------------------------------------------------------------------------------------------------------------
class PyKvState:
    def __init__(self, model, hasher=None):
        self.model, self.db, self.H, self._dflt = model, {}, hasher or pr.poseidon, {}
        self.hash, self.size, self.height = self._default(model), 0, 0

    def _default(self, model):
        """`compress_default` (memoised per sub-model object: the reference recomputes it, the value is the same)"""
        if model[0] == "scalar":
            return 0
        if id(model) not in self._dflt:
            if model[0] == "struct":
                d = self.H([self._default(f) for f in model[1]])
            else:
                d = self._default(model[2])
                for _ in range(model[1]):
                    d = self.H([d] * 4)
            self._dflt[id(model)] = d
        return self._dflt[id(model)]

    def get_data(self, locator):
        locator = tuple(locator)
        return self.db.get(("v", locator), self._default(model_locate(self.model, locator)))

    def set_data(self, locator, value):
        locator, value = list(locator), value % pr.R_MOD
        if model_locate(self.model, locator)[0] != "scalar":
            raise ValueError("NonScalarLocatorError")
        prev = self.get_data(locator)
        if prev == value:
            return self.get_data(())
        if value == 0:
            self.size -= 1 if prev else 0
            self.db.pop(("v", tuple(locator)), None)
        else:
            self.size += 0 if prev else 1
            self.db[("v", tuple(locator))] = value
        while locator:
            at = locator.pop()
            here = model_locate(self.model, locator)
            if here[0] == "list":
                log4, dflt, cur = here[1], self._default(here[2]), at
                for layer in range(log4 - 1, -1, -1):
                    first = cur - cur % 4
                    kids = []
                    for j in range(first, first + 4):
                        if j == cur:
                            kids.append(value)
                        elif layer == log4 - 1:
                            kids.append(self.get_data(locator + [j]))
                        else:
                            kids.append(self.db.get(("a", tuple(locator), (4 ** (layer + 1) - 1) // 3 + j), dflt))
                    value, dflt, cur = self.H(kids), self.H([dflt] * 4), cur // 4
                    if layer > 0:
                        key = ("a", tuple(locator), (4 ** layer - 1) // 3 + cur)
                        if value == dflt:
                            self.db.pop(key, None)
                        else:
                            self.db[key] = value
            else:
                value = self.H([value if f == at else self.get_data(locator + [f]) for f in range(len(here[1]))])
            if value == self._default(here):
                self.db.pop(("v", tuple(locator)), None)
            else:
                self.db[("v", tuple(locator))] = value
        return value

    def update_contract(self, delta, target_height):
        """delta: {locator: int or None}.  All or nothing, like the reference's mirror / fork.  -> the rollback delta"""
        for loc in delta:
            if model_locate(self.model, loc)[0] != "scalar":
                raise ValueError("NonScalarLocatorError")
        rollback = {tuple(loc): self.db.get(("v", tuple(loc))) for loc in delta}
        for loc, v in delta.items():
            self.hash = self.set_data(loc, v or 0)
        self.height = target_height
        return rollback

    def root(self):
        return self.hash, self.size

    def prove(self, tree_loc, index):
        here = model_locate(self.model, tree_loc)
        if here[0] != "list":
            raise ValueError("NonTreeLocatorError")
        log4, dflt, cur, out = here[1], self._default(here[2]), index, []
        for layer in range(log4 - 1, -1, -1):
            first, part = cur - cur % 4, []
            for j in range(first, first + 4):
                if j != cur:
                    part.append(self.get_data(list(tree_loc) + [j]) if layer == log4 - 1
                                else self.db.get(("a", tuple(tree_loc), (4 ** (layer + 1) - 1) // 3 + j), dflt))
            out.append(part)
            cur, dflt = cur // 4, self.H([dflt] * 4)
        return out


def delta_bincode(delta) -> bytes:
    """bincode of `ZkDeltaPairs(HashMap<ZkDataLocator, Option<ZkScalar>>)`: u64 count; per entry Vec<u64>, u8 Option tag, the limbs"""
    out = len(delta).to_bytes(8, "little")
    for loc, v in delta.items():
        out += len(loc).to_bytes(8, "little") + b"".join(int(x).to_bytes(8, "little") for x in loc)
        out += b"\x00" if v is None else b"\x01" + pr.fr_to_mont_bytes(v % pr.R_MOD)
    return out


def delta_decode(blob: bytes) -> dict:
    """inverse of delta_bincode with the checks a bincode reader of `ZkDeltaPairs` makes (ValueError on anything else): count, Vec<u64>
    locators, the Option tag, 32-byte limbs below r, no trailing bytes; a HashMap holds a key once"""
    def u64(pos):
        if pos + 8 > len(blob):
            raise ValueError("truncated")
        return int.from_bytes(blob[pos:pos + 8], "little"), pos + 8
    n, pos = u64(0)
    if n > len(blob):
        raise ValueError("implausible count")
    out = {}
    for _ in range(n):
        k, pos = u64(pos)
        if k > 64:
            raise ValueError("implausible locator")
        loc = []
        for _ in range(k):
            x, pos = u64(pos)
            loc.append(x)
        if pos >= len(blob) or blob[pos] > 1:
            raise ValueError("bad Option tag")
        tag, pos = blob[pos], pos + 1
        v = None
        if tag:
            if pos + 32 > len(blob):
                raise ValueError("truncated")
            limbs = int.from_bytes(blob[pos:pos + 32], "little")
            if limbs >= pr.R_MOD:
                raise ValueError("not a field element")
            v, pos = pr.fr_from_mont_bytes(blob[pos:pos + 32]), pos + 32
        if tuple(loc) in out:
            raise ValueError("duplicate key")
        out[tuple(loc)] = v
    if pos != len(blob):
        raise ValueError("trailing bytes")
    return out
