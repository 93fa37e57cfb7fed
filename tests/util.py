"""Shared helpers for the parity tests (seeded inputs in the C-ABI byte formats)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pyref as pr  # noqa: E402

SEED = pr.SEED


def fr_list(n, seed=SEED):
    rng = pr.SplitMix64(seed)
    return [rng.fr() for _ in range(n)]


def fr_bytes(vals, mont=True):
    f = pr.fr_to_mont_bytes if mont else pr.fr_to_canon_bytes
    return b"".join(f(v) for v in vals)


def rand_scalars_bytes(n, seed=SEED):
    """n uniform Montgomery scalars, fast (numpy): random 32-byte strings are valid Montgomery limbs iff
    < r as integers; clear the top bit and patch the rare overflow."""
    import numpy as np
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    a = rs.randint(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x3F  # < 2^254 < r  -> always a valid residue
    return a.tobytes()


def to_dev(b: bytes):
    import torch
    t = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    return t


def dev_bytes(t) -> bytes:
    return bytes(t.cpu().numpy().tobytes())


def synth_r1cs(n_mul: int, n_in: int = 3, seed: int = SEED, fan: int = 3):
    """Random satisfiable R1CS in the bellman layout: inputs (ONE, x1..), aux; every constraint k is
    <A_k,z> * <B_k,z> = new aux variable, plus a few linear ones; the n_in trailing `input_i * 0 = 0`
    rows are appended as bellman does.  Returns dict(n_in, n_aux, rows=[(A,B,C)], z=[ints])."""
    import random
    rnd = random.Random(seed)
    z = [1] + [rnd.randrange(pr.R_MOD) for _ in range(n_in - 1)]
    rows = []
    for k in range(n_mul):
        nv = len(z)
        A = [(rnd.randrange(nv), rnd.randrange(1, pr.R_MOD) if rnd.random() < 0.5 else rnd.choice([1, 2, pr.R_MOD - 1]))
             for _ in range(rnd.randint(1, fan))]
        B = [(rnd.randrange(nv), rnd.randrange(1, pr.R_MOD) if rnd.random() < 0.5 else 1) for _ in range(rnd.randint(1, fan))]
        if k % 7 == 3:  # boolean-style constraint: b * (1 - b) = 0 with b a fresh 0/1 aux variable
            b = rnd.randint(0, 1)
            z.append(b)
            v = len(z) - 1
            rows.append(([(v, 1)], [(0, 1), (v, pr.R_MOD - 1)], []))
            continue
        av = sum(c * z[v] for v, c in A) % pr.R_MOD
        bv = sum(c * z[v] for v, c in B) % pr.R_MOD
        z.append(av * bv % pr.R_MOD)
        rows.append((A, B, [(len(z) - 1, 1)]))
    for i in range(n_in):
        rows.append(([(i, 1)], [], []))
    return {"n_in": n_in, "n_aux": len(z) - n_in, "rows": rows, "z": z}


def r1cs_to_csr(co, r1):
    out = []
    for which in range(3):
        rp, col, val = [0], [], []
        for row in r1["rows"]:
            for v, c in row[which]:
                col.append(v)
                val.append(pr.fr_to_mont_bytes(c))
            rp.append(len(col))
        out.append(co.CsrHolder(len(r1["rows"]), rp, col, b"".join(val)))
    return out


def log2_ceil(n):
    lg = 0
    while (1 << lg) < n:
        lg += 1
    return lg
