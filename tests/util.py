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
