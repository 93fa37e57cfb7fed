"""Multi-GPU combine of window-sharded MSM partial sums.

RCCL's reduce operators are arithmetic on numeric dtypes and cannot add elliptic-curve points, so the
"all-reduce of partial bucket sums" of the north star is realised as an all-gather of the packed
97-byte (G1) / 193-byte (G2) partial points as raw uint8 over xGMI, followed by a local fold with the
host-side bzk_g1_sum / bzk_g2_sum.  The message is <= 200 B per rank: latency-bound, one collective
per MSM.  Works on any torch.distributed backend (nccl = RCCL on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes as C

from .lib import load_library

_SLOT = {97: 104, 193: 200}  # padded to a multiple of 8 bytes


def window_range(n_windows: int, rank: int, world: int):
    """Contiguous window range owned by `rank` (balanced to within one window)."""
    return n_windows * rank // world, n_windows * (rank + 1) // world


def allgather_fold(part: bytes, device=None, group=None) -> bytes:
    """part: this rank's packed partial point (97 B G1 or 193 B G2).  Returns the folded total (same
    on every rank)."""
    import torch
    import torch.distributed as dist
    size = len(part)
    slot = _SLOT[size]
    world = dist.get_world_size(group)
    src = torch.zeros(slot, dtype=torch.uint8)
    src[:size] = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    if device is not None:
        src = src.to(device)
    dst = torch.empty(slot * world, dtype=torch.uint8, device=src.device)
    dist.all_gather_into_tensor(dst, src, group=group)
    rows = dst.cpu().numpy().reshape(world, slot)[:, :size].tobytes()
    lib = load_library()
    out = C.create_string_buffer(size)
    fn = lib.bzk_g1_sum if size == 97 else lib.bzk_g2_sum
    st = fn(rows, world, out)
    if st != 0:
        raise RuntimeError(f"bzk_g?_sum failed: {st}")
    return out.raw
