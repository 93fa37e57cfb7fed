"""CPU suite: the N>1 combine path (all-gather of packed partial points + local fold) on gloo, world 2.
The partial points come from the CPU oracle (the product has no CPU MSM); what is under test is the
exchange + fold that bench.py and a multi-GPU prover use."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from bazuka_amd.dist import allgather_fold, window_range
    from oracle import coracle as co
    from util import rand_scalars_bytes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 600
    bases, sc = co.g1_bases(61, 0, n), rand_scalars_bytes(n, 61)
    lo, hi = n * rank // world, n * (rank + 1) // world  # any partition of a linear map folds the same way
    part = co.msm_g1(bases[96 * lo:96 * hi], sc[32 * lo:32 * hi])
    total = allgather_fold(part)
    b2 = co.g2_bases(62, 0, 64)
    part2 = co.msm_g2(b2[192 * (32 * rank):192 * (32 * rank + 32)], sc[32 * 32 * rank:32 * (32 * rank + 32)])
    total2 = allgather_fold(part2)
    ok = total == co.msm_g1(bases, sc) and total2 == co.msm_g2(b2, sc[: 32 * 64])
    # window ranges tile [0, W) exactly
    W = 16
    tiles = [window_range(W, r, world) for r in range(world)]
    ok = ok and tiles[0][0] == 0 and tiles[-1][1] == W and all(tiles[i][1] == tiles[i + 1][0] for i in range(world - 1))
    q.put((rank, ok, total))
    dist.destroy_process_group()


def test_allgather_fold_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]  # every rank holds the same folded point


# ---- the group-building protocol of bench.py --gpus N (bazuka_amd/dist.py::build_device_group) on gloo, world 2, with stub groups ----
class _StubGroup:
    def __init__(self, exchange):
        self.exchange = exchange
        self.closed = False

    def close(self):
        self.closed = True


def _group_worker(rank, world, port, scenario, q):
    sys.path.insert(0, ROOT)
    import time

    import torch.distributed as dist
    from bazuka_amd.dist import build_device_group
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def new_uid():
        box = [os.urandom(16) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def make_group(uid, exchange):
        calls.append(exchange)
        if scenario == "rccl_init_hangs_on_rank1" and exchange != 1 and rank == 1:
            time.sleep(30)                       # what a peer that never arrives looks like from inside ncclCommInitRank
        if scenario == "rccl_init_raises_on_rank0" and exchange != 1 and rank == 0:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        if scenario == "nothing_works" and rank == 1:
            raise RuntimeError("no transport")
        return _StubGroup(exchange)

    probe = 1 if (scenario == "rank1_has_no_rccl" and rank == 1) else 3
    t0 = time.perf_counter()
    try:
        mg, x, notes = build_device_group(make_group, probe, rank, 0, 2.0, vote_group=None, new_uid=new_uid)
        q.put((rank, scenario, "ok", x, calls, round(time.perf_counter() - t0, 1)))
    except RuntimeError as e:
        q.put((rank, scenario, "error", str(e), calls, round(time.perf_counter() - t0, 1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario,want", [("all_fine", ("ok", 0, [0])), ("rank1_has_no_rccl", ("ok", 1, [1])),
                                           ("rccl_init_hangs_on_rank1", ("ok", 1, [0, 1])), ("rccl_init_raises_on_rank0", ("ok", 1, [0, 1])),
                                           ("nothing_works", ("error", None, None))])
def test_device_group_protocol_never_hangs_and_falls_back_together(scenario, want):
    """every rank must reach the same decision: RCCL when all can, the shared-memory transport when ONE rank cannot (capability vote before
    the blocking init, or a bounded wait inside it), an error on every rank when nothing works - and never a hang"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, sc_, status, x, calls, secs in got:
        assert status == want[0], got
        assert secs < 20, got
        if status == "ok":
            assert x == want[1] and calls == want[2], got
