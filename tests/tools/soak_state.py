"""Soak of the persistent device state (bzk_state_*): thousands of deltas on the production MPN model with the incremental root checked
against a one-shot compress of everything written so far, handles created and freed in a loop, device memory watched throughout.
usage: python tests/tools/soak_state.py [updates=3000]"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bazuka_amd import Bzk, DeviceState
import pystate as ps
from oracle import pyref as pr

F = pr.fr_to_mont_bytes


def main(updates=3000):
    torch.cuda.init()
    ctx = Bzk(0)
    rnd = random.Random(2024)
    model = ps.model_bincode(ps.mpn_model(15, 3))
    free0 = torch.cuda.mem_get_info()[0]
    dev = DeviceState(ctx, model)
    accts = rnd.sample(range(4 ** 15), 5000)
    live, out, t0 = {}, {"checks": 0}, time.perf_counter()
    mem = []
    for u in range(updates):
        delta = {}
        for a in rnd.sample(accts, rnd.choice((1, 2, 8, 64))):
            delta[(a, rnd.randrange(4))] = rnd.randrange(1 << 62)
            delta[(a, 4, rnd.randrange(4), rnd.randrange(2))] = rnd.choice((0, rnd.randrange(1, 1 << 40)))
        pairs = [(k, F(v)) for k, v in delta.items()]
        h, n = dev.update(pairs, u + 1)
        for k, v in delta.items():
            live[k] = v
        if u % 500 == 499 or u == updates - 1:
            want = ctx.state_compress(model, [(k, F(v)) for k, v in live.items()])
            assert (h, n) == want, ("incremental state differs from the one-shot compress", u)
            out["checks"] += 1
            mem.append(torch.cuda.mem_get_info()[0])
    out["updates"], out["seconds"] = updates, round(time.perf_counter() - t0, 1)
    out["stats"] = dev.stats()
    dev.close()
    for _ in range(300):     # handles come and go
        d = DeviceState(ctx, model)
        d.update([((1, 0), F(5))], 1)
        d.close()
    ctx.sync()
    free1 = torch.cuda.mem_get_info()[0]
    out["free_bytes_before_after"] = [free0, free1]
    out["free_bytes_at_checks"] = mem
    # the workspace of the context may have grown once; nothing else may stay behind
    assert free0 - free1 < 512 << 20, (free0, free1)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
