"""GPU debug helper: runs one G1 MSM case in-process and prints per-kernel event times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bazuka_amd import Bzk
from oracle import coracle as co
from util import rand_scalars_bytes, to_dev, dev_bytes

n = int(sys.argv[1])
check = len(sys.argv) > 2 and sys.argv[2] == "check"
ctx = Bzk(0)
t = time.time()
bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
ctx.g1_synth_bases_dev(1, 0, n, bases); ctx.sync()
print(f"n={n} synth {time.time()-t:.3f}s", flush=True)
sc = to_dev(rand_scalars_bytes(n, 5))
ctx.prof_enable(True)
for it in range(2):
    ctx.prof_reset()
    t = time.time()
    out = ctx.msm_g1_dev(bases, sc, n)
    dt = time.time() - t
    print(f"iter {it}: {dt*1e3:.3f} ms  W={ctx.msm_window_count(n)}", {k: round(v[1], 4) for k, v in ctx.prof_dump().items()}, flush=True)
if check:
    t = time.time()
    want = co.msm_g1(dev_bytes(bases), dev_bytes(sc), nthreads=min(32, co.ncpu()))
    print("oracle", time.time() - t, "match", want == out, flush=True)
