"""How much do INDEPENDENT G1 MSMs overlap on one GPU?  K host threads, each with a context (stream + workspace) of its own, each running
`per` MSMs of 2^log_n points over ONE shared resident base set (the prover's situation: the CRS is static).  Prints one JSON line per K with the
aggregate rate - the GPU-side answer to VERDICT r5 item 2 (`two_msms_in_flight` bought 1.8 %: why, and what changes it).
usage: python tools/overlap_probe.py [ks=1,2,3,4] [per=16] [log_n=20] [flags=0]     (flags: 4 = BZK_F_THROUGHPUT forms)
env (read by libbzk, see msm_impl.cuh): BZK_MSM_CU_SPLIT, BZK_MSM_SORT_WG, GPU_MAX_HW_QUEUES (HIP runtime)"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bazuka_amd import Bzk  # noqa: E402

SEED = 0x42415A554B41


def main(ks=(1, 2, 3, 4), per=16, log_n=20, flags=0):
    import bench
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n = 1 << log_n
    ctx0 = Bzk(0)
    bases = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    ctx0.g1_synth_bases_dev(SEED, 0, n, bases)
    scalars = [bench.uniform_fr_dev(n, SEED + i, dev) for i in range(max(ks))]
    torch.cuda.synchronize()
    rb = ctx0.msm_bases_load_dev(bases, n)
    ctxs = [Bzk(0) for _ in range(max(ks))]
    want = [ctx0.msm_bases_run_dev(rb, scalars[i], n, throughput=bool(flags & 4)) for i in range(max(ks))]
    for i, c in enumerate(ctxs):
        for _ in range(2):
            assert c.msm_bases_run_dev(rb, scalars[i], n, throughput=bool(flags & 4)) == want[i]
    env = {k: os.environ[k] for k in ("BZK_MSM_CU_SPLIT", "BZK_MSM_SORT_WG", "GPU_MAX_HW_QUEUES", "BZK_MSM_TAIL_PRIO") if k in os.environ}
    base = None
    for k in ks:
        outs = [None] * k

        def run(i):
            for _ in range(per):
                outs[i] = ctxs[i].msm_bases_run_dev(rb, scalars[i], n, throughput=bool(flags & 4))

        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=run, args=(i,)) for i in range(k)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert all(outs[i] == want[i] for i in range(k)), "overlapped MSMs differ from the one-at-a-time results"
        ms = best * 1e3 / (k * per)
        base = base or ms
        print(json.dumps({"in_flight": k, "msms": k * per, "ms_per_msm": round(ms, 4), "Mpt_per_s": round(n / ms / 1e3, 2), "vs_one": round(base / ms, 4),
                          "log_n": log_n, "flags": flags, "env": env}), flush=True)
    for c in ctxs:
        c.close()
    ctx0.msm_bases_free(rb)
    ctx0.close()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(tuple(int(x) for x in a[0].split(",")) if a else (1, 2, 3, 4), int(a[1]) if len(a) > 1 else 16, int(a[2]) if len(a) > 2 else 20,
         int(a[3]) if len(a) > 3 else 0)
