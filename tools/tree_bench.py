"""Device-resident account-tree at the production depth: log4 = 15 (2^30 leaves, 46 GB of nodes in HBM) created empty, then
batches of leaf updates and proofs - the level loop of KvStoreStateManager::set_data / prove on the GPU."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bazuka_amd import Bzk

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def main():
    ctx = Bzk(0)
    out = {}
    for log4 in (12, 15):
        t0 = time.perf_counter(); tree = ctx.tree4_create(log4, None, bytes(32)); t_create = time.perf_counter() - t0
        rnd = random.Random(log4)
        res = {"create_empty_s": round(t_create, 3), "nodes_GB": round(((4 ** (log4 + 1) - 1) // 3) * 32 / 1e9, 2)}
        for n in (16, 256, 4096, 65536):
            idx = [rnd.randrange(4 ** log4) for _ in range(n)]
            vals = b"".join(fr(rnd.randrange(1, 1 << 60)) for _ in range(n))
            ctx.tree4_update(tree, idx, vals)  # warm
            t0 = time.perf_counter(); ctx.tree4_update(tree, idx, vals); tu = time.perf_counter() - t0
            t0 = time.perf_counter(); ctx.tree4_prove(tree, idx, log4); tp = time.perf_counter() - t0
            res[f"n={n}"] = {"update_ms": round(tu * 1e3, 2), "prove_ms": round(tp * 1e3, 2), "hashes": None}
        out[f"log4={log4}"] = res
        ctx.tree4_free(tree)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
