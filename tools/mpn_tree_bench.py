"""Timing of the device-resident MPN account state (bzk_mpn_tree_*) at the production depth: bulk load, a block's worth of account
updates, proofs.  usage: python tools/mpn_tree_bench.py"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bazuka_amd import Bzk

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def main():
    L4, T4 = 15, 3
    ctx = Bzk(0)
    rnd = random.Random(1)
    out = {}
    t0 = time.perf_counter(); tree = ctx.mpn_tree_create(L4, T4, 70000); out["create_s (46 GB of default leaf hashes + pool)"] = round(time.perf_counter() - t0, 3)
    idx = rnd.sample(range(4 ** L4), 65536)

    def acct(i, k):
        return (i, [fr(k), fr(0), fr(i + 1), fr(i + 2)], {0: (fr(1), fr(10 ** 9 + k)), (i % 63) + 1: (fr(7), fr(k))})

    batch = [acct(i, 0) for i in idx]
    t0 = time.perf_counter(); ctx.mpn_tree_set_accounts(tree, batch); out["set 65536 new accounts (2 token slots each) s"] = round(time.perf_counter() - t0, 3)
    for n in (256, 4096):
        sub = rnd.sample(idx, n)
        b = [acct(i, 5) for i in sub]
        ctx.mpn_tree_set_accounts(tree, b)
        best = 1e9
        for rep in range(3):
            b = [acct(i, 6 + rep) for i in sub]
            t0 = time.perf_counter(); ctx.mpn_tree_set_accounts(tree, b); best = min(best, time.perf_counter() - t0)
        out[f"update {n} accounts ms"] = round(best * 1e3, 2)
        t0 = time.perf_counter(); ctx.mpn_tree_prove(tree, sub, L4); out[f"prove {n} accounts ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        t0 = time.perf_counter(); ctx.mpn_tree_prove_token(tree, sub, [0] * n, T4); out[f"prove_token {n} ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        t0 = time.perf_counter(); ctx.mpn_tree_get_accounts(tree, sub, T4); out[f"get {n} accounts ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    out["root"] = ctx.mpn_tree_root(tree).hex()[:16]
    ctx.mpn_tree_free(tree)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
