"""Differential fuzzing of the C ABI against the CPU oracle on random shapes (run on a GPU box):
MSM G1/G2 (plain, de-duplicated, window partitions, witness-like / degenerate scalar mixes, repeated and negated bases),
NTT (all four modes, every log size up to 14), Poseidon batches (every arity), 4-ary trees, tree updates, the general state seam one-shot and
device-resident (random models under random deltas), stand-alone calls as window ranges in flight over giant-bucket scalar mixes (round 6).
usage: python tests/tools/fuzz_gpu.py [seconds=60] [seed=1]"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bazuka_amd import Bzk
from oracle import coracle as co, pyref as pr
from util import fr_bytes, fr_list, rand_scalars_bytes


def scalars(rnd, n):
    mode = rnd.randrange(5)
    if mode == 0:
        return rand_scalars_bytes(n, rnd.randrange(1 << 30))
    base = fr_list(max(1, n // rnd.choice((1, 2, 3, 50))), rnd.randrange(1 << 30))
    sc = [base[rnd.randrange(len(base))] for _ in range(n)]
    if mode >= 2:
        for i in range(0, n, rnd.choice((2, 5, 9))):
            sc[i] = rnd.choice((0, 1, 2, pr.R_MOD - 1, 1 << 15, (1 << 16) - 1, 1 << 254))
    if mode == 4:
        sc = [sc[0]] * n
    return fr_bytes(sc)


def neg_y(raw, size):
    half = size // 2
    if size == 96:
        return raw[:48] + pr.fp_to_mont_bytes(-pr.fp_from_mont_bytes(raw[48:96]))
    y0, y1 = pr.fp_from_mont_bytes(raw[96:144]), pr.fp_from_mont_bytes(raw[144:192])
    return raw[:96] + pr.fp_to_mont_bytes(-y0) + pr.fp_to_mont_bytes(-y1)


def main(seconds=60, seed=1):
    import torch
    torch.cuda.init()  # torch bundles its own HIP runtime: let it initialise before libbzk's (system ROCm) does
    rnd = random.Random(seed)
    ctx = Bzk(0)
    nt = co.ncpu()
    counts = {}
    t_end = time.time() + seconds
    while time.time() < t_end:
        kinds = ("msm_g1", "msm_g1", "msm_g2", "ntt", "poseidon", "tree", "windows", "table", "window_size",
                 "bases", "mg", "h_chain", "state", "state_dev", "ranges")
        if os.environ.get("FUZZ_KINDS"):  # e.g. FUZZ_KINDS=ranges,msm_g2 for a soak of chosen paths
            kinds = tuple(k for k in kinds if k in os.environ["FUZZ_KINDS"].split(",")) or kinds
        kind = rnd.choice(kinds)
        counts[kind] = counts.get(kind, 0) + 1
        if kind == "ranges":
            # round 6: one stand-alone call as 1 - 4 window ranges in flight (msm_run_split; a context of its own created under the knobs), resident set and
            # raw bases, over scalar mixes that put GIANT buckets (more than 640 tasks: the two-level fold, msm_fold_wide_kernel and its G2 pair twin) into some
            # or all windows
            import torch
            g2 = rnd.random() < 0.3
            n = rnd.choice((3000, 9000, 22000, 30000, 47000)) if not g2 else rnd.choice((3000, 22000, 30000))
            bases = (co.g2_bases if g2 else co.g1_bases)(rnd.randrange(1 << 30), 0, n, nthreads=nt)
            mode = rnd.randrange(4)
            if mode == 0:
                sc = scalars(rnd, n)
            else:
                vals = fr_list(rnd.choice((1, 2, 3, 7)), rnd.randrange(1 << 30))
                scl = fr_list(n, rnd.randrange(1 << 30)) if mode == 3 else [None] * n
                for i in range(n):
                    if scl[i] is None or i % 2 == 0:
                        scl[i] = vals[i % len(vals)]
                sc = fr_bytes(scl)
            want = (co.msm_g2 if g2 else co.msm_g1)(bases, sc, nthreads=nt)
            env = {"BZK_MSM_SPLIT": str(rnd.randrange(1, 5)), "BZK_MSM_SPLIT_PRIO": str(rnd.randrange(3)), "BZK_MSM_SPLIT_MIN_LOG": "10"}
            if rnd.random() < 0.2:
                env["BZK_MSM_SPLIT_CUTS"] = rnd.choice(("2,3", "5,1", "1,1,1", "4,4,4,4"))  # used when they add up to the window count, else equal ranges
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                c2 = Bzk(0)
            finally:
                for k, v in old.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            try:
                db = torch.frombuffer(bytearray(bases), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
                torch.cuda.synchronize()
                hb = c2.msm_bases_load_dev(db, n, g2=g2)
                for _ in range(2):
                    assert c2.msm_bases_run_dev(hb, ds, n, g2=g2) == want, (kind, "resident", n, g2, mode, env)
                assert (c2.msm_g2_dev if g2 else c2.msm_g1_dev)(db, ds, n) == want, (kind, "raw", n, g2, mode, env)
                m = rnd.randrange(1, n + 1)
                assert c2.msm_bases_run_dev(hb, ds, m, g2=g2) == (co.msm_g2 if g2 else co.msm_g1)(bases[:m * (192 if g2 else 96)], sc[:32 * m], nthreads=nt), \
                    (kind, "prefix", n, m, g2, mode, env)
                c2.msm_bases_free(hb)
            finally:
                c2.close()
            continue
        if kind in ("bases", "mg"):  # round 3: resident base sets, device groups (several contexts on this one GPU)
            import torch
            from bazuka_amd import Mg
            g2 = rnd.random() < 0.25
            n = rnd.choice((1, 3, 300, 5000, 20000)) if not g2 else rnd.choice((1, 300, 5000))
            bases = (co.g2_bases if g2 else co.g1_bases)(rnd.randrange(1 << 30), 0, n, nthreads=nt)
            sc = scalars(rnd, n)
            want = (co.msm_g2 if g2 else co.msm_g1)(bases, sc, nthreads=nt)
            dd = rnd.random() < 0.5
            if kind == "bases":
                db = torch.frombuffer(bytearray(bases), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
                torch.cuda.synchronize()
                hb = ctx.msm_bases_load_dev(db, n, g2=g2)
                assert ctx.msm_bases_run_dev(hb, ds, n, g2=g2, dedup=dd) == want, (kind, n, g2, dd)
                # round 4: the endomorphism form (whole-MSM calls with the throughput hint over a set that carries its images)
                assert ctx.msm_bases_run_dev(hb, ds, n, g2=g2, dedup=dd, throughput=True) == want, (kind, "endo", n, g2, dd)
                m = rnd.randrange(1, n + 1)   # a prefix of the set
                assert ctx.msm_bases_run_dev(hb, ds, m, g2=g2, dedup=rnd.random() < 0.5, throughput=True) == \
                    (co.msm_g2 if g2 else co.msm_g1)(bases[:m * (192 if g2 else 96)], sc[:32 * m], nthreads=nt), (kind, "endo prefix", n, m, g2)
                W = ctx.msm_window_count(n)
                cut = sorted({0, W, rnd.randrange(W + 1)})
                parts = b"".join(ctx.msm_bases_windows_dev(hb, ds, n, a, b, g2=g2) for a, b in zip(cut, cut[1:]))
                assert (ctx.g2_sum if g2 else ctx.g1_sum)(parts) == want, (kind, n, cut)
                ctx.msm_bases_free(hb)
            else:
                mg = Mg(devices=[0] * rnd.randrange(1, 5), exchange=rnd.choice((1, 2)))
                hb = mg.bases_load(bases, n, g2=g2)
                assert mg.msm(hb, sc, n, g2=g2, dedup=dd) == want, (kind, n, g2, dd, mg.world, mg.exchange)
                mg.bases_free(hb)
                mg.close()
            continue
        if kind == "h_chain":  # the fused h chain against the oracle's seven transforms + pointwise
            import torch
            lg = rnd.randrange(1, 15)
            m = 1 << lg
            rows = rnd.randrange(1, m + 1)
            az, bz, cz = (rand_scalars_bytes(rows, rnd.randrange(1 << 30)) for _ in range(3))
            pad = bytes(32 * (m - rows))
            d = [torch.frombuffer(bytearray(x + pad), dtype=torch.uint8).cuda() for x in (az, bz, cz)]
            torch.cuda.synchronize()
            ctx.groth16_h_dev(d[0], d[1], d[2], lg)
            torch.cuda.synchronize()
            assert bytes(d[0].cpu().numpy().tobytes())[: 32 * (m - 1)] == co.groth16_h(az, bz, cz, lg, nthreads=nt), (kind, lg, rows)
            continue
        if kind == "state_dev":  # round 4: the persistent device state against the pair-at-a-time restatement of the reference's state manager
            import pystate as ps
            from bazuka_amd import DeviceState
            def rmodel(depth):
                k = rnd.random()
                if depth == 0 or k < 0.3:
                    return ("scalar",)
                if k < 0.65:
                    return ("struct", [rmodel(depth - 1) for _ in range(rnd.randint(1, 4))])
                return ("list", rnd.randint(0, 3), rmodel(depth - 1))
            def rloc(model, stop=0.0, span=5):
                loc = []
                while model[0] != "scalar" and rnd.random() >= stop:
                    if model[0] == "struct":
                        f = rnd.randrange(len(model[1])); loc.append(f); model = model[1][f]
                    else:
                        loc.append(rnd.randrange(min(span, 4 ** model[1]))); model = model[2]
                return tuple(loc), model
            model = rmodel(3)
            dev, ref = DeviceState(ctx, ps.model_bincode(model)), ps.PyKvState(model)
            for height in range(1, rnd.randrange(2, 6)):
                delta = {rloc(model)[0]: rnd.choice((None, 0, 1, rnd.randrange(pr.R_MOD))) for _ in range(rnd.choice((1, 2, 6, 25)))}
                rb = ref.update_contract(delta, height)
                ps_ = [(k, pr.fr_to_mont_bytes((v or 0) % pr.R_MOD)) for k, v in delta.items()]
                h, n, prev = dev.update(ps_, height, want_rollback=True)
                assert (h, n) == (pr.fr_to_mont_bytes(ref.hash), ref.size), (kind, model, delta)
                assert prev == [pr.fr_to_mont_bytes(rb[k] or 0) for k, _ in ps_], (kind, "rollback", model, delta)
                locs = [rloc(model, stop=0.3)[0] for _ in range(6)]
                assert dev.get(locs) == [pr.fr_to_mont_bytes(ref.get_data(l)) for l in locs], (kind, "get", model, locs)
                tl, sub = rloc(model, stop=0.4)
                if sub[0] == "list":
                    ix = [rnd.randrange(4 ** sub[1]) for _ in range(2)]
                    assert dev.prove(tl, ix) == [[[pr.fr_to_mont_bytes(x) for x in part] for part in ref.prove(tl, i)] for i in ix], (kind, "prove", model, tl, ix)
            dev.close()
            continue
        if kind == "state":  # general ZkStateModel::compress against the general Python restatement (small cases: pure-Python Poseidon)
            import pystate as ps
            def rmodel(depth):
                k = rnd.random()
                if depth == 0 or k < 0.3:
                    return ("scalar",)
                if k < 0.65:
                    return ("struct", [rmodel(depth - 1) for _ in range(rnd.randint(1, 4))])
                return ("list", rnd.randint(0, 3), rmodel(depth - 1))
            def rloc(model):
                loc = []
                while model[0] != "scalar":
                    if model[0] == "struct":
                        f = rnd.randrange(len(model[1])); loc.append(f); model = model[1][f]
                    else:
                        loc.append(rnd.randrange(4 ** model[1])); model = model[2]
                return tuple(loc)
            model = rmodel(3)
            pairs = {rloc(model): rnd.choice((0, 1, rnd.randrange(pr.R_MOD))) for _ in range(rnd.choice((0, 1, 4, 30)))}
            got = ctx.state_compress(ps.model_bincode(model), [(k, pr.fr_to_mont_bytes(v)) for k, v in pairs.items()])
            wh, wn = ps.compress(model, pairs)
            assert got == (pr.fr_to_mont_bytes(wh), wn), (kind, model, pairs)
            continue
        if kind in ("table", "window_size"):
            import torch
            n = rnd.choice((1, 5, 300, 2000, 6000))
            bases = co.g1_bases(rnd.randrange(1 << 30), 0, n, nthreads=nt)
            sc = scalars(rnd, n)
            want = co.msm_g1(bases, sc, nthreads=nt)
            db = torch.frombuffer(bytearray(bases), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
            if kind == "table":  # static-base tables: full (levels 0) and folded (2..6 levels), bucket-set / level shards
                tab = ctx.msm_table_build(db, n, levels=rnd.choice((0, 0, 2, 3, 4, 5, 6)))
                m = rnd.choice((n, max(1, n // 2)))
                assert ctx.msm_table_run_dev(tab, ds, m) == co.msm_g1(bases[:96 * m], sc[:32 * m], nthreads=nt), (kind, n, m)
                S = ctx.msm_table_window_count(tab)
                cut = sorted({0, S, rnd.randrange(S + 1)})
                parts = b"".join(ctx.msm_table_windows_dev(tab, ds, n, a, b) for a, b in zip(cut, cut[1:]))
                assert ctx.g1_sum(parts) == want, (kind, n, cut)
                ctx.msm_table_free(tab)
            else:  # every window size through a context created under BZK_MSM_C
                c = rnd.randrange(4, 19)
                os.environ["BZK_MSM_C"] = str(c)
                try:
                    cx = Bzk(0)
                finally:
                    del os.environ["BZK_MSM_C"]
                assert cx.msm_g1_dev(db, ds, n) == want, (kind, n, c)
                assert cx.msm_g1(bases, sc, dedup=True) == want, (kind, n, c, "dedup")
                cx.close()
            continue
        if kind in ("msm_g1", "msm_g2", "windows"):
            g2 = kind == "msm_g2"
            size = 192 if g2 else 96
            n = rnd.choice((1, 2, 3, 7, 64, 300, 1000, 4096, 5000, 20000)) if not g2 else rnd.choice((1, 5, 300, 4096, 6000))
            bases = bytearray((co.g2_bases if g2 else co.g1_bases)(rnd.randrange(1 << 30), 0, n, nthreads=nt))
            if n >= 4 and rnd.random() < 0.5:  # repeated point, point and its negative
                bases[size:2 * size] = bases[0:size]
                bases[3 * size:4 * size] = neg_y(bytes(bases[2 * size:3 * size]), size)
            bases = bytes(bases)
            sc = scalars(rnd, n)
            want = (co.msm_g2 if g2 else co.msm_g1)(bases, sc, nthreads=nt)
            f = ctx.msm_g2 if g2 else ctx.msm_g1
            assert f(bases, sc) == want, (kind, n, "plain")
            assert f(bases, sc, dedup=True) == want, (kind, n, "dedup")
            if kind == "windows" and n >= 2:
                import torch
                db = torch.frombuffer(bytearray(bases), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
                W = ctx.msm_window_count(n)
                cut = sorted({0, W, *[rnd.randrange(W + 1) for _ in range(rnd.randrange(1, 4))]})
                parts = b"".join(ctx.msm_g1_windows_dev(db, ds, n, a, b) for a, b in zip(cut, cut[1:]))
                assert ctx.g1_sum(parts) == want, (kind, n, cut)
        elif kind == "ntt":
            lg = rnd.randrange(0, 15)
            data = rand_scalars_bytes(1 << lg, rnd.randrange(1 << 30))
            inv, cs = rnd.random() < 0.5, rnd.random() < 0.5
            assert ctx.ntt(data, lg, inv, cs) == co.ntt(data, lg, inv, cs, nthreads=nt), (lg, inv, cs)
        elif kind == "poseidon":
            ar = rnd.randrange(1, 17)
            n = rnd.choice((1, 2, 63, 64, 65, 1000))
            inp = rand_scalars_bytes(n * ar, rnd.randrange(1 << 30))
            assert ctx.poseidon_batch(inp, ar) == co.poseidon_batch(inp, ar, nthreads=nt), (ar, n)
        else:
            log4 = rnd.randrange(1, 6)
            leaves = rand_scalars_bytes(4 ** log4, rnd.randrange(1 << 30))
            root = co.merkle4_root(leaves, log4, nthreads=nt)
            assert ctx.merkle4_root(leaves, log4) == root
            import torch
            tree = ctx.tree4_create(log4, torch.frombuffer(bytearray(leaves), dtype=torch.uint8).cuda())
            idx = [rnd.randrange(4 ** log4) for _ in range(rnd.randrange(1, 40))]
            vals = rand_scalars_bytes(len(idx), rnd.randrange(1 << 30))
            ctx.tree4_update(tree, idx, vals)
            host = bytearray(leaves)
            for k, i in enumerate(idx):
                host[32 * i:32 * i + 32] = vals[32 * k:32 * k + 32]
            assert ctx.tree4_root(tree) == co.merkle4_root(bytes(host), log4, nthreads=nt)
            ctx.tree4_free(tree)
    print(json.dumps({"seconds": seconds, "seed": seed, "cases": counts, "total": sum(counts.values()), "mismatches": 0}))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
