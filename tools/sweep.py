"""GPU measurement helper: MSM parameter sweep (window bits c, reduce chunk) and timings of the other
kernels (Poseidon tree 2^24, NTT 2^20, G2 MSM).  Each configuration runs in a fresh ctx (env-driven knobs)."""
import os, sys, time, subprocess, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def child(mode, log_n):
    import torch
    from bazuka_amd import Bzk
    ctx = Bzk(0)
    n = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(7)
    def rand_fr(cnt):
        t = torch.randint(0, 256, (cnt, 32), dtype=torch.uint8, device="cuda", generator=g); t[:, 31] &= 0x3F
        return t.contiguous()
    def timeit(fn, reps=5):
        fn(); ctx.sync(); best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); ctx.sync(); best = min(best, time.perf_counter() - t)
        return best * 1e3
    if mode == "g1":
        bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda"); ctx.g1_synth_bases_dev(1, 0, n, bases); sc = rand_fr(n)
        ms = timeit(lambda: ctx.msm_g1_dev(bases, sc, n))
        ctx.prof_enable(True); ctx.prof_reset(); ctx.msm_g1_dev(bases, sc, n); prof = {k: round(v[1], 3) for k, v in ctx.prof_dump().items() if v[1] > 0.05}
        print(json.dumps({"mode": mode, "log_n": log_n, "c": os.environ.get("BZK_MSM_C"), "chunk": os.environ.get("BZK_MSM_CHUNK"), "ms": round(ms, 3), "Mpt/s": round(n / ms / 1e3, 1), "prof": prof}))
    elif mode == "g1win":  # what one rank of an 8-GPU window-sharded MSM does: 2 windows over 2^log_n points
        bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda"); ctx.g1_synth_bases_dev(1, 0, n, bases); sc = rand_fr(n)
        W = ctx.msm_window_count(n); w1 = max(1, W // int(os.environ.get("SHARDS", "8")))
        ms = timeit(lambda: ctx.msm_g1_windows_dev(bases, sc, n, 0, w1))
        ctx.prof_enable(True); ctx.prof_reset(); ctx.msm_g1_windows_dev(bases, sc, n, 0, w1); prof = {k: round(v[1], 3) for k, v in ctx.prof_dump().items() if v[1] > 0.05}
        print(json.dumps({"mode": mode, "log_n": log_n, "windows": [0, w1], "of": W, "ms": round(ms, 3), "prof": prof}))
    elif mode == "g1winres":  # the same on a RESIDENT base set (what bzk_mg_msm_g1_dev runs per rank since round 3): conversion is load-time work
        bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda"); ctx.g1_synth_bases_dev(1, 0, n, bases); sc = rand_fr(n)
        hb = ctx.msm_bases_load_dev(bases, n)
        W = ctx.msm_window_count(n); w1 = max(1, W // int(os.environ.get("SHARDS", "8")))
        ms = timeit(lambda: ctx.msm_bases_windows_dev(hb, sc, n, 0, w1))
        ctx.prof_enable(True); ctx.prof_reset(); ctx.msm_bases_windows_dev(hb, sc, n, 0, w1); prof = {k: round(v[1], 3) for k, v in ctx.prof_dump().items() if v[1] > 0.05}
        print(json.dumps({"mode": mode, "log_n": log_n, "windows": [0, w1], "of": W, "ms": round(ms, 3), "prof": prof}))
        ctx.msm_bases_free(hb)
    elif mode in ("g1res", "g2res"):  # whole MSM over a RESIDENT set (endomorphism form where the set carries its images; env BZK_MSM_ENDO_G1 / _G2)
        g2 = mode == "g2res"
        bases = torch.empty(n * (192 if g2 else 96), dtype=torch.uint8, device="cuda")
        (ctx.g2_synth_bases_dev if g2 else ctx.g1_synth_bases_dev)(1, 0, n, bases); sc = rand_fr(n)
        hb = ctx.msm_bases_load_dev(bases, n, g2=g2)
        thr = os.environ.get("THROUGHPUT", "0") == "1"
        ms = timeit(lambda: ctx.msm_bases_run_dev(hb, sc, n, g2=g2, throughput=thr), reps=int(os.environ.get("SWEEP_REPS", "3")))
        k_mean = int(os.environ.get("SWEEP_MEAN_OVER", "0"))  # back-to-back calls, as bench.py times its steps
        mean_ms = None
        if k_mean:
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(k_mean):
                ctx.msm_bases_run_dev(hb, sc, n, g2=g2, throughput=thr)
            torch.cuda.synchronize(); mean_ms = round((time.perf_counter() - t) * 1e3 / k_mean, 4)
        res = ctx.msm_bases_run_dev(hb, sc, n, g2=g2, throughput=thr)
        same = res == (ctx.msm_g2_dev if g2 else ctx.msm_g1_dev)(bases, sc, n)
        ctx.prof_enable(True); ctx.prof_reset(); ctx.msm_bases_run_dev(hb, sc, n, g2=g2, throughput=thr); prof = {k: round(v[1], 3) for k, v in ctx.prof_dump().items() if v[1] > 0.04}
        print(json.dumps({"mode": mode, "log_n": log_n, "endo_g1": os.environ.get("BZK_MSM_ENDO_G1"), "endo_g2": os.environ.get("BZK_MSM_ENDO_G2"), "throughput": thr,
                          "lib": os.path.basename(os.environ.get("BZK_LIBBZK", "")), "seg": os.environ.get("BZK_MSM_SEG"), "rc_leaf": os.environ.get("BZK_MSM_RC_LEAF"), "no_wide": os.environ.get("BZK_MSM_NO_WIDE_FOLD"), "split": os.environ.get("BZK_MSM_SPLIT"), "cuts": os.environ.get("BZK_MSM_SPLIT_CUTS"), "split_prio": os.environ.get("BZK_MSM_SPLIT_PRIO"), "mean_ms": mean_ms,
                          "ms": round(ms, 3), "Mpt/s": round(n / ms / 1e3, 2), "same_as_raw": same, "digest": hashlib.sha256(bytes(res)).hexdigest()[:16], "prof": prof}))
        ctx.msm_bases_free(hb)
    elif mode == "g1tab":
        bases = torch.empty(n * 96, dtype=torch.uint8, device="cuda"); ctx.g1_synth_bases_dev(1, 0, n, bases); sc = rand_fr(n)
        lv = int(os.environ.get("TAB_LEVELS", "0"))
        t0 = time.perf_counter(); tab = ctx.msm_table_build(bases, n, levels=lv); tb = time.perf_counter() - t0
        ms = timeit(lambda: ctx.msm_table_run_dev(tab, sc, n))
        same = ctx.msm_table_run_dev(tab, sc, n) == ctx.msm_g1_dev(bases, sc, n)
        ctx.prof_enable(True); ctx.prof_reset(); ctx.msm_table_run_dev(tab, sc, n); prof = {k: round(v[1], 3) for k, v in ctx.prof_dump().items() if v[1] > 0.05}
        print(json.dumps({"mode": mode, "log_n": log_n, "table_c": os.environ.get("BZK_MSM_TABLE_C"), "levels": ctx.msm_table_levels(tab), "build_s": round(tb, 3), "ms": round(ms, 3), "Mpt/s": round(n / ms / 1e3, 1), "same_as_plain": same, "prof": prof}))
    elif mode == "g2":
        bases = torch.empty(n * 192, dtype=torch.uint8, device="cuda"); ctx.g2_synth_bases_dev(1, 0, n, bases); sc = rand_fr(n)
        ms = timeit(lambda: ctx.msm_g2_dev(bases, sc, n), reps=2)
        ctx.prof_enable(True); ctx.prof_reset(); ctx.msm_g2_dev(bases, sc, n); prof = {k: round(v[1], 3) for k, v in ctx.prof_dump().items() if v[1] > 0.05}
        print(json.dumps({"mode": mode, "log_n": log_n, "occ_alt": os.environ.get("BZK_MSM_OCC_ALT"), "ms": round(ms, 3), "Mpt/s": round(n / ms / 1e3, 2), "prof": prof}))
    elif mode == "tree":
        leaves = rand_fr(n)
        ms = timeit(lambda: ctx.merkle4_root_dev(leaves, log_n // 2), reps=3)
        hashes = (n - 1) // 3
        print(json.dumps({"mode": mode, "leaves": n, "digest": hashlib.sha256(bytes(ctx.merkle4_root_dev(leaves, log_n // 2))).hexdigest()[:16], "ms": round(ms, 3), "Mhash/s": round(hashes / ms / 1e3, 2), "alg_GB/s": round((32 * n + 32 * hashes) / ms / 1e6, 2)}))
    elif mode == "hash":
        ar = int(os.environ.get("ARITY", "4")); inp = rand_fr(n * ar); out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        ms = timeit(lambda: ctx.poseidon_batch_dev(inp, ar, n, out), reps=3)
        print(json.dumps({"mode": mode, "arity": ar, "n": n, "digest": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], "ms": round(ms, 3), "Mhash/s": round(n / ms / 1e3, 2)}))
    elif mode == "ntt":
        d = rand_fr(n)
        ms = timeit(lambda: ctx.ntt_dev(d, log_n, False, bool(int(os.environ.get("NTT_COSET", "0")))))
        print(json.dumps({"mode": mode, "log_n": log_n, "coset": os.environ.get("NTT_COSET", "0"), "variant": os.environ.get("BZK_NTT_VARIANT"),
                          "tile": os.environ.get("BZK_NTT_TILE"), "bmax": os.environ.get("BZK_NTT_BMAX"), "ms": round(ms, 4),
                          "alg_GB/s": round(64 * n / ms / 1e6, 1), "G_fr_products/s": round(log_n * n / 2 / ms / 1e6, 1)}))
    elif mode == "h":
        a, b, c = rand_fr(n), rand_fr(n), rand_fr(n)
        ms = timeit(lambda: ctx.groth16_h_dev(a, b, c, log_n), reps=3)
        print(json.dumps({"mode": mode, "log_m": log_n, "variant": os.environ.get("BZK_NTT_VARIANT"), "unfused": os.environ.get("BZK_H_UNFUSED"),
                          "ms_7_ntts_plus_pointwise": round(ms, 3)}))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3])); sys.exit(0)
    def run(mode, log_n, env=None):
        e = dict(os.environ); e.update(env or {})
        try:
            out = subprocess.run([sys.executable, __file__, "child", mode, str(log_n)], env=e, capture_output=True, text=True,
                                 timeout=int(os.environ.get("SWEEP_CHILD_TIMEOUT", "150")))
            lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(lines[-1] if lines else ("FAILED " + mode + " " + out.stderr[-300:]), flush=True)
        except subprocess.TimeoutExpired:
            print("TIMEOUT", mode, log_n, env, flush=True)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "sweep"):
        for c in (13, 14, 15, 16, 17):
            run("g1", 20, {"BZK_MSM_C": str(c)})
        for ch in (8, 32):
            run("g1", 20, {"BZK_MSM_C": "16", "BZK_MSM_CHUNK": str(ch)})
        for lg in (16, 18, 22, 24):
            run("g1", lg)
    if what in ("r2csort",):  # round 2: counting sort vs radix sort of the pairs, side-stream base conversion
        for lg in (20, 22, 24):
            for cs in ("1", "0"):
                run("g1", lg, {"BZK_MSM_CSORT": cs})
        run("g1", 20, {"BZK_MSM_NO_AUX": "1"})
        run("g1", 18); run("g1", 16)
        for cs in ("1", "0"):
            run("g2", 20, {"BZK_MSM_CSORT": cs})
        run("g1win", 23)
    if what in ("r2sortcfg",):  # round 2: rocPRIM with the explicit 8-bit onesweep configuration vs its untuned gfx950 fallback (4 bits, merge sort)
        for lg in (20, 22, 24, 16):
            for d in ("0", "1"):
                run("g1", lg, {"BZK_MSM_SORT_DEFAULT": d})
        for d in ("0", "1"):
            run("g2", 20, {"BZK_MSM_SORT_DEFAULT": d})
        run("g1win", 23)
    if what in ("r2psort",):  # round 2: LDS two-pass partition vs rocPRIM radix sort of the pairs
        for lg in (20, 22, 24, 18):
            for ps in ("1", "0"):
                run("g1", lg, {"BZK_MSM_PSORT": ps})
        for ps in ("1", "0"):
            run("g2", 20, {"BZK_MSM_PSORT": ps})
        run("g1win", 23)
    if what in ("r6split",):  # round 6 run 18: one stand-alone G1 call as 1 / 2 / 3 / 4 window ranges in flight (msm_run_split), children at normal / highest priority
        for rep in (0, 1):  # alternating
            for lg in (20, 22, 18, 24):
                for sp, pr in (("1", "0"), ("2", "0"), ("2", "1"), ("3", "0"), ("3", "1"), ("4", "0"), ("4", "1")):
                    if lg != 20 and sp in ("3",):
                        continue
                    run("g1res", lg, {"BZK_MSM_SPLIT": sp, "BZK_MSM_SPLIT_PRIO": pr, "SWEEP_REPS": "8", "SWEEP_MEAN_OVER": "20" if lg <= 22 else "5",
                                      "BZK_MSM_SPLIT_MIN_LOG": "16"})
    if what in ("r6split2",):  # run 19: unequal ranges (short first range = short exposed head, short last range = short exposed tail), 2 tasks per lane in a range
        for rep in (0, 1):
            for lg in (20, 22):
                for cuts in ("", "8,8", "4,12", "12,4", "4,8,4", "2,12,2", "2,10,4", "4,10,2", "2,6,6,2", "6,10", "10,6"):
                    env = {"SWEEP_REPS": "8", "SWEEP_MEAN_OVER": "20", "BZK_MSM_SPLIT_MIN_LOG": "16"}
                    env.update({"BZK_MSM_SPLIT_CUTS": cuts} if cuts else {"BZK_MSM_SPLIT": "1"})
                    run("g1res", lg, env)
            for cuts in ("", "9,9", "4,10,4", "6,12", "12,6"):  # 2^18: c = 15?, W = 18 (cuts that do not add up to W fall back to equal ranges)
                env = {"SWEEP_REPS": "8", "SWEEP_MEAN_OVER": "20", "BZK_MSM_SPLIT_MIN_LOG": "16"}
                env.update({"BZK_MSM_SPLIT_CUTS": cuts} if cuts else {"BZK_MSM_SPLIT": "1"})
                run("g1res", 18, env)
    if what in ("r6split3",):  # run 20: priority modes (0 none, 1 later ranges high, 2 later ranges high with their saturating kernels lowest)
        for rep in (0, 1):
            for lg, cutss in ((20, ("8,8", "6,10", "10,6", "4,12")), (22, ("8,8", "6,10")), (18, ("9,9",)), (19, ("8,8",)), (17, ("9,9",)), (16, ("10,10",)), (24, ("8,8",))):
                base = {"SWEEP_REPS": "8", "SWEEP_MEAN_OVER": "20" if lg <= 22 else "5", "BZK_MSM_SPLIT_MIN_LOG": "16"}
                run("g1res", lg, dict(base, BZK_MSM_SPLIT="1"))
                for cuts in cutss:
                    for pr in ("0", "1", "2"):
                        if pr == "1" and cuts != cutss[0]:
                            continue
                        run("g1res", lg, dict(base, BZK_MSM_SPLIT_CUTS=cuts, BZK_MSM_SPLIT_PRIO=pr))
    if what in ("r6seg",):  # run 21: the task cut of mid-size calls (enough_tasks): run length forced against the default
        for rep in (0, 1):
            for lg in (19, 18, 17, 16):
                for seg in ("", "32", "48", "64", "128", "254"):
                    env = {"SWEEP_REPS": "8", "SWEEP_MEAN_OVER": "20", "BZK_MSM_SPLIT": "1"}
                    if seg:
                        env["BZK_MSM_SEG"] = seg
                    run("g1res", lg, env)
    if what in ("r6wide",):  # run 22: two-level fold of giant buckets (msm_fold_wide_kernel) against the one-level fold, unsplit and as two ranges
        for rep in (0, 1):
            for lg in (19, 18, 17, 16, 20, 21):
                for wide_off in ("1", "0"):
                    for sp in ("1", "2"):
                        run("g1res", lg, {"SWEEP_REPS": "8", "SWEEP_MEAN_OVER": "20", "BZK_MSM_SPLIT": sp, "BZK_MSM_SPLIT_MIN_LOG": "16", "BZK_MSM_NO_WIDE_FOLD": wide_off})
    if what in ("r6acclds",):  # run 24: the G1 accumulation's next base through LDS by direct loads (requested at the top of the addition) vs into registers under the tail
        alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bazuka_amd", "libbzk.so.acclds0")
        for rep in (0, 1):
            for lib in (alt, ""):
                env = {"BZK_LIBBZK": lib} if lib else {}
                for lg in (20, 22, 24, 18):
                    run("g1res", lg, dict(env, SWEEP_REPS="8", SWEEP_MEAN_OVER="20" if lg <= 22 else "5"))
                run("g1tab", 20, env)
                run("g1res", 20, dict(env, THROUGHPUT="1", SWEEP_REPS="8"))
    if what in ("r6rcleaf",):  # run 30: buckets per lane of the row / column sums' serial phase
        for rep in (0, 1):
            for lf in ("8", "4", "16", "2"):
                for lg in (20, 22):
                    run("g1res", lg, {"SWEEP_REPS": "8", "SWEEP_MEAN_OVER": "20", "BZK_MSM_RC_LEAF": lf})
    if what in ("occ",):
        for occ in (2, 3, 4):
            run("g1", 20, {"BZK_MSM_ACC_OCC": str(occ)})
            run("g1", 22, {"BZK_MSM_ACC_OCC": str(occ)})
    if what in ("r4pos",):  # round 4: MDS rows as straight-line code (no scratch) vs the rolled loop (bazuka_amd/libbzk.so.rows0, built with -DBZK_POSEIDON_ROWS_STRAIGHT=0)
        alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bazuka_amd", "libbzk.so.rows0")
        for lib in (alt, ""):
            env = {"BZK_LIBBZK": lib} if lib else {}
            run("tree", 24, env); run("tree", 20, env)
            for ar in ("2", "3", "4", "5", "7"):
                run("hash", 22, dict(env, ARITY=ar))
    if what in ("r4inl",):  # round 4, late: products inlined into the G1 accumulation / Poseidon's partial rounds, quotient digit by v_mad_u64_u32: same-box A/B,
        # alternating twice.  base = all three switches off, inl = inlined forms without the multiply-add digit, "" = the default build, mdsr = default + MDS re-read per round
        here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bazuka_amd")
        L = lambda name: {"BZK_LIBBZK": os.path.join(here, "libbzk.so." + name)} if name else {}
        for rnd in range(2):
            for name in ("base", "inl", ""):
                print("## lib", name or "default", flush=True); run("g1res", 20, L(name))
            for name in ("base", ""):
                print("## lib", name or "default", flush=True); run("g2res", 20, L(name))
            for name in ("base", "", "mdsr"):
                print("## lib", name or "default", flush=True); run("tree", 24, L(name)); run("hash", 22, dict(L(name), ARITY="4"))
        for name in ("base", ""):
            print("## lib", name or "default", flush=True); run("hash", 22, dict(L(name), ARITY="2")); run("g1tab", 20, L(name))
        print("## same inputs in every child (seeded): equal digests across libraries = the new forms compute what the committed build computes")
    if what in ("r4ntt",):  # round 4: 32-byte inter-pass elements (BZK_NTT_IP32) vs the 48-byte padded limb form
        for ip in ("0", "1"):
            for lg in (20, 22, 24):
                run("ntt", lg, {"BZK_NTT_IP32": ip})
            run("h", 20, {"BZK_NTT_IP32": ip}); run("h", 24, {"BZK_NTT_IP32": ip})
    if what in ("r4endo2",):  # round 4: why the endomorphism form's accumulation is slower - task length and reduction chunk
        for sg in ("32", "64", "128", "256"):
            run("g2res", 20, {"BZK_MSM_ENDO_G2": "1", "BZK_MSM_SEG": sg})
        for sg in ("16", "32", "64"):
            run("g2res", 20, {"BZK_MSM_ENDO_G2": "0", "BZK_MSM_SEG": sg})
        for ch in ("2", "4"):
            run("g2res", 20, {"BZK_MSM_ENDO_G2": "1", "BZK_MSM_CHUNK": ch, "BZK_MSM_SEG": "64"})
        for sg in ("32", "64", "128"):
            run("g1res", 20, {"BZK_MSM_ENDO_G1": "1", "BZK_MSM_SEG": sg})
        for ch in ("2", "4"):
            run("g1res", 20, {"BZK_MSM_ENDO_G1": "1", "BZK_MSM_CHUNK": ch})
        run("g1res", 20, {"BZK_MSM_ENDO_G1": "0", "BZK_MSM_SEG": "16"})
    if what in ("r4endo",):  # round 4: endomorphism form vs plain form on resident sets, stand-alone (latency form) and with the throughput hint
        for e in ("0", "1"):
            run("g1res", 20, {"BZK_MSM_ENDO_G1": e}); run("g1res", 20, {"BZK_MSM_ENDO_G1": e, "THROUGHPUT": "1"})
            run("g2res", 20, {"BZK_MSM_ENDO_G2": e}); run("g2res", 20, {"BZK_MSM_ENDO_G2": e, "THROUGHPUT": "1"})
        run("g1res", 22, {"BZK_MSM_ENDO_G1": "0"}); run("g1res", 22, {"BZK_MSM_ENDO_G1": "1"})
    if what in ("r4strong",):  # round 4: SURVEY C4 (i) on one GPU: the whole 2^26-point MSM, and what one rank of 8 / 4 / 2 does of it (its window share)
        run("g1res", 26); run("g1winres", 26); run("g1winres", 26, {"SHARDS": "4"}); run("g1winres", 26, {"SHARDS": "2"})
        run("g1res", 24); run("g1winres", 24); run("g1winres", 24, {"SHARDS": "4"}); run("g1winres", 24, {"SHARDS": "2"})
    if what in ("r4rank8",):  # round 4: what one rank of 8 / 4 / 2 does at 2^23 / 2^22 / 2^21 points (weak scaling), raw and resident bases
        run("g1win", 23); run("g1winres", 23); run("g1winres", 22, {"SHARDS": "4"}); run("g1winres", 21, {"SHARDS": "2"}); run("g1", 20)
    if what in ("r39",):
        run("tree", 24); run("tree", 20); run("tree", 16); run("tree", 12)
    if what in ("r36",):
        run("g1", 20); run("g1", 22); run("g2", 20); run("g1win", 23)
    if what in ("r35",):
        run("g1", 20); run("g1", 22); run("g1", 24); run("g1win", 23); run("g1win", 22, {"SHARDS": "4"}); run("g1win", 21, {"SHARDS": "2"})
    if what in ("r31",):
        run("g2", 20); run("g2", 18); run("g1", 20)
    if what in ("r28",):
        for lg in (18, 20, 22, 24):
            run("g1", lg)
        run("g2", 20); run("g2", 18)
    if what in ("r3ntt2",):  # round 3, second step: inlined products, XCD-aware column tiles
        for v in ("2", "4", "5"):
            for lg in (20, 22, 24):
                run("ntt", lg, {"BZK_NTT_VARIANT": v})
            run("h", 20, {"BZK_NTT_VARIANT": v})
        for v in ("2", "4"):
            for lg in (20, 24):
                run("ntt", lg, {"BZK_NTT_VARIANT": v, "BZK_NTT_NO_XCD": "1"})
            run("h", 20, {"BZK_NTT_VARIANT": v, "BZK_NTT_NO_XCD": "1"})
    if what in ("r3ntt",):  # round 3: 36-byte LDS tiles (4 workgroups per CU), raw first loads, fused h chain; kernel variants (occupancy, twiddle prefetch)
        for v in ("0", "1", "2", "3"):
            for lg in (20, 24):
                run("ntt", lg, {"BZK_NTT_VARIANT": v})
            run("h", 20, {"BZK_NTT_VARIANT": v})
        run("ntt", 20, {"NTT_COSET": "1"}); run("ntt", 24, {"NTT_COSET": "1"})
        run("h", 20, {"BZK_H_UNFUSED": "1"}); run("h", 24); run("h", 24, {"BZK_H_UNFUSED": "1"})
        for lg in (16, 18, 22):
            run("ntt", lg)
        run("ntt", 20, {"BZK_NTT_TILE": "2048"}); run("ntt", 24, {"BZK_NTT_TILE": "2048"})
    if what in ("r26",):
        for lg in (16, 18, 20, 22, 24):
            run("ntt", lg)
        run("ntt", 20, {"BZK_NTT_TILE": "2048"}); run("ntt", 24, {"BZK_NTT_TILE": "2048"})
        run("h", 20); run("h", 24)
    if what in ("r19",):
        for lg in (16, 18, 20, 22, 24):
            run("ntt", lg)
        for tile in ("1024", "4096"):
            run("ntt", 20, {"BZK_NTT_TILE": tile}); run("ntt", 24, {"BZK_NTT_TILE": tile})
        for bm in ("7", "8", "9"):
            run("ntt", 20, {"BZK_NTT_BMAX": bm}); run("ntt", 24, {"BZK_NTT_BMAX": bm})
        run("h", 20); run("h", 24)
    if what in ("r18",):
        run("tree", 24); run("tree", 20); run("hash", 22, {"ARITY": "2"}); run("hash", 22, {"ARITY": "4"}); run("hash", 22, {"ARITY": "7"})
    if what in ("r16",):
        run("g1", 20); run("g1", 22); run("g1", 24); run("g2", 20); run("g2", 18); run("tree", 24); run("h", 20)
    if what in ("g2occ",):
        for alt in ("0", "1"):
            run("g2", 20, {"BZK_MSM_OCC_ALT": alt})
            run("g2", 18, {"BZK_MSM_OCC_ALT": alt})
    if what in ("chunk",):
        for ch in (2, 4, 8, 16):
            run("g1", 20, {"BZK_MSM_CHUNK": str(ch)})
        run("g1", 22, {"BZK_MSM_CHUNK": "4"}); run("g1", 16, {"BZK_MSM_CHUNK": "4"})
        run("g2", 20, {"BZK_MSM_CHUNK": "4"}); run("g2", 20, {"BZK_MSM_CHUNK": "8"})
    if what in ("fold",):  # folded tables: 1 / L of the buckets for L x the base memory
        run("g1", 20)
        for lv in (2, 4, 8):
            run("g1tab", 20, {"TAB_LEVELS": str(lv)})
        run("g1tab", 20, {"TAB_LEVELS": "2", "BZK_MSM_TABLE_C": "17"})
        run("g1tab", 20, {"TAB_LEVELS": "4", "BZK_MSM_TABLE_C": "17"})
        run("g1", 22); run("g1tab", 22, {"TAB_LEVELS": "2"}); run("g1tab", 22, {"TAB_LEVELS": "4"})
    if what in ("r5g1tails",):  # round 5, run 5: what the reduction's existing forms give the stand-alone headline MSM (one level / two levels, chunk)
        for rep in range(2):
            for r2 in ("-1", "1"):
                for ch in ("8", "4"):
                    run("g1res", 20, {"BZK_MSM_ENDO_G1": "0", "BZK_MSM_CHUNK": ch, "BZK_MSM_REDUCE2": r2})
    if what in ("r5g2b",):  # round 5, run 6: pair tails with the two-level reduction for every call (level-2 chunk 2 / 4 / 8) against one level
        for rep in range(2):
            run("g2", 20, {"BZK_MSM_REDUCE2": "-1"})
            for ch in ("2", "4", "8"):
                run("g2", 20, {"BZK_MSM_PAIR_L2_CH": ch})
        for ch in ("2", "4", "8"):
            run("g2res", 20, {"BZK_MSM_PAIR_L2_CH": ch, "BZK_MSM_ENDO_G2": "1", "THROUGHPUT": "1"})
        run("g2", 18); run("g2", 16); run("g2", 22)
    if what in ("r6bitsum",):  # round 6, run 4: the multiplication-free bucket reduction (row / column sums + bit sums, weights in the host Horner) against the chunked running sum, same box, alternating
        for rep in range(2):
            for bs in ("0", "1"):
                run("g1res", 20, {"BZK_MSM_BITSUM": bs})
                run("g1res", 20, {"BZK_MSM_BITSUM": bs, "BZK_MSM_ENDO_G1": "1", "THROUGHPUT": "1"})
                run("g1tab", 20, {"BZK_MSM_BITSUM": bs, "BZK_MSM_TABLE_C": "20"})
        for bs in ("0", "1"):
            run("g1", 18, {"BZK_MSM_BITSUM": bs}); run("g1", 16, {"BZK_MSM_BITSUM": bs}); run("g1res", 22, {"BZK_MSM_BITSUM": bs}); run("g1res", 24, {"BZK_MSM_BITSUM": bs})
    if what in ("r6g2lds",):  # round 6, run 2: the G2 pair accumulation with the next base through LDS (direct loads) against the round-5 form (through registers, parked in scratch), alternating libraries
        libs = (os.path.join(ROOT, "bazuka_amd", "libbzk.so.g2reg"), os.path.join(ROOT, "bazuka_amd", "libbzk.so"))
        for rep in range(2):
            for lib in libs:
                tag = {"BZK_LIBBZK": lib}
                print("# lib", os.path.basename(lib), flush=True)
                run("g2", 20, tag)
                run("g2res", 20, dict(tag, BZK_MSM_ENDO_G2="1", THROUGHPUT="1"))
        for lib in libs:
            print("# lib", os.path.basename(lib), flush=True)
            run("g2", 18, {"BZK_LIBBZK": lib}); run("g2", 22, {"BZK_LIBBZK": lib})
    if what in ("r5g2",):  # round 5, run 4: one-lane G2 / pair accumulation with one-lane tails / pairs everywhere, same box, alternating (the container of runs 2 - 3 was lost)
        cfgs = ({"BZK_G2_PAIR": "0"}, {"BZK_G2_PAIR": "1", "BZK_G2_PAIR_TAILS": "0"}, {"BZK_G2_PAIR": "1", "BZK_G2_PAIR_TAILS": "1"})
        for rep in range(2):
            for cfg in cfgs:
                run("g2", 20, cfg)
                run("g2res", 20, dict(cfg, BZK_MSM_ENDO_G2="1", THROUGHPUT="1"))
        for cfg in cfgs:
            run("g2", 18, cfg)
    if what in ("r5tails",):  # round 5, run 3: the G2 tails (folds, bucket reduction, window sums) on pairs of lanes against the one-lane kernels, same box, alternating
        for rep in range(2):
            for tails in ("0", "1"):
                run("g2", 20, {"BZK_G2_PAIR_TAILS": tails})
                run("g2res", 20, {"BZK_G2_PAIR_TAILS": tails, "BZK_MSM_ENDO_G2": "0"})
                run("g2res", 20, {"BZK_G2_PAIR_TAILS": tails, "BZK_MSM_ENDO_G2": "1", "THROUGHPUT": "1"})
        for tails in ("0", "1"):
            run("g2", 16, {"BZK_G2_PAIR_TAILS": tails}); run("g2", 18, {"BZK_G2_PAIR_TAILS": tails})
        for ch in ("4", "16"):
            run("g2", 20, {"BZK_MSM_CHUNK": ch})
    if what in ("r5pair",):  # round 5, run 2: the G2 accumulation on pairs of lanes (BZK_G2_PAIR=1, the default) against the one-lane kernel, same box, alternating
        for rep in range(2):
            for pair in ("0", "1"):
                run("g2", 20, {"BZK_G2_PAIR": pair})
                run("g2res", 20, {"BZK_G2_PAIR": pair, "BZK_MSM_ENDO_G2": "0"})
                run("g2res", 20, {"BZK_G2_PAIR": pair, "BZK_MSM_ENDO_G2": "1"})
        for pair in ("0", "1"):
            run("g2", 16, {"BZK_G2_PAIR": pair}); run("g2", 18, {"BZK_G2_PAIR": pair}); run("g2", 22, {"BZK_G2_PAIR": pair})
    if what in ("r5knobs",):  # round 5, run 1: what the existing switches give a stand-alone 2^20 MSM over a resident set - endomorphism form x reduce chunk x one / two-level reduction
        for mode in ("g1res", "g2res"):
            e = "BZK_MSM_ENDO_G1" if mode == "g1res" else "BZK_MSM_ENDO_G2"
            for endo in ("0", "1"):
                for ch in ("8", "4", "2"):
                    for r2 in ("-1", "1"):
                        run(mode, 20, {e: endo, "BZK_MSM_CHUNK": ch, "BZK_MSM_REDUCE2": r2})
    if what in ("all", "tab"):
        for c in (15, 16, 17, 18, 19):
            run("g1tab", 20, {"BZK_MSM_TABLE_C": str(c)})
    if what in ("all", "others"):
        run("tree", 24); run("tree", 20); run("ntt", 20); run("ntt", 24); run("h", 20); run("g2", 16); run("g2", 20)
