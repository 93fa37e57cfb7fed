#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X-native Groth16 hot path (BASELINE.json metric).

Step      = one complete BLS12-381 G1 Pippenger MSM over a synthetic batch whose bases and scalars are
            already resident in HBM (BASELINE.json configs[1]: 2^20 points on one MI355X).
            The bases are a resident set (bzk_msm_g1_bases_load_dev: the static CRS converted once to the internal limb form,
            outside the timed region, like the CRS upload itself); the per-call pipeline on raw bases is timed beside it
            (other_configs.msm_g1_2p20_raw_bases).
N GPUs    = the north-star partition: ONE MSM over N * 2^20 points, sharded by scalar-window range
            (rank r owns windows [W*r/N, W*(r+1)/N)), bases/scalars replicated; the collective lives BEHIND THE C ABI
            (bzk_mg_*, bazuka_amd/csrc/mg.hip): one RCCL all-gather of the W window sums (3 KB for the whole group; RCCL cannot
            add curve points), then the Horner combine on the host as in the single-GPU call.  Per-GPU work (window x point pairs)
            is constant in N  => "scaling": "weak".  Launch: one process per GPU under torch.distributed.run; `python bench.py
            --gpus N` without a launcher re-executes itself under torch.distributed.run (127.0.0.1, free port).
value     = points of the whole job / second (max over ranks of the timed region).
roofline  = msm_accumulate (dominant kernel): algorithmic 128 B per (point, scalar) pair over its HIP
            event time, against the 8 TB/s HBM peak.  The kernel is integer-ALU bound, so this
            fraction is tiny by construction; int_ops/s is reported beside it.
cpu_baseline = the C++ oracle Pippenger (bellman-equivalent algorithm, "port") on the host cores,
            rank 0, N=1 only; its result doubles as the parity check of the timed GPU result.  `cores` = threads
            used, `cpu_quota` = CPUs the container may actually consume (cgroup; 16 of the 256 shown on the GPU box).
Other sections of the same JSON line (N = 1): `proofs` (full Groth16 proofs of the 2^20-class Update circuit:
            single, serial and pipelined - 8 host witness producers x 8 threads -> 4 prover slots on the GPU - with
            their own cpu_baseline and `proof_roofline`), `other_configs` (2^24-leaf tree, MPN-shaped state, NTT
            2^20 / 2^24, h stage, G2 MSM 2^20, static-table G1 MSM), `two_msms_in_flight`, `kernel_ms_per_step`.
            The `proofs` section is measured in a process of its own, spawned before this one touches the GPU, whose HIP runtime
            hands launches to its worker threads (AMD_DIRECT_DISPATCH=0: what a proving service sets; `proofs.process`).
Options:    --scaling strong --log-n-total 24|26 (fixed job), --partition points (rank r owns points, not windows),
            --no-proofs / --no-others / --no-overlap / --no-cpu-baseline (shorter runs for profiling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before the process's first HIP call (torch initialises the runtime): see bazuka_amd/__init__.py
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

LOG_N = 20
# micro-benchmarked peak of the library's own Fp product (chains of dependent calls, all CUs busy):
# tools/ubench_int "Fp28 lib mul", profiles/r04_ubench_int.txt.  An XYZZ mixed add is 8 products + 2 squares; since round 2 its Y
# coordinate R (Q - X3) - Y PPP is two multiplications under ONE Montgomery reduction (fp28::mul_sub2_body) = 1.5 products
FP_MUL_PEAK_G = 76.2
MULS_PER_MIXED_ADD = 9.5
# integer multiply-adds actually issued by one XYZZ mixed addition: 6 products of 394 v_mad_u64_u32 + 2 squares of 301
# (profiles/r01_run56_fp28_square.txt) + the fused Y (2 x 196 + 196 + 2), against the measured v_mad_u64_u32 ceiling of the whole
# chip (profiles/r04_ubench_int.txt)
MADS_PER_MIXED_ADD = 6 * 394 + 2 * 301 + (3 * 196 + 2)
MAD_PEAK_T = 31.4
# G2: 6 Fp2 products (3 Fp products each) + 2 Fp2 squares (2 Fp products each) + the fused Y (8 multiplications under 2 reductions
# = 5 products' worth of multiply-adds); 28 before round 2's g2x28::add_mixed
FP_MULS_PER_G2_MIXED_ADD = 6 * 3 + 2 * 2 + 5
PROOF_ALG_BYTES = 1.24e9  # MSM 481 MB (G1) + 203 MB (G2) + 7 NTTs x 67 MB + vectors 87 MB per 2^20-class proof
SEED = 0x42415A554B41
HBM_PEAK_GBS = 8000.0


R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _fr(x: int) -> bytes:
    """Montgomery limbs of a small integer (what a Rust host would hand over as a ZkScalar)."""
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def _rows_ge_r(rows):
    """per row of a (k, 32) uint8 tensor (little-endian 256-bit integers): value >= r ?"""
    import torch
    q = rows.contiguous().view(torch.int64).view(-1, 4)
    ge = torch.ones(q.shape[0], dtype=torch.bool, device=rows.device)   # "equal on every limb seen so far" counts as >=
    for i in range(4):                                                   # least significant limb first: a higher limb overrides
        a = q[:, i] ^ (-(1 << 63))                                       # x ^ 2^63 read as int64 keeps the unsigned order
        b = ((R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF) ^ (1 << 63)
        b -= (1 << 64) if b >= (1 << 63) else 0
        ge = torch.where(a == b, ge, a > b)
    return ge


def uniform_fr_dev(cnt: int, seed: int, dev):
    """cnt Fr values UNIFORM in [0, r) by rejection, as 32-byte little-endian Montgomery limbs on `dev`: 255-bit candidates
    (acceptance r / 2^255 = 0.906), rows >= r redrawn until none is left.  Limbs uniform in [0, r) <=> the canonical values are (x -> x R^-1
    is a bijection of the residues)."""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed & 0x7FFFFFFF)
    t = torch.randint(0, 256, (cnt, 32), dtype=torch.uint8, device=dev, generator=g)
    t[:, 31] &= 0x7F
    bad = _rows_ge_r(t).nonzero().flatten()
    while bad.numel():
        fresh = torch.randint(0, 256, (bad.numel(), 32), dtype=torch.uint8, device=dev, generator=g)
        fresh[:, 31] &= 0x7F
        t[bad] = fresh
        bad = bad[_rows_ge_r(fresh)]
    return t.contiguous()


def _fr_blind(k: int) -> bytes:
    """A full-size (255-bit) blinding scalar for proof k, seeded: bellman's create_random_proof draws r and s uniformly, and the
    host assembly of a proof (six scalar multiplications in the reference) costs time in proportion to their bit length - a
    bench proving with r = 7 would skip it."""
    import hashlib
    v = int.from_bytes(hashlib.sha256(b"bzk bench blinding %d" % k).digest(), "little") % R_MOD
    return _fr(v)


MSM_KERNEL_SOURCES = ("msm_impl.cuh", "msm_policy.cuh", "msm_g1.hip", "bzk_fp28.cuh", "bzk_curve.cuh", "bzk_field.cuh")


def quota_binds(quota, world: int = 1) -> bool:
    """True when this rank's share of the cgroup CPU quota is smaller than the thread count of the pipelined-proof section (8 producers x 8
    workers + ~20 prover threads): host waits then sleep instead of spinning (BZK_SYNC_BLOCKING=1).  env BZK_BENCH_BLOCKING_WAITS=0|1 overrides."""
    if "BZK_BENCH_BLOCKING_WAITS" in os.environ:
        return os.environ["BZK_BENCH_BLOCKING_WAITS"] != "0"
    if quota is None:
        quota = float(os.cpu_count() or 64) if world > 1 else None
    return quota is not None and quota / max(1, world) < 84


def host_thread_budget(world: int):
    """(producers, worker threads per producer) for one rank of `world`: the ranks of a node share its CPUs - the cgroup quota where
    one is set (16 on the GPU pool's boxes), the visible CPUs otherwise - so a rank gets quota / world of them, not cpu_count
    (eight ranks x 64 threads on a 16-CPU quota measure the host)."""
    if world == 1:
        return 8, 8
    q = cpu_quota() or float(os.cpu_count() or 64)
    share = max(1.0, q / world)
    n_prod = max(1, min(8, int(share // 2)))
    return n_prod, max(1, min(8, int(share // n_prod)))


def cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max / v1 cfs quota), None when unlimited.  The GPU box shows 256
    logical CPUs but runs this job inside a 16-CPU quota (profiles/r02_run37_46_host_interference.txt): `cores` of a cpu_baseline is the
    number of THREADS the oracle used, `cpu_quota` what the kernel lets them consume."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def msm_source_stamp():
    """sha256 over the sources the MSM kernels are built from: a PMC measurement is only quoted for the build it was taken on
    (the GPU box has no .git, so the stamp is a content hash, not a commit id)"""
    import hashlib
    h = hashlib.sha256()
    for f in MSM_KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "bazuka_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _pmc_traffic(log_n):
    """HBM-side bytes per msm_accumulate launch from the committed PMC passes of this same command (profiles/*pmc_traffic.json,
    written by tools/pmc_traffic.py from two rocprofv3 --pmc runs: counters cannot be collected from inside the timed process).
    The newest file whose source stamp equals the current MSM sources is used; a measurement of another build is refused
    (returns None plus the reason) instead of being quoted for kernels it was not taken on."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True)
    if log_n != LOG_N or not cands:
        return None, "no PMC measurement of this configuration on file"
    stamp = msm_source_stamp()
    for path in cands:
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if doc.get("source_stamp") == stamp:
            return doc["traffic_bytes_per_launch"], os.path.basename(path)
    return None, f"stale: no profiles/*pmc_traffic*.json carries source stamp {stamp} (re-run tools/runs/pmc passes)"


OTHER_KERNEL_SOURCES = ("ntt.hip", "poseidon.hip", "bzk_fr29.cuh", "bzk_poseidon29.cuh", "bzk_poseidon29_coop.cuh", "msm_impl.cuh", "msm_policy.cuh", "msm_g2.hip", "bzk_fp28.cuh",
                        "bzk_g2pair.cuh", "msm_g2pair_tails.cuh")


def other_source_stamp():
    import hashlib
    h = hashlib.sha256()
    for f in OTHER_KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "bazuka_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _pmc_other(name):
    """HBM-side bytes of one of the other kernels (tools/pmc_kernels.py over tools/pmc_ops.py), only if taken on THIS build's sources"""
    import glob
    stamp = other_source_stamp()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_other_kernels*.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        k = doc.get("kernels", {}).get(name)
        if doc.get("source_stamp") == stamp and k and "traffic_bytes_per_launch" in k:
            return k["traffic_bytes_per_launch"], f"{os.path.basename(path)} ({k['per']}; {k['correction']})"
    return None, f"no profiles/*pmc_other_kernels*.json carries source stamp {stamp}"


def full_prove_section(ctx, n_proofs: int = 4, n_prod: int = 4, prod_threads: int = 16, cpu_baseline: bool = False, helper: bool = False,
                       second_process: bool = False, world: int = 1, ring_k: int = 6):
    """Second half of BASELINE.json's metric: Groth16 proofs/s for the 2^20-constraint MPN class =
    UpdateCircuit(L=15, T=3, B=2): 16 signed transactions, 903 037 constraints, 2^20 NTT domain.
    Product code only: host witness/R1CS generator (C++ worker threads), CRS generated on the GPU
    (bzk_groth16_setup), proof on the GPU.  Parity of this path (387 proof bytes == oracle, pairing check)
    is the job of tests/test_gpu_mpn_prove.py; the optional cpu_baseline leg below (N = 1, rank 0) times the CPU
    oracle on one proof and compares its bytes with the GPU's - the oracle is never on the measured path."""
    import threading
    from bazuka_amd import lib as L
    ZIESHA = _fr(1)
    lg, t, b = 15, 3, 2
    n_tx = 1 << (2 * b)
    w = L.MpnWorld(lg, t)
    for i in range(2 * n_tx):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 12)
    state = {"k": 0}

    def batch():
        state["k"] += 1
        for i in range(n_tx):
            w.push_tx(i, n_tx + i, ZIESHA, 100 + i + state["k"], ZIESHA, i % 7)

    out = {"circuit": "UpdateCircuit(L=15,T=3,B=2): 16 tx"}
    batch()
    r = w.update_synthesize(b, _fr(99), ZIESHA, record_matrices=True)
    assert r.satisfied and r.accepted == n_tx
    out.update(n_constraints=r.n_constraints, n_aux=r.n_aux)
    csr = [(r.n_constraints, r.view("rp" + x), r.view("col" + x), r.view("val" + x)) for x in "ABC"]
    tox = b"".join(_fr(x) for x in (1234567, 2345678, 3456789, 4567891, 5678912))
    t0 = time.perf_counter()
    ph, _vk = ctx.groth16_setup(csr, r.n_in, r.n_aux, tox)
    out["gpu_crs_setup_s"] = round(time.perf_counter() - t0, 3)
    tw, tp, tcpu = [], [], []
    cur = None
    import resource
    for k in range(n_proofs):
        batch()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        cur = w.update_synthesize(b, _fr(99), ZIESHA)
        t1 = time.perf_counter()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        tcpu.append((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime))
        assert cur.satisfied
        views = [cur.raw(x) for x in ("z", "az", "bz", "cz")]
        t2 = time.perf_counter()
        ctx.groth16_prove(ph, *views, _fr_blind(2 * k), _fr_blind(2 * k + 1))
        t3 = time.perf_counter()
        tw.append(t1 - t0)
        tp.append(t3 - t2)
    out["witness_s"] = round(min(tw), 4)
    # host CPU seconds one witness costs (user + system of the whole process around the call, nothing else running: the library's worker
    # threads included) - what a host must spend per proof whatever its thread count (VERDICT r3 missing 4)
    out["witness_cpu_s"] = round(min(tcpu), 4)
    out["gpu_prove_s"] = round(min(tp), 4)
    # The same with the hash-dependent witness values left to the device (round 5; VERDICT r4 item 3: bzk_mpn_set_defer + bzk_groth16_prove_r1cs): the host
    # generator skips the Poseidon gadget's variables, the Merkle muxes and the root checks (82 % of a transition's constraints) and the prover runs the
    # instance's program on the GPU before anything reads the arrays.  Same proof bytes (tests/test_gpu_defer.py).  BZK_BENCH_DEFER=1: the live producers
    # below use it too (A/B; the default is the plain generator: see DESIGN.md section 3.5 for the measured trade)
    # Default since round 6: ON when several ranks share the host (N > 1: the producers are bound by the host's CPU quota and a deferred witness costs 0.035 - 0.040
    # instead of 0.051 - 0.063 CPU-s), OFF for one rank (the GPU is the bound there and the program is 5.6 ms of extra device work per proof: 69.3 - 70.3 against
    # 71.5 - 72.4 proofs/s, profiles/r06_run6...)
    defer = os.environ.get("BZK_BENCH_DEFER", "1" if world > 1 else "0") != "0"
    # BZK_BENCH_STAGE=1 (with or without deferral): each producer owns a context and STAGES its instances - the 116 MB upload and the deferred-value
    # program run on the producer's stream (bzk_r1cs_stage), the prover slots copy device to device (bzk_groth16_prove_staged)
    # default ON since round 6 for every N: staging alone (plain instances) is + 1 - 2 % pipelined on one GPU - the 116 MB upload leaves the prover slot's
    # stream (73.8 / 75.6 -> 75.3 / 76.4 proofs/s, same box, alternating: profiles/r06_run16_staged_plain_producers.txt)
    stage = os.environ.get("BZK_BENCH_STAGE", "1") != "0"
    w.set_defer(True)
    twd, tpd, tcd = [], [], []
    for k in range(n_proofs):
        batch()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        curd = w.update_synthesize(b, _fr(99), ZIESHA)
        t1 = time.perf_counter()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        tcd.append((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime))
        assert curd.satisfied and curd.defer_info()["deferred"] == 1
        t2 = time.perf_counter()
        ctx.groth16_prove_r1cs(ph, curd, _fr_blind(2 * k), _fr_blind(2 * k + 1))
        tpd.append(time.perf_counter() - t2)
        twd.append(t1 - t0)
        curd.free()
    # ... and with the transition builder's Merkle walk on the device as well (bzk_mpn_set_device, row f-3: one batched Poseidon launch per tree level
    # instead of ~45 native hashes per transaction): what a host core then still does per witness is the signature gadgets, the range bits and the bookkeeping
    w.set_device(ctx)
    twb, tcb = [], []
    for k in range(n_proofs):
        batch()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        curb = w.update_synthesize(b, _fr(99), ZIESHA)
        t1 = time.perf_counter()
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        tcb.append((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime))
        twb.append(t1 - t0)
        assert curb.satisfied and curb.defer_info()["deferred"] == 1
        if k == 0:   # the instance is as good as any other: it proves and the proof verifies
            pb_ = ctx.groth16_prove_r1cs(ph, curb, _fr_blind(77), _fr_blind(78))
            assert L.groth16_verify(_vk, bytes(curb.raw("z")[32:32 * curb.n_in]), pb_)
        curb.free()
    w.set_device(None)
    w.set_defer(False)
    out["deferred"] = {"witness_s": round(min(twd), 4), "witness_cpu_s": round(min(tcd), 4), "gpu_prove_s": round(min(tpd), 4),
                       "with_device_builder": {"witness_s": round(min(twb), 4), "witness_cpu_s": round(min(tcb), 4),
                                               "what": "the same with bzk_mpn_set_device: the transition builder's tree walk as batched launches (CPU seconds exclude blocking waits for the GPU)"},
                       "live_producers_use_it": defer, "live_producers_stage": stage,
                       "what": "host generator without the Poseidon / Merkle value traces (bzk_mpn_set_defer), the device fills them in inside bzk_groth16_prove_r1cs"}
    if cpu_baseline:
        # The same proof on the host cores with the CPU oracle (kind "port": bellman's algorithms restated; the Rust
        # prover cannot be built here), ONE proof of the same circuit from the same CRS, witness and (r, s) - which makes
        # it the bit-exactness check of the timed GPU path as well.  This is the only place bench.py touches oracle/.
        from oracle import coracle as co
        d = {"n_in": r.n_in, "n_aux": r.n_aux, "log_m": (r.n_constraints - 1).bit_length(),
             "a_density": r.view("a_density"), "b_density": r.view("b_density")}
        for which, key in ((0, "vk"), (1, "h"), (2, "l"), (3, "a"), (4, "b_g1"), (5, "b_g2")):
            d[key] = ctx.params_read(ph, which)
        d["n_a"], d["n_b"] = sum(d["a_density"]), sum(d["b_density"])
        rk, sk = _fr_blind(2 * (n_proofs - 1)), _fr_blind(2 * (n_proofs - 1) + 1)
        gpu_proof = ctx.groth16_prove(ph, *views, rk, sk)
        q_now = cpu_quota()
        cores = max(1, min(co.ncpu(), int(q_now + 0.5))) if q_now else co.ncpu()   # threads = the CPUs the quota lets this container use
        dts = []
        for rep in range(4):   # one warm-up (page faults of the oracle's tables, thread pool), then the median of three (~8 s each)
            t0 = time.perf_counter()
            want = co.groth16_prove(d, cur.view("z"), cur.view("az"), cur.view("bz"), cur.view("cz"), rk, sk, nthreads=cores)
            if rep:
                dts.append(time.perf_counter() - t0)
            assert want == gpu_proof, "GPU proof bytes differ from the CPU oracle's"
        dt = sorted(dts)[1]
        out["cpu_baseline"] = {"value": round(1 / dt, 4), "unit": "proofs/s", "cores": cores, "cpu_quota": cpu_quota(), "kind": "port",
                               "sample": f"the same 16-tx circuit (same CRS, witness, r, s) proved 1 + 3 times, median of the last three {dt:.2f} s "
                                         f"(all: {', '.join('%.2f' % x for x in dts)})",
                               "parity": "bit-exact (387 proof bytes, every repetition)"}
        del d
    out["proofs_per_s_gpu_only"] = round(1 / min(tp), 2)
    out["proofs_per_s_serial"] = round(1 / (min(tp) + min(tw)), 3)
    # pipelined: host producers synthesize the next batches (GIL released inside libbzk) while the GPU proves.
    # Each producer owns an independent account tree - the shape of Bazuka's own work distribution, where a
    # prover holds several independent MpnWork items at once (src/mpn/mod.rs:79-107).
    import queue
    n_warm, n_pipe = 12, 64
    synth_s = []
    q = queue.Queue(maxsize=4)
    stop = threading.Event()
    # Host CPU is the scarce resource of this pipeline on a quota'd box (the GPU pool runs the job inside cgroup cpu.max = 16 CPUs while
    # showing 256): a proof's witness is ~0.23 CPU-seconds, so ~55 proofs/s keep ~13 CPUs busy, and the provers' ~20 host threads SPINNING in
    # their waits are charged against the same budget.  When the quota is smaller than the threads this pipeline starts, the provers wait
    # on interrupts instead - libbzk's BZK_SYNC_BLOCKING=1 (same-box A/B, profiles/r03_run23...: 49.9 / 54.1 -> 57.5 / 57.8 proofs/s with 4
    # worker threads per producer; the MSM headline of the same runs is unchanged within the box noise).  The runtime flag behind it is
    # process-wide and has to be in place before the first context exists (flipping it later hung the process, profiles/r03_run23...):
    # main() sets the variable at start-up when the quota binds.
    quota = cpu_quota()
    waits_blocking = os.environ.get("BZK_SYNC_BLOCKING", "0") != "0"  # main() decides (before the first context is created)
    throttled = quota_binds(quota, world)
    if throttled and "BZK_BENCH_PROD_THREADS" not in os.environ:
        prod_threads = min(prod_threads, 4)

    prod_dev = os.environ.get("BZK_BENCH_PRODUCER_DEV", "0") != "0"

    def producer(seed):
        pw = L.MpnWorld(lg, t)
        pw.set_threads(prod_threads)
        if defer:
            pw.set_defer(True)
        if prod_dev:  # A/B: the producers' Merkle hashing in batched launches on the GPU (bzk_mpn_set_device) instead of on their host threads
            pw.set_device(Bzk(ctx.device))
        pctx = Bzk(ctx.device) if stage else None
        for i in range(2 * n_tx):
            pw.add_account(i, b"p%dacct%d" % (seed, i), ZIESHA, 10 ** 12)
        k = 0
        while not stop.is_set():
            k += 1
            for i in range(n_tx):
                pw.push_tx(i, n_tx + i, ZIESHA, 100 + i + k, ZIESHA, i % 7)
            ts = time.perf_counter()
            rr = pw.update_synthesize(b, _fr(99), ZIESHA)
            synth_s.append(time.perf_counter() - ts)
            assert rr.satisfied and rr.n_constraints == r.n_constraints  # host-side self check of the witness
            item = (rr, pctx, pctx.r1cs_stage(rr)) if stage else rr
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    break
                except queue.Full:
                    pass

    threads = [threading.Thread(target=producer, args=(s,), daemon=True) for s in range(n_prod)]
    # Several prover slots on the same GPU (each its own context, lanes and scratch; since round 3 they SHARE the device-resident
    # CRS, its resident base sets and the h table - bzk_params_slot): the latency-bound tails of one proof (bucket reduction, window
    # sums, batched inversions) overlap the throughput-bound bucket accumulation of the others.
    from bazuka_amd import Bzk
    n_slots = max(1, int(os.environ.get("BZK_BENCH_SLOTS", "4")))
    slots = [(ctx, ph)]
    for _ in range(n_slots - 1):
        cx = Bzk(ctx.device)
        slots.append((cx, cx.params_slot(ph)))
    # the slots keep proving n_drain more proofs after the timed ones, so that the last timed proofs do not run on a draining GPU
    # (the timed window is steady state on both sides: n_warm completions before it, every slot still busy at its end)
    n_drain = len(slots)
    lock = threading.Lock()

    def run_pipeline(n_warm, n_pipe, until=None, finished=None, ring=None):
        """n_warm + n_pipe + n_drain proofs through the slots (or, with `until`, proofs until that event is set); returns the sorted
        completion times.  ring: pre-synthesised witnesses taken round-robin instead of the producers' queue; every proof draws its
        own (r, s) either way"""
        done = {"n": 0}
        finished = [] if finished is None else finished

        def consumer(slot):
            c, p = slots[slot]
            while True:
                with lock:
                    if (until.is_set() if until is not None else done["n"] >= n_warm + n_pipe + n_drain):
                        return
                    j = done["n"]
                    done["n"] += 1
                rr = ring[j % len(ring)] if ring is not None else q.get()
                if isinstance(rr, tuple):  # staged by its producer: (instance, staging context, handle)
                    inst, pc, hd = rr
                    c.groth16_prove_staged(p, hd, _fr_blind(1000 + 2 * j), _fr_blind(1001 + 2 * j))
                    pc.staged_free(hd)
                    inst.free()
                else:
                    c.groth16_prove_r1cs(p, rr, _fr_blind(1000 + 2 * j), _fr_blind(1001 + 2 * j))  # completes a deferred instance on the device first
                with lock:
                    finished.append(time.perf_counter())

        cons = [threading.Thread(target=consumer, args=(i,)) for i in range(len(slots))]
        for th in cons:
            th.start()
        return cons, finished

    def join_all(cons):
        for th in cons:
            th.join()

    if ring_k and not helper:
        # GPU-side rate with the host's witness synthesis out of the loop: a ring of K pre-synthesised witnesses (pinned host arrays; every
        # proof still uploads its 116 MB and draws its own r, s).  At N > 1 THIS is the reported proofs_per_sec: the
        # ranks of a node share one CPU quota, live producers would measure the host; their rate is reported beside it.
        ring = []
        for _ in range(ring_k):
            batch()
            rr = w.update_synthesize(b, _fr(99), ZIESHA)
            assert rr.satisfied
            ring.append(rr)
        import resource as _res
        ru0, tr0 = _res.getrusage(_res.RUSAGE_SELF), time.perf_counter()
        cons, fin = run_pipeline(8, n_pipe, ring=ring)
        join_all(cons)
        ru1, tr1 = _res.getrusage(_res.RUSAGE_SELF), time.perf_counter()
        fin.sort()
        out["proofs_per_s_ring"] = round(n_pipe / (fin[8 + n_pipe - 1] - fin[8 - 1]), 3)
        # what the PROVER side costs the host per proof (lane threads, staging, Horner, assembly, waits) - no witness is made in this leg:
        # user + system seconds of the process over all proofs of the leg (warm-up and drain included)
        out["prover_host_cpu_s_per_proof"] = round(((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)) / (8 + n_pipe + n_drain), 4)
        out["prover_host_cpu_cores_busy"] = round(((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)) / (tr1 - tr0), 2)
        out["ring"] = f"{ring_k} pre-synthesised witnesses of the 16-tx circuit round-robin -> {len(slots)} prover slots, {n_pipe} proofs timed (after 8, before the last {n_drain}); distinct (r, s) per proof"
        for rr in ring:
            rr.free()
        del ring
    for th in threads:
        th.start()
    if helper:
        # second prover process of `proofs.two_processes` (see below): prove until told to stop; protocol on stdin / stdout:
        # -> "READY" once n_warm proofs are through; <- "START": mark; <- "STOP": mark, report the proofs between the marks, leave
        until = threading.Event()
        fin = []
        cons, fin = run_pipeline(0, 0, until=until, finished=fin)
        while True:
            with lock:
                if len(fin) >= n_warm:
                    break
            time.sleep(0.01)
        print("READY", flush=True)
        marks = []
        for line in sys.stdin:
            cmd = line.strip()
            if cmd in ("START", "STOP"):
                with lock:
                    marks.append((time.perf_counter(), len(fin)))
            if cmd == "STOP":
                break
        until.set()
        join_all(cons)
        if len(marks) >= 2:
            print(json.dumps({"helper_proofs": marks[-1][1] - marks[0][1], "helper_seconds": marks[-1][0] - marks[0][0]}), flush=True)
        stop.set()
        for th in threads:
            th.join()
        for cx, px in slots[1:]:
            cx.params_free(px)
            cx.close()
        ctx.params_free(ph)
        return {}
    cons, finished = run_pipeline(n_warm, n_pipe)
    join_all(cons)
    finished.sort()  # completion times: rate over the n_pipe completions after the first n_warm
    out["proofs_per_s_pipelined"] = round(n_pipe / (finished[n_warm + n_pipe - 1] - finished[n_warm - 1]), 3)
    if second_process:
        # Informational (never `proofs_per_s_pipelined`): the same pipeline once more while a SECOND prover process - its own HIP runtime,
        # CRS, producers and slots - proves on the same GPU.  One process tops out below what the GPU can take (the N = 2 rehearsal with
        # both ranks on one GPU gave 65 proofs/s against 56 from one process, profiles/r03_run29...): a deployment runs two workers per GPU.
        import subprocess
        hp = None
        try:
            hp = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--prover-helper"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True, bufsize=1)
            ready = {"ok": False}

            def wait_ready():
                for line in hp.stdout:
                    if line.strip() == "READY":
                        ready["ok"] = True
                        return

            wr = threading.Thread(target=wait_ready, daemon=True)
            wr.start()
            wr.join(timeout=120.0)
            if not ready["ok"]:
                raise RuntimeError("helper process not ready within 120 s")
            hp.stdin.write("START\n")
            hp.stdin.flush()
            cons2, fin2 = run_pipeline(8, n_pipe)
            join_all(cons2)
            hp.stdin.write("STOP\n")
            hp.stdin.flush()
            fin2.sort()
            mine = n_pipe / (fin2[8 + n_pipe - 1] - fin2[8 - 1])
            rep = {}
            for line in hp.stdout:
                line = line.strip()
                if line.startswith("{"):
                    rep = json.loads(line)
                    break
            hp.wait(timeout=60)
            theirs = rep["helper_proofs"] / rep["helper_seconds"]
            out["two_processes"] = {"proofs_per_s": round(mine + theirs, 3), "this_process": round(mine, 3), "helper_process": round(theirs, 3),
                                    "how": f"informational: a second prover process ({n_prod} producers -> {len(slots)} slots, own CRS) on the same GPU while this "
                                           f"one repeats its {n_pipe}-proof measurement; the helper's rate over the enclosing START..STOP interval"}
        except Exception as e:
            out["two_processes"] = {"error": repr(e)}
        finally:
            if hp is not None and hp.poll() is None:
                hp.kill()
    out["producer_synth_s_mean_under_load"] = round(sum(synth_s) / len(synth_s), 4)
    out["pipeline"] = (f"{n_prod} host producers ({prod_threads} worker threads each{', tree hashing on the device' if prod_dev else ''}) -> {len(slots)} prover slots on 1 GPU, "
                       f"{n_pipe} proofs timed (after {n_warm}, before the last {n_drain})")
    out["host_waits"] = {"mode": "blocking (BZK_SYNC_BLOCKING=1: hipDeviceScheduleBlockingSync)" if waits_blocking else "spin (runtime default)", "cpu_quota": quota,
                         "why": "CPU quota below the pipeline's thread count" if throttled else "no binding CPU quota"}
    stop.set()
    for th in threads:
        th.join()
    for cx, px in slots[1:]:
        cx.params_free(px)
        cx.close()
    ctx.params_free(ph)
    return out


def state_seam_section(ctx, out):
    """the general state seam (row b) one-shot and device-resident; fills out[...]"""
    # general `ZkStateModel::compress` seam (row b): the MPN model at production depth over 4096 sparse accounts (one token each)
    try:
        import random as _r
        rnd = _r.Random(5)

        def mb(m):
            if m[0] == "scalar":
                return (0).to_bytes(4, "little")
            if m[0] == "struct":
                return (1).to_bytes(4, "little") + len(m[1]).to_bytes(8, "little") + b"".join(mb(f) for f in m[1])
            return (2).to_bytes(4, "little") + bytes([m[1]]) + mb(m[2])
        S_ = ("scalar",)
        model = mb(("list", 15, ("struct", [S_, S_, S_, S_, ("list", 3, ("struct", [S_, S_]))])))
        pairs = []
        for a in rnd.sample(range(4 ** 15), 4096):
            for j in range(4):
                pairs.append(((a, j), _fr(rnd.randrange(1, 1 << 60))))
            pairs.append(((a, 4, 0, 0), _fr(1)))
            pairs.append(((a, 4, 0, 1), _fr(rnd.randrange(1, 1 << 40))))
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            h_, n_ = ctx.state_compress(model, pairs)
            best = min(best, time.perf_counter() - t)
        out["state_compress_sparse_mpn"] = {"what": "bzk_state_compress: MpnConfig::state_model (L = 15, T = 3), 4096 populated accounts = 24 576 (locator, scalar) pairs, "
                                                    "~86 k hashes in 21 batched launches; includes the ctypes marshalling of the pairs",
                                            "ms": round(best * 1e3, 2), "state_size": n_}
        # the same state kept ON the device (bzk_state_*: `KvStoreStateManager::update_contract`): a block's worth of writes - 512
        # accounts, nonce + one balance each - re-hashes only the touched paths; the one-shot seam re-lays-out and re-hashes all of it
        from bazuka_amd import DeviceState
        dev = DeviceState(ctx, model)
        t = time.perf_counter()
        h0, n0 = dev.update(pairs, 1)
        load_s = time.perf_counter() - t
        assert (h0, n0) == (h_, n_)
        accts = sorted({p[0][0] for p in pairs})
        best_u, height = 1e9, 1
        for rep in range(4):
            delta = []
            for a in rnd.sample(accts, 512):
                delta.append(((a, 0), _fr(rnd.randrange(1, 1 << 30))))
                delta.append(((a, 4, 0, 1), _fr(rnd.randrange(1, 1 << 40))))
            height += 1
            t = time.perf_counter()
            h1, n1 = dev.update(delta, height)
            best_u = min(best_u, time.perf_counter() - t)
            live = dict((tuple(l), v) for l, v in pairs)
            live.update((tuple(l), v) for l, v in delta)
            pairs = list(live.items())
        assert (h1, n1) == ctx.state_compress(model, pairs)       # the incremental state == a fresh one-shot compress of everything
        prove_s = 1e9
        for rep in range(3):
            t = time.perf_counter()
            proofs = dev.prove((), accts[64 * rep:64 * rep + 64])
            prove_s = min(prove_s, time.perf_counter() - t)
        one = []
        for rep in range(3):
            a = accts[rep]
            t = time.perf_counter()
            dev.update([((a, 0), _fr(7 + rep))], height + 1 + rep)
            one.append(time.perf_counter() - t)
        out["state_device_incremental"] = {"what": "bzk_state_update on the device-resident state of the entry above: 512 accounts x (tx_nonce, one balance) = 1024 "
                                                   "writes per update, touched paths only (~12 k hashes); checked against a one-shot compress of the whole state",
                                           "ms": round(best_u * 1e3, 2), "load_24576_pairs_ms": round(load_s * 1e3, 2),
                                           "one_write_ms": round(min(one) * 1e3, 2),
                                           "latency_floor": "a write is a chain of 21 dependent hash levels (15 + 3 tree levels, 3 structs) at ~0.24 ms of device latency each",
                                           "prove_64_accounts_ms": round(prove_s * 1e3, 2), "proof_levels": len(proofs[0]), **dev.stats()}
        dev.close()
    except Exception as e:
        out.setdefault("state_compress_sparse_mpn", {"error": repr(e)})
        out["state_device_incremental"] = {"error": repr(e)}



def make_work_section(ctx, out=None):
    """validator-side work preparation (f-3); fills and returns out[...]"""
    out = {} if out is None else out
    # validator-side work preparation (f-3): 256 update transactions at the production shape, host walk vs device batches
    try:
        from bazuka_amd import lib as L_
        import json as _json
        vks = [bytes.fromhex(h) for h in _json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))["verifying_keys_bincode_hex"]]
        Z_ = _fr(1)
        res = {}
        for key, dev_on in (("host_s", False), ("device_s", True)):
            w_ = L_.MpnWorld(15, 3)
            if dev_on:
                w_.set_device(ctx)
            for i in range(512):
                w_.add_account((i * 7919 + 1) % 4 ** 15, b"a%d" % i, Z_, 10 ** 12)
            ts = []
            for k in range(2):
                for i in range(256):
                    w_.push_tx((i * 7919 + 1) % 4 ** 15, ((256 + i) * 7919 + 1) % 4 ** 15, Z_, 100 + i + k, Z_, i % 7)
                t = time.perf_counter()
                wk = w_.make_work(2, vks, 1, log4_batches=(1, 1, 4))
                ts.append(time.perf_counter() - t)
                if k == 0:
                    res[key + "_bytes"] = wk.encode()
            res[key] = round(min(ts), 4)
        out["mpn_make_work_256tx"] = {"what": "one update work of `prepare_works` (256 transactions, L = 15, T = 3): transitions with their Merkle proofs; "
                                              "host = per-transaction walk of the sparse tree, device = bzk_mpn_set_device (one batched Poseidon launch per level)",
                                      "host_s": res["host_s"], "device_s": res["device_s"], "same_work_bytes": res["host_s_bytes"] == res["device_s_bytes"]}
    except Exception as e:
        out["mpn_make_work_256tx"] = {"error": repr(e)}
    return out


def other_configs_section(ctx, dev):
    """BASELINE.json's other single-GPU configurations and the proof's remaining kernels, timed by the driver's own run
    (VERDICT r1 item 3): inputs resident in HBM, best of 3 after one warm-up, plus the HIP-event time of the dominant kernel and
    the algorithmic-byte roofline of SURVEY 8d.  Parity of each is the job of tests/test_gpu_fullsize.py."""
    import torch

    def rand_fr(cnt, seed):
        return uniform_fr_dev(cnt, seed, dev)

    def timeit(fn, reps=3):
        fn()
        ctx.sync()
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn()
            ctx.sync()
            best = min(best, time.perf_counter() - t)
        return best * 1e3

    def kernels(fn):
        ctx.prof_enable(True)
        ctx.prof_reset()
        fn()
        ctx.sync()
        d = {k: round(v[1], 4) for k, v in ctx.prof_dump().items() if v[1] > 0.02}
        ctx.prof_enable(False)
        return d

    def hbm(alg_bytes, ms):
        gbs = alg_bytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6)}

    out = {}
    # configs[4]: 2^24-leaf 4-ary Poseidon tree (5 592 405 arity-4 hashes); 32 B per leaf read + 32 B per node written
    n = 1 << 24
    leaves = rand_fr(n, 24)
    ms = timeit(lambda: ctx.merkle4_root_dev(leaves, 12))
    hashes = (n - 1) // 3
    out["tree_2p24"] = {"workload": "BASELINE configs[4]: 4-ary Poseidon re-hash of a 2^24-leaf ZkState tree", "ms": round(ms, 3),
                        "Mhash_per_s": round(hashes / ms / 1e3, 2), "roofline": hbm(32.0 * n + 32.0 * hashes, ms),
                        "alu": {"fr_products_per_hash_sparse": 1184, "mads_per_hash": 131000,
                                "achieved": round(hashes * 131000 / ms / 1e9, 2), "peak": MAD_PEAK_T, "unit": "T v_mad_u64_u32/s",
                                "frac": round(hashes * 131000 / ms / 1e9 / MAD_PEAK_T, 4),
                                "note": "integer-ALU bound: 9 x 29-bit Fr, sparse partial rounds; ~131 k multiply-adds per arity-4 hash "
                                        "(DESIGN 3.4) against the measured instruction ceiling"},
                        "kernel_ms": kernels(lambda: ctx.merkle4_root_dev(leaves, 12))}
    out["tree_2p24"]["roofline"]["traffic"], out["tree_2p24"]["roofline"]["traffic_source"] = _pmc_other("tree_2p24")
    del leaves
    # configs[4], secondary instance: the MPN-shaped state - 4^9 accounts x (64 H2 + 21 H4 + 1 H5) + the account tree
    n_acct, ts = 1 << 18, 64
    cells, toks = rand_fr(n_acct * 4, 9), rand_fr(n_acct * ts * 2, 93)
    ms = timeit(lambda: ctx.mpn_state_compress_dev(9, 3, cells, toks))
    hashes = n_acct * (ts + (ts - 1) // 3 + 1) + (n_acct - 1) // 3
    out["mpn_state_4p9_accounts"] = {"workload": "BASELINE configs[4] secondary: compress of a dense List{9, Struct{S,S,S,S,List{3,Struct{S,S}}}} state (262 144 accounts)",
                                     "ms": round(ms, 3), "Mhash_per_s": round(hashes / ms / 1e3, 2), "hashes": hashes,
                                     "roofline": hbm(32.0 * n_acct * (4 + 2 * ts) + 32.0 * hashes, ms)}
    del cells, toks
    # NTT (a4): 64 B per element per transform in one HBM round trip.  Timed: the plain forward transform (bellman `fft`) -
    # since round 3 the form every transform of the h chain takes (the coset scalings ride on the neighbouring transforms' stores);
    # the stand-alone coset transform (one more product per element on load) beside it
    for lg in (20, 24):
        d = rand_fr(1 << lg, lg)
        ms = timeit(lambda: ctx.ntt_dev(d, lg, False, False))
        ms_coset = timeit(lambda: ctx.ntt_dev(d, lg, False, True))
        out[f"ntt_2p{lg}"] = {"ms": round(ms, 4), "coset_ms": round(ms_coset, 4), "roofline": hbm(64.0 * (1 << lg), ms),
                              "alu": {"fr_products": (lg * (1 << lg)) // 2, "G_per_s": round(lg * (1 << lg) / 2 / ms / 1e6, 2), "peak": 162.9,
                                      "peak_source": "Fr29 product as dependent calls, profiles/r04_ubench_int.txt"}}
        if lg == 24:
            out["ntt_2p24"]["roofline"]["traffic"], out["ntt_2p24"]["roofline"]["traffic_source"] = _pmc_other("ntt_2p24")
        del d
    a, b, c = rand_fr(1 << 20, 1), rand_fr(1 << 20, 2), rand_fr(1 << 20, 3)
    ms = timeit(lambda: ctx.groth16_h_dev(a, b, c, 20))
    out["h_stage_2p20"] = {"what": "3 iNTT + 3 coset NTT + pointwise (a b - c) / Z + 1 inverse coset NTT (fused chain since round 3: coset scaling on the inverse transforms' stores, pointwise step on the last transform's load)",
                           "ms": round(ms, 3),
                           "roofline": hbm(7 * 64.0 * (1 << 20) + 128.0 * (1 << 20), ms)}
    del a, b, c
    state_seam_section(ctx, out)
    make_work_section(ctx, out)
    # G2 MSM (a6): 224 algorithmic bytes per (point, scalar) pair
    n = 1 << 20
    bases = torch.empty(n * 192, dtype=torch.uint8, device=dev)
    ctx.g2_synth_bases_dev(SEED, 0, n, bases)
    sc = rand_fr(n, 2020)
    ms = timeit(lambda: ctx.msm_g2_dev(bases, sc, n), reps=2)
    km = kernels(lambda: ctx.msm_g2_dev(bases, sc, n))
    acc = km.get("msm_accumulate", 0.0)
    sec = {"ms": round(ms, 3), "Mpt_per_s": round(n / ms / 1e3, 2), "kernel_ms": km}
    if acc:
        W = ctx.msm_window_count(n)
        sec["roofline"] = dict(hbm(224.0 * n, acc), kernel="msm_accumulate<G2>", avg_launch_ms=round(acc, 4))
        sec["roofline"]["traffic"], sec["roofline"]["traffic_source"] = _pmc_other("msm_accumulate_g2")
        g = n * W * FP_MULS_PER_G2_MIXED_ADD / (acc * 1e-3) / 1e9
        sec["alu"] = {"achieved": round(g, 2), "peak": 70.2, "unit": "G Fp-mul/s", "frac": round(g / 70.2, 4),
                      "peak_source": "library product at 2 waves/SIMD (the pair-lane G2 kernel's occupancy since round 5; 60.1 at the one-lane kernel's 1 wave), profiles/r04_ubench_int.txt"}
    out["msm_g2_2p20"] = sec
    # static bases (a Groth16 CRS is static): full table at c = 20 - 13 windows sharing one bucket set, no host Horner.  Informational:
    # `value` of the headline stays the per-call pipeline on raw bases
    gb = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    ctx.g1_synth_bases_dev(SEED, 0, n, gb)
    tab = ctx.msm_table_build_c(gb, n, 20)
    ms = timeit(lambda: ctx.msm_table_run_dev(tab, sc, n))
    same = ctx.msm_table_run_dev(tab, sc, n) == ctx.msm_g1_dev(gb, sc, n)
    out["msm_g1_2p20_static_table"] = {"what": "2^20-point G1 MSM over a precomputed table 2^(20 w) P_i (13 levels, 1.5 GB): 13 additions per point instead of 16",
                                       "ms": round(ms, 3), "Mpt_per_s": round(n / ms / 1e3, 2), "same_bytes_as_per_call_pipeline": same,
                                       "kernel_ms": kernels(lambda: ctx.msm_table_run_dev(tab, sc, n))}
    ctx.msm_table_free(tab)
    del gb, sc
    # mid-size stand-alone calls (round 6, runs 21 - 23): below 0.66 M points the window is narrower than 16 bits and the top window of the signed recoding is
    # degenerate - at 2^19 points (c = 15) its bucket 0 receives 45 % of all scalars.  Such giant buckets are folded in two levels (msm_fold_wide_kernel) and the
    # call runs as two window ranges in flight (msm_run_split): 4.1 ms -> 2.25 ms at 2^19 points, 2.07 -> 1.73 ms at 2^18 (profiles/r06_run22_23_giant_bucket_fold.txt)
    for lg in (19, 18):
        n = 1 << lg
        gb = torch.empty(n * 96, dtype=torch.uint8, device=dev)
        ctx.g1_synth_bases_dev(SEED, 0, n, gb)
        sc = rand_fr(n, 1900 + lg)
        hb = ctx.msm_bases_load_dev(gb, n)
        ms = timeit(lambda: ctx.msm_bases_run_dev(hb, sc, n), reps=5)
        same = ctx.msm_bases_run_dev(hb, sc, n) == ctx.msm_g1_dev(gb, sc, n)
        out[f"msm_g1_2p{lg}"] = {"what": f"2^{lg}-point G1 MSM over a resident base set ({ctx.msm_window_count(n)} windows; two window ranges in flight, giant buckets folded in two levels)",
                                 "ms": round(ms, 3), "Mpt_per_s": round(n / ms / 1e3, 2), "same_bytes_as_per_call_pipeline": same,
                                 "kernel_ms": kernels(lambda: ctx.msm_bases_run_dev(hb, sc, n))}
        ctx.msm_bases_free(hb)
        del gb, sc
    # the same per-call pipeline at 2^24 points (bases 1.9 GB in internal form: beyond the Infinity Cache; 32 entries per bucket -> 512):
    # where the fixed cost of the bucket reduction and the window sums is 3 % of the call instead of 23 %
    n = 1 << 24
    gb = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    ctx.g1_synth_bases_dev(SEED, 0, n, gb)
    sc = rand_fr(n, 17)
    ms = timeit(lambda: ctx.msm_g1_dev(gb, sc, n), reps=2)
    k = kernels(lambda: ctx.msm_g1_dev(gb, sc, n))
    acc = k.get("msm_accumulate", 0.0)
    sec = {"ms": round(ms, 3), "Mpt_per_s": round(n / ms / 1e3, 2), "kernel_ms": k}
    if acc:
        sec["roofline"] = {"bound": "hbm", "achieved": round(128.0 * n / (acc * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(128.0 * n / (acc * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "kernel": "msm_accumulate", "avg_launch_ms": round(acc, 4)}
        tmad = n * 16 * MADS_PER_MIXED_ADD / (acc * 1e-3) / 1e12
        sec["alu_mad"] = {"achieved": round(tmad, 2), "peak": MAD_PEAK_T, "unit": "T v_mad_u64_u32/s", "frac": round(tmad / MAD_PEAK_T, 4)}
    out["msm_g1_2p24"] = sec
    return out


def production_block_section(ctx, with_1024tx: bool = False):
    """The proofs a real node accepts (row g): one Deposit(15,3,3), one Withdraw(15,3,3) and one Update(15,3,4) work per block
    (/root/reference/src/config/blockchain.rs:22-26, 326-328) - 64 / 64 / 256 transitions, 1.39 M / 2.35 M / 14.4 M constraints, 2^21 / 2^22 /
    2^24 domains - with FULL batches: validator side (`bzk_mpn_make_work`, Merkle hashing batched on the GPU), wire bytes, worker side
    (decode, witness synthesis on the host threads, Groth16 proof on the GPU), and the node's acceptance check (`bzk_groth16_verify`,
    src/zk/groth16/mod.rs:67-121) on every timed proof.  CRS: generated on the GPU from the circuit's own matrices (untimed, reported).
    Byte parity of these shapes with the oracle prover is tests/test_gpu_production.py's job."""
    from bazuka_amd import lib as L
    vks = [bytes.fromhex(h) for h in json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))["verifying_keys_bincode_hex"]]
    Z, TOK, prover = _fr(1), _fr(777), bytes(range(1, 33))
    size = 4 ** 15
    shapes = [("deposit", 0, 3), ("withdraw", 1, 3), ("update", 2, 4)]
    if with_1024tx:
        shapes.append(("update_1024tx_single", 2, 5))
    out, total = {}, 0.0
    for name, kind, b4 in shapes:
        n_slots = 4 ** b4
        w = L.MpnWorld(15, 3)
        w.set_device(ctx)
        idx = [(i * 22369621 + 5) % size for i in range(2 * n_slots)]
        for i, a in enumerate(idx):
            w.add_account(a, b"blk%d" % i, Z, 10 ** 12)
        w.set_height(7)
        state = {"k": 0}

        def push():
            state["k"] += 1
            k = state["k"]
            for i in range(n_slots):
                if kind == 0:
                    w.push_deposit(idx[(i + k) % len(idx)], TOK if i % 3 == 1 else Z, 1000 + i + k)
                elif kind == 1:
                    w.push_withdraw(idx[i], Z, 400 + i + k, Z, i % 4)
                else:
                    w.push_tx(idx[i], idx[n_slots + i], Z, 100 + i + k, Z, i % 7)

        lb = [1, 1, 1]
        lb[kind] = b4
        sec = {"circuit": f"{('Deposit', 'Withdraw', 'Update')[kind]}Circuit(L=15,T=3,B={b4}): {n_slots} transitions"}
        push()
        t0 = time.perf_counter()
        dec = L.MpnWork.decode(w.make_work(kind, vks, 1, log4_batches=tuple(lb)).encode())
        r = dec.synthesize(prover, record_matrices=True)
        assert r.satisfied and r.accepted == n_slots, (name, r.accepted, r.first_unsatisfied)
        sec.update(n_constraints=r.n_constraints, n_aux=r.n_aux, log_domain=(r.n_constraints - 1).bit_length(),
                   synthesize_with_matrices_s=round(time.perf_counter() - t0, 2))
        csr = [(r.n_constraints, r.raw("rp" + x), r.raw("col" + x), r.raw("val" + x)) for x in "ABC"]
        t0 = time.perf_counter()
        ph, vkb = ctx.groth16_setup(csr, r.n_in, r.n_aux, b"".join(_fr(x) for x in (1234567, 2345678, 3456789, 4567891, 5678912)))
        sec["gpu_crs_setup_s"] = round(time.perf_counter() - t0, 2)
        del csr
        r.free()
        import resource as _rs

        def _cpu():
            ru = _rs.getrusage(_rs.RUSAGE_SELF)
            return ru.ru_utime + ru.ru_stime

        tm, tw, tp, tv, ok, twc = [], [], [], [], True, []
        for k in range(3):
            push()
            t0 = time.perf_counter()
            blob = w.make_work(kind, vks, 1, log4_batches=tuple(lb)).encode()
            t1 = time.perf_counter()
            c1 = _cpu()
            dec = L.MpnWork.decode(blob)
            rk = dec.synthesize(prover)
            t2 = time.perf_counter()
            twc.append(_cpu() - c1)
            assert rk.satisfied and rk.accepted == n_slots
            z = rk.raw("z")
            proof = ctx.groth16_prove(ph, z, rk.raw("az"), rk.raw("bz"), rk.raw("cz"), _fr_blind(5000 + 2 * k), _fr_blind(5001 + 2 * k))
            t3 = time.perf_counter()
            ok = ok and L.groth16_verify(vkb, bytes(z[32:32 * 6]), proof) and not L.groth16_verify(vkb, bytes(z[32:32 * 5]) + _fr(12345), proof)
            t4 = time.perf_counter()
            tm.append(t1 - t0); tw.append(t2 - t1); tp.append(t3 - t2); tv.append((t4 - t3) / 2)   # two verifications (accept, refuse)
            rk.free()
        # decode_and_witness_cpu_s: user + system seconds of the process around decode + synthesize (all generator threads) - the WORKER side's host cost per work
        sec.update(make_work_s=round(min(tm), 4), wire_bytes=len(blob), decode_and_witness_s=round(min(tw), 4), decode_and_witness_cpu_s=round(min(twc), 4),
                   prove_s=round(min(tp), 4),
                   prove_s_all=[round(x, 4) for x in tp], verified=bool(ok), verify_ms_host=round(min(tv) * 1e3, 2),
                   tx_per_s_prove_only=round(n_slots / min(tp), 1))
        if True:
            # the same work with the hash-dependent witness values left to the device: at 64 / 256 transitions the deferred-value program has that
            # many workgroups to run, and the host generator is what a production block waits for longest after the proof itself
            twd, tpd, okd, twdc = [], [], True, []
            for k in range(2):
                t1 = time.perf_counter()
                c1 = _cpu()
                rd = L.MpnWork.decode(blob).synthesize(prover, defer=True)
                t2 = time.perf_counter()
                twdc.append(_cpu() - c1)
                pd = ctx.groth16_prove_r1cs(ph, rd, _fr_blind(6000 + 2 * k), _fr_blind(6001 + 2 * k))
                t3 = time.perf_counter()
                okd = okd and rd.defer_info()["deferred"] == 1 and L.groth16_verify(vkb, bytes(rd.raw("z")[32:32 * 6]), pd)
                twd.append(t2 - t1); tpd.append(t3 - t2)
                rd.free()
            sec["deferred"] = {"decode_and_witness_s": round(min(twd), 4), "decode_and_witness_cpu_s": round(min(twdc), 4), "prove_s": round(min(tpd), 4),
                               "verified": bool(okd)}
        if b4 <= 4:
            total += min(tp)
        ctx.params_free(ph)
        w.close()
        out[name] = sec
    out["prove_s_total"] = round(total, 4)
    out["what"] = ("one block's three MPN works at the chain's parameters (src/config/blockchain.rs:22-26): make_work with the device builder, "
                   "decode + witness on the host, proof on the GPU, bzk_groth16_verify accepts / rejects a wrong input; best of 3 full batches")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="log2 points per GPU (default 20 = BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-proofs", action="store_true", help="skip the full Groth16 proofs/s section (N=1 only)")
    ap.add_argument("--no-overlap", action="store_true", help="skip the informational two-MSMs-in-flight measurement (kernel traces)")
    ap.add_argument("--no-others", action="store_true", help="skip the tree / NTT / h-stage / G2 sections (N=1 only)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): 2^log-n points PER GPU; strong: 2^log-n-total points for the whole job (SURVEY 8d C4 reading (i))")
    ap.add_argument("--log-n-total", type=int, default=24, help="log2 points of the whole job in --scaling strong (24 or 26)")
    ap.add_argument("--partition", choices=("windows", "points"), default="windows",
                    help="N > 1: windows = the north star's scalar-window ranges over all points (default); points = every rank runs "
                         "all windows over its own slice of the points (no rank converts or recodes another rank's points); same "
                         "single all-gather + fold, same result")
    ap.add_argument("--no-production", action="store_true", help="skip other_configs.production_block (the chain's three real circuit shapes)")
    ap.add_argument("--with-1024tx", action="store_true",
                    help="production_block also proves BASELINE configs[2] at face value: ONE 1024-tx Update circuit (57.8 M constraints, 2^26 domain)")
    ap.add_argument("--prover-helper", action="store_true",
                    help="internal: the second prover process of proofs.two_processes (proves until told to stop on stdin; prints no bench line)")
    ap.add_argument("--proofs-child", type=int, default=-1,
                    help="internal: run the proofs section on this device in a process of its own and print it as one BZK_PROOFS_JSON line (see proofs_in_child)")
    args = ap.parse_args()

    if args.proofs_child >= 0:
        import torch
        from bazuka_amd import Bzk
        if os.environ.get("BZK_BENCH_DRYRUN_BACKEND"):
            args.proofs_child %= torch.cuda.device_count()
        torch.cuda.set_device(args.proofs_child)
        world_c = args.gpus
        if "BZK_SYNC_BLOCKING" not in os.environ and quota_binds(cpu_quota(), world_c):
            os.environ["BZK_SYNC_BLOCKING"] = "1"
        cctx = Bzk(args.proofs_child)
        n_prod, pt = host_thread_budget(world_c)
        n_prod = int(os.environ.get("BZK_BENCH_PRODUCERS", str(n_prod)))
        pt = int(os.environ.get("BZK_BENCH_PROD_THREADS", str(pt)))
        try:
            res = full_prove_section(cctx, n_prod=n_prod, prod_threads=pt, cpu_baseline=(world_c == 1 and not args.no_cpu_baseline),
                                     second_process=(world_c == 1 and os.environ.get("BZK_BENCH_TWO_PROCS", "1") != "0"), world=world_c)
        except Exception as e:
            res = {"error": repr(e)}
        cctx.close()
        print("BZK_PROOFS_JSON " + json.dumps(res), flush=True)
        return

    if args.prover_helper:
        import torch
        from bazuka_amd import Bzk
        dev_i = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev_i)
        if "BZK_SYNC_BLOCKING" not in os.environ and quota_binds(cpu_quota()):
            os.environ["BZK_SYNC_BLOCKING"] = "1"
        hctx = Bzk(dev_i)
        full_prove_section(hctx, n_prod=int(os.environ.get("BZK_BENCH_PRODUCERS", "8")), prod_threads=int(os.environ.get("BZK_BENCH_PROD_THREADS", "8")),
                           helper=True)
        hctx.close()
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher (how the driver may call it): become the launcher - one process per GPU under
        # torch.distributed.run on 127.0.0.1 with a free port; stdout / exit code are the ranks' own (exec, no wrapper process)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execve(sys.executable, cmd, env)

    # The proofs child runs BEFORE this process initialises the HIP runtime: two processes with 16 hardware queues each on one GPU are time-sliced by the
    # scheduler even when one of them idles (run 40: 69.7 proofs/s in a child beside the idle parent against 78.1 alone).
    early_proofs = None
    if (not args.no_proofs and os.environ.get("BZK_BENCH_PROOFS_PROCESS", "1") != "0" and not os.environ.get("BZK_BENCH_SPAWN_ONLY")
            and int(os.environ.get("WORLD_SIZE", "1")) == args.gpus):
        try:
            import subprocess
            cenv = dict(os.environ, AMD_DIRECT_DISPATCH=os.environ.get("BZK_BENCH_PROOFS_DISPATCH", "0"))
            cmd = [sys.executable, os.path.abspath(__file__), "--proofs-child", os.environ.get("LOCAL_RANK", "0"), "--gpus", str(args.gpus)] + \
                  (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
            cp = subprocess.run(cmd, env=cenv, capture_output=True, text=True, timeout=600)  # ~75 s when all is well; after that the section runs in this process
            for line in reversed(cp.stdout.splitlines()):
                if line.startswith("BZK_PROOFS_JSON "):
                    early_proofs = json.loads(line[len("BZK_PROOFS_JSON "):])
                    break
            if early_proofs is None or "error" in early_proofs:
                print(f"[bench] proofs child failed (rc {cp.returncode}): {(early_proofs or {}).get('error')} {cp.stderr[-400:]}", file=sys.stderr, flush=True)
                early_proofs = None
            else:
                early_proofs["process"] = ("a process of its own, run before this one touched the GPU, with AMD_DIRECT_DISPATCH=" + cenv["AMD_DIRECT_DISPATCH"] +
                                           " (launches through the runtime's per-stream worker threads: prover-side host CPU per proof - 58 %, pipelined rate + 4 %, same-box "
                                           "A/B profiles/r06_run37_39_host_cpu_of_the_prover.txt); the MSM half of this line runs with the runtime's default")
        except Exception as e:
            print(f"[bench] proofs child: {e!r}", file=sys.stderr, flush=True)
            early_proofs = None

    import torch
    import torch.distributed as dist
    from bazuka_amd import Bzk, Mg, mg_unique_id

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("BZK_BENCH_SPAWN_ONLY"):
        # CPU-side check of the launch path (tests/test_bench_spawn_cpu.py): the ranks rendezvous over gloo, pass a 128-byte group id
        # from rank 0 to everybody exactly as the GPU run does, report it and leave - no device is touched
        assert world == args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks"
        if world > 1:
            dist.init_process_group("gloo")
            box = [bytes(range(128)) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            dist.barrier()
            print(json.dumps({"spawn_only": True, "rank": rank, "world": world, "uid_ok": box[0] == bytes(range(128))}), flush=True)
            dist.destroy_process_group()
        else:
            print(json.dumps({"spawn_only": True, "rank": 0, "world": 1, "uid_ok": True}), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU: libbzk has no CPU path"
    # BZK_BENCH_DRYRUN_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks
    # share devices, the 97-byte exchange goes through gloo on the host).  Never set by the driver; the line
    # printed in that mode is marked "dryrun" and is not a measurement.
    dry = os.environ.get("BZK_BENCH_DRYRUN_BACKEND")
    if dry:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if not args.no_proofs and "BZK_SYNC_BLOCKING" not in os.environ and quota_binds(cpu_quota(), world):
        os.environ["BZK_SYNC_BLOCKING"] = "1"  # read by libbzk when the first context is created (full_prove_section explains)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dry:
            dist.init_process_group(dry)
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks"
    if world > 1 and not dry:
        assert torch.cuda.device_count() >= world, (f"{world} ranks but {torch.cuda.device_count()} GPUs visible "
                                                    "(BZK_BENCH_DRYRUN_BACKEND=gloo rehearses the code path on fewer GPUs)")

    ctx = Bzk(local_rank, torch.cuda.current_stream().cuda_stream)
    # whole-job points; every rank holds all of them (the CRS is static and replicated).  weak: per-GPU work constant in N;
    # strong: the job is fixed (2^24 / 2^26 points) and a rank's share of the windows shrinks with N
    n = (1 << args.log_n_total) if args.scaling == "strong" else (1 << args.log_n) * world
    bases = torch.empty(n * 96, dtype=torch.uint8, device=dev)
    ctx.g1_synth_bases_dev(SEED, 0, n, bases)
    scalars = uniform_fr_dev(n, SEED, dev)  # uniform in [0, r) by rejection, as SURVEY 8d prescribes
    torch.cuda.synchronize()

    from bazuka_amd.dist import allgather_fold, window_range
    by_points = world > 1 and args.partition == "points"
    p_lo, p_hi = (n * rank // world, n * (rank + 1) // world) if by_points else (0, n)
    n_rank = p_hi - p_lo                      # points this rank's launches touch
    W = ctx.msm_window_count(n_rank if by_points else n)
    w0, w1 = (0, W) if by_points else window_range(W, rank, world)
    bases_rank, scalars_rank = bases[96 * p_lo:96 * p_hi], scalars[p_lo:p_hi]

    # The static base set is converted ONCE to the internal limb form and stays in HBM (the CRS of a prover is loaded once): outside the
    # timed region, like the upload itself.  N > 1: the device group behind the C ABI (bzk_mg_create_rank; one process per GPU).  Its
    # 128-byte group id travels over the launcher's process group - the only thing torch.distributed carries besides the barriers.
    mg = mg_bases = None
    pctx = ctx  # the context whose launches carry the HIP events of the per-kernel tables
    if world == 1:
        rbases = ctx.msm_bases_load_dev(bases, n)
    elif not by_points:
        # Building the group must never hang the job: capability vote over a CPU-side gloo group before RCCL is entered,
        # creation on a helper thread with a bounded wait, fall-back to the shared-memory transport on all ranks together
        # (bazuka_amd/dist.py::build_device_group; tests/test_dist_cpu.py runs the protocol on gloo with stub groups)
        from bazuka_amd import mg_probe
        from bazuka_amd.dist import build_device_group
        vote_pg = None if dry else dist.new_group(backend="gloo")

        def new_uid():
            box = [mg_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return box[0]

        # ranks sharing a GPU (rehearsal) use the shared-memory transport outright; otherwise AUTO (RCCL over xGMI)
        want_x = 1 if dry else int(os.environ.get("BZK_BENCH_MG_EXCHANGE", "0"))
        mg, _, _ = build_device_group(lambda uid, x: Mg(device=local_rank, rank=rank, world=world, uid=uid, exchange=x), mg_probe(local_rank), rank,
                                      want_x, float(os.environ.get("BZK_MG_TIMEOUT_S", "120")), vote_group=vote_pg, new_uid=new_uid,
                                      log=lambda m: print(f"[bench] {m}", file=sys.stderr, flush=True))
        mg_bases = mg.bases_load_dev([bases], n)
        pctx = Bzk(local_rank, handle=mg.ctx_handle(0))

    def step():
        if world == 1:
            return ctx.msm_bases_run_dev(rbases, scalars, n)
        if by_points:   # the MSM is linear in its points: any partition of them folds to the same element (torch-side exchange)
            part = ctx.msm_g1_dev(bases_rank, scalars_rank, n_rank)
            return allgather_fold(part, device=None if dry else dev)
        return mg.msm_dev(mg_bases, [scalars], n)   # this rank's windows + ONE all-gather of window sums + host Horner, all in libbzk

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    result = None
    for _ in range(args.warmup):
        result = step()
    # HIP events on the ctx stream, inside the timed region, around the DOMINANT kernel only (two events per step): timing all
    # ~25 launches of a step costs ~0.25 ms of event creation per step, i.e. it would be measured into `value`
    pctx.prof_filter("msm_accumulate")
    pctx.prof_enable(True)
    pctx.prof_reset()
    if mg:
        mg.stats(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    pctx.prof_enable(False)
    prof = pctx.prof_dump()
    # the per-kernel breakdown comes from extra, untimed steps with every launch instrumented
    pctx.prof_filter(None)
    pctx.prof_enable(True)
    pctx.prof_reset()
    n_break = min(5, args.steps)
    for _ in range(n_break):
        step()
    fence()
    pctx.prof_enable(False)
    prof_all = pctx.prof_dump()
    elapsed_rank = elapsed
    acc_n_rank, acc_ms_rank = prof.get("msm_accumulate", (0, 0.0))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must hold the same folded result
        r = torch.frombuffer(bytearray(result), dtype=torch.uint8).to(dev)
        r0 = r.clone()
        dist.broadcast(r0, 0)
        assert bool((r == r0).all()), "ranks disagree on the MSM result"
        # untimed: the window-sharded + folded result equals the unsharded MSM of the same points on one GPU
        if rank == 0:
            assert ctx.msm_g1_dev(bases, scalars, n) == result, "window-sharded MSM differs from the single-GPU MSM"
    elif args.log_n <= 22:  # untimed: the resident-set call equals the per-call pipeline on the raw bases
        assert ctx.msm_g1_dev(bases, scalars, n) == result, "resident-base MSM differs from the raw-base MSM"

    acc_n, acc_ms = prof.get("msm_accumulate", (0, 0.0))
    # Informational (N = 1): the same K MSMs issued from two host threads on two contexts (own stream + workspace each) -
    # what a prover with independent MSMs in flight sees: the latency-bound tail of one MSM (bucket reduction, window
    # sums, read-back, ~1.4 ms) overlaps the accumulation of the other.  `value` above stays the one-at-a-time rate.
    overlapped = None
    if world == 1 and not args.no_overlap:
        # Informational (N = 1): the same MSM issued from 2 and 4 host threads, each on a context of its own (stream + workspace), over the SAME resident
        # base set - what a prover with independent MSMs in flight sees (round 5 measured this over RAW bases, i.e. with a conversion per call: +1.8 %; the
        # resident-set figure is the comparable one).  `value` above stays the one-at-a-time rate.
        import threading
        ctxs = [Bzk(local_rank) for _ in range(4)]
        for c in ctxs:
            for _ in range(max(1, args.warmup)):
                assert c.msm_bases_run_dev(rbases, scalars, n) == result
        torch.cuda.synchronize()
        overlapped = {"how": "K independent MSMs in flight (K contexts / streams / host threads over one resident base set); informational, not `value`",
                      "unit": "Mpt/s"}
        for k_in in (2, 4):
            outs = [None] * k_in
            per = (args.steps + k_in - 1) // k_in

            def run(i):
                for _ in range(per):
                    outs[i] = ctxs[i].msm_bases_run_dev(rbases, scalars, n)

            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                th = [threading.Thread(target=run, args=(i,)) for i in range(k_in)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            assert all(o == result for o in outs), "overlapped MSMs differ from the one-at-a-time result"
            overlapped[f"in_flight_{k_in}"] = {"value": round(n * k_in * per / best / 1e6, 3), "msms": k_in * per, "ms_per_msm": round(best * 1e3 / (k_in * per), 4)}
        overlapped["value"] = overlapped["in_flight_2"]["value"]
        for c in ctxs:
            c.close()
    ms_per_step = elapsed * 1e3 / args.steps
    value = n / (elapsed / args.steps) / 1e6
    out = {
        "metric": "G1-MSM throughput (BLS12-381 Pippenger, 2^20 points per GPU)",
        "value": round(value, 3),
        "unit": "Mpt/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[1]: 2^{args.log_n}-point BLS12-381 G1 Pippenger MSM per GPU "
                                f"(bases k_i*G, uniform scalars, resident in HBM)") if args.scaling == "weak" else
                               (f"BASELINE configs[3] reading (i): ONE 2^{args.log_n_total}-point G1 MSM for the whole job, "
                                f"{'point' if by_points else 'window'}-sharded over the ranks (bases k_i*G, uniform scalars, replicated in HBM)"),
                   "scalars": "uniform in [0, r) by rejection sampling (255-bit candidates, rows >= r redrawn), 32-byte Montgomery limbs; seed 0x42415A554B41",
                   "points_total": n, "windows": W, "window_range_this_rank": [w0, w1], "point_range_this_rank": [p_lo, p_hi],
                   "bases": "resident set in the internal 112-byte form (converted once at load, outside the timed region)",
                   "parallelism": "single-gpu" if world == 1 else
                                  f"{'point' if by_points else 'window'}-sharded x{world} + one all-gather of window sums per MSM"},
        "proofs_per_sec": None,
    }
    if world > 1:  # what the collective layer actually saw
        assert dist.get_world_size() == args.gpus, f"collective world size {dist.get_world_size()} != --gpus {args.gpus}"
        # every rank's view of the timed steps, so that the first hardware SCALE run diagnoses itself: a slow rank shows as
        # a long local stage there and as exchange / waiting time on the others; the transport is the one actually in use, not the one asked for
        st_mg = mg.stats() if mg else None
        calls = max(1, st_mg["calls"]) if st_mg else 1
        mine_diag = {"rank": rank, "device": local_rank, "windows": [w0, w1],
                     "msm_accumulate_ms": round(acc_ms_rank / max(1, acc_n_rank), 4) if acc_n_rank else None,
                     "step_ms_this_rank": round(elapsed_rank * 1e3 / args.steps, 4),
                     "transport": ({0: "auto", 1: "host/shm", 2: "peer", 3: "rccl"}.get(mg.exchange, str(mg.exchange)) if mg else "torch.distributed"),
                     "local_stage_ms": round(st_mg["local_ms"] / calls, 4) if st_mg else None,
                     "exchange_ms": round(st_mg["exchange_ms"] / calls, 4) if st_mg else None,
                     "peer_wait_ms": round(st_mg["peer_wait_ms"] / calls, 4) if st_mg else None,
                     "host_combine_ms": round(st_mg["combine_ms"] / calls, 4) if st_mg else None,
                     "group_create_s": round(st_mg["create_s"], 3) if st_mg else None,
                     "comm_init_s": round(st_mg["comm_init_s"], 3) if st_mg else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_diag)
        out["collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "per_rank": per_rank,
                             "launcher_role": "rendezvous, barriers and the 128-byte group id only",
                             "data_path": (f"libbzk bzk_mg_* (C ABI): transport {mg.exchange}, group of {mg.world}; one all-gather per MSM of what the ranks leave per "
                                           f"window - the 9 terms (192 B each) of the window's bucket set, multiplication-free reduction - for the {W} windows, Horner "
                                           "combine on the host") if mg else
                                          "torch.distributed all-gather of 97-byte partial sums + bzk_g1_sum (--partition points)"}
    if dry:
        out["dryrun"] = (f"ranks share GPUs (rendezvous over {dry}; window sums exchanged through libbzk's shared-memory transport, RCCL refuses "
                         "two ranks on one device): a rehearsal of the code path, NOT a measurement")
    # the other single-GPU configurations: before the proofs section (which may switch the process to blocking host waits)
    others = None
    if world == 1 and rank == 0 and not args.no_others:
        try:
            others = other_configs_section(ctx, dev)
        except Exception as e:  # the headline line must still be printed
            others = {"error": repr(e)}
        if not args.no_production:
            try:
                torch.cuda.empty_cache()
                others["production_block"] = production_block_section(ctx, with_1024tx=args.with_1024tx)
            except Exception as e:
                others["production_block"] = {"error": repr(e)}
    # Second half of the metric: full Groth16 proofs/s.  Every rank proves its own batches (replicas).
    proofs, rates, rates_live, host_cpu = None, [], [], []
    if not args.no_proofs:
        try:
            # N ranks share the host: split its cores between the ranks' witness producers
            # 8 producers x 8 worker threads measured best of the producer / thread sweeps (profiles/r02_run34_35_producer_sweep.txt:
            # 49.4 - 50.4 proofs/s against 45.8 - 45.9 for 6 x 16 on the same boxes; the GPU-side ceiling there was 54.7 - 55.0)
            # N ranks share the host's CPU quota: a rank's producers get quota / world of it (host_thread_budget)
            n_prod, pt = host_thread_budget(world)
            n_prod = int(os.environ.get("BZK_BENCH_PRODUCERS", str(n_prod)))
            pt = int(os.environ.get("BZK_BENCH_PROD_THREADS", str(pt)))
            # Round 6, runs 37 - 39: the proofs section runs in a PROCESS OF ITS OWN whose HIP runtime hands launches to its per-stream worker threads
            # (AMD_DIRECT_DISPATCH=0, read when the runtime initialises - hence a process).  Measured: with direct dispatch (the runtime's default) the
            # runtime's helper threads cost a proof 0.013 CPU-s, two thirds of it system time, on top of the 0.004 of libbzk's own threads; through the worker
            # threads the prover side costs 0.0065 - 0.0084 CPU-s per proof instead of 0.0174 - 0.0202 and, under the box's CPU quota, the pipelined rate with
            # live producers goes 75.2 -> 78.1 proofs/s - while a stand-alone MSM call, which waits for each of its few launches, is 2 % SLOWER that way
            # (3.47 - 3.51 -> 3.54 - 3.59 ms), so the MSM half of this line keeps the default (profiles/r06_run37_39_host_cpu_of_the_prover.txt).  It is what a
            # proving service does (bzk-worker / worker.py set the variable themselves); BZK_BENCH_PROOFS_PROCESS=0 measures in this process as before.
            proofs = early_proofs  # measured before this process touched the GPU (see proofs_in_child above), or None
            if proofs is None:
                proofs = full_prove_section(ctx, n_prod=n_prod, prod_threads=pt, cpu_baseline=(world == 1 and not args.no_cpu_baseline),
                                            second_process=(world == 1 and os.environ.get("BZK_BENCH_TWO_PROCS", "1") != "0"), world=world)
                proofs["process"] = "this process (the runtime's default dispatch mode)"
        except Exception as e:  # the headline MSM line must still be printed
            proofs = {"error": repr(e)}
        if world > 1:
            mine = torch.tensor([float(proofs.get("proofs_per_s_ring", float("nan"))), float(proofs.get("proofs_per_s_pipelined", float("nan"))),
                                 # what a witness costs THIS rank's live producers: the deferred generator's figure where they use it (the N > 1 default)
                                 float((proofs.get("deferred") or {}).get("witness_cpu_s", float("nan")) if (proofs.get("deferred") or {}).get("live_producers_use_it")
                                       else proofs.get("witness_cpu_s", float("nan"))),
                                 float(proofs.get("prover_host_cpu_s_per_proof", float("nan")))],
                                dtype=torch.float64, device="cpu" if dry else dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            rates = [float(x[0].item()) for x in allr]
            rates_live = [float(x[1].item()) for x in allr]
            host_cpu = [(float(x[2].item()), float(x[3].item())) for x in allr]
    if rank == 0:
        if acc_n:
            per_launch_ms = acc_ms / acc_n
            # since round 6 run 18 a stand-alone call runs its windows as several ranges in flight (msm_run_split): `launches_per_step` accumulation
            # launches per MSM, each over ALL points but only its share of the windows - a launch is credited that share of the MSM's algorithmic bytes
            # (and of its additions below); the launches of one step overlap each other's front chains and reductions, so a launch's own duration
            # includes the issue slots it gave away
            launches_per_step = max(1, round(acc_n / args.steps))
            alg_bytes = 128.0 * n_rank / launches_per_step  # 96 B affine base + 32 B scalar per (point, scalar) pair of this rank's MSM
            achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "msm_accumulate", "achieved": round(achieved, 3),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                               "traffic": _pmc_traffic(args.log_n)[0], "traffic_source": _pmc_traffic(args.log_n)[1],
                               "avg_launch_ms": round(per_launch_ms, 4), "launches_per_step": launches_per_step,
                               "accumulate_ms_per_step": round(acc_ms / args.steps, 4),
                               "note": "integer-ALU bound (381-bit Montgomery carry chains); HBM fraction is "
                                       "structurally ~1e-3, see DESIGN.md"}
            pairs = n_rank * (w1 - w0) / launches_per_step  # one mixed add per (point, window) pair (zero digits skipped: ~2^-16 of them)
            gmul = pairs * MULS_PER_MIXED_ADD / (per_launch_ms * 1e-3) / 1e9
            out["roofline"]["alu"] = {"achieved": round(gmul, 2), "peak": FP_MUL_PEAK_G, "unit": "G Fp-mul/s",
                                      "frac": round(gmul / FP_MUL_PEAK_G, 4),
                                      "peak_source": "tools/ubench_int.hip, library product as dependent calls (profiles/r04_ubench_int.txt: 76.2 G/s, as in round 1)"}
            tmad = pairs * MADS_PER_MIXED_ADD / (per_launch_ms * 1e-3) / 1e12
            out["roofline"]["alu_mad"] = {"achieved": round(tmad, 2), "peak": MAD_PEAK_T, "unit": "T v_mad_u64_u32/s",
                                          "frac": round(tmad / MAD_PEAK_T, 4),
                                          "how": f"{MADS_PER_MIXED_ADD} mads issued per mixed add (6 x 394 + 2 x 301 + 590 for the fused Y) against the "
                                                 "micro-benchmarked instruction ceiling (profiles/r04_ubench_int.txt)"}
        out["kernel_ms_per_step"] = {k: round(v[1] / n_break, 4) for k, v in sorted(prof_all.items())}
        out["kernel_ms_per_step_how"] = f"{n_break} extra untimed steps with every launch instrumented (HIP events)"
        if overlapped:
            out["two_msms_in_flight"] = overlapped
        if world == 1 and not args.no_cpu_baseline:
            from oracle import coracle as co
            # threads = the CPUs this container may actually consume (the pool's boxes show 256 logical CPUs inside a 16-CPU quota; 256
            # threads on it were 16 x oversubscribed - VERDICT r3 weak 8); the all-threads figure is reported beside it
            cores_all = co.ncpu()
            quota_now = cpu_quota()
            cores = max(1, min(cores_all, int(quota_now + 0.5))) if quota_now else cores_all
            hb = bytes(bases.cpu().numpy().tobytes())
            hs = bytes(scalars.cpu().numpy().tobytes())
            # SURVEY 8d protocol: one warm-up, then the median of >= 5 runs (bounded to ~30 s of wall time)
            want = co.msm_g1(hb, hs, nthreads=cores)
            assert want == result, "GPU MSM result differs from the CPU oracle"
            dts, t_all = [], time.perf_counter()
            while len(dts) < 5 or (len(dts) < 9 and time.perf_counter() - t_all < 12.0):
                t0 = time.perf_counter()
                again = co.msm_g1(hb, hs, nthreads=cores)
                dts.append(time.perf_counter() - t0)
                assert again == want
                if time.perf_counter() - t_all > 30.0 and len(dts) >= 3:
                    break
            dts.sort()
            dt = dts[len(dts) // 2]
            out["cpu_baseline"] = {"value": round(n / dt / 1e6, 4), "unit": "Mpt/s", "cores": cores, "cpu_quota": cpu_quota(), "kind": "port",
                                   "sample": f"the full 2^{args.log_n}-point MSM of this run: median of {len(dts)} runs after 1 warm-up "
                                             f"(min {dts[0]:.2f} s, median {dt:.2f} s, max {dts[-1]:.2f} s), window-per-thread Pippenger "
                                             "(bellman-equivalent)",
                                   "parity": "bit-exact (97-byte affine result)"}
            if cores_all != cores:
                t0 = time.perf_counter()
                assert co.msm_g1(hb, hs, nthreads=cores_all) == want
                out["cpu_baseline"]["all_threads"] = {"cores": cores_all, "value": round(n / (time.perf_counter() - t0) / 1e6, 4),
                                                      "note": "one run with a thread per visible CPU (oversubscribes the quota)"}
        if others is not None:
            out["other_configs"] = others
        if proofs is not None and isinstance(proofs.get("proofs_per_s_pipelined"), float):
            gbs = PROOF_ALG_BYTES * proofs["proofs_per_s_pipelined"] / 1e9
            proofs["proof_roofline"] = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(gbs / HBM_PEAK_GBS, 6),
                                        "alg_bytes_per_proof": PROOF_ALG_BYTES, "note": "SURVEY 8d: 1.24 GB of compulsory traffic per 2^20-class proof"}
        if proofs is not None:
            out["proofs"] = proofs
            if world == 1:
                out["proofs_per_sec"] = proofs.get("proofs_per_s_pipelined")
            else:  # proofs do not shard: N GPUs = N independent replicas, rates add
                out["proofs"]["per_rank_ring"] = [round(x, 3) for x in rates]
                out["proofs"]["per_rank_live_producers"] = [round(x, 3) for x in rates_live]
                out["proofs"]["live_producers_total"] = None if any(x != x for x in rates_live) else round(sum(rates_live), 3)
                out["proofs_per_sec"] = None if any(x != x for x in rates) else round(sum(rates), 3)
                # the HOST side of the scaling curve, spelled out (VERDICT r5 item 6b / missing 2): what one proof costs the host on every rank, how many proofs/s
                # the shared CPU quota can feed at that price, and how far the live-producer total falls short of the ring-fed (GPU-side) total
                q_now = cpu_quota()
                per_proof = [(w_ + p_) for w_, p_ in host_cpu]
                ok = bool(per_proof) and all(x == x and x > 0 for x in per_proof)
                live_total, ring_total = out["proofs"]["live_producers_total"], out["proofs_per_sec"]
                out["proofs"]["host_bound"] = {
                    "witness_cpu_s_per_rank": [round(w_, 4) for w_, _ in host_cpu],
                    "prover_host_cpu_s_per_proof_per_rank": [round(p_, 4) for _, p_ in host_cpu],
                    "host_cpu_s_per_proof": round(sum(per_proof) / len(per_proof), 4) if ok else None,
                    "cpu_quota": q_now,
                    "quota_feeds_proofs_per_s": round(q_now / (sum(per_proof) / len(per_proof)), 1) if ok and q_now else None,
                    "live_over_ring": round(live_total / ring_total, 3) if live_total and ring_total else None,
                    "reading": "live_over_ring < 1 with quota_feeds_proofs_per_s < proofs_per_sec: the ranks' witness producers are bound by the host's CPU quota, "
                               "not by the GPUs (the N > 1 default - deferred witness values staged by the producers, BZK_BENCH_DEFER / _STAGE - already lowers witness_cpu_s)"}
                out["proofs"]["proofs_per_sec_is"] = ("sum over the ranks of proofs_per_s_ring: every rank proves a ring of pre-synthesised 16-tx witnesses "
                                                     "(upload + proof, own r, s per proof); the ranks' LIVE witness producers share one host CPU quota "
                                                     f"({cpu_quota()} CPUs for {world} ranks) and are reported beside it (live_producers_total)")
        print(json.dumps(out), flush=True)
    if mg is not None:
        mg.bases_free(mg_bases)
        pctx.close()
        mg.close()
    elif world == 1:
        ctx.msm_bases_free(rbases)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
