"""Where is the pipelined proof rate bound?  N prover slots on one GPU proving the SAME pre-built witness in a loop (no host producers, no
queue): the GPU-side ceiling of bench.py's pipelined figure.  With bg_producers > 0 that many witness producers (bg_threads worker threads
each) synthesize batches in the background and THROW THEM AWAY: what the producers' mere presence on the host costs the prover.
bg_procs = 1 runs those producers in ONE SEPARATE PROCESS (no GPU visible to it) instead of threads of this one: the same host work beside
the prover, but no shared interpreter lock, allocator or HIP runtime.
usage: python tools/pipe_probe.py [slots=4] [proofs_per_slot=16] [bg_producers=0] [bg_threads=8] [bg_procs=0]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bazuka_amd import Bzk, lib as L

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def main(n_slots=4, per=16, bg_producers=0, bg_threads=8, bg_procs=0):
    torch.cuda.init()
    ZIESHA = fr(1)
    w = L.MpnWorld(15, 3)
    for i in range(32):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 12)
    for i in range(16):
        w.push_tx(i, 16 + i, ZIESHA, 100 + i, ZIESHA, i % 7)
    r = w.update_synthesize(2, fr(99), ZIESHA, record_matrices=True)
    csr = [(r.n_constraints, r.view("rp" + x), r.view("col" + x), r.view("val" + x)) for x in "ABC"]
    tox = b"".join(fr(x) for x in (1234567, 2345678, 3456789, 4567891, 5678912))
    slots = []
    for _ in range(n_slots):
        cx = Bzk(0)
        slots.append((cx, cx.groth16_setup(csr, r.n_in, r.n_aux, tox)[0]))
    views = [r.raw(x) for x in ("z", "az", "bz", "cz")]

    def run(i, k):
        c, p = slots[i]
        for j in range(k):
            c.groth16_prove(p, *views, fr(3 + j), fr(5 + j))

    for i in range(n_slots):
        run(i, 2)
    stop = threading.Event()
    made = [0]

    def background(seed):
        pw = L.MpnWorld(15, 3)
        pw.set_threads(bg_threads)
        for i in range(32):
            pw.add_account(i, b"bg%dacct%d" % (seed, i), ZIESHA, 10 ** 12)
        k = 0
        while not stop.is_set():
            k += 1
            for i in range(16):
                pw.push_tx(i, 16 + i, ZIESHA, 100 + i + k, ZIESHA, i % 7)
            rr = pw.update_synthesize(2, fr(99), ZIESHA)
            assert rr.satisfied
            made[0] += 1

    child = None
    if bg_procs and bg_producers:
        import subprocess
        env = dict(os.environ, ROCR_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", BZK_PROBE_CHILD="1")
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(bg_producers), str(bg_threads)], env=env,
                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        bg = []
        time.sleep(4.0)  # interpreter start + worlds + pool warm-up (page faults of the first witnesses) in the child
    else:
        bg = [threading.Thread(target=background, args=(s,), daemon=True) for s in range(bg_producers)]
    for t in bg:
        t.start()
    if bg:
        time.sleep(1.0)  # let them reach their steady state
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, per)) for i in range(n_slots)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    stop.set()
    if child is not None:
        try:
            made[0] = int(child.communicate(input="stop\n", timeout=30)[0].strip().splitlines()[-1])  # witnesses it made in all
        except Exception:
            child.kill()
    out = {"slots": n_slots, "proofs": n_slots * per, "proofs_per_s_same_witness_no_producers": round(n_slots * per / dt, 2)}
    if bg_producers:
        out = {"slots": n_slots, "proofs": n_slots * per, "background_producers": bg_producers, "threads_each": bg_threads,
               "producers_in": "a separate process (count = all it made since its start)" if child is not None else "threads of this process",
               "proofs_per_s_same_witness_with_idle_producers": round(n_slots * per / dt, 2), "witnesses_discarded": made[0]}
    print(json.dumps(out))
    for t in bg:
        t.join()


def child_main(n_producers, n_threads):
    """producers only, no prover, no GPU: runs until a line arrives on stdin, then prints how many witnesses it made"""
    if os.environ.get("BZK_PROBE_CHILD_NICE"):
        os.nice(int(os.environ["BZK_PROBE_CHILD_NICE"]))
    ZIESHA = fr(1)
    stop = threading.Event()
    made = [0]

    def background(seed):
        pw = L.MpnWorld(15, 3)
        pw.set_threads(n_threads)
        for i in range(32):
            pw.add_account(i, b"bg%dacct%d" % (seed, i), ZIESHA, 10 ** 12)
        k = 0
        while not stop.is_set():
            k += 1
            for i in range(16):
                pw.push_tx(i, 16 + i, ZIESHA, 100 + i + k, ZIESHA, i % 7)
            rr = pw.update_synthesize(2, fr(99), ZIESHA)
            assert rr.satisfied
            made[0] += 1

    th = [threading.Thread(target=background, args=(s,), daemon=True) for s in range(n_producers)]
    for t in th:
        t.start()
    sys.stdin.readline()
    stop.set()
    for t in th:
        t.join()
    print(made[0], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child_main(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main(*(int(x) for x in sys.argv[1:]))
