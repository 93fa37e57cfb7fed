"""Where does the PROVER side's host CPU go?  N prover slots prove the same pre-built witness in a loop (tools/pipe_probe.py's set-up: no producers), and the
CPU seconds of every thread of the process over the timed region are read from /proc/self/task/*/stat, grouped by thread name: the slots' own threads (named here),
libbzk's lane threads (`bzk-lane`, named in ctx.hip) and everything else (runtime helper threads, the interpreter's main thread).
usage: python tools/host_cpu_probe.py [slots=4] [proofs_per_slot=24]      env BZK_SYNC_BLOCKING=0|1 (read when the first context is created)"""
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bazuka_amd import Bzk, lib as L

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
TCK = os.sysconf("SC_CLK_TCK")
libc = ctypes.CDLL(None)


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def task_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            comm = open(f"/proc/self/task/{tid}/comm").read().strip()
            st = open(f"/proc/self/task/{tid}/stat").read()
            f = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(f[11]), int(f[12])))  # utime, stime in ticks
        except (OSError, ValueError):
            pass
    return out


def main(n_slots=4, per=24):
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

    slot_cpu = [0.0] * n_slots

    def run(i, k, timed=True):
        c, p = slots[i]
        t = time.thread_time()
        for j in range(k):
            c.groth16_prove(p, *views, fr(3 + j), fr(5 + j))
        if timed:
            slot_cpu[i] = time.thread_time() - t  # CLOCK_THREAD_CPUTIME_ID of the calling thread: what bzk_groth16_prove costs the CALLER's thread

    for i in range(n_slots):
        run(i, 2, timed=False)
    import resource
    a = task_cpu()
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, per)) for i in range(n_slots)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    b = task_cpu()  # the slot threads have exited (their own clocks are in slot_cpu); lane threads and the runtime's helpers persist
    by = {}
    for tid, (comm, (u, s)) in b.items():
        u0, s0 = a.get(tid, (comm, (0, 0)))[1]
        e = by.setdefault(comm, [0.0, 0.0, 0])
        e[0] += (u - u0) / TCK
        e[1] += (s - s0) / TCK
        e[2] += 1
    n = n_slots * per
    print(json.dumps({"slots": n_slots, "proofs": n, "proofs_per_s": round(n / dt, 2), "blocking": os.environ.get("BZK_SYNC_BLOCKING"),
                      "process_cpu_s_per_proof": round(((r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)) / n, 5),
                      "process_sys_share": round((r1.ru_stime - r0.ru_stime) / max(1e-9, (r1.ru_utime + r1.ru_stime) - (r0.ru_utime + r0.ru_stime)), 3),
                      "callers_threads_cpu_s_per_proof": round(sum(slot_cpu) / n, 5),
                      "persisting_threads_cpu_s_per_proof_by_name": {k: {"user": round(v[0] / n, 5), "sys": round(v[1] / n, 5), "threads": v[2]} for k, v in sorted(by.items())}}))


if __name__ == "__main__":
    main(*[int(x) for x in sys.argv[1:]])
