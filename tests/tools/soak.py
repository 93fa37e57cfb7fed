"""Soak test of the proof path: N proofs back to back through `slots` prover slots fed by witness producers; reports proofs/s,
host RSS and free device memory at the start, the middle and the end (leaks show up as drift).  EVERY proof is checked against its own
public inputs with the product's host verifier (bzk_groth16_verify, ~20 ms each, on checker threads - a race that corrupts one proof in a
thousand shows up here), and the last one also with the oracle's independent pairing verifier.
usage: python tests/tools/soak.py [n_proofs=600] [slots=1]"""
import json, os, queue, resource, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def main(n_proofs=600, n_slots=1):
    import torch
    torch.cuda.init()
    from bazuka_amd import Bzk, lib as L
    ctx = Bzk(0)
    Z = fr(1)
    lg, t, b, n_tx = 15, 3, 2, 16

    def world(seed):
        w = L.MpnWorld(lg, t)
        for i in range(2 * n_tx):
            w.add_account(i, b"s%dacct%d" % (seed, i), Z, 10 ** 12)
        return w
    w0 = world(0)
    for i in range(n_tx):
        w0.push_tx(i, n_tx + i, Z, 100 + i, Z, i % 7)
    r = w0.update_synthesize(b, fr(99), Z, record_matrices=True)
    csr = [(r.n_constraints, r.view("rp" + x), r.view("col" + x), r.view("val" + x)) for x in "ABC"]
    ph, vk = ctx.groth16_setup(csr, r.n_in, r.n_aux, b"".join(fr(x) for x in (11, 22, 33, 44, 55)))
    del csr
    q = queue.Queue(maxsize=4)
    stop = threading.Event()

    def producer(seed):
        pw = world(seed)
        k = 0
        while not stop.is_set():
            k += 1
            for i in range(n_tx):
                pw.push_tx(i, n_tx + i, Z, 100 + i + k, Z, i % 7)
            rr = pw.update_synthesize(b, fr(99), Z)
            while not stop.is_set():
                try:
                    q.put(rr, timeout=0.05); break
                except queue.Full:
                    pass
    th = [threading.Thread(target=producer, args=(s + 1,), daemon=True) for s in range(3)]
    for x in th:
        x.start()
    slots = [(ctx, ph)]
    for _ in range(n_slots - 1):  # further slots share the device-resident CRS of the first (bzk_params_slot, round 3)
        cx = Bzk(0)
        slots.append((cx, cx.params_slot(ph)))
    marks = []
    lock = threading.Lock()
    state = {"taken": 0, "done": 0, "bad": 0, "checked": 0}
    check_q = queue.Queue()
    last = [None]

    def checker():
        while True:
            item = check_q.get()
            if item is None:
                return
            inputs, proof = item
            good = L.groth16_verify(vk, inputs, proof)
            with lock:
                state["checked"] += 1
                state["bad"] += 0 if good else 1

    checkers = [threading.Thread(target=checker) for _ in range(3)]
    for c in checkers:
        c.start()
    t0 = time.perf_counter()

    def prover(slot):
        c, p = slots[slot]
        while True:
            with lock:
                if state["taken"] >= n_proofs:
                    return
                k = state["taken"]
                state["taken"] += 1
            rr = q.get()
            proof = c.groth16_prove(p, rr.raw("z"), rr.raw("az"), rr.raw("bz"), rr.raw("cz"), fr(3 + k), fr(5 + k))
            z = rr.raw("z")
            check_q.put((bytes(z[32:32 * 6]), proof))
            with lock:
                state["done"] += 1
                d = state["done"]
                last[0] = (rr, proof)
                if d in (20, n_proofs // 2, n_proofs):
                    free, total = torch.cuda.mem_get_info()
                    marks.append({"proof": d, "host_rss_MB": round(rss_mb()), "device_free_GB": round(free / 1e9, 3), "elapsed_s": round(time.perf_counter() - t0, 2)})

    provers = [threading.Thread(target=prover, args=(i,)) for i in range(len(slots))]
    for x in provers:
        x.start()
    for x in provers:
        x.join()
    dt = time.perf_counter() - t0
    stop.set()
    for _ in checkers:
        check_q.put(None)
    for c in checkers:
        c.join()
    last = last[0]
    from oracle import pyref as pr
    z = last[0].raw("z")
    pub = [pr.fr_from_mont_bytes(z[32 * i:32 * i + 32]) for i in range(1, 6)]
    ok = pr.groth16_verify(pr.vk_from_bytes(vk), pub, pr.proof_from_bytes(last[1]))
    print(json.dumps({"proofs": n_proofs, "slots": len(slots), "proofs_per_s": round(n_proofs / dt, 2), "marks": marks,
                      "verified_by_bzk_groth16_verify": state["checked"], "rejected": state["bad"], "last_proof_verifies_oracle": ok}))
    assert ok and state["bad"] == 0 and state["checked"] == n_proofs


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
