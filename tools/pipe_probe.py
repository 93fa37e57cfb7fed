"""Where is the pipelined proof rate bound?  N prover slots on one GPU proving the SAME pre-built witness in a loop (no host producers, no
queue): the GPU-side ceiling of bench.py's pipelined figure.  usage: python tools/pipe_probe.py [slots=4] [proofs_per_slot=16]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bazuka_amd import Bzk, lib as L

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def main(n_slots=4, per=16):
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
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, per)) for i in range(n_slots)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print(json.dumps({"slots": n_slots, "proofs": n_slots * per, "proofs_per_s_same_witness_no_producers": round(n_slots * per / dt, 2)}))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
