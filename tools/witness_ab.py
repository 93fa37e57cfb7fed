"""A/B of the witness builders' tree work (VERDICT r2 task 7): production shape L = 15, T = 3, `n_tx` update transactions between
distinct accounts, host path (per-transaction walk of the sparse tree) against the device path (bzk_mpn_set_device: one batched
Poseidon launch per tree level).  Prints make_work time (validator side: the transitions with their proofs) and update_synthesize
time (witness + circuit instance) for both, and checks that the work bytes agree.
usage: python tools/witness_ab.py [n_tx=256]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import r1cs_scenarios as sc
    from bazuka_amd import Bzk, lib as L
    from oracle import pyref as pr
    F = pr.fr_to_mont_bytes
    Z = F(1)
    n_tx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    b = 0
    while 4 ** b < n_tx:
        b += 1
    ctx = Bzk(0)

    def world(dev):
        w = L.MpnWorld(15, 3)
        if dev:
            w.set_device(ctx)
        for i in range(2 * n_tx):
            w.add_account((i * 7919 + 1) % 4 ** 15, b"a%d" % i, Z, 10 ** 12)
        return w

    def queue(w, k):
        for i in range(n_tx):
            w.push_tx(((i) * 7919 + 1) % 4 ** 15, ((n_tx + i) * 7919 + 1) % 4 ** 15, Z, 100 + i + k, Z, i % 7)

    out = {"n_tx": n_tx, "log4_batch": b, "shape": "L=15, T=3"}
    blobs = {}
    for dev in (False, True):
        w = world(dev)
        key = "device" if dev else "host"
        ts = []
        for k in range(3):
            queue(w, k)
            t0 = time.perf_counter()
            work = w.make_work(2, sc.VKS, 1, log4_batches=(1, 1, b))
            ts.append(time.perf_counter() - t0)
            if k == 0:
                blobs[key] = work.encode()
        out[f"make_work_s_{key}"] = [round(t, 4) for t in ts]
        ts = []
        for k in range(3):
            queue(w, 10 + k)
            t0 = time.perf_counter()
            r = w.update_synthesize(b, F(7), Z)
            ts.append(time.perf_counter() - t0)
            assert r.satisfied and r.accepted == n_tx
        out[f"update_synthesize_s_{key}"] = [round(t, 4) for t in ts]
    out["same_work_bytes"] = blobs["host"] == blobs["device"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
