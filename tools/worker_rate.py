"""What a ONE-SLOT worker gets per second (bazuka_amd/worker.py): N update works of the 2^20 class (16 transactions each, consecutive
states), proved (a) one after the other - synthesize, then prove: the round-3 loop - and (b) by Worker.run_once, which synthesizes work
k + 1 on a host thread while the GPU proves work k.  Every proof is checked with the work's own key (`MpnWork::verify`).
usage: python tools/worker_rate.py [n_works=24]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bazuka_amd import Bzk, lib as L, worker as W
from bench import _fr

ADDR = bytes(range(1, 33))


def main(n_works=24):
    Z = _fr(1)
    ctx = Bzk(0)
    tox = {k: b"".join(_fr(1234567 * (k + 1) + 7919 * j + 11) for j in range(5)) for k in range(3)}   # (tau = 1 would sit on the domain)
    keys = W.DevSetup(ctx, tox)
    vks = [keys.keys(k, 15, 3, b)[1] for k, b in ((0, 1), (1, 1), (2, 2))]
    w = L.MpnWorld(15, 3)
    for i in range(32):
        w.add_account(i, b"acct%d" % i, Z, 10 ** 12)
    works = {}
    for k in range(n_works):
        for i in range(16):
            w.push_tx(i, 16 + i, Z, 100 + i + k, Z, i % 7)
        works[k] = L.MpnWork.decode(w.make_work(2, vks, 100 + k, log4_batches=(1, 1, 2)).encode())
    out = {"works": n_works, "circuit": "UpdateCircuit(L=15,T=3,B=2): 16 tx"}
    wk = W.Worker(ctx, ADDR, ("127.0.0.1", 1), keys)
    wk.prove(works[0])  # warm-up: h table, workspaces
    t0 = time.perf_counter()
    serial = {k: wk.prove(v) for k, v in works.items()}
    out["serial_works_per_s"] = round(n_works / (time.perf_counter() - t0), 2)
    posted = {}
    wk.fetch = lambda: works
    wk.submit = lambda proofs: posted.update(proofs) or len(proofs)
    t0 = time.perf_counter()
    assert wk.run_once() == n_works
    out["run_once_works_per_s"] = round(n_works / (time.perf_counter() - t0), 2)
    out["all_verified"] = all(works[k].verify(ADDR, p) for k, p in posted.items()) and all(works[k].verify(ADDR, p) for k, p in serial.items())
    out["stats"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in wk.stats.items() if k in ("proved", "synth_s", "prove_s", "unsat")}
    print(json.dumps(out))
    keys.close()
    ctx.close()


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
