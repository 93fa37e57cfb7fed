"""N proofs of the 16-tx Update circuit (15,3,2), one after the other on one prover slot - the un-pipelined caller.  With
BZK_PROVE_SERIAL=1 the five MSMs of a proof also run one after the other, so a rocprofv3 kernel trace of this script is a clean
per-kernel table of ONE proof (what the rocPRIM sorts, the de-duplication and every MSM stage really cost):

    BZK_PROVE_SERIAL=1 rocprofv3 --kernel-trace --stats -d out -- python tools/prove_serial.py 6

PROVE_DEFER=1: the witness generator leaves the hash-dependent values to the device (bzk_mpn_set_defer) and the proofs go through
bzk_groth16_prove_r1cs - the trace then also shows the witness-fill kernels (wf_hash / wf_poseidon / wf_small).

usage: python tools/prove_serial.py [n_proofs=6]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bazuka_amd import Bzk, lib as L
from bench import _fr, _fr_blind


def main(n_proofs=6):
    Z = _fr(1)
    ctx = Bzk(0)
    w = L.MpnWorld(15, 3)
    for i in range(32):
        w.add_account(i, b"acct%d" % i, Z, 10 ** 12)

    def batch(k):
        for i in range(16):
            w.push_tx(i, 16 + i, Z, 100 + i + k, Z, i % 7)

    batch(0)
    r = w.update_synthesize(2, _fr(99), Z, record_matrices=True)
    csr = [(r.n_constraints, r.raw("rp" + x), r.raw("col" + x), r.raw("val" + x)) for x in "ABC"]
    ph, _ = ctx.groth16_setup(csr, r.n_in, r.n_aux, b"".join(_fr(x) for x in (1234567, 2345678, 3456789, 4567891, 5678912)))
    ts = []
    defer = os.environ.get("PROVE_DEFER", "0") != "0"
    w.set_defer(defer)
    for k in range(n_proofs):
        batch(k + 1)
        rk = w.update_synthesize(2, _fr(99), Z)
        t0 = time.perf_counter()
        if defer:
            ctx.groth16_prove_r1cs(ph, rk, _fr_blind(2 * k), _fr_blind(2 * k + 1))
        else:
            ctx.groth16_prove(ph, rk.raw("z"), rk.raw("az"), rk.raw("bz"), rk.raw("cz"), _fr_blind(2 * k), _fr_blind(2 * k + 1))
        ts.append(time.perf_counter() - t0)
        rk.free()
    print(json.dumps({"n_proofs": n_proofs, "serial_msms": os.environ.get("BZK_PROVE_SERIAL", "0"), "deferred": defer, "prove_s": [round(t, 4) for t in ts]}))
    ctx.params_free(ph)
    ctx.close()


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
