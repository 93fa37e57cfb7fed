"""Production-size MPN Update proof on ONE GPU: UpdateCircuit(L=15, T=3, B) with B = 4 -> 256 signed transactions,
14 443 117 constraints, NTT domain 2^24 (src/config/blockchain.rs:22-26) - or any other B given on the command line.
Product code end to end (host generator, CRS on the GPU, proof on the GPU); the proof is then checked with the
oracle's pairing verifier against the batch's public inputs, and (optionally) byte-compared with the oracle prover.
usage: python tests/tools/prove_production.py [log4_batch=4] [n_proofs=2] [compare_oracle=0] [device_builder=0]
device_builder=1: the witness builder batches its Merkle hashing on the GPU (bzk_mpn_set_device)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bazuka_amd import Bzk, lib as L

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def main(b=4, n_proofs=2, compare=0, device_builder=0):
    ZIESHA = fr(1)
    lg, t = 15, 3
    ctx = Bzk(0)
    n_tx = 1 << (2 * b)
    w = L.MpnWorld(lg, t)
    if device_builder:
        w.set_device(Bzk(0))
    for i in range(2 * n_tx):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 12)

    def batch(k):
        for i in range(n_tx):
            w.push_tx(i, n_tx + i, ZIESHA, 100 + i + k, ZIESHA, i % 7)

    out = {"circuit": f"UpdateCircuit(L={lg},T={t},B={b}): {n_tx} tx", "witness_builder": "device (bzk_mpn_set_device)" if device_builder else "host"}
    t0 = time.perf_counter(); batch(0); out["sign_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter(); r = w.update_synthesize(b, fr(99), ZIESHA, record_matrices=True)
    out["synthesize_with_matrices_s"] = round(time.perf_counter() - t0, 2)
    assert r.satisfied and r.accepted == n_tx
    out.update(n_constraints=r.n_constraints, n_aux=r.n_aux)
    csr = [(r.n_constraints, r.raw("rp" + x), r.raw("col" + x), r.raw("val" + x)) for x in "ABC"]  # zero-copy (val > 2 GiB at B = 5)
    out["nnz"] = [r.nnzA, r.nnzB, r.nnzC]
    tox = b"".join(fr(x) for x in (1234567, 2345678, 3456789, 4567891, 5678912))
    t0 = time.perf_counter(); ph, vk = ctx.groth16_setup(csr, r.n_in, r.n_aux, tox); out["gpu_crs_setup_s"] = round(time.perf_counter() - t0, 2)
    print(json.dumps(out), flush=True)
    tw, tp = [], []
    proof = z = None
    cur = r
    for k in range(n_proofs):
        if k:
            batch(k)
            if cur is not r:
                cur.free()  # hand the pinned witness arrays back to the pool BEFORE the next instance asks for its own
            t0 = time.perf_counter(); cur = w.update_synthesize(b, fr(99), ZIESHA); tw.append(time.perf_counter() - t0)
            assert cur.accepted == n_tx
        z, az, bz, cz = cur.raw("z"), cur.raw("az"), cur.raw("bz"), cur.raw("cz")
        t0 = time.perf_counter(); proof = ctx.groth16_prove(ph, z, az, bz, cz, fr(7 + k), fr(9 + k)); tp.append(time.perf_counter() - t0)
    out["witness_s"] = round(min(tw), 3) if tw else None
    out["gpu_prove_s"] = [round(x, 4) for x in tp]
    out["tx_per_s_gpu_only"] = round(n_tx / min(tp), 1)
    ctx.prof_enable(True); ctx.prof_reset()
    ctx.groth16_prove(ph, z, az, bz, cz, fr(1), fr(2))
    out["prove_kernels_ms"] = {k: round(v[1], 3) for k, v in ctx.prof_dump().items() if v[1] > 0.5}
    # parity: pairing check (size-independent) against the batch's public inputs
    from oracle import pyref as pr
    U = pr.fr_from_mont_bytes
    pub = [U(z[32 * i:32 * i + 32]) for i in range(1, 6)]
    t0 = time.perf_counter()
    ok = pr.groth16_verify(pr.vk_from_bytes(vk), pub, pr.proof_from_bytes(proof))
    bad = pr.groth16_verify(pr.vk_from_bytes(vk), [pub[0] + 1] + pub[1:], pr.proof_from_bytes(proof))
    out["pairing_check"] = {"accepts": ok, "rejects_wrong_input": not bad, "s": round(time.perf_counter() - t0, 2)}
    if compare:
        from oracle import coracle as co
        d = {"n_in": cur.n_in, "n_aux": cur.n_aux, "log_m": (r.n_constraints - 1).bit_length(),
             "a_density": r.view("a_density"), "b_density": r.view("b_density")}
        for which, key in ((0, "vk"), (1, "h"), (2, "l"), (3, "a"), (4, "b_g1"), (5, "b_g2")):
            d[key] = ctx.params_read(ph, which)
        d["n_a"], d["n_b"] = sum(d["a_density"]), sum(d["b_density"])
        t0 = time.perf_counter()
        want = co.groth16_prove(d, bytes(z), bytes(az), bytes(bz), bytes(cz), fr(7 + n_proofs - 1), fr(9 + n_proofs - 1), nthreads=co.ncpu())
        out["oracle_prove"] = {"s": round(time.perf_counter() - t0, 1), "cores": co.ncpu(), "bytes_equal": want == proof}
    print(json.dumps(out), flush=True)
    assert ok and not bad


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
