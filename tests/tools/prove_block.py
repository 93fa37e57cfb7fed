"""One production MPN block on ONE GPU: the three proofs a validator needs per block (src/config/blockchain.rs:22-26,
326-328): a deposit batch (L=15, T=3, B=3: 64 tx, 2^21 domain), a withdraw batch (64 tx, 2^22 domain) and - unless
`small` is given - an update batch (B=4: 256 tx, 2^24 domain).  Product code end to end; every proof is checked with the
oracle's pairing verifier against its batch's public inputs.
usage: python tests/tools/prove_block.py [small]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bazuka_amd import Bzk, lib as L

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr(x):
    return (x * ((1 << 256) % R_MOD) % R_MOD).to_bytes(32, "little")


def main(small=False):
    from oracle import pyref as pr
    U = pr.fr_from_mont_bytes
    ZIESHA = fr(1)
    lg, t = 15, 3
    ctx = Bzk(0)
    w = L.MpnWorld(lg, t)
    n_acc = 512
    for i in range(n_acc):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 12)
    tox = b"".join(fr(x) for x in (1234567, 2345678, 3456789, 4567891, 5678912))
    report = {}

    def run(kind, make_batch, synth, log4_batch):
        out = {}
        make_batch(0)
        t0 = time.perf_counter(); r = synth(True); out["synthesize_with_matrices_s"] = round(time.perf_counter() - t0, 2)
        assert r.satisfied and r.accepted == 1 << (2 * log4_batch), (kind, r.accepted, r.rejected)
        out.update(tx=r.accepted, n_constraints=r.n_constraints, log_m=(r.n_constraints - 1).bit_length())
        csr = [(r.n_constraints, r.view("rp" + x), r.view("col" + x), r.view("val" + x)) for x in "ABC"]
        t0 = time.perf_counter(); ph, vk = ctx.groth16_setup(csr, r.n_in, r.n_aux, tox); out["gpu_crs_setup_s"] = round(time.perf_counter() - t0, 2)
        del csr
        tp, tw = [], []
        cur, proof = r, None
        for k in range(3):
            if k:
                make_batch(k)
                if cur is not r:
                    cur.free()  # pinned witness arrays back to the pool before the next instance takes its own
                t0 = time.perf_counter(); cur = synth(False); tw.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            proof = ctx.groth16_prove(ph, cur.raw("z"), cur.raw("az"), cur.raw("bz"), cur.raw("cz"), fr(7 + k), fr(9 + k))
            tp.append(time.perf_counter() - t0)
        z = cur.raw("z")
        pub = [U(z[32 * i:32 * i + 32]) for i in range(1, 6)]
        ok = pr.groth16_verify(pr.vk_from_bytes(vk), pub, pr.proof_from_bytes(proof))
        bad = pr.groth16_verify(pr.vk_from_bytes(vk), pub[:4] + [(pub[4] + 1) % R_MOD], pr.proof_from_bytes(proof))
        out.update(witness_s=round(min(tw), 3), gpu_prove_s=round(min(tp), 4), pairing_accepts=ok, rejects_wrong_next_state=not bad)
        ctx.params_free(ph)
        report[kind] = out
        print(json.dumps({kind: out}), flush=True)
        assert ok and not bad

    nd = 64
    run("deposit", lambda k: [w.push_deposit(i, ZIESHA, 1000 + i + k) for i in range(nd)],
        lambda rec: w.deposit_synthesize(3, fr(11), record_matrices=rec), 3)
    run("withdraw", lambda k: [w.push_withdraw(i, ZIESHA, 10 + i + k, ZIESHA, 1, fr(777 + i + 1000 * k)) for i in range(nd)],
        lambda rec: w.withdraw_synthesize(3, fr(12), record_matrices=rec), 3)
    if not small:
        nu = 256
        run("update", lambda k: [w.push_tx(i, nu + i, ZIESHA, 100 + i + k, ZIESHA, i % 7) for i in range(nu)],
            lambda rec: w.update_synthesize(4, fr(13), ZIESHA, record_matrices=rec), 4)
    report["block_gpu_prove_s"] = round(sum(v["gpu_prove_s"] for v in report.values()), 4)
    print(json.dumps(report), flush=True)


if __name__ == "__main__":
    main(len(sys.argv) > 1 and sys.argv[1] == "small")
