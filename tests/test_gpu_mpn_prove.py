"""GPU end-to-end: MPN Update batch -> host witness/R1CS (C++) -> CRS on the GPU -> Groth16 proof on the GPU;
the 387 proof bytes equal the CPU oracle's on the same (CRS, witness, r, s) and the oracle's pairing check
accepts them for the batch's public inputs [commitment, height, state, aux, next_state]
(what `groth16_verify`, src/zk/groth16/mod.rs:67-121, checks in the node)."""
import pytest

from bazuka_amd import lib as L
from oracle import pyref as pr
from util import fr_bytes, fr_list

pytestmark = pytest.mark.gpu
F, U = pr.fr_to_mont_bytes, pr.fr_from_mont_bytes
ZIESHA = F(1)


def _csr(r):
    return [(r.n_constraints, r.view("rp" + w), r.view("col" + w), r.view("val" + w)) for w in "ABC"]


def _oracle_params(bzk, ph, r, log_m):
    d = {"n_in": r.n_in, "n_aux": r.n_aux, "log_m": log_m, "a_density": r.view("a_density"), "b_density": r.view("b_density")}
    for which, key in ((0, "vk"), (1, "h"), (2, "l"), (3, "a"), (4, "b_g1"), (5, "b_g2")):
        d[key] = bzk.params_read(ph, which)
    d["n_a"], d["n_b"] = sum(d["a_density"]), sum(d["b_density"])
    return d


def test_update_batch_prove_on_gpu_verifies_and_matches_oracle(bzk, co):
    w = L.MpnWorld(3, 3)
    for i in range(4):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 9)
    w.push_tx(0, 1, ZIESHA, 1000, ZIESHA, 7)
    w.push_tx(1, 2, ZIESHA, 500, ZIESHA, 3)
    w.push_tx(2, 3, ZIESHA, 10, ZIESHA, 1)
    w.set_height(11)
    r = w.update_synthesize(1, F(456), ZIESHA, record_matrices=True)
    assert r.satisfied and r.accepted == 3
    tox = fr_bytes(fr_list(5, 31337))
    ph, vkb = bzk.groth16_setup(_csr(r), r.n_in, r.n_aux, tox)
    rs = fr_bytes(fr_list(2, 77))
    proof = bzk.groth16_prove(ph, r.view("z"), r.view("az"), r.view("bz"), r.view("cz"), rs[:32], rs[32:])
    # 1. pairing check with the VK the setup returned (bincode Groth16VerifyingKey layout)
    vk = pr.vk_from_bytes(vkb)
    z = r.view("z")
    pub = [U(z[32 * i:32 * i + 32]) for i in range(1, 6)]
    assert pub[0] == 456 and pub[1] == 11 and pub[4] == U(w.root())
    assert pr.groth16_verify(vk, pub, pr.proof_from_bytes(proof))
    assert not pr.groth16_verify(vk, pub[:4] + [pub[4] + 1], pr.proof_from_bytes(proof))
    # 2. byte parity with the oracle prover on the device-generated CRS
    op = _oracle_params(bzk, ph, r, 17)
    want = co.groth16_prove(op, z, r.view("az"), r.view("bz"), r.view("cz"), rs[:32], rs[32:], nthreads=co.ncpu())
    assert proof == want
    # 3. a tampered witness (one aux value changed) must not verify
    zbad = bytearray(z)
    zbad[32 * 100:32 * 101] = F(12345)
    bad = bzk.groth16_prove(ph, bytes(zbad), r.view("az"), r.view("bz"), r.view("cz"), rs[:32], rs[32:])
    assert not pr.groth16_verify(vk, pub, pr.proof_from_bytes(bad))
    bzk.params_free(ph)


def test_deposit_and_withdraw_batches_prove_on_gpu(bzk, co):
    """The other two MpnWorkData variants (src/mpn/mod.rs:243-248): a deposit batch then a withdraw batch, each
    set up + proved on the GPU, verified by the oracle's pairing check, proof bytes == the oracle prover's."""
    w = L.MpnWorld(3, 3)
    for i in range(2):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 9)
    w.add_key(6, b"fresh")
    w.push_deposit(0, ZIESHA, 1000)
    w.push_deposit(6, ZIESHA, 5)
    dep = w.deposit_synthesize(1, F(456), record_matrices=True)
    w.push_withdraw(0, ZIESHA, 400, ZIESHA, 2, F(4242))
    w.push_withdraw(1, ZIESHA, 9, ZIESHA, 0, F(7))
    wd = w.withdraw_synthesize(1, F(457), record_matrices=True)
    for r, log_m, seed in ((dep, 16, 5), (wd, 17, 6)):
        assert r.satisfied and r.accepted == 2
        ph, vkb = bzk.groth16_setup(_csr(r), r.n_in, r.n_aux, fr_bytes(fr_list(5, 1000 + seed)))
        rs = fr_bytes(fr_list(2, seed))
        z = r.view("z")
        proof = bzk.groth16_prove(ph, z, r.view("az"), r.view("bz"), r.view("cz"), rs[:32], rs[32:])
        pub = [U(z[32 * i:32 * i + 32]) for i in range(1, 6)]
        assert pr.groth16_verify(pr.vk_from_bytes(vkb), pub, pr.proof_from_bytes(proof))
        assert not pr.groth16_verify(pr.vk_from_bytes(vkb), [pub[0] + 1] + pub[1:], pr.proof_from_bytes(proof))
        op = _oracle_params(bzk, ph, r, log_m)
        assert proof == co.groth16_prove(op, z, r.view("az"), r.view("bz"), r.view("cz"), rs[:32], rs[32:], nthreads=co.ncpu())
        bzk.params_free(ph)
