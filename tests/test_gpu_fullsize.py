"""BASELINE-size parity against the CPU oracle inside `pytest -m gpu`: the sizes BASELINE.json's configs
name, compared with the oracle itself - not with the GPU's own composition properties - on the GPU box's host cores.

  C5  2^24-leaf 4-ary Poseidon tree            root == oracle root (5 592 405 hashes on all cores)
  a4  NTT 2^24 (coset forward, the h-stage's)  bytes == oracle NTT
  a6  G2 MSM 2^20                              193 bytes == oracle Pippenger
  a3+a7  16-tx Update circuit (15,3,2), 903 037 constraints, 2^20 domain: R1CS hashes == the independent Python restatement's
         (tests/golden/r1cs_sha256.json), CRS on the GPU, 387 proof bytes == oracle prover, oracle pairing check accepts."""
import hashlib
import json
import os

import pytest
import torch

import r1cs_scenarios as S
from bazuka_amd import lib as L
from oracle import pyref as pr
from util import dev_bytes, fr_bytes, fr_list, rand_scalars_bytes, to_dev

pytestmark = pytest.mark.gpu
U = pr.fr_from_mont_bytes
FIX = json.load(open(os.path.join(S.G, "r1cs_sha256.json")))


def _rand_scalars_dev(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    t = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    t[:, 31] &= 0x3F
    t = t.contiguous()
    torch.cuda.synchronize()
    return t


def test_tree_2p24_root_equals_oracle(bzk, co):
    leaves = _rand_scalars_dev(1 << 24, 24)
    root = bzk.merkle4_root_dev(leaves, 12)
    assert root == co.merkle4_root(dev_bytes(leaves), 12, nthreads=co.ncpu())


def test_ntt_2p24_equals_oracle(bzk, co):
    log_n = 24
    d = _rand_scalars_dev(1 << log_n, 2424)
    data = dev_bytes(d)
    bzk.ntt_dev(d, log_n, False, True)
    torch.cuda.synchronize()
    assert dev_bytes(d) == co.ntt(data, log_n, False, True, nthreads=co.ncpu())
    del data
    bzk.ntt_dev(d, log_n, True, False)          # a second mode at full size: plain inverse
    torch.cuda.synchronize()
    got = dev_bytes(d)
    d2 = _rand_scalars_dev(1 << log_n, 2424)    # same seed: the original input again
    bzk.ntt_dev(d2, log_n, False, True)
    torch.cuda.synchronize()
    assert got == co.ntt(dev_bytes(d2), log_n, True, False, nthreads=co.ncpu())


def test_msm_g2_2p20_equals_oracle(bzk, co):
    n = 1 << 20
    bases = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    bzk.g2_synth_bases_dev(2020, 0, n, bases)
    sc = _rand_scalars_dev(n, 2020)
    want = co.msm_g2(dev_bytes(bases), dev_bytes(sc), nthreads=co.ncpu())
    assert bzk.msm_g2_dev(bases, sc, n) == want
    assert bzk.msm_g2_dev(bases, sc, n, dedup=True) == want


@pytest.mark.parametrize("name", [n for n in S.SCENARIOS if n not in S.PRODUCTION])   # the production shapes: test_gpu_production.py
def test_r1cs_fixture_replay_on_the_gpu_box(name):
    """the generator is host code, but it is THIS build on THIS box that proves: replay the pinned hashes here too"""
    blob = S.make_work(name)
    assert hashlib.sha256(blob).hexdigest() == FIX[name]["work_sha256"]
    r, views, _ = S.product_views(blob)
    assert (r.n_in, r.n_aux, r.n_constraints) == (FIX[name]["n_in"], FIX[name]["n_aux"], FIX[name]["n_constraints"])
    for key in L.R1cs.VIEWS:
        assert hashlib.sha256(views[key]).hexdigest() == FIX[name]["sha256"][key], (name, key)


def test_16tx_update_proof_equals_oracle_and_verifies(bzk, co):
    """BASELINE configs[1..2]'s circuit class as a test, not only a bench assert: MpnWork bytes -> worker-side synthesis ->
    GPU CRS -> GPU proof; bytes == oracle prover on the same (CRS, witness, r, s); pairing check accepts / rejects."""
    name = "update_15_3_2"
    blob = S.make_work(name)
    dec = L.MpnWork.decode(blob)
    r = dec.synthesize(S.PROVER, record_matrices=True)
    assert r.satisfied and (r.n_aux, r.n_constraints) == (904870, 903037)
    for key in L.R1cs.VIEWS:
        assert hashlib.sha256(r.view(key)).hexdigest() == FIX[name]["sha256"][key], key
    csr = [(r.n_constraints, r.view("rp" + w), r.view("col" + w), r.view("val" + w)) for w in "ABC"]
    ph, vkb = bzk.groth16_setup(csr, r.n_in, r.n_aux, fr_bytes(fr_list(5, 20201)))
    rs = fr_bytes(fr_list(2, 20202))
    z, az, bz, cz = (r.view(k) for k in ("z", "az", "bz", "cz"))
    proof = bzk.groth16_prove(ph, z, az, bz, cz, rs[:32], rs[32:])
    op = {"n_in": r.n_in, "n_aux": r.n_aux, "log_m": 20, "a_density": r.view("a_density"), "b_density": r.view("b_density")}
    for which, key in ((0, "vk"), (1, "h"), (2, "l"), (3, "a"), (4, "b_g1"), (5, "b_g2")):
        op[key] = bzk.params_read(ph, which)
    op["n_a"], op["n_b"] = sum(op["a_density"]), sum(op["b_density"])
    assert proof == co.groth16_prove(op, z, az, bz, cz, rs[:32], rs[32:], nthreads=co.ncpu())
    vk = pr.vk_from_bytes(vkb)
    pub = [U(z[32 * i:32 * i + 32]) for i in range(1, 6)]
    assert pub[0] == U(dec.commitment(S.PROVER)) and pub[1] == 11 and pub[2] == U(dec.state) and pub[4] == U(dec.next_state)
    assert pr.groth16_verify(vk, pub, pr.proof_from_bytes(proof))
    assert not pr.groth16_verify(vk, pub[:4] + [pub[4] + 1], pr.proof_from_bytes(proof))
    bzk.params_free(ph)
