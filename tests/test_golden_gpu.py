"""GPU suite: the HIP path reproduces the committed fixtures of tests/golden/oracle_vectors.json byte for byte (no oracle
call at run time: the expected bytes are in the repository)."""
import hashlib
import json
import os

import pytest

from util import fr_bytes, fr_list, log2_ceil, r1cs_to_csr, rand_scalars_bytes, synth_r1cs

pytestmark = pytest.mark.gpu
ORC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.json")))


@pytest.mark.parametrize("arity", [1, 2, 4, 7, 16])
def test_poseidon_batch_fixture(bzk, arity):
    inp = fr_bytes(fr_list(arity * 5, 1000 + arity))
    assert bzk.poseidon_batch(inp, arity).hex() == ORC[f"poseidon_arity{arity}_seed{1000 + arity}_n5"]


def test_merkle_and_ntt_fixtures(bzk):
    assert bzk.merkle4_root(fr_bytes(fr_list(64, 2000)), 3).hex() == ORC["merkle4_log3_seed2000"]
    data = fr_bytes(fr_list(64, 3000))
    for inv in (0, 1):
        for cs in (0, 1):
            assert hashlib.sha256(bzk.ntt(data, 6, bool(inv), bool(cs))).hexdigest() == ORC[f"ntt_log6_seed3000_inv{inv}_coset{cs}"]
    big = rand_scalars_bytes(1 << 12, 3001)
    assert hashlib.sha256(bzk.ntt(big, 12, False, True)).hexdigest() == ORC["ntt_log12_seed3001_fwd_coset_sha256"]


def test_msm_fixtures(bzk, co):
    # bases k_i * G come from the device generator (itself checked against the oracle in test_gpu_msm.py)
    import torch
    sc = rand_scalars_bytes(300, 4001)
    for g, size, key in (("g1", 96, "msm_g1_bases4000_scalars4001_n300"), ("g2", 192, "msm_g2_bases4000_scalars4001_n300")):
        d = torch.empty(300 * size, dtype=torch.uint8, device="cuda")
        getattr(bzk, f"{g}_synth_bases_dev")(4000, 0, 300, d)
        torch.cuda.synchronize()
        bases = bytes(d.cpu().numpy().tobytes())
        assert getattr(bzk, f"msm_{g}")(bases, sc).hex() == ORC[key]
    d = torch.empty(5000 * 96, dtype=torch.uint8, device="cuda")
    bzk.g1_synth_bases_dev(4002, 0, 5000, d)
    torch.cuda.synchronize()
    bases = bytes(d.cpu().numpy().tobytes())
    sck = rand_scalars_bytes(5000, 4003)
    assert bzk.msm_g1(bases, sck).hex() == ORC["msm_g1_bases4002_scalars4003_n5000"]
    assert bzk.msm_g1(bases, sck, dedup=True).hex() == ORC["msm_g1_bases4002_scalars4003_n5000"]


def test_groth16_setup_and_proof_fixture(bzk, pr):
    import array
    r = synth_r1cs(200, 3, 5000)
    csr = []
    for which in range(3):
        rp, col, val = array.array("I", [0]), array.array("I"), []
        for row in r["rows"]:
            for v, c in row[which]:
                col.append(v)
                val.append(pr.fr_to_mont_bytes(c))
            rp.append(len(col))
        csr.append((len(r["rows"]), rp.tobytes(), col.tobytes(), b"".join(val)))
    ph, vkb = bzk.groth16_setup(csr, r["n_in"], r["n_aux"], fr_bytes(fr_list(5, 5001)))
    want = ORC["groth16_synth200_seed5000_tox5001_rs5002"]
    assert hashlib.sha256(vkb[:870]).hexdigest() == want["vk_sha256"]
    assert hashlib.sha256(bzk.params_read(ph, 1)).hexdigest() == want["h_sha256"]
    # witness evaluations from the rows (plain Python; the fixture's proof came from the oracle's evaluation)
    z = r["z"]
    ev = [[sum(c * z[v] for v, c in row[w]) % pr.R_MOD for row in r["rows"]] for w in range(3)]
    rs = fr_bytes(fr_list(2, 5002))
    proof = bzk.groth16_prove(ph, fr_bytes(z), fr_bytes(ev[0]), fr_bytes(ev[1]), fr_bytes(ev[2]), rs[:32], rs[32:])
    assert proof.hex() == want["proof"]
    bzk.params_free(ph)
