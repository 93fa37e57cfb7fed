"""CPU suite: the committed fixtures of tests/golden/ (made by tests/golden/make_vectors.py) are reproduced by the oracle
(C++ and pure Python) and by the host half of libbzk - without the reference tree, so this also runs on the GPU box."""
import hashlib
import json
import os

from bazuka_amd import lib as L
from util import fr_bytes, fr_list, rand_scalars_bytes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(G, "reference_vectors.json")))
ORC = json.load(open(os.path.join(G, "oracle_vectors.json")))


def test_reference_poseidon_kats_everywhere(co, pr):
    for k in range(1, 17):
        want = int(REF["poseidon_kat"][k - 1], 16)
        inp = fr_bytes(list(range(k)))
        assert pr.poseidon(list(range(k))) == want
        assert pr.fr_from_mont_bytes(co.poseidon_batch(inp, k)) == want
        assert pr.fr_from_mont_bytes(L.host_poseidon(inp)) == want


def test_reference_verifying_keys_decode(co, pr):
    for h in REF["verifying_keys_bincode_hex"]:
        b = bytes.fromhex(h)
        vk = pr.vk_from_bytes(b)
        assert len(vk["ic"]) == 6 and pr.vk_to_bytes(vk) == b
        assert all(pr.g1_on_curve(vk[k]) for k in ("alpha_g1", "beta_g1", "delta_g1"))
        assert all(pr.g2_on_curve(vk[k]) for k in ("beta_g2", "gamma_g2", "delta_g2"))
    # alpha, beta, gamma, delta are shared by the three circuits (first 878 bytes identical)
    assert len({h[:2 * 870] for h in REF["verifying_keys_bincode_hex"]}) == 1


def test_reference_jubjub_case(pr):
    j = REF["jubjub_abc"]
    key = L.host_jubjub_keys(b"ABC")
    U = pr.fr_from_mont_bytes
    assert [hex(U(key[:32])), hex(U(key[32:64]))] == j["pub"]
    sig = L.host_jubjub_sign(key, pr.fr_to_mont_bytes(j["message"]))
    assert [hex(U(sig[:32])), hex(U(sig[32:64]))] == j["sig_r"] and hex(U(sig[64:])) == j["sig_s"]
    assert L.host_jubjub_verify(key[:64], pr.fr_to_mont_bytes(j["message"]), sig)


def test_oracle_vectors_reproduced_by_the_oracle(co):
    nt = co.ncpu()
    for arity in (1, 2, 4, 7, 16):
        inp = fr_bytes(fr_list(arity * 5, 1000 + arity))
        assert co.poseidon_batch(inp, arity).hex() == ORC[f"poseidon_arity{arity}_seed{1000 + arity}_n5"]
    assert co.merkle4_root(fr_bytes(fr_list(64, 2000)), 3).hex() == ORC["merkle4_log3_seed2000"]
    data = fr_bytes(fr_list(64, 3000))
    for inv in (0, 1):
        for cs in (0, 1):
            assert hashlib.sha256(co.ntt(data, 6, bool(inv), bool(cs))).hexdigest() == ORC[f"ntt_log6_seed3000_inv{inv}_coset{cs}"]
    sc = rand_scalars_bytes(300, 4001)
    assert co.msm_g1(co.g1_bases(4000, 0, 300, nthreads=nt), sc, nthreads=nt).hex() == ORC["msm_g1_bases4000_scalars4001_n300"]
    assert co.msm_g2(co.g2_bases(4000, 0, 300, nthreads=nt), sc, nthreads=nt).hex() == ORC["msm_g2_bases4000_scalars4001_n300"]


def test_poseidon_vectors_host_half():
    for arity in (1, 2, 4, 7, 16):
        inp = fr_bytes(fr_list(arity * 5, 1000 + arity))
        got = b"".join(L.host_poseidon(inp[32 * arity * i:32 * arity * (i + 1)]) for i in range(5))
        assert got.hex() == ORC[f"poseidon_arity{arity}_seed{1000 + arity}_n5"]
