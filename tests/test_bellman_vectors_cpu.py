"""Consumer of tests/golden/bellman_vectors.json - vectors made by the REAL bellman 0.14 running the reference's own circuits with seeded
rngs (rust/fixture-dump).  The image that builds libbzk has no Rust toolchain, so the file may be absent: the tests that
need it then SKIP with one line, and a rehearsal feeds the same consumer code with a document of the same schema made from the
independent Python restatement (oracle/pycircuit.py) - that exercises the encodings and the plumbing, it pins nothing on bellman.

What the real file pins (CPU half): for UpdateCircuit / DepositCircuit / WithdrawCircuit at (3, 3, 1) with the public inputs of
/root/reference/src/mpn/circuits/test.rs:117-237 - the host witness generator's z, A.z, B.z, C.z and the density vectors hash to what
bellman's ProvingAssignment saw (sizes included: the trailing `input * 0 = 0` rows, the ONE input).  GPU half (needs the parameter
files, BZK_BELLMAN_PARAMS_DIR): proof bytes == bellman's on the same CRS, witness, r, s."""
import hashlib
import json
import os

import pytest

from bazuka_amd import lib as L
from oracle import pyref as pr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VECTORS = os.path.join(ROOT, "tests", "golden", "bellman_vectors.json")
F = pr.fr_to_mont_bytes
KIND = {"update_3_3_1": 2, "deposit_3_3_1": 0, "withdraw_3_3_1": 1}
R_INV = pow(1 << 256, -1, pr.R_MOD)


def _canon(mont_bytes: bytes) -> bytes:
    """n x 32-byte Montgomery limbs -> n x `Scalar::to_bytes` (canonical, little-endian)"""
    out = bytearray(len(mont_bytes))
    for i in range(0, len(mont_bytes), 32):
        out[i:i + 32] = (int.from_bytes(mont_bytes[i:i + 32], "little") * R_INV % pr.R_MOD).to_bytes(32, "little")
    return bytes(out)


def encode_like_the_dumper(n_in, views):
    """sha256 of the arrays in rust/fixture-dump's encodings, from libbzk's layout (z = inputs then aux, densities over inputs + aux)"""
    sha = lambda b: hashlib.sha256(b).hexdigest()
    a_d, b_d = views["a_density"], views["b_density"]
    return {"z": sha(_canon(views["z"])), "az": sha(_canon(views["az"])), "bz": sha(_canon(views["bz"])), "cz": sha(_canon(views["cz"])),
            "a_aux_density": sha(a_d[n_in:]), "b_input_density": sha(b_d[:n_in]), "b_aux_density": sha(b_d[n_in:])}


def product_instance(name, public_le):
    com, height, st, aux, nxt = [int.from_bytes(bytes.fromhex(h), "little") for h in public_le]
    args = (F(com), height, F(st), F(aux), F(nxt))
    if KIND[name] == 2:
        return L.mpn_update_empty(3, 3, 1, *args, F(1), record_matrices=True)   # fee_token: ContractId::Ziesha -> 1; densities come with the matrices
    return L.mpn_circuit_empty(KIND[name], 3, 3, 1, *args, record_matrices=True)


def check_r1cs_against(doc):
    for c in doc["circuits"]:
        r = product_instance(c["circuit"], c["public_inputs_le"])
        try:
            assert r.satisfied
            assert (r.n_in, r.n_aux, r.n_constraints) == (c["n_inputs"], c["n_aux"], c["n_constraints"]), c["circuit"]
            got = encode_like_the_dumper(r.n_in, {k: r.view(k) for k in ("z", "az", "bz", "cz", "a_density", "b_density")})
            for key, want in c["sha256"].items():
                assert got[key] == want, (c["circuit"], key)
        finally:
            r.free()


def test_rehearsal_with_the_python_restatement_standing_in_for_bellman():
    """the consumer above on a document of the dumper's schema made from oracle/pycircuit.py (NOT bellman): encodings, density split,
    size bookkeeping and the public-input plumbing all run on every CPU pass"""
    from oracle import pycircuit as pc
    com, st = 456, 123
    docs = []
    for name, cs in (("update_3_3_1", pc.update_circuit(3, 3, com, 0, st, pr.poseidon([1, 0]), st, 1, [pc.null_update_transition(3, 3)] * 4)),):
        v = pc.all_views(cs)
        n_in = 6
        aux = pr.poseidon([1, 0])
        docs.append({"circuit": name, "public_inputs_le": [x.to_bytes(32, "little").hex() for x in (com, 0, st, aux, st)],
                     "n_inputs": n_in, "n_aux": len(v["z"]) // 32 - n_in, "n_constraints": len(v["az"]) // 32,
                     "sha256": encode_like_the_dumper(n_in, v)})
    check_r1cs_against({"circuits": docs})
    # and the encoding itself on known values: Montgomery 1 -> canonical 1, little-endian
    assert _canon(F(1) + F(pr.R_MOD - 1)) == (1).to_bytes(32, "little") + (pr.R_MOD - 1).to_bytes(32, "little")


def test_r1cs_hashes_equal_bellmans():
    if not os.path.exists(VECTORS):
        pytest.skip("tests/golden/bellman_vectors.json absent: made by `cargo run --release` in rust/fixture-dump (needs a Rust toolchain)")
    check_r1cs_against(json.load(open(VECTORS)))


def _g1_uncompressed(b97: bytes) -> bytes:
    if b97[96]:
        return bytes([0x40]) + bytes(95)
    return b"".join(pr.fp_from_mont_bytes(b97[o:o + 48]).to_bytes(48, "big") for o in (0, 48))


def _g2_uncompressed(b193: bytes) -> bytes:
    if b193[192]:
        return bytes([0x40]) + bytes(191)
    x0, x1, y0, y1 = (pr.fp_from_mont_bytes(b193[o:o + 48]).to_bytes(48, "big") for o in (0, 48, 96, 144))
    return x1 + x0 + y1 + y0   # bls12_381 writes the c1 coefficient first


@pytest.mark.gpu
def test_proof_bytes_equal_bellmans(bzk):
    pdir = os.environ.get("BZK_BELLMAN_PARAMS_DIR")
    if not os.path.exists(VECTORS) or not pdir:
        pytest.skip("needs tests/golden/bellman_vectors.json and BZK_BELLMAN_PARAMS_DIR (rust/fixture-dump --params-dir)")
    for c in json.load(open(VECTORS))["circuits"]:
        blob = open(os.path.join(pdir, c["circuit"] + ".params"), "rb").read()
        assert hashlib.sha256(blob).hexdigest() == c["setup"]["parameters_write_sha256"]
        r = product_instance(c["circuit"], c["public_inputs_le"])
        ph = bzk.params_load_bellman(blob, r.n_in, r.n_aux, r.view("a_density"), r.view("b_density"))
        rs = [F(int.from_bytes(bytes.fromhex(c["proof"][k]), "little")) for k in ("r_le", "s_le")]
        proof = bzk.groth16_prove(ph, r.raw("z"), r.raw("az"), r.raw("bz"), r.raw("cz"), rs[0], rs[1])
        assert _g1_uncompressed(proof[:97]).hex() == c["proof"]["a"], c["circuit"]
        assert _g2_uncompressed(proof[97:290]).hex() == c["proof"]["b"], c["circuit"]
        assert _g1_uncompressed(proof[290:]).hex() == c["proof"]["c"], c["circuit"]
        bzk.params_free(ph)
        r.free()
