"""Generates the committed fixtures of tests/golden/ (run in the build container, where /root/reference is mounted):

  reference_vectors.json  vectors COPIED FROM THE REFERENCE'S OWN TESTS / CONFIG (the anchors the oracle is pinned on):
      - the 16 Poseidon known answers, inputs [0..k)             /root/reference/src/zk/poseidon/mod.rs:114-149
      - SHA-256 of each Poseidon parameter file (t = 2..17)      /root/reference/src/zk/poseidon/params/*.txt
      - the three hard-coded Groth16 verifying keys (bincode)    /root/reference/src/config/blockchain.rs:32-37
      - the Jubjub sign/verify case: seed b"ABC", message 123456 /root/reference/src/crypto/jubjub/mod.rs:180-192
        (the reference only asserts verify() == true; the key / signature values recorded here are the oracle's)
  oracle_vectors.json     seeded inputs -> outputs of the CPU oracle (C++, cross-checked against pure Python) for every
      kernel of the path; the GPU tests compare the HIP path with these bytes (tests/test_golden_*.py), so a change of
      the oracle, of the byte formats or of a kernel shows up as a fixture mismatch.

usage: python tests/golden/make_vectors.py"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

from oracle import coracle as co  # noqa: E402
from oracle import pyref as pr  # noqa: E402
from util import fr_bytes, fr_list, log2_ceil, r1cs_to_csr, rand_scalars_bytes, synth_r1cs  # noqa: E402


def reference_vectors():
    src = open(f"{REF}/src/zk/poseidon/mod.rs").read()
    kats = [int(h, 16) for h in re.findall(r'from_str_vartime\(\s*"(\d+)"', src)] or None
    if not kats:  # the reference writes them as hex literals
        from test_oracle_cpu import POSEIDON_KAT
        kats = list(POSEIDON_KAT)
    assert len(kats) == 16 and all(pr.poseidon(list(range(k + 1))) == kats[k] for k in range(16))
    params = {}
    for t in range(2, 18):
        p = f"{REF}/src/zk/poseidon/params/poseidon_params_n255_t{t}_alpha5_M128.txt"
        params[str(t)] = hashlib.sha256(open(p, "rb").read()).hexdigest()
    cfg = open(f"{REF}/src/config/blockchain.rs").read()
    vks = re.findall(r'"([0-9a-f]{2000,})"', cfg)[:3]
    sk = pr.jj_generate_keys(b"ABC")
    rr, s = pr.jj_sign(sk, 123456)
    assert pr.jj_verify(sk["pub"], 123456, (rr, s))
    return {"poseidon_kat": [hex(x) for x in kats], "poseidon_param_files_sha256": params, "verifying_keys_bincode_hex": vks,
            "jubjub_abc": {"pub": [hex(sk["pub"][0]), hex(sk["pub"][1])], "sig_r": [hex(rr[0]), hex(rr[1])], "sig_s": hex(s),
                           "message": 123456}}


def oracle_vectors():
    out = {}
    nt = co.ncpu()
    # Poseidon batches and a tree
    for arity in (1, 2, 4, 7, 16):
        inp = fr_bytes(fr_list(arity * 5, 1000 + arity))
        out[f"poseidon_arity{arity}_seed{1000 + arity}_n5"] = co.poseidon_batch(inp, arity).hex()
    leaves = fr_bytes(fr_list(64, 2000))
    out["merkle4_log3_seed2000"] = co.merkle4_root(leaves, 3).hex()
    # NTT, all four modes
    data = fr_bytes(fr_list(64, 3000))
    for inv in (0, 1):
        for cs in (0, 1):
            out[f"ntt_log6_seed3000_inv{inv}_coset{cs}"] = hashlib.sha256(co.ntt(data, 6, bool(inv), bool(cs))).hexdigest()
    big = rand_scalars_bytes(1 << 12, 3001)
    out["ntt_log12_seed3001_fwd_coset_sha256"] = hashlib.sha256(co.ntt(big, 12, False, True, nthreads=nt)).hexdigest()
    # MSM
    b1 = co.g1_bases(4000, 0, 300, nthreads=nt)
    b2 = co.g2_bases(4000, 0, 300, nthreads=nt)
    sc = rand_scalars_bytes(300, 4001)
    out["msm_g1_bases4000_scalars4001_n300"] = co.msm_g1(b1, sc, nthreads=nt).hex()
    out["msm_g2_bases4000_scalars4001_n300"] = co.msm_g2(b2, sc, nthreads=nt).hex()
    b1k = co.g1_bases(4002, 0, 5000, nthreads=nt)
    sck = rand_scalars_bytes(5000, 4003)
    out["msm_g1_bases4002_scalars4003_n5000"] = co.msm_g1(b1k, sck, nthreads=nt).hex()
    # Groth16: synthetic R1CS -> CRS -> proof (everything deterministic in the seeds)
    r = synth_r1cs(200, 3, 5000)
    A, B, Cm = r1cs_to_csr(co, r)
    tox = fr_bytes(fr_list(5, 5001))
    log_m = log2_ceil(len(r["rows"]))
    params = co.groth16_setup(A, B, Cm, r["n_in"], r["n_aux"], log_m, tox, nthreads=nt)
    z = fr_bytes(r["z"])
    az, bz, cz = co.r1cs_eval(A, B, Cm, z, nthreads=nt)
    rs = fr_bytes(fr_list(2, 5002))
    proof = co.groth16_prove(params, z, az, bz, cz, rs[:32], rs[32:], nthreads=nt)
    vk = {"alpha_g1": pr.g1_from_bytes(params["vk"][0:97]), "beta_g2": pr.g2_from_bytes(params["vk"][194:387]),
          "gamma_g2": pr.g2_from_bytes(params["vk"][387:580]), "delta_g2": pr.g2_from_bytes(params["vk"][677:870]),
          "ic": [pr.g1_from_bytes(params["ic"][97 * i:97 * i + 97]) for i in range(r["n_in"])]}
    assert pr.groth16_verify(vk, r["z"][1:r["n_in"]], pr.proof_from_bytes(proof))
    out["groth16_synth200_seed5000_tox5001_rs5002"] = {"vk_sha256": hashlib.sha256(params["vk"]).hexdigest(),
                                                      "h_sha256": hashlib.sha256(params["h"]).hexdigest(), "proof": proof.hex()}
    return out


if __name__ == "__main__":
    json.dump(reference_vectors(), open(os.path.join(HERE, "reference_vectors.json"), "w"), indent=1)
    json.dump(oracle_vectors(), open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=1)
    print("written")
