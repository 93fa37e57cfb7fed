"""Row a8 in the product: bzk_groth16_verify (host C++, bazuka_amd/csrc/host_pairing.hip) = `groth16_verify`
(src/zk/groth16/mod.rs:67-121) against the oracle's Python verifier on oracle-made proofs (C++ oracle setup + prove on the CPU):
same verdict on a valid proof, a wrong public input, a tampered proof, swapped proof elements, an off-curve point; the reference's
own acceptance case - the empty 4-slot Update circuit with inputs [456, 0, 123, H2(1, 0), 123] (src/mpn/circuits/test.rs:117-149)."""
import time

import pytest

from bazuka_amd import lib as L
from oracle import pyref as pr
from util import fr_bytes, fr_list, log2_ceil, r1cs_to_csr, synth_r1cs

F = pr.fr_to_mont_bytes


def _vk_bytes(params):
    return params["vk"] + (len(params["ic"]) // 97).to_bytes(8, "little") + params["ic"]


@pytest.mark.parametrize("n_mul,n_in", [(40, 3), (300, 6)])
def test_same_verdict_as_the_oracle_verifier(co, n_mul, n_in):
    r1 = synth_r1cs(n_mul, n_in=n_in, seed=500 + n_mul)
    A, B, Cm = r1cs_to_csr(co, r1)
    params = co.groth16_setup(A, B, Cm, r1["n_in"], r1["n_aux"], log2_ceil(len(r1["rows"])), fr_bytes(fr_list(5, 77)))
    zb = fr_bytes(r1["z"])
    az, bz, cz = co.r1cs_eval(A, B, Cm, zb)
    rs = fr_bytes(fr_list(2, 9))
    proof = co.groth16_prove(params, zb, az, bz, cz, rs[:32], rs[32:])
    vkb = _vk_bytes(params)
    vk = pr.vk_from_bytes(vkb)
    pub = r1["z"][1:n_in]
    inputs = fr_bytes(pub)
    t0 = time.perf_counter()
    assert L.groth16_verify(vkb, inputs, proof) is True
    dt = time.perf_counter() - t0
    assert pr.groth16_verify(vk, pub, pr.proof_from_bytes(proof))
    cases = {
        "wrong input": (vkb, fr_bytes([pub[0] + 1] + pub[1:]), proof),
        "a and c swapped": (vkb, inputs, proof[290:387] + proof[97:290] + proof[0:97]),
        "tampered c.x": (vkb, inputs, proof[:290] + bytes([proof[290] ^ 1]) + proof[291:]),      # off the curve
        "identity a": (vkb, inputs, pr.g1_to_bytes(None) + proof[97:]),
        "fewer inputs": (vkb, inputs[:-32], proof),
        # the same point with x's Montgomery limbs raised by p: not a value Fp([u64; 6]) may hold - refused, never computed with
        "limbs >= p": (vkb, inputs, (int.from_bytes(proof[:48], "little") + pr.P_MOD).to_bytes(48, "little") + proof[48:]),
    }
    for name, (v, i, p) in cases.items():
        assert L.groth16_verify(v, i, p) is False, name
    # a DIFFERENT valid proof of the same statement (other r, s) also verifies; negating a and b together does too (e(-A, -B) = e(A, B))
    rs2 = fr_bytes(fr_list(2, 10))
    assert L.groth16_verify(vkb, inputs, co.groth16_prove(params, zb, az, bz, cz, rs2[:32], rs2[32:]))
    a, b = pr.g1_from_bytes(proof[:97]), pr.g2_from_bytes(proof[97:290])
    neg = pr.g1_to_bytes((a[0], (-a[1]) % pr.P_MOD)) + pr.g2_to_bytes((b[0], ((-b[1][0]) % pr.P_MOD, (-b[1][1]) % pr.P_MOD))) + proof[290:]
    assert L.groth16_verify(vkb, inputs, neg) and pr.groth16_verify(vk, pub, pr.proof_from_bytes(neg))
    assert dt < 2.0


def test_the_references_own_acceptance_case(co):
    """src/mpn/circuits/test.rs:117-149: MpnCircuit::empty(3, 3, 1) with commitment 456, height 0, state = next_state = 123,
    aux = H2(fee_token Ziesha = 1, fee sum 0): proved on the CPU oracle, accepted by the product verifier"""
    aux = pr.poseidon([1, 0])
    r = L.mpn_update_empty(3, 3, 1, F(456), 0, F(123), F(aux), F(123), F(1), record_matrices=True)
    assert r.satisfied
    csr = [co.CsrHolder(r.n_constraints, list(memoryview(r.view("rp" + w)).cast("I")), list(memoryview(r.view("col" + w)).cast("I")), r.view("val" + w))
           for w in "ABC"]
    params = co.groth16_setup(*csr, r.n_in, r.n_aux, 17, fr_bytes(fr_list(5, 123)), nthreads=co.ncpu())
    z = r.view("z")
    rs = fr_bytes(fr_list(2, 11))
    proof = co.groth16_prove(params, z, r.view("az"), r.view("bz"), r.view("cz"), rs[:32], rs[32:], nthreads=co.ncpu())
    vkb = _vk_bytes(params)
    inputs = F(456) + F(0) + F(123) + F(aux) + F(123)
    assert z[32:192] == inputs
    assert L.groth16_verify(vkb, inputs, proof)
    assert not L.groth16_verify(vkb, F(456) + F(1) + F(123) + F(aux) + F(123), proof)   # another height
    # the three hard-coded verifying keys of the reference decode (src/config/blockchain.rs:32-37): a proof for another key is refused
    import json, os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for hexvk in json.load(open(os.path.join(G, "reference_vectors.json")))["verifying_keys_bincode_hex"]:
        assert L.groth16_verify(bytes.fromhex(hexvk), inputs, proof) is False


def test_mpn_work_verify_binds_the_prover(co):
    """bzk_mpn_work_verify = `MpnWork::verify` (src/mpn/mod.rs:281-295): a deposit work proved on the CPU oracle against an oracle-made key is
    accepted for its prover and refused for another address, another work, a tampered proof"""
    ZIESHA = F(1)
    alice, mallory = bytes(range(1, 33)), bytes(range(101, 133))
    shape = L.mpn_circuit_empty(0, 3, 3, 1, bytes(32), 0, bytes(32), bytes(32), bytes(32), record_matrices=True)
    csr = [co.CsrHolder(shape.n_constraints, list(memoryview(shape.view("rp" + w)).cast("I")), list(memoryview(shape.view("col" + w)).cast("I")),
                        shape.view("val" + w)) for w in "ABC"]
    params = co.groth16_setup(*csr, shape.n_in, shape.n_aux, 16, fr_bytes(fr_list(5, 321)), nthreads=co.ncpu())
    vkb = _vk_bytes(params)
    w = L.MpnWorld(3, 3)
    w.add_account(0, b"acct0", ZIESHA, 10 ** 9)
    w.add_key(5, b"newcomer")
    w.set_height(3)
    w.push_deposit(0, ZIESHA, 10)
    w.push_deposit(5, ZIESHA, 20)
    work = L.MpnWork.decode(w.make_work(0, [vkb, vkb, vkb], 77).encode())
    r = work.synthesize(alice)
    assert r.satisfied
    rs = fr_bytes(fr_list(2, 12))
    proof = co.groth16_prove(params, r.view("z"), r.view("az"), r.view("bz"), r.view("cz"), rs[:32], rs[32:], nthreads=co.ncpu())
    assert work.verify(alice, proof) is True
    assert work.verify(mallory, proof) is False                       # the commitment binds the proof to its prover
    assert work.verify(alice, proof[:100] + bytes([proof[100] ^ 1]) + proof[101:]) is False
    w.push_deposit(0, ZIESHA, 11)
    other = L.MpnWork.decode(w.make_work(0, [vkb, vkb, vkb], 77).encode())
    assert other.verify(alice, proof) is False                        # another state transition


def test_edge_scalars_in_the_public_input_combination(co):
    """X = IC_0 + sum x_i IC_i runs through shared 4-bit windows in the product: edge scalars (0, 1, r - 1, single bits, all-ones nibbles)
    must give the point the oracle's arithmetic gives.  For a valid (vk, x, proof) and any other input vector x', the key whose IC_0 is moved
    by sum (x_i - x'_i) IC_i has the same X under x' - so it must still verify, and must not with one scalar changed."""
    r1 = synth_r1cs(60, n_in=6, seed=900)
    A, B, Cm = r1cs_to_csr(co, r1)
    params = co.groth16_setup(A, B, Cm, r1["n_in"], r1["n_aux"], log2_ceil(len(r1["rows"])), fr_bytes(fr_list(5, 78)))
    zb = fr_bytes(r1["z"])
    az, bz, cz = co.r1cs_eval(A, B, Cm, zb)
    rs = fr_bytes(fr_list(2, 11))
    proof = co.groth16_prove(params, zb, az, bz, cz, rs[:32], rs[32:])
    vkb = _vk_bytes(params)
    x = r1["z"][1:6]
    assert L.groth16_verify(vkb, fr_bytes(x), proof)
    ic = [pr.g1_from_bytes(vkb[878 + 97 * i:878 + 97 * (i + 1)]) for i in range(6)]
    R = pr.R_MOD
    for xp in ([0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [R - 1, R - 1, 0, 1, R - 1], [1 << 252, (1 << 254) % R, 15, 0xF0F0F0F0, (1 << 128) - 1],
               [int("f" * 63, 16) % R, int("8" * 63, 16) % R, 16, 1 << 4, R - 16]):
        moved = ic[0]
        for i in range(5):
            moved = pr.g1_add(moved, pr.g1_mul(ic[i + 1], (x[i] - xp[i]) % R))
        vk2 = vkb[:878] + pr.g1_to_bytes(moved) + vkb[878 + 97:]
        assert L.groth16_verify(vk2, fr_bytes(xp), proof) is True, xp
        bumped = list(xp)
        bumped[3] = (bumped[3] + 1) % R
        assert L.groth16_verify(vk2, fr_bytes(bumped), proof) is False, xp
    # an input whose Montgomery limbs are not below r is not a ZkScalar: refused even where its residue would verify
    over = (int.from_bytes(fr_bytes(x[:1]), "little") + pr.R_MOD).to_bytes(32, "little") + fr_bytes(x[1:])
    assert L.groth16_verify(vkb, over, proof) is False
    # an IC point at infinity contributes nothing whatever its scalar
    inf = pr.g1_to_bytes(None)
    moved = pr.g1_add(ic[0], pr.g1_mul(ic[2], x[1]))                       # fold input 2 into IC_0, blank its IC
    vk3 = vkb[:878] + pr.g1_to_bytes(moved) + vkb[975:1072] + inf + vkb[1169:]
    assert L.groth16_verify(vk3, fr_bytes([x[0], 12345] + x[2:]), proof) is True
