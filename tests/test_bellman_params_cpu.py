"""bellman `Parameters` reader / writer (bzk_bellman_params_*, host code) against the oracle's WRITER of the same format
(oracle/pyref.py::bellman_params_bytes, restated from the published bellman 0.14 / bls12_381 0.8 crates - no Rust here, so the
format itself is `parity unpinned`; what IS checked: two independent implementations of the layout agree byte for byte in both
directions, every decoded point lands in the reference's raw Montgomery form, and malformed input is refused)."""
import pytest

from bazuka_amd import lib as L
from oracle import pyref as pr
from util import synth_r1cs


def _params():
    r1 = synth_r1cs(6, n_in=3, seed=77)
    r = pr.R1CS(r1["n_in"], r1["n_aux"], r1["rows"][:-r1["n_in"]])
    return pr.groth16_setup(r, 11111, 22222, 33333, 44444, 55555)


def test_oracle_written_parameters_decode_to_the_raw_forms():
    p = _params()
    vk = {k: p[k] for k in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2", "ic")}
    blob = pr.bellman_params_bytes(vk, p["h"], p["l"], p["a"], p["b_g1"], p["b_g2"])
    d = L.bellman_params_decode(blob + b"tail")       # trailing bytes are the caller's business: `consumed` says where the blob ends
    assert d["consumed"] == len(blob)
    assert (d["n_ic"], d["n_h"], d["n_l"], d["n_a"], d["n_b_g1"], d["n_b_g2"]) == (3, len(p["h"]), len(p["l"]), len(p["a"]), len(p["b_g1"]), len(p["b_g2"]))
    want_vk = (pr.g1_to_bytes(p["alpha_g1"]) + pr.g1_to_bytes(p["beta_g1"]) + pr.g2_to_bytes(p["beta_g2"]) + pr.g2_to_bytes(p["gamma_g2"]) +
               pr.g1_to_bytes(p["delta_g1"]) + pr.g2_to_bytes(p["delta_g2"]))
    assert d["vk"] == want_vk
    assert d["ic"] == b"".join(pr.g1_to_bytes(q) for q in p["ic"])
    for k in ("h", "l", "a", "b_g1"):
        assert d[k] == b"".join(pr.g1_raw96(q) for q in p[k]), k
    assert d["b_g2"] == b"".join(pr.g2_raw192(q) for q in p["b_g2"])
    # and back: the product's writer reproduces the oracle writer's bytes
    assert L.bellman_params_encode(d["vk"], d["ic"], d["h"], d["l"], d["a"], d["b_g1"], d["b_g2"]) == blob
    # the vk the file carries is the one the oracle verifier accepts proofs under
    z = [1, 5, 7] + [0] * p["n_aux"]  # not a witness - only the encoding of ic is at stake here
    assert pr.vk_from_bytes(d["vk"] + (3).to_bytes(8, "little") + d["ic"])["ic"] == p["ic"] and len(z) == 3 + p["n_aux"]


def test_infinity_and_malformed_points():
    p = _params()
    vk = {k: p[k] for k in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2", "ic")}
    args = (p["h"], p["l"], p["a"], p["b_g1"], p["b_g2"])
    good = pr.bellman_params_bytes(vk, *args)
    # an ic entry at infinity is legal (a public input that no constraint uses) and decodes to bls12_381's (0, 1, inf) form
    vk_inf = dict(vk, ic=[vk["ic"][0], None, vk["ic"][2]])
    d = L.bellman_params_decode(pr.bellman_params_bytes(vk_inf, *args))
    assert d["ic"][97:194] == pr.g1_to_bytes(None)
    assert L.bellman_params_encode(d["vk"], d["ic"], d["h"], d["l"], d["a"], d["b_g1"], d["b_g2"]) == pr.bellman_params_bytes(vk_inf, *args)
    # ... but not inside a query (Parameters::read: "point at infinity")
    with pytest.raises(L.BzkError):
        L.bellman_params_decode(pr.bellman_params_bytes(vk, [None] + p["h"][1:], *args[1:]))
    off_h = 3 * 96 + 3 * 192 + 4 + 96 * 3 + 4
    for mutate in (lambda b: b[:off_h] + bytes([b[off_h] | 0x80]) + b[off_h + 1:],        # compression flag
                   lambda b: b[:off_h] + bytes([b[off_h] | 0x20]) + b[off_h + 1:],        # sort flag
                   lambda b: b[:off_h + 95] + bytes([b[off_h + 95] ^ 1]) + b[off_h + 96:],  # y off the curve
                   lambda b: b[:off_h] + bytes([0x1f]) + b"\xff" * 47 + b[off_h + 48:],    # x >= p
                   lambda b: b[:-1],                                                       # truncated
                   lambda b: b[:off_h - 4] + (2 ** 31).to_bytes(4, "big") + b[off_h:]):    # length beyond the input
        with pytest.raises(L.BzkError):
            L.bellman_params_decode(mutate(good))
    assert L.bellman_params_decode(good)["n_h"] == len(p["h"])


def test_reader_survives_mutated_files():
    """random byte flips, truncations, splices and length-field edits of a valid file: decoded (and then re-encodable) or refused with an error -
    never a crash, never a read beyond the input (the CPU suite also runs against an ASAN build: tools/sanitize_cpu.sh)"""
    import random
    rnd = random.Random(4242)
    p = _params()
    vk = {k: p[k] for k in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2", "ic")}
    good = pr.bellman_params_bytes(vk, p["h"], p["l"], p["a"], p["b_g1"], p["b_g2"])
    decoded = refused = 0
    for it in range(400):
        b = bytearray(good)
        mode = it % 5
        if mode == 0:
            for _ in range(rnd.randint(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            b = b[:rnd.randrange(len(b))]
        elif mode == 2:
            i, j = sorted(rnd.randrange(len(b)) for _ in range(2))
            b = b[:i] + b[j:]
        elif mode == 3:
            i = rnd.randrange(len(b) - 4)
            b[i:i + 4] = rnd.choice([0, 1, 2 ** 31, 2 ** 32 - 1, rnd.randrange(2 ** 32)]).to_bytes(4, "big")
        else:
            i = rnd.randrange(len(b))
            b = b[:i] + bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 200))) + b[i:]
        try:
            d = L.bellman_params_decode(bytes(b))
        except L.BzkError:
            refused += 1
            continue
        decoded += 1
        assert d["consumed"] <= len(b)
        assert L.bellman_params_encode(d["vk"], d["ic"], d["h"], d["l"], d["a"], d["b_g1"], d["b_g2"]) == bytes(b[:d["consumed"]])
    assert refused > 200 and decoded + refused == 400
