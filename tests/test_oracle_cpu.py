"""CPU suite (-m "not gpu"): pins the oracle against the reference's own vectors and cross-checks the
C++ oracle with the independent pure-Python restatement."""
import os
import random

import pytest

from util import fr_bytes, fr_list

REF = "/root/reference"

# /root/reference/src/zk/poseidon/mod.rs:116-133 (inputs [0..k), k = 1..16)
POSEIDON_KAT = [
    27570695323925995271701303589514430472678239829854264417883970952440292573348,
    6587584068506488869767403662460111870851709789694140241572542699619538605403,
    11065162352055215342882956665028806373710857144056793315618843991574034541745,
    27235437669367044799899874028200860893259633691548428184978833555844239099210,
    39122459949963443953695513827515422590145971775731164693081784821001500765271,
    14822541353598610072073758561600133199190898904019472753356348939736178856242,
    32119039894111509393883349238591117345166479914896997011437787663480858229324,
    43492451727584886720328582747486156090763899250669626113572962177392830153672,
    23782521420058920239581486714235942233162905749917547091367129332109148150964,
    1950261058989975858181381159018748926889722679795466088362775920975943983890,
    47763254094198808066374497304963224993617822320088130264863862435119574697678,
    44035521596650126254580286193043646937530018324533162959282567836364656349620,
    45248278075433906869650374149660178834237900630357739057386839430392516698709,
    30558481537294127342952125056358924225581206938869947160862017954746718634085,
    10702554392571105609953066033536365418563149392782994983402406449789876497692,
    34319425623279664398659085846739236990635100324667226409415519671072072962346,
]


def test_poseidon_kat_python(pr):
    for k in (1, 2, 4, 5, 7, 16):
        assert pr.poseidon(list(range(k))) == POSEIDON_KAT[k - 1]


def test_poseidon_kat_c(co, pr):
    for k in range(1, 17):
        out = co.poseidon_batch(fr_bytes(range(k)), k)
        assert pr.fr_from_mont_bytes(out) == POSEIDON_KAT[k - 1], k


def test_poseidon_reflects_changes(co, pr):
    # mirrors test_hash_reflects_changes (src/zk/poseidon/mod.rs:100-111)
    for arity in range(1, 17):
        vals = [0] * arity
        base = co.poseidon_batch(fr_bytes(vals), arity)
        for i in range(arity):
            vals[i] = 1
            assert co.poseidon_batch(fr_bytes(vals), arity) != base


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_poseidon_params_match_reference_files(co, pr):
    """Grain-LFSR re-derivation == the reference's params/*.txt for every width."""
    import re
    for t in range(2, 18):
        path = f"{REF}/src/zk/poseidon/params/poseidon_params_n255_t{t}_alpha5_M128.txt"
        lines = open(path).read().splitlines()
        rc = [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", lines[3])]
        mds = [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", lines[15])]
        blob = co.poseidon_params(t)
        got = [pr.fr_from_mont_bytes(blob[32 * i:32 * i + 32]) for i in range(len(blob) // 32)]
        assert got[: len(rc)] == rc, t
        assert got[len(rc):] == mds, t
        if t <= 6:
            prc, pmds = pr.poseidon_params(t)
            assert prc == rc and [x for row in pmds for x in row] == mds


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")
def test_reference_vk_blobs_decode_on_curve(co, pr):
    """The three hard-coded VKs (src/config/blockchain.rs:32-37) decode, under the Montgomery x|y|inf
    layout the oracle and libbzk use, to points on G1 / G2 with ic.len() == 6."""
    import re
    src = open(f"{REF}/src/config/blockchain.rs").read()
    blobs = [bytes.fromhex(h) for h in re.findall(r'"([0-9a-f]{2000,})"', src)]
    assert len(blobs) >= 3
    for b in blobs[:3]:
        assert len(b) == 1460
        vk = pr.vk_from_bytes(b)
        assert len(vk["ic"]) == 6
        for k in ("alpha_g1", "beta_g1", "delta_g1"):
            assert pr.g1_on_curve(vk[k]) and vk[k] is not None
        for k in ("beta_g2", "gamma_g2", "delta_g2"):
            assert pr.g2_on_curve(vk[k]) and vk[k] is not None
        assert all(pr.g1_on_curve(p) for p in vk["ic"])
        assert pr.vk_to_bytes(vk) == b
        assert co.g1_on_curve(b[0:97]) and co.g2_on_curve(b[194:387])


def test_field_ops_c_vs_python(co, pr):
    rnd = random.Random(1)
    for _ in range(50):
        a, b = rnd.randrange(pr.R_MOD), rnd.randrange(pr.R_MOD)
        A, B = pr.fr_to_mont_bytes(a), pr.fr_to_mont_bytes(b)
        assert pr.fr_from_mont_bytes(co.fr_op(2, A, B)) == a * b % pr.R_MOD
        assert pr.fr_from_mont_bytes(co.fr_op(0, A, B)) == (a + b) % pr.R_MOD
        assert pr.fr_from_mont_bytes(co.fr_op(1, A, B)) == (a - b) % pr.R_MOD
        assert pr.fr_from_mont_bytes(co.fr_op(3, A)) == pow(a, -1, pr.R_MOD)
        assert co.fr_op(4, A) == pr.fr_to_canon_bytes(a)
        a, b = rnd.randrange(pr.P_MOD), rnd.randrange(pr.P_MOD)
        A, B = pr.fp_to_mont_bytes(a), pr.fp_to_mont_bytes(b)
        assert pr.fp_from_mont_bytes(co.fp_op(2, A, B)) == a * b % pr.P_MOD
        assert pr.fp_from_mont_bytes(co.fp_op(3, A)) == pow(a, -1, pr.P_MOD)


def test_zkscalar_u64_roundtrip(co, pr):
    # mirrors src/zk/test/mod.rs:31-41 at the byte level: Montgomery(from u64) -> canonical == u64
    for v in (0, 1, 123456, 2**64 - 1):
        m = co.fr_op(5, v.to_bytes(32, "little"))
        assert int.from_bytes(co.fr_op(4, m), "little") == v


def test_msm_c_vs_python_and_naive(co, pr):
    n = 24
    bases = co.g1_bases(pr.SEED, 0, n)
    sc = fr_list(n)
    sc[0], sc[1], sc[2] = 0, 1, pr.R_MOD - 1
    scb = fr_bytes(sc)
    pts = [pr.g1_from_bytes(bases[96 * i:96 * i + 96] + b"\0") for i in range(n)]
    assert all(pr.g1_on_curve(p) for p in pts)
    want = pr.g1_to_bytes(pr.ec_msm(pr.FP, pts, sc))
    assert co.msm_g1(bases, scb) == want
    assert co.msm_g1(bases, scb, naive=True) == want
    assert co.msm_g1(bases, fr_bytes(sc, mont=False), mont=False) == want
    b2 = co.g2_bases(pr.SEED, 0, n)
    pts2 = [pr.g2_from_bytes(b2[192 * i:192 * i + 192] + b"\0") for i in range(n)]
    assert all(pr.g2_on_curve(p) for p in pts2)
    assert co.msm_g2(b2, scb) == pr.g2_to_bytes(pr.ec_msm(pr.FP2, pts2, sc))


def test_msm_pippenger_vs_naive_4096(co):
    n = 4096
    from util import rand_scalars_bytes
    bases = co.g1_bases(3, 0, n, nthreads=co.ncpu())
    sc = rand_scalars_bytes(n, 9)
    assert co.msm_g1(bases, sc, nthreads=co.ncpu()) == co.msm_g1(bases, sc, naive=True)
    import numpy as np  # the zero-copy entry the 2^24 / 2^26 GPU checks use
    assert co.msm_g1_np(np.frombuffer(bases, dtype=np.uint8), np.frombuffer(sc, dtype=np.uint8), nthreads=2) == co.msm_g1(bases, sc)


def test_msm_linearity(co, pr):
    n = 300
    bases = co.g1_bases(5, 0, n, nthreads=co.ncpu())
    s, t = fr_list(n, 1), fr_list(n, 2)
    u = [(a + b) % pr.R_MOD for a, b in zip(s, t)]
    ms, mt, mu = (co.msm_g1(bases, fr_bytes(x)) for x in (s, t, u))
    assert co.g1_add(ms, mt) == mu


def test_ntt_c_vs_python(co, pr):
    for lg in (0, 1, 2, 5, 8):
        v = fr_list(1 << lg, lg + 1)
        vb = fr_bytes(v)
        for inv in (False, True):
            for cs in (False, True):
                assert co.ntt(vb, lg, inv, cs) == fr_bytes(pr.ntt(v, lg, inv, cs)), (lg, inv, cs)


def test_ntt_roundtrip_and_identities(co, pr):
    lg = 12
    vb = fr_bytes(fr_list(1 << lg, 7))
    for cs in (False, True):
        assert co.ntt(co.ntt(vb, lg, False, cs, nthreads=4), lg, True, cs, nthreads=4) == vb
    w = pr.omega_for(lg)
    assert pow(w, 1 << lg, pr.R_MOD) == 1 and pow(w, 1 << (lg - 1), pr.R_MOD) != 1
    # delta at 0 -> all ones
    d = fr_bytes([1] + [0] * 15)
    assert co.ntt(d, 4) == fr_bytes([1] * 16)


def test_merkle_dense_vs_python_and_default_chain(co, pr):
    leaves = fr_list(64, 11)
    nodes = []
    root = pr.merkle4_root(leaves, 3, nodes)
    r, nd = co.merkle4_root(fr_bytes(leaves), 3, True)
    assert r == pr.fr_to_mont_bytes(root)
    assert nd == fr_bytes(nodes)
    # empty tree root == compress_default chain (src/zk/mod.rs:401-423): H(d,d,d,d) iterated from 0
    d = 0
    for _ in range(3):
        d = pr.poseidon([d] * 4)
    assert co.merkle4_root(fr_bytes([0] * 64), 3) == pr.fr_to_mont_bytes(d)


def test_pairing_bilinear(pr):
    e1 = pr.pairing(pr.g1_mul(pr.G1_GEN, 5), pr.g2_mul(pr.G2_GEN, 7))
    e2 = pr.f12_pow(pr.pairing(pr.G1_GEN, pr.G2_GEN), 35)
    assert e1 == e2 and e1 != pr.F12_ONE


def _toy_r1cs(pr):
    # x*x = y ; y*x = out ; (x+5)*1 = x+5      inputs: ONE, out ; aux: x, y
    return pr.R1CS(2, 2, [([(2, 1)], [(2, 1)], [(3, 1)]), ([(3, 1)], [(2, 1)], [(1, 1)]),
                          ([(2, 1), (0, 5)], [(0, 1)], [(2, 1), (0, 5)])])


def test_groth16_python_roundtrip(pr):
    r1 = _toy_r1cs(pr)
    z = [1, 27, 3, 9]
    assert r1.is_satisfied(z)
    rng = pr.SplitMix64()
    params = pr.groth16_setup(r1, *[rng.fr() for _ in range(5)])
    proof = pr.groth16_prove(r1, params, z, rng.fr(), rng.fr())
    assert pr.groth16_verify(params, [27], proof)
    assert not pr.groth16_verify(params, [28], proof)
    assert pr.proof_from_bytes(pr.proof_to_bytes(proof)) == proof
    bad = pr.groth16_prove(r1, params, [1, 27, 3, 10], 5, 6)
    assert not pr.groth16_verify(params, [27], bad)


def _csr(co, pr, r1, which):
    rp, col, val = [0], [], b""
    for cons in r1.constraints:
        for v, c in cons[which]:
            col.append(v)
            val += pr.fr_to_mont_bytes(c)
        rp.append(len(col))
    return co.CsrHolder(len(r1.constraints), rp, col, val)


def test_groth16_c_vs_python_bytes(co, pr):
    """C++ oracle setup+prove == Python setup+prove, byte for byte, on the same toxic waste / r / s."""
    r1 = _toy_r1cs(pr)
    z = [1, 27, 3, 9]
    rng = pr.SplitMix64(77)
    tox = [rng.fr() for _ in range(5)]
    r, s = rng.fr(), rng.fr()
    pp = pr.groth16_setup(r1, *tox)
    want = pr.proof_to_bytes(pr.groth16_prove(r1, pp, z, r, s))
    A, B, Cm = (_csr(co, pr, r1, k) for k in range(3))
    cp = co.groth16_setup(A, B, Cm, r1.n_in, r1.n_aux, r1.log_m(), fr_bytes(tox))
    assert cp["vk"][:97] == pr.g1_to_bytes(pp["alpha_g1"])
    assert cp["ic"] == b"".join(pr.g1_to_bytes(p) for p in pp["ic"])
    assert cp["h"] == b"".join(pr.g1_raw96(p) for p in pp["h"])
    assert cp["l"] == b"".join(pr.g1_raw96(p) for p in pp["l"])
    assert cp["a"] == b"".join(pr.g1_raw96(p) for p in pp["a"])
    assert cp["b_g2"] == b"".join(pr.g2_raw192(p) for p in pp["b_g2"])
    zb = fr_bytes(z)
    az, bz, cz = co.r1cs_eval(A, B, Cm, zb)
    got = co.groth16_prove(cp, zb, az, bz, cz, pr.fr_to_mont_bytes(r), pr.fr_to_mont_bytes(s))
    assert got == want
    assert pr.groth16_verify(pp, [27], pr.proof_from_bytes(got))
