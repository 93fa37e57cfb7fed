"""GPU parity: Groth16 prove (a4-a7) - proof bytes bit-exact vs the CPU oracle on the same
(CRS, witness, r, s); the oracle verifier (pairing check) accepts them.  Proof bytes are unpinned by
the reference (OsRng everywhere), so the oracle on identical inputs is the parity anchor."""
import pytest
import torch

from util import dev_bytes, fr_bytes, fr_list, log2_ceil, r1cs_to_csr, rand_scalars_bytes, synth_r1cs, to_dev

pytestmark = pytest.mark.gpu


def _setup(co, pr, n_mul, seed):
    r1 = synth_r1cs(n_mul, seed=seed)
    A, B, Cm = r1cs_to_csr(co, r1)
    log_m = log2_ceil(len(r1["rows"]))
    tox = fr_bytes(fr_list(5, seed + 1))
    params = co.groth16_setup(A, B, Cm, r1["n_in"], r1["n_aux"], log_m, tox, nthreads=co.ncpu())
    zb = fr_bytes(r1["z"])
    az, bz, cz = co.r1cs_eval(A, B, Cm, zb, nthreads=co.ncpu())
    return r1, params, zb, az, bz, cz


@pytest.mark.parametrize("n_mul", [5, 100, 2000])
def test_groth16_prove_bytes_vs_oracle(bzk, co, pr, n_mul):
    r1, params, zb, az, bz, cz = _setup(co, pr, n_mul, 1000 + n_mul)
    r, s = fr_bytes(fr_list(2, 5))[:32], fr_bytes(fr_list(2, 5))[32:]
    want = co.groth16_prove(params, zb, az, bz, cz, r, s, nthreads=co.ncpu())
    ph = bzk.params_load(params)
    got = bzk.groth16_prove(ph, zb, az, bz, cz, r, s)
    bzk.params_free(ph)
    assert got == want
    if n_mul <= 100:
        vk = {"alpha_g1": pr.g1_from_bytes(params["vk"][0:97]), "beta_g2": pr.g2_from_bytes(params["vk"][194:387]),
              "gamma_g2": pr.g2_from_bytes(params["vk"][387:580]), "delta_g2": pr.g2_from_bytes(params["vk"][677:870]),
              "ic": [pr.g1_from_bytes(params["ic"][97 * i:97 * i + 97]) for i in range(r1["n_in"])]}
        pub = r1["z"][1:r1["n_in"]]
        assert pr.groth16_verify(vk, pub, pr.proof_from_bytes(got))
        assert not pr.groth16_verify(vk, [pub[0] + 1] + pub[1:], pr.proof_from_bytes(got))


def test_groth16_h_stage_vs_oracle_2p16(bzk, co):
    log_m = 16
    m = 1 << log_m
    n_rows = m - 1000
    az, bz, cz = (rand_scalars_bytes(n_rows, k) for k in (1, 2, 3))
    pad = b"\0" * (32 * (m - n_rows))
    da, db, dc = to_dev(az + pad), to_dev(bz + pad), to_dev(cz + pad)
    bzk.groth16_h_dev(da, db, dc, log_m)
    torch.cuda.synchronize()
    assert dev_bytes(da)[: 32 * (m - 1)] == co.groth16_h(az, bz, cz, log_m, nthreads=co.ncpu())


@pytest.mark.parametrize("log_m", [1, 2, 5, 10, 11, 14, 21])
def test_groth16_h_chain_sizes_vs_oracle(bzk, co, log_m):
    """the fused h chain (coset scaling on the inverse transforms' final stores, pointwise step on the last transform's first
    load) over 1-, 2- and 3-pass plans == the oracle's seven transforms + pointwise, coefficient for coefficient"""
    m = 1 << log_m
    n_rows = max(1, m - (m // 3))
    az, bz, cz = (rand_scalars_bytes(n_rows, 10 * log_m + k) for k in (1, 2, 3))
    pad = b"\0" * (32 * (m - n_rows))
    da, db, dc = to_dev(az + pad), to_dev(bz + pad), to_dev(cz + pad)
    torch.cuda.synchronize()
    bzk.groth16_h_dev(da, db, dc, log_m)
    torch.cuda.synchronize()
    assert dev_bytes(da)[: 32 * (m - 1)] == co.groth16_h(az, bz, cz, log_m, nthreads=co.ncpu())


def _csr_bytes(co, r1):
    import array
    out = []
    for which in range(3):
        rp, col, val = array.array("I", [0]), array.array("I"), []
        for row in r1["rows"]:
            for v, c in row[which]:
                col.append(v)
                val.append(pr_mod.fr_to_mont_bytes(c))
            rp.append(len(col))
        out.append((len(r1["rows"]), rp.tobytes(), col.tobytes(), b"".join(val)))
    return out


from oracle import pyref as pr_mod  # noqa: E402


@pytest.mark.parametrize("n_mul", [30, 1500])
def test_gpu_setup_matches_oracle_crs(bzk, co, pr, n_mul):
    """bzk_groth16_setup (device fixed-base multiplications) == oracle setup on the same toxic waste, byte for byte"""
    r1 = synth_r1cs(n_mul, seed=4242 + n_mul)
    A, B, Cm = r1cs_to_csr(co, r1)
    log_m = log2_ceil(len(r1["rows"]))
    tox = fr_bytes(fr_list(5, 99))
    want = co.groth16_setup(A, B, Cm, r1["n_in"], r1["n_aux"], log_m, tox, nthreads=co.ncpu())
    ph, vk = bzk.groth16_setup(_csr_bytes(co, r1), r1["n_in"], r1["n_aux"], tox)
    assert vk[:870] == want["vk"]
    assert vk[870:878] == r1["n_in"].to_bytes(8, "little") and vk[878:] == want["ic"]
    for which, key in ((1, "h"), (2, "l"), (3, "a"), (4, "b_g1"), (5, "b_g2")):
        assert bzk.params_read(ph, which) == want[key], key
    # and the device-resident CRS proves: same bytes as the oracle prover
    zb = fr_bytes(r1["z"])
    az, bz, cz = co.r1cs_eval(A, B, Cm, zb, nthreads=co.ncpu())
    r, s = fr_bytes(fr_list(2, 8))[:32], fr_bytes(fr_list(2, 8))[32:]
    assert bzk.groth16_prove(ph, zb, az, bz, cz, r, s) == co.groth16_prove(want, zb, az, bz, cz, r, s, nthreads=co.ncpu())
    bzk.params_free(ph)


def test_prove_refuses_an_assignment_of_another_circuit_shape(bzk, co, pr):
    """bzk_assignment carries n_vars; a witness of another circuit than the CRS is BZK_E_ARG, never an
    out-of-bounds read of `z` or a silently wrong proof"""
    from bazuka_amd.lib import BzkError
    r1, params, zb, az, bz, cz = _setup(co, pr, 100, 4242)
    r, s = fr_bytes(fr_list(2, 5))[:32], fr_bytes(fr_list(2, 5))[32:]
    ph = bzk.params_load(params)
    for bad_z in (zb[:-32], zb + bytes(32), zb[:32]):
        with pytest.raises(BzkError, match="bad argument"):
            bzk.groth16_prove(ph, bad_z, az, bz, cz, r, s)
    with pytest.raises(BzkError):
        bzk.groth16_prove(ph, zb, az, bz[:-32], cz, r, s)      # ragged evaluation vectors: refused by the driver
    assert bzk.groth16_prove(ph, zb, az, bz, cz, r, s) == co.groth16_prove(params, zb, az, bz, cz, r, s)  # ctx still healthy
    bzk.params_free(ph)


def test_bellman_parameters_file_round_trip_through_the_prover(bzk, co, pr):
    """a CRS written in bellman's `Parameters` format (here: the GPU-generated one, encoded by libbzk and
    cross-checked against the oracle writer on the small case of tests/test_bellman_params_cpu.py) loads through
    bzk_params_load_bellman and proves the same bytes; the vk it returns is the bincode `Groth16VerifyingKey`."""
    from bazuka_amd import lib as L
    from bazuka_amd.lib import BzkError
    r1 = synth_r1cs(1500, seed=99)
    csr = _csr_bytes(co, r1)
    tox = fr_bytes(fr_list(5, 4321))
    ph, vkb = bzk.groth16_setup(csr, r1["n_in"], r1["n_aux"], tox)
    A, B, Cm = r1cs_to_csr(co, r1)
    nv = r1["n_in"] + r1["n_aux"]
    a_d, b_d = co.r1cs_density(A, nv), co.r1cs_density(B, nv)
    parts = [bzk.params_read(ph, which) for which in range(6)]
    ic = vkb[878:]
    blob = L.bellman_params_encode(parts[0], ic, *parts[1:])
    ph2, vkb2 = bzk.params_load_bellman(blob, r1["n_in"], r1["n_aux"], a_d, b_d)
    assert vkb2 == vkb
    zb = fr_bytes(r1["z"])
    az, bz, cz = co.r1cs_eval(A, B, Cm, zb, nthreads=co.ncpu())
    r, s = fr_bytes(fr_list(2, 8))[:32], fr_bytes(fr_list(2, 8))[32:]
    p1 = bzk.groth16_prove(ph, zb, az, bz, cz, r, s)
    assert bzk.groth16_prove(ph2, zb, az, bz, cz, r, s) == p1
    assert pr.groth16_verify(pr.vk_from_bytes(vkb2), r1["z"][1:r1["n_in"]], pr.proof_from_bytes(p1))
    # a key of another circuit shape is refused by its lengths, not loaded
    with pytest.raises(BzkError, match="bad argument"):
        bzk.params_load_bellman(blob, r1["n_in"], r1["n_aux"] - 1, a_d[:-1], b_d[:-1])
    with pytest.raises(BzkError, match="bad argument"):
        bzk.params_load_bellman(blob[:-7], r1["n_in"], r1["n_aux"], a_d, b_d)
    bzk.params_free(ph)
    bzk.params_free(ph2)


def test_groth16_slots_share_one_crs_and_prove_concurrently(bzk, co, pr):
    """bzk_params_slot: several prover slots (own ctx + scratch) over ONE device-resident CRS prove different witnesses at the same
    time from different host threads; every proof equals the oracle's; the h table can be dropped / rebuilt by a sole owner"""
    import threading
    from bazuka_amd import Bzk
    r1, params, zb, az, bz, cz = _setup(co, pr, 40000, 4242)   # 2^16 domain: the h query runs through its static table
    rs = [(fr_bytes(fr_list(2, 50 + k))[:32], fr_bytes(fr_list(2, 50 + k))[32:]) for k in range(4)]
    want = [co.groth16_prove(params, zb, az, bz, cz, r, s, nthreads=co.ncpu()) for r, s in rs]
    ph = bzk.params_load(params)
    assert bzk.groth16_prove(ph, zb, az, bz, cz, *rs[0]) == want[0]
    ctxs = [Bzk(0) for _ in range(3)]
    slots = [c.params_slot(ph) for c in ctxs]
    got = [None] * 4

    def run(k):
        c, p = ([bzk] + ctxs)[k], ([ph] + slots)[k]
        for _ in range(3):
            got[k] = c.groth16_prove(p, zb, az, bz, cz, *rs[k])

    th = [threading.Thread(target=run, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got == want
    # the CRS outlives its first handle: slots keep it alive (reference count)
    bzk.params_free(ph)
    assert ctxs[0].groth16_prove(slots[0], zb, az, bz, cz, *rs[1]) == want[1]
    with pytest.raises(Exception):
        ctxs[0].params_h_table(slots[0], False)      # shared: refused
    ctxs[1].params_free(slots[1])
    ctxs[2].params_free(slots[2])
    ctxs[0].params_h_table(slots[0], False)          # sole owner now: dropped ...
    assert ctxs[0].groth16_prove(slots[0], zb, az, bz, cz, *rs[2]) == want[2]
    ctxs[0].params_h_table(slots[0], True)           # ... and rebuilt
    assert ctxs[0].groth16_prove(slots[0], zb, az, bz, cz, *rs[3]) == want[3]
    ctxs[0].params_free(slots[0])
    for c in ctxs:
        c.close()
