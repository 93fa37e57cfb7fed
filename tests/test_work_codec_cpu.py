"""CPU suite for SURVEY 8f-2 (the proving worker's wire format): libbzk's bincode codec of `MpnWork` / `ZkProof`
(bazuka_amd/csrc/host_bincode.h, C ABI `bzk_mpn_work_*`) against an independent schema-driven Python restatement of
the reference's `#[derive(Serialize)]` types (tests/bincode_ref.py), anchored on the one thing the reference pins for this
format: the hard-coded verifying keys (src/config/blockchain.rs:32-37 -> tests/golden/reference_vectors.json), which sit
inside every MpnWork's config.  Also: a decoded work synthesizes the very circuit instance the world it came from does."""
import hashlib
import json
import os

import pytest

import bincode_ref as B
from bazuka_amd import lib as L
from oracle import pyref as pr

F, U = pr.fr_to_mont_bytes, pr.fr_from_mont_bytes
ZIESHA = F(1)
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(G, "reference_vectors.json")))
VKS = [bytes.fromhex(h) for h in REF["verifying_keys_bincode_hex"]]  # the reference's three hard-coded keys, used as (deposit, withdraw, update)
PROVER = bytes(range(1, 33))  # a worker's L1 address (ed25519 public key bytes)


def _world(n_acct=4, L4=3, T4=3):
    w = L.MpnWorld(L4, T4)
    for i in range(n_acct):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 9)
    w.add_key(6, b"fresh")
    w.set_height(11)
    return w


def _queue(w, kind):
    if kind == 2:
        w.push_tx(0, 1, ZIESHA, 1000, ZIESHA, 7)
        w.push_tx(1, 2, ZIESHA, 500, ZIESHA, 3)
        w.push_tx(2, 6, ZIESHA, 10, ZIESHA, 1)      # to an account that does not exist yet
    elif kind == 0:
        w.push_deposit(0, ZIESHA, 1000)
        w.push_deposit(6, F(777), 5)                # new account, custom token
    else:
        w.push_withdraw(0, ZIESHA, 400, ZIESHA, 2)
        w.push_withdraw(1, ZIESHA, 9, ZIESHA, 0)


def _make(kind, reward=5000):
    w = _world()
    _queue(w, kind)
    root0 = w.root()
    work = w.make_work(kind, VKS, reward, log4_batches=(1, 1, 1), num_batches=(1, 2, 3), state_size=42)
    return w, root0, work


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_encoded_work_parses_under_the_reference_schema(kind):
    w, root0, work = _make(kind)
    blob = work.encode()
    v = B.decode(B.MpnWork, blob)  # consumes every byte
    c = v["config"]
    assert (c["log4_tree_size"], c["log4_token_tree_size"]) == (3, 3)
    assert (c["mpn_num_update_batches"], c["mpn_num_deposit_batches"], c["mpn_num_withdraw_batches"]) == (1, 2, 3)
    # the reference's verifying keys sit in the config byte for byte, each behind the ZkVerifierKey::Groth16 tag
    for name, vk in zip(("deposit_vk", "withdraw_vk", "update_vk"), VKS):
        assert B.encode(B.ZkVerifierKey, c[name]) == b"\x00\x00\x00\x00" + vk
        assert len(c[name][1]["ic"]) == 6
    pi = v["public_inputs"]
    assert pi["height"] == 11 and pi["state"] == root0 and pi["next_state"] == w.root() != root0
    assert v["new_root"] == {"state_hash": w.root(), "state_size": 42} and v["reward"] == 5000
    name, trs = v["data"]
    assert name == ("Deposit", "Withdraw", "Update")[kind] and len(trs) == (2, 2, 3)[kind] and all(t["enabled"] for t in trs)
    # the schema encoder reproduces the bytes (maps are written in ascending key order by both)
    assert B.encode(B.MpnWork, v) == blob
    # spot checks against the reference's semantics
    if kind == 2:
        t = trs[0]
        assert t["tx"]["nonce"] == 1 and t["tx"]["amount"] == {"token_id": ("Ziesha", None), "amount": 1000}
        assert t["src_before"]["tokens"] == {0: {"token_id": ("Ziesha", None), "amount": 10 ** 9}}
        assert len(t["src_proof"]) == 3 and len(t["src_balance_proof"]) == 3
        assert pi["aux_data"] == F(pr.poseidon([1, 7 + 3 + 1]))  # H(fee_token, sum of fees) - src/mpn/update.rs:274-285
        assert trs[2]["dst_before"]["address"] == {"x": F(0), "y": F(0)} and trs[2]["dst_before"]["tokens"] == {}
    if kind == 0:
        assert trs[1]["tx"]["payment"]["amount"] == {"token_id": ("Custom", F(777)), "amount": 5}
        assert trs[1]["tx"]["payment"]["sig"] is None and trs[1]["before"]["tokens"] == {}
    if kind == 1:
        pay = trs[0]["tx"]["payment"]
        assert pay["amount"]["amount"] == 400 and pay["fee"]["amount"] == 2
        # ContractWithdraw::fingerprint (src/core/transaction.rs:204-211) = ZkScalar::new(sha3(bincode(payment, calldata := 0)))
        # is what the circuit signs over; the calldata the wallet writes is H6(address, nonce, sig) (verify_calldata :176-181)
        sig = trs[0]["tx"]["mpn_sig"]
        unsigned = dict(pay, calldata=F(0))
        fp = int.from_bytes(hashlib.sha3_256(B.encode(B.ContractWithdraw, unsigned)).digest(), "little") % pr.R_MOD
        key = L.host_jubjub_keys(b"acct0")
        msg = F(pr.poseidon([fp, trs[0]["tx"]["mpn_withdraw_nonce"]]))
        assert L.host_jubjub_verify(key[:64], msg, sig["r"]["x"] + sig["r"]["y"] + sig["s"])
        pub = [U(key[:32]), U(key[32:64])]
        assert pay["calldata"] == F(pr.poseidon(pub + [trs[0]["tx"]["mpn_withdraw_nonce"], U(sig["r"]["x"]), U(sig["r"]["y"]), U(sig["s"])]))


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_decode_roundtrip_and_same_circuit_as_the_world(kind):
    """worker side: bytes -> bzk_mpn_work_decode -> bzk_mpn_work_synthesize gives the assignment an identical world gives
    through bzk_mpn_*_synthesize with the commitment H(prover, reward) - z, A.z, B.z, C.z byte for byte, all satisfied."""
    w, root0, work = _make(kind)
    blob = work.encode()
    dec = L.MpnWork.decode(blob + b"trailing")   # a response holds several works back to back
    assert dec.consumed == len(blob)
    assert (dec.kind, dec.log4_tree, dec.log4_token_tree, dec.log4_batch, dec.height, dec.reward, dec.new_root_size) == (kind, 3, 3, 1, 11, 5000, 42)
    assert dec.n_transitions == (2, 2, 3)[kind] and dec.state == root0 and dec.next_state == w.root() == dec.new_root_hash
    assert (dec.num_update_batches, dec.num_deposit_batches, dec.num_withdraw_batches) == (1, 2, 3)
    assert dec.encode() == blob
    assert [dec.vk(i) for i in range(3)] == VKS and dec.vk() == VKS[kind]
    # commitment: ZkScalar::new(Hasher::hash(bincode((prover, reward))))  (src/mpn/mod.rs:283-285)
    pre = B.encode(B.Address, PROVER) + B.encode(B.U64, 5000)
    com = F(int.from_bytes(hashlib.sha3_256(pre).digest(), "little") % pr.R_MOD)
    assert dec.commitment(PROVER) == com
    r_work = dec.synthesize(PROVER)
    twin = _world()
    _queue(twin, kind)
    if kind == 0:
        r_twin = twin.deposit_synthesize(1, com)
    elif kind == 1:
        r_twin = twin.withdraw_synthesize(1, com)
    else:
        r_twin = twin.update_synthesize(1, com, ZIESHA)
    assert r_work.satisfied and r_twin.satisfied
    assert (r_work.n_in, r_work.n_aux, r_work.n_constraints) == (r_twin.n_in, r_twin.n_aux, r_twin.n_constraints)
    for name in ("z", "az", "bz", "cz"):
        assert r_work.view(name) == r_twin.view(name), name
    z = r_work.view("z")
    pub = [z[32 * i:32 * i + 32] for i in range(1, 6)]
    assert pub == [com, F(11), root0, dec.aux_data, dec.next_state]
    # one worker thread (the sequential walk) produces the same assignment as the parallel per-transition workers
    assert dec.synthesize(PROVER, threads=1).view("z") == z


def test_python_encoded_work_decodes_in_libbzk():
    """the other direction: a work assembled with the reference-schema encoder (no libbzk involved) - an Update work with
    no transitions, exactly what prepare_works emits for an idle block - decodes, pads to null transitions and satisfies"""
    L4, T4 = 3, 3
    empty_root = _world(0).root()
    aux = F(pr.poseidon([1, 0]))
    vk = lambda b: ("Groth16", B.decode(B.Groth16VerifyingKey, b))  # noqa: E731
    work = {
        "config": {"log4_tree_size": L4, "log4_token_tree_size": T4, "log4_deposit_batch_size": 1, "log4_withdraw_batch_size": 1,
                   "log4_update_batch_size": 1, "mpn_contract_id": ("Custom", F(99)), "mpn_num_update_batches": 1,
                   "mpn_num_deposit_batches": 1, "mpn_num_withdraw_batches": 1, "deposit_vk": vk(VKS[0]), "withdraw_vk": vk(VKS[1]),
                   "update_vk": vk(VKS[2])},
        "public_inputs": {"height": 0, "state": empty_root, "aux_data": aux, "next_state": empty_root},
        "data": ("Update", []),
        "new_root": {"state_hash": empty_root, "state_size": 0},
        "reward": 1,
    }
    blob = B.encode(B.MpnWork, work)
    assert len(blob) == 5 + 36 + 24 + 3 * 1464 + 104 + 4 + 8 + 40 + 8
    dec = L.MpnWork.decode(blob)
    assert dec.consumed == len(blob) and dec.kind == 2 and dec.n_transitions == 0 and dec.contract_id == F(99)
    r = dec.synthesize(PROVER)
    assert r.satisfied
    e = L.mpn_update_empty(L4, T4, 1, dec.commitment(PROVER), 0, empty_root, aux, empty_root, ZIESHA)
    assert (r.n_aux, r.n_constraints) == (e.n_aux, e.n_constraints) == (126166, 126001)  # SURVEY App. B at (3, 3, 1)
    assert r.view("z") == e.view("z")


def test_deposit_with_l1_signature_both_encodings():
    """ContractDeposit.sig = Some(ed25519 signature): 64 raw bytes (ed25519 >= 1.3) or a length-prefixed byte string
    (older releases; BZK_WORK_SIG_LEN_PREFIXED)"""
    _, _, work = _make(0)
    v = B.decode(B.MpnWork, work.encode())
    sig = bytes(range(64))
    for t in v["data"][1]:
        t["tx"]["payment"]["sig"] = sig
        t["tx"]["payment"]["memo"] = "deposit to the payment network"
        t["tx"]["payment"]["src"] = bytes(range(100, 132))
    b_tuple = B.encode(B.MpnWork, v)
    b_len = B.encode(B.mpn_work(B.L1SignatureLenPrefixed), v)
    assert len(b_len) == len(b_tuple) + 8 * 2
    d1 = L.MpnWork.decode(b_tuple)
    d2 = L.MpnWork.decode(b_len, flags=1)
    assert d1.consumed == len(b_tuple) and d2.consumed == len(b_len)
    assert d1.encode() == b_tuple and d2.encode() == b_len      # payments are carried verbatim
    z1, z2 = d1.synthesize(PROVER), d2.synthesize(PROVER)
    assert z1.satisfied and z1.view("z") == z2.view("z")        # L1 memo / key / signature are invisible to the circuit
    with pytest.raises(L.BzkError):
        L.MpnWork.decode(b_len)                                  # wrong flavour: the parse derails and is refused


def test_malformed_works_are_refused_not_crashed():
    _, _, work = _make(2)
    blob = work.encode()
    for cut in (0, 1, 40, 5 + 36 + 24 + 100, len(blob) // 2, len(blob) - 1):
        with pytest.raises(L.BzkError):
            L.MpnWork.decode(blob[:cut])
    v = B.decode(B.MpnWork, blob)
    bad = bytearray(blob)
    bad[5:9] = (7).to_bytes(4, "little")           # ContractId variant 7
    with pytest.raises(L.BzkError, match="ContractId"):
        L.MpnWork.decode(bytes(bad))
    off_tag = 5 + len(B.encode(B.ContractId, v["config"]["mpn_contract_id"])) + 24 + 3 * 1464 + 104
    bad = bytearray(blob)
    assert bad[off_tag:off_tag + 4] == (2).to_bytes(4, "little")
    bad[off_tag:off_tag + 4] = (3).to_bytes(4, "little")  # MpnWorkData variant 3
    with pytest.raises(L.BzkError, match="MpnWorkData"):
        L.MpnWork.decode(bytes(bad))
    bad = bytearray(blob)
    bad[off_tag + 4:off_tag + 12] = (1 << 40).to_bytes(8, "little")  # absurd transition count
    with pytest.raises(L.BzkError):
        L.MpnWork.decode(bytes(bad))
    bad = bytearray(blob)
    bad[off_tag + 12] = 2                             # `enabled` bool = 2
    with pytest.raises(L.BzkError, match="bool"):
        L.MpnWork.decode(bytes(bad))
    # a proof of the wrong depth cannot feed the circuit of this config
    v["data"][1][0]["src_proof"].append(b"\x00" * 96)
    with pytest.raises(L.BzkError, match="proof depth"):
        L.MpnWork.decode(B.encode(B.MpnWork, v))
    # a public key that is not on the curve (the reference would panic in decompress().unwrap())
    v = B.decode(B.MpnWork, blob)
    v["data"][1][0]["tx"]["dst_pub_key"]["x"] = F(5)
    d = (-10240 * pow(10241, -1, pr.R_MOD)) % pr.R_MOD  # Jubjub d (src/crypto/jubjub/curve.rs:146-164)
    xs = [x for x in range(2, 40) if pow((1 + x * x) * pow(1 - d * x * x, -1, pr.R_MOD) % pr.R_MOD, (pr.R_MOD - 1) // 2, pr.R_MOD) != 1]
    v["data"][1][0]["tx"]["dst_pub_key"]["x"] = F(xs[0])  # y^2 = (1 + x^2) / (1 - d x^2) has no root
    with pytest.raises(L.BzkError, match="decompress"):
        L.MpnWork.decode(B.encode(B.MpnWork, v))


def test_tampered_witness_gives_an_unsatisfied_instance():
    """a work whose transition does not match its proofs still decodes (the reference would prove garbage and the node's
    verifier would reject it); the generator reports the first violated constraint instead"""
    _, _, work = _make(2)
    v = B.decode(B.MpnWork, work.encode())
    v["data"][1][1]["src_before_balance"]["amount"] += 1
    dec = L.MpnWork.decode(B.encode(B.MpnWork, v))
    r = dec.synthesize(PROVER)
    assert not r.satisfied and r.first_unsatisfied >= 0
    r1 = dec.synthesize(PROVER, threads=1)
    assert r1.first_unsatisfied == r.first_unsatisfied


def test_zkproof_and_protocol_messages():
    proof = bytes((i * 7) & 0xFF for i in range(387))
    proof = proof[:96] + b"\x00" + proof[97:289] + b"\x01" + proof[290:386] + b"\x00"
    enc = L.zkproof_encode(proof)
    assert enc == B.encode(B.ZkProof, ("Groth16", {"a": proof[:97], "b": proof[97:290], "c": proof[290:]}))
    assert L.zkproof_decode(enc) == proof
    with pytest.raises(L.BzkError):
        L.zkproof_decode(b"\x01\x00\x00\x00" + proof)        # variant 1 does not exist outside cfg(test)
    with pytest.raises(L.BzkError):
        L.zkproof_decode(enc[:4] + proof[:96] + b"\x02" + proof[97:])  # bool = 2
    # the three messages of the worker protocol (src/client/messages.rs:368-386)
    from bazuka_amd import worker as W
    _, _, work = _make(2)
    blob = work.encode()
    resp = B.encode(B.GetMpnWorkResponse, {"works": {3: B.decode(B.MpnWork, blob), 9: B.decode(B.MpnWork, blob)}})
    works = W.parse_work_response(resp)
    assert sorted(works) == [3, 9] and all(x.encode() == blob for x in works.values())
    assert W.work_request(PROVER) == B.encode(B.GetMpnWorkRequest, {"address": PROVER})
    sol = W.solution_request(PROVER, {3: proof, 9: proof})
    assert B.decode(B.PostMpnSolutionRequest, sol) == {
        "prover": PROVER, "proofs": {k: ("Groth16", {"a": proof[:97], "b": proof[97:290], "c": proof[290:]}) for k in (3, 9)}}
    assert W.parse_solution_response(B.encode(B.PostMpnSolutionResponse, {"accepted": 2})) == 2


def test_worker_http_round_trip_against_a_mock_node():
    """the worker's plumbing without a GPU: register, fetch over HTTP (GET with a bincode body, as the reference's
    `bincode_get` does), parse the HashMap of works; a junk proof is posted and - the node runs the pairing check - refused"""
    from bazuka_amd import worker as W
    from mock_node import MockNode
    _, _, wu = _make(2)
    _, _, wd = _make(0)
    node = MockNode({4: wu.encode(), 17: wd.encode()})
    try:
        wk = W.Worker(None, PROVER, ("127.0.0.1", node.port), params_for=None)
        assert wk.register() is True
        works = wk.fetch()
        assert sorted(works) == [4, 17] and works[4].kind == 2 and works[17].kind == 0 and wk.stats["fetched"] == 2
        junk = VKS[0][:97] + VKS[0][194:387] + VKS[0][97:194]   # well-formed points that prove nothing
        assert wk.submit({4: junk}) == 0
        assert [e[0] for e in node.log] == ["worker", "work", "solution"] and node.log[2][1:] == (PROVER, [4], 0)
    finally:
        node.close()


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_decoder_survives_mutated_payloads(kind):
    """differential fuzzing of the decoder: random byte flips, truncations and splices of a valid work either decode in
    BOTH codecs to the same thing or are refused by libbzk - never a crash, never a silent disagreement.  (The schema
    codec accepts a few inputs libbzk refuses on purpose: keys that do not decompress, proofs of the wrong depth.)"""
    import random
    rnd = random.Random(1234 + kind)
    _, _, work = _make(kind)
    blob = work.encode()
    agree = refused = 0
    for it in range(300):
        b = bytearray(blob)
        mode = it % 4
        if mode == 0:      # flip 1-3 bytes anywhere
            for _ in range(rnd.randint(1, 3)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif mode == 1:    # overwrite an aligned u64 / u32 (lengths, tags, indices live there)
            off = rnd.randrange(0, len(b) - 8)
            b[off:off + 8] = rnd.choice([0, 1, 2, 255, 1 << 31, (1 << 64) - 1]).to_bytes(8, "little")
        elif mode == 2:    # truncate
            b = b[:rnd.randrange(len(b))]
        else:              # splice a chunk from elsewhere
            a, c = sorted(rnd.randrange(len(b)) for _ in range(2))
            off = rnd.randrange(len(b))
            b[off:off + (c - a)] = blob[a:c]
        b = bytes(b)
        try:
            dec = L.MpnWork.decode(b)
        except L.BzkError:
            refused += 1
            continue
        # accepted: the reference-schema codec must read the same prefix and agree on what it says
        v, used = B.decode_prefix(B.MpnWork, b)
        assert used == dec.consumed
        assert v["public_inputs"]["height"] == dec.height and v["reward"] == dec.reward
        assert v["public_inputs"]["state"] == dec.state and v["new_root"]["state_hash"] == dec.new_root_hash
        assert len(v["data"][1]) == dec.n_transitions and ("Deposit", "Withdraw", "Update").index(v["data"][0]) == dec.kind
        again = L.MpnWork.decode(dec.encode())
        assert again.encode() == dec.encode()   # encode . decode is idempotent on whatever was accepted
        agree += 1
    assert agree > 20 and refused > 20, (agree, refused)


def test_crs_cache_file_is_validated_before_it_reaches_libbzk(tmp_path):
    """DevSetup._load checks every part length against (n_in, n_aux, log_m, n_a, n_b) and the density maps against
    n_a / n_b; a damaged cache file is a BzkError, not a short buffer handed to bzk_params_load"""
    import struct
    from bazuka_amd import worker as W

    class FakeBzk:  # records what would be uploaded; no GPU on this path
        def params_load(self, d):
            self.loaded = d
            return "handle"

    n_in, n_aux, log_m = 2, 5, 3
    a_d, b_d = bytes([1, 0, 1, 1, 0, 0, 1]), bytes([0, 1, 0, 0, 1, 0, 0])
    n_a, n_b = sum(a_d), sum(b_d)
    parts = [bytes(878 + 97 * n_in), bytes(870), bytes(96 * 7), bytes(96 * n_aux), bytes(96 * n_a), bytes(96 * n_b), bytes(192 * n_b), a_d, b_d]

    def write(path, parts, head=(n_in, n_aux, log_m, n_a, n_b), tail=b""):
        with open(path, "wb") as f:
            f.write(W.DevSetup._MAGIC + struct.pack("<5I", *head))
            for p in parts:
                f.write(struct.pack("<Q", len(p)) + p)
            f.write(tail)

    ds = W.DevSetup(FakeBzk(), {})
    good = tmp_path / "good.bzkcrs"
    write(good, parts)
    assert ds._load(str(good))[0] == "handle" and ds.bzk.loaded["n_a"] == n_a
    for name, kw in (("short_part", dict(parts=parts[:2] + [bytes(96 * 6)] + parts[3:])),
                     ("wrong_counts", dict(parts=parts, head=(n_in, n_aux, log_m, n_a + 1, n_b))),
                     ("density", dict(parts=parts[:7] + [bytes([1, 1, 1, 1, 1, 0, 1]), b_d])),
                     ("density_value", dict(parts=parts[:7] + [bytes([2, 0, 1, 1, 0, 0, 0]), b_d])),
                     ("trailing", dict(parts=parts, tail=b"x")),
                     ("log_m", dict(parts=parts, head=(n_in, n_aux, 40, n_a, n_b)))):
        bad = tmp_path / (name + ".bzkcrs")
        write(bad, **kw)
        with pytest.raises(L.BzkError):
            ds._load(str(bad))
    trunc = tmp_path / "trunc.bzkcrs"
    trunc.write_bytes(good.read_bytes()[:-10])
    with pytest.raises(L.BzkError):
        ds._load(str(trunc))


def test_work_response_is_parsed_by_offset():
    """several works in one response body are decoded in place (no per-work re-slicing of the body)"""
    import struct
    from bazuka_amd import worker as W
    blobs = [_make(k)[2].encode() for k in (0, 1, 2)]
    body = struct.pack("<Q", 3) + b"".join(struct.pack("<Q", 10 + i) + b for i, b in enumerate(blobs))
    works = W.parse_work_response(body)
    assert sorted(works) == [10, 11, 12] and [works[10 + i].encode() for i in range(3)] == blobs
    with pytest.raises(L.BzkError):
        W.parse_work_response(body + b"\\0")
    with pytest.raises(L.BzkError):
        W.parse_work_response(body[:-5])
