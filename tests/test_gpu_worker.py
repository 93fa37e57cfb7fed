"""GPU end-to-end for SURVEY 8f-2: a proving worker against a (mock) Bazuka node.  The node hands out three works - one
deposit, one withdraw, one update batch over consecutive states, as `prepare_works` does (src/mpn/mod.rs:352-414) - as
bincode `GetMpnWorkResponse`; the worker (bazuka_amd/worker.py) decodes them with libbzk, synthesizes the circuits with
the commitment H(prover, reward), proves on the GPU and posts `PostMpnSolutionRequest`; the node accepts a solution iff
the oracle's pairing check passes for [commitment, height, state, aux_data, next_state] - `MpnWork::verify`."""
import pytest

from bazuka_amd import lib as L
from bazuka_amd import worker as W
from mock_node import MockNode
from oracle import pyref as pr
from util import fr_bytes, fr_list

pytestmark = pytest.mark.gpu
F = pr.fr_to_mont_bytes
ZIESHA = F(1)
ALICE = bytes(range(1, 33))
MALLORY = bytes(range(101, 133))


def test_worker_proves_a_block_of_works_and_the_node_accepts(bzk):
    keys = W.DevSetup(bzk, {k: fr_bytes(fr_list(5, 9000 + k)) for k in range(3)})
    vks = [keys.keys(k, 3, 3, 1)[1] for k in range(3)]
    assert len({v for v in vks}) == 3 and all(len(v) == 1460 for v in vks)
    w = L.MpnWorld(3, 3)
    for i in range(4):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 9)
    w.add_key(7, b"newcomer")
    w.set_height(21)
    blobs, roots = {}, [w.root()]
    w.push_deposit(0, ZIESHA, 1000)
    w.push_deposit(7, ZIESHA, 55)
    blobs[0] = w.make_work(0, vks, 100).encode()
    roots.append(w.root())
    w.push_withdraw(1, ZIESHA, 400, ZIESHA, 2)
    blobs[1] = w.make_work(1, vks, 200).encode()
    roots.append(w.root())
    w.push_tx(0, 1, ZIESHA, 1000, ZIESHA, 7)
    w.push_tx(7, 2, ZIESHA, 5, ZIESHA, 1)      # the account the deposit just created spends
    w.push_tx(2, 3, ZIESHA, 10, ZIESHA, 0)
    blobs[2] = w.make_work(2, vks, 300).encode()
    roots.append(w.root())
    assert len(set(roots)) == 4
    node = MockNode(blobs)
    try:
        rnd = iter(range(1, 1000))
        # self_check: every proof passes the product's own host-side groth16_verify (the node's check) before it is posted
        alice = W.Worker(bzk, ALICE, ("127.0.0.1", node.port), keys, rng=lambda n: bytes([next(rnd)]) * n, self_check=True)
        # a solution is bound to its prover: Mallory re-posting Alice's proofs under her own address gets nothing
        proofs = {wid: alice.prove(work) for wid, work in alice.fetch().items()}
        assert all(p is not None and len(p) == 387 for p in proofs.values()) and alice.stats["proved"] == 3
        mallory = W.Worker(bzk, MALLORY, ("127.0.0.1", node.port), keys)
        assert mallory.submit(proofs) == 0 and not node.solved
        # wrong work id (update proof offered for the deposit work): refused; then the honest round
        assert alice.submit({0: proofs[2]}) == 0
        assert alice.run_once() == 3
        assert node.solved == {0: ALICE, 1: ALICE, 2: ALICE}
        assert alice.fetch() == {} and alice.run_once() == 0     # nothing left to do
        assert alice.stats["accepted"] == 3 and alice.stats["unsat"] == 0 and alice.stats["self_check_failed"] == 0
        # the product verifier and the mock node's oracle pairing check agree on every posted proof, and it binds the prover:
        # Alice's proof fails under Mallory's commitment
        for wid, blob in blobs.items():
            wk = L.MpnWork.decode(blob)
            ok_inputs = wk.commitment(ALICE) + pr.fr_to_mont_bytes(wk.height) + wk.state + wk.aux_data + wk.next_state
            bad_inputs = wk.commitment(MALLORY) + ok_inputs[32:]
            assert L.groth16_verify(wk.vk(), ok_inputs, proofs[wid]) and not L.groth16_verify(wk.vk(), bad_inputs, proofs[wid])
            assert wk.verify(ALICE, proofs[wid]) and not wk.verify(MALLORY, proofs[wid])
    finally:
        node.close()
        keys.close()


def test_dev_setup_disk_cache_gives_the_same_keys_and_proofs(bzk, tmp_path):
    """a restarted worker loads its proving key from the cache file (bzk_params_load) instead of regenerating it: same
    verifying key, and the SAME proof bytes for the same (work, r, s) as the worker that generated the key"""
    tox = {k: fr_bytes(fr_list(5, 7000 + k)) for k in range(3)}
    first = W.DevSetup(bzk, tox, cache_dir=str(tmp_path))
    ph1, vk1 = first.keys(0, 3, 3, 1)
    files = list(tmp_path.iterdir())
    assert len(files) == 1 and files[0].suffix == ".bzkcrs" and files[0].stat().st_size > 1 << 20
    again = W.DevSetup(bzk, tox, cache_dir=str(tmp_path))
    ph2, vk2 = again.keys(0, 3, 3, 1)          # from disk
    assert vk2 == vk1 and ph2.value != ph1.value
    w = L.MpnWorld(3, 3)
    w.add_account(0, b"acct0", ZIESHA, 10 ** 9)
    w.push_deposit(0, ZIESHA, 17)
    work = L.MpnWork.decode(w.make_work(0, [vk1, vk1, vk1], 5).encode())
    proofs = []
    for keys in (first, again):
        rnd = iter(range(1, 100))
        wk = W.Worker(bzk, ALICE, ("127.0.0.1", 1), keys, rng=lambda n: bytes([next(rnd)]) * n)
        proofs.append(wk.prove(work))
    assert proofs[0] is not None and proofs[0] == proofs[1]
    first.close()
    again.close()


def test_worker_with_bellman_parameter_files(bzk, tmp_path):
    """a network ships its proving keys as bellman `Parameters` files: the worker loads them (worker.BellmanKeys ->
    bzk_params_load_bellman with the density maps of the circuit shape), proves a work, the node's pairing check accepts; a file of
    another circuit kind is refused by its lengths / verifying key instead of producing a proof nobody accepts"""
    dev = W.DevSetup(bzk, {k: fr_bytes(fr_list(5, 9100 + k)) for k in range(3)})
    paths, vks = {}, []
    for kind in range(3):
        ph, vk = dev.keys(kind, 3, 3, 1)
        vks.append(vk)
        parts = [bzk.params_read(ph, which) for which in range(6)]
        path = tmp_path / f"kind{kind}.params"
        path.write_bytes(L.bellman_params_encode(parts[0], vk[878:], *parts[1:]))
        paths[kind] = str(path)
    w = L.MpnWorld(3, 3)
    for i in range(3):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 9)
    w.set_height(5)
    w.push_tx(0, 1, ZIESHA, 77, ZIESHA, 2)
    w.push_tx(1, 2, ZIESHA, 5, ZIESHA, 0)
    blobs = {4: w.make_work(2, vks, 900).encode()}
    node = MockNode(blobs)
    keys = W.BellmanKeys(bzk, paths)
    try:
        alice = W.Worker(bzk, ALICE, ("127.0.0.1", node.port), keys, self_check=True)
        assert alice.run_once() == 1 and node.solved == {4: ALICE} and alice.stats["self_check_failed"] == 0
        # the deposit key offered for the update circuit: array lengths do not fit the shape
        wrong = W.BellmanKeys(bzk, {2: paths[0]})
        with pytest.raises(L.BzkError):
            wrong(L.MpnWork.decode(blobs[4]))
    finally:
        node.close()
        keys.close()
        dev.close()


def test_multi_slot_worker_proves_a_round_side_by_side(bzk):
    """replicas: a worker with three prover slots (two further contexts on this GPU sharing the first one's device-resident keys through
    bzk_params_slot - on a multi-GPU node the extra slots sit on other devices) proves the works of a round concurrently; the node
    accepts all of them and more than one slot did work"""
    from bazuka_amd import Bzk
    keys = W.DevSetup(bzk, {k: fr_bytes(fr_list(5, 9100 + k)) for k in range(3)})
    vks = [keys.keys(k, 3, 3, 1)[1] for k in range(3)]
    w = L.MpnWorld(3, 3)
    for i in range(6):
        w.add_account(i, b"slot%d" % i, ZIESHA, 10 ** 9)
    w.set_height(5)
    blobs = {}
    for k in range(6):   # six update works over consecutive states
        w.push_tx(k % 6, (k + 1) % 6, ZIESHA, 10 + k, ZIESHA, 1)
        w.push_tx((k + 2) % 6, (k + 3) % 6, ZIESHA, 20 + k, ZIESHA, 0)
        blobs[k] = w.make_work(2, vks, 50 + k).encode()
    node = MockNode(blobs)
    extra_ctx = [Bzk(0), Bzk(0)]
    extra = [(c, W.SlotKeys(c, keys)) for c in extra_ctx]
    try:
        worker = W.Worker(bzk, ALICE, ("127.0.0.1", node.port), keys, self_check=True, extra_slots=extra)
        assert worker.run_once() == 6
        assert node.solved == {k: ALICE for k in range(6)}
        st = worker.stats
        assert st["proved"] == 6 and sum(st["proved_by_slot"]) == 6 and sum(1 for x in st["proved_by_slot"] if x) >= 2
        assert st["self_check_failed"] == 0 and st["unsat"] == 0
    finally:
        node.close()
        for _, sk in extra:
            sk.close()
        keys.close()
        for c in extra_ctx:
            c.close()


# ---- the NATIVE worker (bazuka_amd/bzk-worker: C++ over the C ABI, no Python in the loop) ------------------------------------------------
def _native(args, timeout=600):
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([os.path.join(root, "bazuka_amd", "bzk-worker")] + args, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


def _block_of_works(vks, reward0=100):
    w = L.MpnWorld(3, 3)
    for i in range(4):
        w.add_account(i, b"acct%d" % i, ZIESHA, 10 ** 9)
    w.add_key(7, b"newcomer")
    w.set_height(21)
    blobs = {}
    w.push_deposit(0, ZIESHA, 1000)
    w.push_deposit(7, ZIESHA, 55)
    blobs[0] = w.make_work(0, vks, reward0).encode()
    w.push_withdraw(1, ZIESHA, 400, ZIESHA, 2)
    blobs[1] = w.make_work(1, vks, reward0 + 100).encode()
    w.push_tx(0, 1, ZIESHA, 1000, ZIESHA, 7)
    w.push_tx(7, 2, ZIESHA, 5, ZIESHA, 1)
    blobs[2] = w.make_work(2, vks, reward0 + 200).encode()
    for k in range(3, 6):    # three more update batches: work for a second slot
        w.push_tx(2, 3, ZIESHA, 10 + k, ZIESHA, 0)
        blobs[k] = w.make_work(2, vks, reward0 + 100 * k).encode()
    return blobs


def test_native_worker_proves_a_block_and_the_node_accepts(bzk):
    """the same round as the first test, run by the native program: it registers, fetches the six works over HTTP, derives the dev-mode
    keys from the seed on the GPU (bzk_groth16_setup), synthesizes ahead of the proofs, proves on two slots sharing the CRS, checks every
    proof with the work's own key and posts; the node accepts a solution iff the ORACLE's pairing check passes"""
    seed = "native-dev"
    keys = W.DevSetup(bzk, {k: W.dev_toxic(seed, k) for k in range(3)})   # what a node set up with this seed holds
    vks = [keys.keys(k, 3, 3, 1)[1] for k in range(3)]
    keys.close()
    blobs = _block_of_works(vks)
    node = MockNode(blobs)
    try:
        st, err = _native(["--node", f"127.0.0.1:{node.port}", "--address", ALICE.hex(), "--dev-toxic", seed, "--slots-per-device", "2",
                           "--self-check", "--rounds", "2", "--poll", "0.05"])
        assert st["fetched"] == 6 and st["proved"] == 6 and st["accepted"] == 6 and st["unsat"] == 0 and st["self_check_failed"] == 0 and st["errors"] == 0, (st, err)
        assert sum(st["proved_by_slot"]) == 6 and len(st["proved_by_slot"]) == 2
        assert node.solved == {k: ALICE for k in blobs}
        assert [e[0] for e in node.log] == ["worker", "work", "solution", "work"]      # second round: nothing left
        # a worker set up with ANOTHER seed holds other keys: it refuses the works (their VK is not its key's) and posts nothing
        node2 = MockNode({0: blobs[2]})
        try:
            st2, err2 = _native(["--node", f"127.0.0.1:{node2.port}", "--address", ALICE.hex(), "--dev-toxic", "another-seed", "--rounds", "1"])
            assert st2["accepted"] == 0 and st2["proved"] == 0 and st2["errors"] == 1 and "verifying key" in st2["last_error"] and not node2.solved
        finally:
            node2.close()
    finally:
        node.close()


def test_native_worker_with_bellman_parameter_files(bzk, tmp_path):
    dev = W.DevSetup(bzk, {k: fr_bytes(fr_list(5, 9300 + k)) for k in range(3)})
    paths, vks = [], []
    for kind in range(3):
        ph, vk = dev.keys(kind, 3, 3, 1)
        vks.append(vk)
        parts = [bzk.params_read(ph, which) for which in range(6)]
        path = tmp_path / f"kind{kind}.params"
        path.write_bytes(L.bellman_params_encode(parts[0], vk[878:], *parts[1:]))
        paths.append(str(path))
    dev.close()
    blobs = _block_of_works(vks, reward0=7)
    node = MockNode(blobs)
    try:
        st, err = _native(["--node", f"127.0.0.1:{node.port}", "--address", ALICE.hex(), "--params"] + paths + ["--self-check", "--rounds", "1"])
        assert st["accepted"] == 6 and st["proved"] == 6 and st["self_check_failed"] == 0 and st["errors"] == 0, (st, err)
        assert node.solved == {k: ALICE for k in blobs}
    finally:
        node.close()


def test_native_worker_over_several_devices(bzk):
    """`--devices a,b`: one key source and one prover slot per device (replicas).  A one-GPU box names its device twice - two independent
    contexts with their own CRS, the code path of a multi-GPU node"""
    seed = "native-two-devices"
    keys = W.DevSetup(bzk, {k: W.dev_toxic(seed, k) for k in range(3)})
    vks = [keys.keys(k, 3, 3, 1)[1] for k in range(3)]
    keys.close()
    blobs = _block_of_works(vks, reward0=11)
    node = MockNode(blobs)
    try:
        st, err = _native(["--node", f"127.0.0.1:{node.port}", "--address", ALICE.hex(), "--dev-toxic", seed, "--devices", "0,0", "--self-check", "--rounds", "1"])
        assert st["accepted"] == 6 and st["proved"] == 6 and st["errors"] == 0 and len(st["proved_by_slot"]) == 2 and sum(st["proved_by_slot"]) == 6, (st, err)
        assert err.count("proving key for kind 2") == 2      # each device generated its own
        assert node.solved == {k: ALICE for k in blobs}
    finally:
        node.close()


def test_workers_with_deferred_witness_values(bzk):
    """--defer: both workers synthesize with BZK_SYNTH_DEFER and prove through bzk_groth16_prove_r1cs - all three kinds of work of a block,
    every proof checked with the work's own key before posting and accepted by the mock node's ORACLE pairing check"""
    seed = "native-dev"
    keys = W.DevSetup(bzk, {k: W.dev_toxic(seed, k) for k in range(3)})
    vks = [keys.keys(k, 3, 3, 1)[1] for k in range(3)]
    blobs = _block_of_works(vks)
    node = MockNode(blobs)
    try:
        alice = W.Worker(bzk, ALICE, ("127.0.0.1", node.port), keys, self_check=True, defer=True)
        assert alice.run_once() == len(blobs)
        assert alice.stats["unsat"] == 0 and alice.stats["self_check_failed"] == 0 and node.solved == {k: ALICE for k in blobs}
    finally:
        node.close()
    node = MockNode(blobs)
    try:
        st, err = _native(["--node", f"127.0.0.1:{node.port}", "--address", ALICE.hex(), "--dev-toxic", seed, "--slots-per-device", "2", "--defer",
                           "--self-check", "--rounds", "1", "--poll", "0.05"])
        assert st["proved"] == len(blobs) and st["accepted"] == len(blobs) and st["self_check_failed"] == 0 and st["errors"] == 0, (st, err)
        assert node.solved == {k: ALICE for k in blobs}
    finally:
        node.close()
        keys.close()
