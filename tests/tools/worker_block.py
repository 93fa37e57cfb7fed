"""One production MPN block through the WORKER over HTTP (row f-2 at the chain's real shapes, src/config/blockchain.rs:22-26, 326-328):
a mock node (tests/mock_node.py) hands out a deposit work (64 tx, 2^21), a withdraw work (64 tx, 2^22) and an update work (256 tx, 2^24)
over consecutive states as bincode `GetMpnWorkResponse`; bazuka_amd/worker.py decodes them, synthesizes ahead of the proofs, proves on
the GPU, checks every proof with the work's own key (--self-check) and posts `PostMpnSolutionRequest`; the node accepts a solution iff
the ORACLE's pairing check passes.  Run on a GPU box: python tests/tools/worker_block.py [--native]
--native: the round is run by the native program bazuka_amd/bzk-worker (C++ over the C ABI; --dev-toxic SEED, so it derives the same
dev-mode keys on the GPU itself) instead of bazuka_amd/worker.py."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bazuka_amd import Bzk, lib as L, worker as W
from mock_node import MockNode
from oracle import pyref as pr

F = pr.fr_to_mont_bytes
ADDR = bytes(range(1, 33))


def main():
    Z = F(1)
    ctx = Bzk(0)
    out = {}
    t0 = time.perf_counter()
    native = "--native" in sys.argv
    seed = "production-block"
    keys = W.DevSetup(ctx, {k: W.dev_toxic(seed, k) for k in range(3)} if native else
                      {k: b"".join(F(1234567 * (k + 1) + 7919 * j + 11) for j in range(5)) for k in range(3)})
    vks = [keys.keys(0, 15, 3, 3)[1], keys.keys(1, 15, 3, 3)[1], keys.keys(2, 15, 3, 4)[1]]
    out["crs_three_shapes_s"] = round(time.perf_counter() - t0, 1)
    w = L.MpnWorld(15, 3)
    w.set_device(ctx)
    size = 4 ** 15
    idx = [(i * 22369621 + 5) % size for i in range(512)]
    for i, a in enumerate(idx):
        w.add_account(a, b"blk%d" % i, Z, 10 ** 12)
    w.set_height(9)
    blobs = {}
    t0 = time.perf_counter()
    for i in range(64):
        w.push_deposit(idx[i], Z, 1000 + i)
    blobs[0] = w.make_work(0, vks, 10, log4_batches=(3, 3, 4)).encode()
    for i in range(64):
        w.push_withdraw(idx[64 + i], Z, 400 + i, Z, i % 4)
    blobs[1] = w.make_work(1, vks, 20, log4_batches=(3, 3, 4)).encode()
    for i in range(256):
        w.push_tx(idx[i], idx[256 + i], Z, 100 + i, Z, i % 7)
    blobs[2] = w.make_work(2, vks, 30, log4_batches=(3, 3, 4)).encode()
    out["make_three_works_s"] = round(time.perf_counter() - t0, 3)
    out["wire_bytes"] = {k: len(v) for k, v in blobs.items()}
    node = MockNode(blobs)
    try:
        if native:
            w.close()
            keys.close()          # the native worker generates its own CRS from the seed: free this process's copy first
            ctx.close()
            t0 = time.perf_counter()
            p = subprocess.run([os.path.join(ROOT, "bazuka_amd", "bzk-worker"), "--node", f"127.0.0.1:{node.port}", "--address", ADDR.hex(),
                                "--dev-toxic", seed, "--self-check", "--rounds", "1"], capture_output=True, text=True, timeout=900)
            out["native_wall_s_incl_crs_generation"] = round(time.perf_counter() - t0, 1)
            out["native_stderr"] = [l for l in p.stderr.splitlines() if l.startswith("[bzk-worker]")]
            assert p.returncode == 0, p.stderr[-2000:]
            out["worker_stats"] = json.loads(p.stdout.strip().splitlines()[-1])
            out["accepted"] = out["worker_stats"]["accepted"]
            out["node_solved"] = sorted(node.solved)
            assert out["accepted"] == 3 and node.solved == {0: ADDR, 1: ADDR, 2: ADDR}
            print(json.dumps(out))
            return
        wk = W.Worker(ctx, ADDR, ("127.0.0.1", node.port), keys, self_check=True)
        assert wk.register()
        t0 = time.perf_counter()
        accepted = wk.run_once()
        out["run_once_s_incl_http_and_node_pairing_checks"] = round(time.perf_counter() - t0, 2)
        out["accepted"] = accepted
        out["node_solved"] = sorted(node.solved)
        out["worker_stats"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in wk.stats.items()}
        assert accepted == 3 and node.solved == {0: ADDR, 1: ADDR, 2: ADDR}
    finally:
        node.close()
    print(json.dumps(out))
    keys.close()
    ctx.close()


if __name__ == "__main__":
    main()
