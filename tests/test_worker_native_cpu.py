"""CPU suite: the native proving worker (bazuka_amd/bzk-worker, bazuka_amd/csrc/worker_main.cpp - plain C++ over include/bzk.h) up to
the GPU: its HTTP client, the bincode framing of `GetMpnWorkRequest` / `GetMpnWorkResponse` (src/client/messages.rs:368-376) and the
decode + synthesis of every work through libbzk's host half, against tests/mock_node.py.  `--dry-run` stops before the device: no
context is created, nothing is posted.  The works carry the reference's own hard-coded verifying keys
(/root/reference/src/config/blockchain.rs:32-37 via tests/golden/reference_vectors.json)."""
import json
import os
import subprocess

import pytest

from bazuka_amd import lib as L
from mock_node import MockNode
from oracle import pyref as pr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("BZK_WORKER_BIN") or os.path.join(ROOT, "bazuka_amd", "bzk-worker")
Z = pr.fr_to_mont_bytes(1)
ALICE = bytes(range(1, 33))


def _run(args, timeout=120):
    p = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


def _reference_vks():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        g = json.load(f)
    out = [bytes.fromhex(v) for v in g["verifying_keys_bincode_hex"]]    # update, deposit, withdraw in the reference's file; any valid key does here
    assert all(len(v) == 1460 for v in out)
    return out


def _works():
    vks = _reference_vks()
    w = L.MpnWorld(3, 3)
    for i in range(4):
        w.add_account(i, b"acct%d" % i, Z, 10 ** 9)
    w.add_key(7, b"newcomer")
    w.set_height(21)
    blobs = {}
    w.push_deposit(0, Z, 1000)
    w.push_deposit(7, Z, 55)
    blobs[3] = w.make_work(0, vks, 100).encode()
    w.push_withdraw(1, Z, 400, Z, 2)
    blobs[1] = w.make_work(1, vks, 200).encode()
    w.push_tx(0, 1, Z, 1000, Z, 7)
    w.push_tx(7, 2, Z, 5, Z, 1)
    blobs[2 ** 40 + 5] = w.make_work(2, vks, 300).encode()     # work ids are usize: not only small ones
    return blobs


def test_the_binary_is_built_and_explains_itself():
    assert os.path.exists(BIN), "bazuka_amd/bzk-worker missing: run `make -C bazuka_amd/csrc` (__graft_entry__.build does)"
    rc, out, err = _run([])
    assert rc == 2 and "usage: bzk-worker" in err
    rc, out, err = _run(["--node", "127.0.0.1:1", "--address", "zz"])
    assert rc == 2 and "32 bytes of hex" in err
    rc, out, err = _run(["--node", "127.0.0.1:1", "--address", ALICE.hex()])
    assert rc == 2 and "exactly one of --dev-toxic / --params" in err


def test_dry_run_fetches_decodes_and_synthesizes_every_work():
    node = MockNode(_works())
    try:
        rc, out, err = _run(["--node", f"127.0.0.1:{node.port}", "--address", ALICE.hex(), "--dry-run", "--rounds", "2", "--poll", "0.05"])
        assert rc == 0, err
        st = json.loads(out.strip().splitlines()[-1])
        assert st["dry_run"] is True and st["rounds"] == 2 and st["fetched"] == 6 and st["unsat"] == 0 and st["errors"] == 0 and st["proved"] == 0
        assert st["synth_s"] > 0 and st["accepted"] == 0
        # the node saw the registration and two work requests from this address, and no solution
        assert [e[0] for e in node.log] == ["worker", "work", "work"] and all(e[1] == ALICE for e in node.log)
    finally:
        node.close()


def test_a_node_that_is_away_or_talks_nonsense_does_not_kill_the_loop():
    # nothing listens: registration fails -> fatal, exit code 1, still one JSON line of statistics
    rc, out, err = _run(["--node", "127.0.0.1:9", "--address", ALICE.hex(), "--dry-run", "--rounds", "1", "--timeout", "2"])
    st = json.loads(out.strip().splitlines()[-1])
    assert rc == 1 and "cannot connect" in st["last_error"]
    # a node whose work response is garbage: the round fails, the loop goes on to the next one
    node = MockNode({0: b"\x07" * 40})
    try:
        rc, out, err = _run(["--node", f"127.0.0.1:{node.port}", "--address", ALICE.hex(), "--dry-run", "--rounds", "2", "--poll", "0.05"])
        st = json.loads(out.strip().splitlines()[-1])
        assert rc == 0 and st["rounds"] == 2 and st["errors"] == 2 and "work response" in st["last_error"]
    finally:
        node.close()


def _hostile_server(reply: bytes, repeat: int = 1):
    """a one-connection-at-a-time TCP server that reads a request head and answers with `reply` x repeat, whatever was asked"""
    import socket
    import threading
    srv = socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(4)
    srv.settimeout(20)

    def serve():
        try:
            while True:
                c, _ = srv.accept()
                try:
                    c.settimeout(5)
                    c.recv(65536)
                    for _ in range(repeat):
                        c.sendall(reply)
                except OSError:
                    pass
                finally:
                    c.close()
        except OSError:
            return
    threading.Thread(target=serve, daemon=True).start()
    return srv, srv.getsockname()[1]


@pytest.mark.parametrize("reply,repeat,phrase", [
    (b"HTTP/1.1 200 OK\r\nTransfer-Encoding: chunked\r\n\r\nffffffffffffffff\r\nabc\r\n0\r\n\r\n", 1, "chunked"),   # pos + n would wrap
    (b"HTTP/1.1 200 OK\r\nContent-Length: 99999999999999\r\n\r\nxx", 1, "implausible Content-Length"),
    (b"HTTP/1.1 200 OK\r\nContent-Length: -7\r\n\r\nxx", 1, "implausible Content-Length"),
    (b"a" * 4096, 20, "no end of the HTTP header"),                                                                   # 80 KiB and no header end
    (b"HTTP/1.1 200 OK\r\nX-Content-Length: 1\r\n\r\n", 1, "registered"),                                            # not the length header: body until close
])
def test_a_hostile_http_peer_is_refused_not_obeyed(reply, repeat, phrase):
    srv, port = _hostile_server(reply, repeat)
    try:
        rc, out, err = _run(["--node", f"127.0.0.1:{port}", "--address", ALICE.hex(), "--dry-run", "--rounds", "1", "--timeout", "5"], timeout=60)
        st = json.loads(out.strip().splitlines()[-1])          # one line of statistics whatever happened: the process ended on its own terms
        if phrase == "registered":
            assert "registered: refused" in err and rc == 0 and st["errors"] == 1   # the (empty) work response is then "truncated"
        else:
            assert rc == 1 and phrase in st["last_error"], (rc, st, err)
    finally:
        srv.close()
