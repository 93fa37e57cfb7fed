"""GPU parity of the device path of the witness builders (bzk_mpn_set_device; SURVEY 8f-3: the per-transaction tree walk of
`prepare_works`, src/mpn/mod.rs:353-414, replaced by one batched Poseidon launch per tree level): the SAME MpnWork bytes as the
host builder on every pinned scenario, on follow-up batches over the state a device batch left behind, on batches whose
transactions share accounts (chains, a self-transfer, a new account used twice), and the same circuit instance from
bzk_mpn_update_synthesize."""
import os
import subprocess
import sys

import pytest

import r1cs_scenarios as sc
from bazuka_amd import lib as L
from oracle import pyref as pr

pytestmark = pytest.mark.gpu
F = pr.fr_to_mont_bytes
ZIESHA = F(1)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list(sc.SCENARIOS))
def test_device_builder_makes_the_same_work_bytes(bzk, name):
    assert sc.make_work(name, dev=bzk) == sc.make_work(name)


def _world(Lg, T, n, dev):
    w = L.MpnWorld(Lg, T)
    if dev is not None:
        w.set_device(dev)
    idx = [(i * 7919 + 3) % (4 ** Lg) for i in range(n)]
    for i, a in enumerate(idx):
        w.add_account(a, b"acct%d" % i, ZIESHA, 10 ** 9)
    w.add_key(4 ** Lg - 5, b"newcomer")
    w.set_height(3)
    return w, idx


def _run(dev):
    Lg, T = 15, 3
    w, idx = _world(Lg, T, 20, dev)
    new = 4 ** Lg - 5
    out = []
    # batch 1 (16 slots): a chain through shared accounts, a self-transfer, a brand-new account that receives twice and then sends,
    # a transaction that must be REJECTED (wrong token) in the middle
    for t in range(6):
        w.push_tx(idx[t], idx[t + 1], ZIESHA, 100 + t, ZIESHA, 1 + t)
    w.push_tx(idx[3], idx[3], ZIESHA, 5, ZIESHA, 2)
    w.push_tx(idx[0], new, ZIESHA, 77, ZIESHA, 0)
    w.push_tx(idx[1], idx[2], F(999), 1, ZIESHA, 0)          # sender has no such token: rejected
    w.push_tx(idx[5], new, ZIESHA, 33, ZIESHA, 4)
    w.push_tx(new, idx[9], ZIESHA, 50, ZIESHA, 1)
    out.append(w.make_work(2, sc.VKS, 10, log4_batches=(1, 1, 2)).encode())
    out.append(w.root())
    # deposits: existing account / custom token into an existing account / a new account; then withdrawals; each on the state the
    # previous DEVICE batch left in the host-side tree
    w.push_deposit(idx[2], ZIESHA, 500)
    w.push_deposit(idx[2], F(4242), 9)
    w.push_deposit(idx[11], F(4242), 1)
    out.append(w.make_work(0, sc.VKS, 11, log4_batches=(1, 1, 2)).encode())
    w.push_withdraw(idx[2], F(4242), 4, ZIESHA, 3)
    w.push_withdraw(idx[4], ZIESHA, 40, ZIESHA, 1)
    w.push_withdraw(idx[4], ZIESHA, 41, ZIESHA, 1)
    out.append(w.make_work(1, sc.VKS, 12, log4_batches=(1, 1, 2)).encode())
    out.append(w.root())
    # a further update batch, synthesized as a circuit instance: same assignment bytes
    for t in range(9):
        w.push_tx(idx[(3 * t) % 20], idx[(3 * t + 7) % 20], ZIESHA, 10 + t, ZIESHA, t % 3)
    r = w.update_synthesize(2, F(99), ZIESHA)
    assert r.satisfied and r.accepted == 9
    out.append(bytes(r.view("z")))
    out.append(w.root())
    return out


def test_shared_accounts_follow_up_batches_and_circuit_instance(bzk):
    host, dev = _run(None), _run(bzk)
    assert len(host) == len(dev)
    for k, (a, b) in enumerate(zip(host, dev)):
        assert a == b, f"item {k} differs between the host and the device builder"


def _fault_run(bzk, fail):
    w, idx = _world(15, 3, 8, bzk)
    out = []
    for kind, push in ((2, lambda: [w.push_tx(idx[t], idx[t + 1], ZIESHA, 100 + t, ZIESHA, 1) for t in range(5)]),
                       (0, lambda: [w.push_deposit(idx[2], ZIESHA, 500), w.push_deposit(4 ** 15 - 5, F(4242), 9)]),
                       (1, lambda: [w.push_withdraw(idx[4], ZIESHA, 40, ZIESHA, 1), w.push_withdraw(idx[4], ZIESHA, 41, ZIESHA, 1)])):
        push()
        if fail:
            root = w.root()
            os.environ["BZK_MPN_TEST_FAULT"] = "1"      # read per call by the hooks build only
            try:
                w.make_work(kind, sc.VKS, 10, log4_batches=(1, 1, 2))
                raise AssertionError("the injected fault did not fail the batch (is this the -DBZK_TEST_HOOKS build?)")
            except L.BzkError:
                pass
            finally:
                del os.environ["BZK_MPN_TEST_FAULT"]
            assert w.root() == root
        out.append(w.make_work(kind, sc.VKS, 10, log4_batches=(1, 1, 2)).encode())
        out.append(w.root())
    return out


def test_failed_device_step_leaves_the_world_untouched(bzk):
    """the device builders decide a batch (balances, nonces, queue) BEFORE the batched hashing step; if that step fails
    nothing of the decision may stay behind.  One injected failure per work kind, then the same call again: the works must equal the
    ones of a world that never saw a failure, and so must the roots.  The fault hook exists only in bazuka_amd/libbzk_testhooks.so
    (-DBZK_TEST_HOOKS): the failing world runs in a child process that loads that build; in the shipped library the
    environment variable does nothing (checked here too)."""
    hooks = os.path.join(ROOT, "bazuka_amd", "libbzk_testhooks.so")
    assert os.path.exists(hooks), "build() makes bazuka_amd/libbzk_testhooks.so"
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import torch\nfrom bazuka_amd import Bzk\nimport test_gpu_mpn_devtree as T\n"
            "ctx = Bzk(0)\nprint('OUT', ' '.join(x.hex() for x in T._fault_run(ctx, True)))\n") % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZK_LIBBZK=hooks), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    got = [bytes.fromhex(x) for x in [ln for ln in out.stdout.splitlines() if ln.startswith("OUT")][0].split()[1:]]
    assert got == _fault_run(bzk, False)
    # the shipped library has no hook: the variable changes nothing
    os.environ["BZK_MPN_TEST_FAULT"] = "1"
    try:
        w, idx = _world(15, 3, 8, bzk)
        w.push_tx(idx[0], idx[1], ZIESHA, 100, ZIESHA, 1)
        assert w.make_work(2, sc.VKS, 10, log4_batches=(1, 1, 2)).encode()
    finally:
        del os.environ["BZK_MPN_TEST_FAULT"]
