"""CPU suite: `python bench.py --gpus N` with no launcher (how the driver may call it - VERDICT r2 item 1) becomes its own
launcher: it re-executes under torch.distributed.run with N ranks on 127.0.0.1, the ranks rendezvous and rank 0's 128-byte group
id reaches every rank (what bzk_mg_create_rank needs).  BZK_BENCH_SPAWN_ONLY stops the ranks before they touch a device."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reports(stdout):
    """the ranks share one stdout: their one-line reports may land on the same line"""
    return [json.loads(m) for m in re.findall(r'\{"spawn_only".*?\}', stdout)]


def _run(args, extra_env=None):
    env = dict(os.environ, BZK_BENCH_SPAWN_ONLY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    return r, _reports(r.stdout)


def test_bench_spawns_its_own_ranks():
    r, lines = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(ln["rank"] for ln in lines) == [0, 1]
    assert all(ln["world"] == 2 and ln["uid_ok"] for ln in lines)


def test_bench_single_gpu_needs_no_launcher():
    r, lines = _run(["--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert lines == [{"spawn_only": True, "rank": 0, "world": 1, "uid_ok": True}]


def test_bench_under_an_external_launcher_is_not_respawned():
    """the documented driver form: python -m torch.distributed.run ... bench.py --gpus 2"""
    env = dict(os.environ, BZK_BENCH_SPAWN_ONLY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _reports(r.stdout)
    assert sorted(ln["rank"] for ln in lines) == [0, 1] and all(ln["uid_ok"] for ln in lines)


def test_host_thread_budget_divides_the_quota_between_the_ranks(monkeypatch):
    """N ranks share one host: a rank's witness producers get cpu_quota / N threads, not cpu_count"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.host_thread_budget(1) == (8, 8)
    monkeypatch.setattr(bench, "cpu_quota", lambda: 16.0)
    for world in (2, 4, 8):
        n_prod, threads = bench.host_thread_budget(world)
        assert 1 <= n_prod * threads <= max(1, 16 // world), (world, n_prod, threads)
    monkeypatch.setattr(bench, "cpu_quota", lambda: None)
    monkeypatch.setattr(bench.os, "cpu_count", lambda: 256)
    n_prod, threads = bench.host_thread_budget(8)
    assert n_prod * threads <= 32 and n_prod <= 8 and threads <= 8
    # blocking host waits whenever a rank's share of the CPUs is below the pipeline's thread count
    assert bench.quota_binds(16.0) and bench.quota_binds(256.0, 8) and not bench.quota_binds(256.0, 1) and not bench.quota_binds(None, 1)
