"""Synthetic host load for tools/runs diagnostics: `cpu N SECONDS` = N processes spinning on integer arithmetic (no memory traffic),
`mem N SECONDS` = N processes streaming 256 MB numpy copies (memory bandwidth, little arithmetic).  Every process ends by itself after
SECONDS (nothing to kill, nothing orphaned: an orphan that inherits a pipe keeps a `cmd | filter` from ever finishing)."""
import multiprocessing as mp, sys, time
import numpy as np


def cpu_spin(seconds):
    x = 1
    end = time.time() + seconds
    while time.time() < end:
        for _ in range(1000000):
            x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF


def mem_stream(seconds):
    a = np.ones(32 << 20, dtype=np.uint64)
    b = np.empty_like(a)
    end = time.time() + seconds
    while time.time() < end:
        np.copyto(b, a)
        np.copyto(a, b)


if __name__ == "__main__":
    kind, n, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    ps = [mp.Process(target=cpu_spin if kind == "cpu" else mem_stream, args=(seconds,)) for _ in range(n)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
