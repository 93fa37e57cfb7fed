"""CPU suite: the scheduling of bazuka_amd/worker.py::Worker.run_once with stub works and a stub prover (no GPU): witness synthesis runs
ahead of the proofs on a producer thread, so a one-slot worker is bound by max(synthesis, proof) per work, not by their sum; several
slots prove side by side; a work whose synthesis or proof fails is counted and reported while the others are still posted."""
import threading
import time

import pytest

from bazuka_amd import lib as L
from bazuka_amd import worker as W

SYNTH_S, PROVE_S = 0.05, 0.05


class _R1cs:
    satisfied = True

    def __init__(self, log):
        self.log = log

    def raw(self, name):
        return b""

    def free(self):
        self.log.append("free")


class _Work:
    kind, log4_tree, log4_token_tree, log4_batch = 2, 15, 3, 2

    def __init__(self, wid, log, fail_synth=False, unsat=False):
        self.wid, self.log, self.fail_synth, self.unsat = wid, log, fail_synth, unsat

    def synthesize(self, address, threads=0):
        time.sleep(SYNTH_S)
        if self.fail_synth:
            raise L.BzkError("injected synthesis failure")
        r = _R1cs(self.log)
        r.satisfied = not self.unsat
        return r

    def verify(self, address, proof):
        return True


class _Bzk:
    def __init__(self, fail_on=None):
        self.in_flight, self.max_in_flight, self.lock, self.fail_on, self.calls = 0, 0, threading.Lock(), fail_on, 0

    def groth16_prove(self, ph, z, az, bz, cz, r, s):
        with self.lock:
            self.in_flight += 1
            self.calls += 1
            n = self.calls
            self.max_in_flight = max(self.max_in_flight, self.in_flight)
        time.sleep(PROVE_S)
        with self.lock:
            self.in_flight -= 1
        if self.fail_on == n:
            raise L.BzkError("injected prover failure")
        return bytes(387)


def _worker(works, slots=1, fail_on=None):
    bz = [_Bzk(fail_on) for _ in range(slots)]
    w = W.Worker(bz[0], bytes(32), ("127.0.0.1", 1), lambda work: None, extra_slots=[(b, W.SlotKeys.__new__(W.SlotKeys)) for b in bz[1:]])
    for k in range(1, slots):   # stub key source of the extra slots
        w.slots[k] = (bz[k], lambda work: None)
    posted = {}
    w.fetch = lambda: works
    w.submit = lambda proofs: posted.update(proofs) or len(proofs)
    return w, bz, posted


def test_one_slot_worker_overlaps_synthesis_with_proving():
    log = []
    works = {i: _Work(i, log) for i in range(8)}
    w, bz, posted = _worker(works)
    t0 = time.perf_counter()
    assert w.run_once() == 8
    dt = time.perf_counter() - t0
    assert sorted(posted) == list(range(8)) and bz[0].max_in_flight == 1
    assert dt < 8 * (SYNTH_S + PROVE_S) * 0.8, dt          # serial would be 0.8 s; pipelined ~ 0.45 s
    assert log.count("free") == 8                          # every witness handed back
    assert w.stats["proved"] == 8 and abs(w.stats["synth_s"] - 8 * SYNTH_S) < 0.2


def test_slots_prove_side_by_side_from_one_synthesis_queue():
    works = {i: _Work(i, []) for i in range(8)}
    w, bz, posted = _worker(works, slots=2)
    assert w.run_once() == 8 and sorted(posted) == list(range(8))
    assert sum(w.stats["proved_by_slot"]) == 8 and all(n > 0 for n in w.stats["proved_by_slot"])


def test_failures_are_counted_and_the_rest_is_posted(capsys):
    log = []
    works = {0: _Work(0, log), 1: _Work(1, log, fail_synth=True), 2: _Work(2, log, unsat=True), 3: _Work(3, log), 4: _Work(4, log)}
    w, bz, posted = _worker(works, fail_on=2)      # the second proof call fails too
    assert w.run_once() == 2
    assert len(posted) == 2 and 1 not in posted and 2 not in posted
    assert w.stats["slot_errors"] == 2 and w.stats["unsat"] == 1 and "work" in w.stats["last_slot_error"]
    assert "works failed this round" in capsys.readouterr().err
    # nothing provable at all: the round raises
    w2, _, _ = _worker({0: _Work(0, [], fail_synth=True)})
    with pytest.raises(L.BzkError):
        w2.run_once()
    # no work: nothing to do
    w3, _, _ = _worker({})
    assert w3.run_once() == 0
