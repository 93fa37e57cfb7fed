"""Proving worker for a Bazuka node: the loop the reference's external provers run against

    GET  /bincode/mpn/work      GetMpnWorkRequest{address}             -> GetMpnWorkResponse{works: HashMap<usize, MpnWork>}
    POST /bincode/mpn/solution  PostMpnSolutionRequest{prover, proofs} -> PostMpnSolutionResponse{accepted}
    POST /bincode/mpn/worker    PostMpnWorkerRequest{address}          -> PostMpnWorkerResponse{accepted}

(/root/reference/src/node/mod.rs:393-413, src/client/messages.rs:368-396, src/client/mod.rs:428-463; bodies are bincode,
a GET carries its request in the body too).  A solution is accepted iff `MpnWork::verify` passes with the commitment
bound to the prover's address and the work's reward (src/mpn/mod.rs:281-295, src/node/api/post_mpn_solution.rs).

Everything that computes is libbzk: `bzk_mpn_work_decode` / `_synthesize` (host C++) and `bzk_groth16_prove` (HIP).
This module is the plumbing around it: HTTP, the HashMap framing of the two messages, the proving-key cache.  It needs a
GPU context (`Bzk`); there is no CPU prover to fall back to.

Proving keys: the reference's provers load bellman `Parameters` files produced by the network's setup.  The key source is a
callable `params_for(work) -> bzk_params handle`: `BellmanKeys` reads those files (`--params DEPOSIT WITHDRAW UPDATE`;
`bzk_params_load_bellman`, bazuka_amd/csrc/host_bellman.hip), `DevSetup` generates the CRS on the GPU from the circuit's
matrices and a given toxic waste, as the reference's dev-mode setup does (src/config/blockchain.rs:355-417; `--dev-toxic`).
"""
from __future__ import annotations

import http.client
import os
import struct
import sys
import time

from . import lib as L


# ---- message framing (the maps' payload types are decoded / encoded by libbzk) -------------------------------------
def work_request(address: bytes) -> bytes:
    """bincode(GetMpnWorkRequest{address}); Address = ed25519 public key = byte string of 32"""
    assert len(address) == 32
    return struct.pack("<Q", 32) + address


worker_request = work_request  # PostMpnWorkerRequest has the same single field


def parse_work_response(body: bytes, flags: int = 0) -> dict[int, L.MpnWork]:
    """bincode(GetMpnWorkResponse{works: HashMap<usize, MpnWork>}) -> {work id: MpnWork}"""
    if len(body) < 8:
        raise L.BzkError("work response: truncated")
    (n,) = struct.unpack_from("<Q", body, 0)
    pos, out = 8, {}
    for _ in range(n):
        if len(body) - pos < 8:
            raise L.BzkError("work response: truncated")
        (wid,) = struct.unpack_from("<Q", body, pos)
        w = L.MpnWork.decode(body, flags, offset=pos + 8)  # by offset: no re-slicing of a multi-MB response per work
        pos += 8 + w.consumed
        out[wid] = w
    if pos != len(body):
        raise L.BzkError(f"work response: {len(body) - pos} trailing bytes")
    return out


def solution_request(prover: bytes, proofs: dict[int, bytes]) -> bytes:
    """bincode(PostMpnSolutionRequest{prover, proofs: HashMap<usize, ZkProof>}); proofs = {work id: 387 proof bytes}"""
    assert len(prover) == 32
    out = [struct.pack("<Q", 32), prover, struct.pack("<Q", len(proofs))]
    for wid, proof in proofs.items():
        out += [struct.pack("<Q", wid), L.zkproof_encode(proof)]
    return b"".join(out)


def parse_solution_response(body: bytes) -> int:
    if len(body) != 8:
        raise L.BzkError("solution response: expected 8 bytes")
    return struct.unpack("<Q", body)[0]


# ---- proving keys ----------------------------------------------------------------------------------------------------
class DevSetup:
    """Proving keys generated on the GPU (bzk_groth16_setup) from the matrices of the all-disabled circuit of the work's
    shape - the reference's dev-mode CRS (src/config/blockchain.rs:355-417) with caller-supplied toxic waste
    (tau | alpha | beta | gamma | delta, 5 x 32 Montgomery bytes per circuit kind).  Cached per (kind, L, T, B)."""

    def __init__(self, bzk: L.Bzk, toxic_by_kind: dict[int, bytes], cache_dir: str | None = None):
        """cache_dir: keep each generated CRS on disk (one file per circuit shape and toxic waste) so that a restarted
        worker uploads it with bzk_params_load instead of regenerating it (minutes for the production circuits)"""
        self.bzk, self.toxic, self.cache, self.cache_dir = bzk, toxic_by_kind, {}, cache_dir
        import threading
        self._lock = threading.Lock()  # several prover slots may ask for the same shape at once: generate / load it once

    def shape_circuit(self, kind: int, L4: int, T4: int, B4: int) -> L.R1cs:
        z = bytes(32)
        if kind == 2:
            return L.mpn_update_empty(L4, T4, B4, z, 0, z, z, z, z, record_matrices=True)
        return L.mpn_circuit_empty(kind, L4, T4, B4, z, 0, z, z, z, record_matrices=True)

    def keys(self, kind: int, L4: int, T4: int, B4: int):
        """(params handle, bincode Groth16VerifyingKey) for a circuit shape"""
        key = (kind, L4, T4, B4)
        with self._lock:
            return self._keys_locked(key, kind, L4, T4, B4)

    def _keys_locked(self, key, kind, L4, T4, B4):
        if key not in self.cache:
            path = None
            if self.cache_dir:
                import hashlib
                tag = hashlib.sha3_256(self.toxic[kind]).hexdigest()[:16]
                path = os.path.join(self.cache_dir, f"crs_k{kind}_L{L4}_T{T4}_B{B4}_{tag}.bzkcrs")
            if path and os.path.exists(path):
                self.cache[key] = self._load(path)
            else:
                r = self.shape_circuit(kind, L4, T4, B4)
                csr = [(r.n_constraints, r.raw("rp" + w), r.raw("col" + w), r.raw("val" + w)) for w in "ABC"]
                self.cache[key] = self.bzk.groth16_setup(csr, r.n_in, r.n_aux, self.toxic[kind])
                if path:
                    self._save(path, self.cache[key], r)
                r.free()
        return self.cache[key]

    # file = magic | 5 x u32 (n_in, n_aux, log_m, n_a, n_b) | 9 x (u64 length + bytes): vk(bincode), vk points(870), h, l, a, b_g1,
    # b_g2, a_density, b_density - the arrays bzk_params_read returns / bzk_params_load takes (include/bzk.h)
    _MAGIC = b"BZKCRS01"

    def _save(self, path, entry, r):
        ph, vk = entry
        a_d, b_d = r.view("a_density"), r.view("b_density")
        parts = [vk] + [self.bzk.params_read(ph, which) for which in range(6)] + [a_d, b_d]
        log_m = max(0, (r.n_constraints - 1).bit_length())
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(self._MAGIC + struct.pack("<5I", r.n_in, r.n_aux, log_m, sum(a_d), sum(b_d)))
            for p in parts:
                f.write(struct.pack("<Q", len(p)))
                f.write(p)
        os.replace(tmp, path)

    def _load(self, path):
        with open(path, "rb") as f:
            head = f.read(8 + 20)
            if head[:8] != self._MAGIC:
                raise L.BzkError(f"{path}: not a bzk CRS file")
            if len(head) != 28:
                raise L.BzkError(f"{path}: truncated header")
            n_in, n_aux, log_m, n_a, n_b = struct.unpack("<5I", head[8:])
            if log_m > 28 or n_in == 0:
                raise L.BzkError(f"{path}: implausible header")
            # the file is a cache, not an input format: still, nothing in it is trusted before it reaches bzk_params_load
            want = [878 + 97 * n_in, 870, 96 * ((1 << log_m) - 1), 96 * n_aux, 96 * n_a, 96 * n_b, 192 * n_b, n_in + n_aux, n_in + n_aux]
            parts = []
            for k in range(9):
                raw = f.read(8)
                if len(raw) != 8 or struct.unpack("<Q", raw)[0] != want[k]:
                    raise L.BzkError(f"{path}: part {k} has the wrong length for (n_in, n_aux, log_m, n_a, n_b)")
                part = f.read(want[k])
                if len(part) != want[k]:
                    raise L.BzkError(f"{path}: truncated")
                parts.append(part)
            if f.read(1):
                raise L.BzkError(f"{path}: trailing bytes")
        vk_bincode, vk_pts, h, l, a, b_g1, b_g2, a_d, b_d = parts
        if sum(a_d) != n_a or sum(b_d) != n_b or max(a_d + b_d, default=0) > 1:
            raise L.BzkError(f"{path}: density maps do not match n_a / n_b")
        ph = self.bzk.params_load({"n_in": n_in, "n_aux": n_aux, "log_m": log_m, "n_a": n_a, "n_b": n_b, "vk": vk_pts, "h": h, "l": l,
                                   "a": a, "b_g1": b_g1, "b_g2": b_g2, "a_density": a_d, "b_density": b_d})
        return ph, vk_bincode

    def __call__(self, work: L.MpnWork):
        ph, vk = self.keys(work.kind, work.log4_tree, work.log4_token_tree, work.log4_batch)
        if vk != work.vk():
            raise L.BzkError("the work's verifying key is not the one of this worker's proving key")
        return ph

    def close(self):
        for ph, _ in self.cache.values():
            self.bzk.params_free(ph)
        self.cache = {}


class BellmanKeys:
    """Proving keys of a real network: bellman `Parameters` files (the format the reference's external prover loads,
    README.md:26-28), one per circuit kind.  The file does not carry the density maps (bellman's prover derives them from the
    circuit), so they come from the all-disabled circuit of the work's shape; bzk_params_load_bellman checks every array
    length of the file against them.  `__call__(work)` returns the params handle like DevSetup."""

    def __init__(self, bzk: L.Bzk, path_by_kind: dict[int, str]):
        self.bzk, self.paths, self.cache = bzk, path_by_kind, {}
        import threading
        self._lock = threading.Lock()

    def keys(self, kind: int, L4: int, T4: int, B4: int):
        key = (kind, L4, T4, B4)
        with self._lock:
            return self._keys_locked(key, kind, L4, T4, B4)

    def _keys_locked(self, key, kind, L4, T4, B4):
        if key not in self.cache:
            r = DevSetup.shape_circuit(None, kind, L4, T4, B4)
            with open(self.paths[kind], "rb") as f:
                blob = f.read()
            self.cache[key] = self.bzk.params_load_bellman(blob, r.n_in, r.n_aux, r.view("a_density"), r.view("b_density"))
            r.free()
        return self.cache[key]

    def __call__(self, work: L.MpnWork):
        ph, vk = self.keys(work.kind, work.log4_tree, work.log4_token_tree, work.log4_batch)
        if vk != work.vk():
            raise L.BzkError("the work's verifying key is not the one inside this worker's parameter file")
        return ph

    def close(self):
        for ph, _ in self.cache.values():
            self.bzk.params_free(ph)
        self.cache = {}


class SlotKeys:
    """Key source of a FURTHER prover slot on a device whose first slot already holds the keys: the slot shares the device-resident
    CRS (its resident base sets and h table included) through bzk_params_slot and owns only its per-proof scratch."""

    def __init__(self, bzk: L.Bzk, first_keys):
        self.bzk, self.first, self.cache = bzk, first_keys, {}

    def __call__(self, work: L.MpnWork):  # called from this slot's thread only
        ph0 = self.first(work)
        key = ph0.value
        if key not in self.cache:
            self.cache[key] = self.bzk.params_slot(ph0)
        return self.cache[key]

    def close(self):
        for ph in self.cache.values():
            self.bzk.params_free(ph)
        self.cache = {}


# ---- the loop ----------------------------------------------------------------------------------------------------------
class Worker:
    def __init__(self, bzk: L.Bzk, address: bytes, node: tuple[str, int], params_for, flags: int = 0, threads: int = 0,
                 rng=os.urandom, timeout_s: float = 30.0, self_check: bool = False, extra_slots=(), defer: bool = False):
        """self_check: verify every proof on the host with the work's own verifying key before posting it (bzk_groth16_verify =
        the check the node will run, src/mpn/mod.rs:281-295; ~20 ms of one core per proof) - a proof that fails is not posted.
        extra_slots: further (Bzk, params_for) prover slots - more slots on the same GPU (params_for = SlotKeys: shared CRS) and / or
        slots on other GPUs of the node (their own key source).  The works of a round are then proved side by side, one host thread
        per slot taking works from a common queue: replicas, as the node's own pool hands a block's proofs to several provers
        (src/mpn/mod.rs:79-107)."""
        self.bzk, self.address, self.node, self.params_for = bzk, address, node, params_for
        self.slots = [(bzk, params_for)] + list(extra_slots)
        self.flags, self.threads, self.rng, self.timeout_s, self.self_check = flags, threads, rng, timeout_s, self_check
        # defer: the host generator leaves the hash-dependent witness values to the device (BZK_SYNTH_DEFER + bzk_groth16_prove_r1cs: DESIGN.md 3.5) -
        # fewer host CPU seconds per work, ~15 ms more GPU latency per proof; same proof statement, same acceptance test
        self.defer = defer
        self.stats = {"fetched": 0, "proved": 0, "accepted": 0, "unsat": 0, "self_check_failed": 0, "synth_s": 0.0, "prove_s": 0.0,
                      "proved_by_slot": [0] * len(self.slots)}
        import threading
        self._lock = threading.Lock()

    def _http(self, method: str, path: str, body: bytes) -> bytes:
        conn = http.client.HTTPConnection(self.node[0], self.node[1], timeout=self.timeout_s)
        try:
            conn.request(method, path, body=body, headers={"Content-Type": "application/octet-stream"})
            resp = conn.getresponse()
            data = resp.read()
            if resp.status != 200:
                raise L.BzkError(f"{method} {path}: HTTP {resp.status}")
            return data
        finally:
            conn.close()

    def register(self) -> bool:
        body = self._http("POST", "/bincode/mpn/worker", worker_request(self.address))
        return body == b"\x01"

    def fetch(self) -> dict[int, L.MpnWork]:
        works = parse_work_response(self._http("GET", "/bincode/mpn/work", work_request(self.address)), self.flags)
        self.stats["fetched"] += len(works)
        return works

    def prove(self, work: L.MpnWork, slot: int = 0) -> bytes | None:
        """387 proof bytes for the work, or None when the work's witness does not satisfy its circuit (a proof of it
        could only be rejected by the node).  slot: which prover slot runs it."""
        t0 = time.perf_counter()
        r1cs = self._synthesize(work)
        return self._prove_synthesized(work, r1cs, time.perf_counter() - t0, slot)

    def _synthesize(self, work: L.MpnWork):
        if self.defer:
            return work.synthesize(self.address, threads=self.threads, defer=True)
        return work.synthesize(self.address, threads=self.threads)

    def _prove_synthesized(self, work: L.MpnWork, r1cs, synth_s: float, slot: int = 0) -> bytes | None:
        """the GPU half of `prove`: the witness arrays are already there (run_once synthesizes the next work on a host thread while the
        GPU proves this one)"""
        bzk, params_for = self.slots[slot]
        t1 = time.perf_counter()
        if not r1cs.satisfied:
            with self._lock:
                self.stats["synth_s"] += synth_s
                self.stats["unsat"] += 1
            return None
        ph = params_for(work)
        r, s = L.host_scalar_new(self.rng(64)), L.host_scalar_new(self.rng(64))  # bellman: `E::Fr::random(rng)` twice
        if self.defer:
            try:
                proof = bzk.groth16_prove_r1cs(ph, r1cs, r, s)   # the deferred instance is completed on the device first
            except L.BzkError as e:
                # a violated DEFERRED row is only seen by the device-side fill: the same outcome as the host scan's `not satisfied` above
                if e.status != L.BZK_E_UNSAT:
                    raise
                with self._lock:
                    self.stats["synth_s"] += synth_s
                    self.stats["prove_s"] += time.perf_counter() - t1
                    self.stats["unsat"] += 1
                return None
        else:
            proof = bzk.groth16_prove(ph, r1cs.raw("z"), r1cs.raw("az"), r1cs.raw("bz"), r1cs.raw("cz"), r, s)
        ok = (not self.self_check) or work.verify(self.address, proof)   # MpnWork::verify: the node's own acceptance test
        with self._lock:
            self.stats["synth_s"] += synth_s
            self.stats["prove_s"] += time.perf_counter() - t1
            self.stats["proved"] += 1
            self.stats["proved_by_slot"][slot] += 1
            if not ok:
                self.stats["self_check_failed"] += 1
        return proof if ok else None

    def submit(self, proofs: dict[int, bytes]) -> int:
        acc = parse_solution_response(self._http("POST", "/bincode/mpn/solution", solution_request(self.address, proofs)))
        self.stats["accepted"] += acc
        return acc

    def run_once(self) -> int:
        """one round: fetch the works assigned to this address, prove them, post the solutions; returns `accepted`.
        Witness synthesis (host threads inside libbzk, GIL released) runs AHEAD of the proofs: a producer thread synthesizes work k + 1
        while the GPU proves work k - a one-slot worker is then bound by max(synthesis, proof) per work instead of their sum (the bench's
        `proofs_per_s_serial` vs `proofs_per_s_pipelined`) - and with several slots the works are proved side by side,
        one host thread per slot taking synthesized works from the common queue."""
        import queue
        import threading
        works = self.fetch()
        proofs = {}
        if not works:
            return 0
        n_slots = min(len(self.slots), len(works))
        if n_slots > 1:
            # key sources generate / load a shape's CRS on THEIR context the first time it is asked for: do that here, before the slot
            # threads start, so that no context is used from two threads (further slots of a device only add a scratch set: SlotKeys)
            warmed = set()
            for work in works.values():
                shape = (work.kind, work.log4_tree, work.log4_token_tree, work.log4_batch)
                if shape not in warmed:
                    warmed.add(shape)
                    for _, params_for in self.slots:
                        if not isinstance(params_for, SlotKeys):
                            params_for(work)
        ready = queue.Queue(maxsize=n_slots)   # synthesized ahead: at most one witness per slot waits (they are 0.1 - 2 GB of pinned memory)
        errors = []

        def failed(where, wid, e):
            errors.append(e)
            with self._lock:
                self.stats["slot_errors"] = self.stats.get("slot_errors", 0) + 1
                self.stats["last_slot_error"] = f"{where}, work {wid}: {e!r}"

        def producer():
            for wid, work in works.items():
                t0 = time.perf_counter()
                try:
                    r1cs = self._synthesize(work)
                except Exception as e:  # noqa: BLE001 - counted and reported; the other works go on
                    failed("synthesis", wid, e)
                    continue
                ready.put((wid, work, r1cs, time.perf_counter() - t0))
            for _ in range(n_slots):
                ready.put(None)

        def consumer(slot):
            while True:
                item = ready.get()
                if item is None:
                    return
                wid, work, r1cs, synth_s = item
                try:
                    p = self._prove_synthesized(work, r1cs, synth_s, slot)
                except Exception as e:  # noqa: BLE001 - the slot keeps taking works, the others keep going
                    failed(f"slot {slot}", wid, e)
                    continue
                finally:
                    r1cs.free()   # hand the pinned witness arrays back to the pool before the next one is taken
                if p is not None:
                    with self._lock:
                        proofs[wid] = p

        prod = threading.Thread(target=producer, daemon=True)
        prod.start()
        th = [threading.Thread(target=consumer, args=(k,)) for k in range(1, n_slots)]
        for t in th:
            t.start()
        consumer(0)   # slot 0 proves on the calling thread
        for t in th:
            t.join()
        prod.join()
        if errors:
            print(f"[worker] {len(errors)} of {len(works)} works failed this round ({self.stats.get('last_slot_error')})", file=sys.stderr, flush=True)
        if errors and not proofs:
            e = errors[0]
            raise e if isinstance(e, L.BzkError) else L.BzkError(f"worker failure: {e!r}")
        return self.submit(proofs) if proofs else 0

    def run_forever(self, poll_s: float = 1.0, rounds: int | None = None):
        done = 0
        while rounds is None or done < rounds:
            try:
                self.run_once()
            except (OSError, L.BzkError, http.client.HTTPException, struct.error) as e:
                # node away, a short read (IncompleteRead) or a bad payload: keep polling, as a worker daemon does
                self.stats["last_error"] = str(e)
            done += 1
            time.sleep(poll_s)


def cpu_quota():
    """CPUs this container may use per scheduling period (cgroup v2 cpu.max / v1 cfs quota); None when unlimited"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        return None if q == "max" else int(q) / int(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def dev_toxic(seed: str, kind: int) -> bytes:
    """the dev-mode toxic waste of circuit `kind` under `--dev-toxic SEED` (tau | alpha | beta | gamma | delta): scalar i =
    ZkScalar::new(sha3_256("SEED/kind/i") twice).  bazuka_amd/csrc/worker_main.cpp derives the same bytes."""
    import hashlib
    return b"".join(L.host_scalar_new(hashlib.sha3_256(f"{seed}/{kind}/{i}".encode()).digest() * 2) for i in range(5))


def main(argv=None):
    """python -m bazuka_amd.worker --node 127.0.0.1:8765 --address <64 hex> --dev-toxic <seed>
    Dev-mode worker: proving keys are generated on the GPU from a toxic-waste seed shared with the node's setup
    (development networks only - a real network's keys come from its ceremony)."""
    import argparse
    import hashlib
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--node", required=True, help="host:port of the Bazuka node")
    ap.add_argument("--address", required=True, help="this worker's L1 address: 32-byte ed25519 public key, hex")
    ap.add_argument("--dev-toxic", help="seed of the dev-mode CRS (tau, alpha, beta, gamma, delta per circuit kind)")
    ap.add_argument("--params", nargs=3, metavar=("DEPOSIT", "WITHDRAW", "UPDATE"),
                    help="bellman `Parameters` files of the network's three circuits (instead of --dev-toxic)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--devices", default=None, help="comma-separated GPU ordinals of this node to prove on (replicas); overrides --device")
    ap.add_argument("--slots-per-device", type=int, default=1, help="prover slots per GPU (they share the device-resident CRS)")
    ap.add_argument("--poll", type=float, default=1.0)
    ap.add_argument("--rounds", type=int, default=None)
    ap.add_argument("--self-check", action="store_true", help="verify every proof on the host (pairing check) before posting it")
    ap.add_argument("--defer", action="store_true", help="leave the hash-dependent witness values to the GPU (less host CPU per work, ~15 ms more GPU latency per proof)")
    ap.add_argument("--sig-len-prefixed", action="store_true", help="node built against ed25519 < 1.3 (BZK_WORK_SIG_LEN_PREFIXED)")
    a = ap.parse_args(argv)
    host, port = a.node.rsplit(":", 1)
    address = bytes.fromhex(a.address)
    if len(address) != 32:
        ap.error("--address must be 32 bytes of hex")

    def toxic(kind):
        return dev_toxic(a.dev_toxic, kind)

    if bool(a.dev_toxic) == bool(a.params):
        ap.error("exactly one of --dev-toxic / --params")
    devices = [int(x) for x in a.devices.split(",")] if a.devices else [a.device]
    # A prover slot keeps ~5 host threads waiting on the GPU.  Inside a CPU-quota'd container (cgroup cpu.max) spinning waits are charged
    # against the same budget as the witness generation: when the quota is smaller than the threads this worker runs, the waits sleep on
    # interrupts instead (libbzk reads BZK_SYNC_BLOCKING once, when the first context is created - so it is set here, before any exists;
    # measured + 4 - 7 % proofs/s under a 16-CPU quota, profiles/r03_run23_27...).
    quota = cpu_quota()
    if quota is not None and quota < len(devices) * a.slots_per_device * 5 + (os.cpu_count() or 1) // 2:
        os.environ.setdefault("BZK_SYNC_BLOCKING", "1")
    # A proving service hands its launches to the HIP runtime's per-stream worker threads (AMD_DIRECT_DISPATCH=0, read when the runtime initialises: before the
    # first context): the prover side then costs the host 0.0065 - 0.0084 CPU-s per proof instead of 0.0174 - 0.0202 (with direct dispatch the runtime's helper
    # threads spend 0.009 s of system time per proof) and four slots prove 4 % more per second under a CPU quota; a lone latency-bound call is 2 % slower that way,
    # which is why the library does not decide this for its callers (profiles/r06_run37_39_host_cpu_of_the_prover.txt).  An operator's own setting wins.
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")

    def key_source(bzk):
        return BellmanKeys(bzk, dict(enumerate(a.params))) if a.params else DevSetup(bzk, {k: toxic(k) for k in range(3)})

    slots, closers = [], []
    for d in devices:
        first_bzk = L.Bzk(d)  # raises without a gfx950 device: there is no CPU prover
        first_keys = key_source(first_bzk)
        slots.append((first_bzk, first_keys))
        closers += [first_keys, first_bzk]
        for _ in range(max(1, a.slots_per_device) - 1):
            bz = L.Bzk(d)
            sk = SlotKeys(bz, first_keys)
            slots.append((bz, sk))
            closers = [sk, bz] + closers   # slots go before the keys they share
    w = Worker(slots[0][0], address, (host, int(port)), slots[0][1], flags=1 if a.sig_len_prefixed else 0, self_check=a.self_check,
               extra_slots=slots[1:], defer=a.defer)
    try:
        w.register()
        w.run_forever(a.poll, a.rounds)
    finally:
        print(w.stats)
        for c in closers:
            c.close()


if __name__ == "__main__":
    main()
