"""ctypes binding of libbzk.so - mirrors include/bzk.h one to one."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbzk.so")

BZK_F_CANONICAL = 1
_lib = None

# name -> (restype, argtypes); must list EVERY symbol declared in include/bzk.h
_vp, _u8p, _u32, _u64, _i32 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int32
SIGNATURES = {
    "bzk_ctx_create": (_i32, [_i32, _vp, C.POINTER(_vp)]),
    "bzk_ctx_destroy": (None, [_vp]),
    "bzk_sync": (_i32, [_vp]),
    "bzk_strerror": (C.c_char_p, [_i32]),
    "bzk_last_error": (C.c_char_p, [_vp]),
    "bzk_abi_version": (_u32, []),
    "bzk_dev_alloc": (_i32, [_vp, _u64, C.POINTER(_vp)]),
    "bzk_dev_free": (_i32, [_vp, _vp]),
    "bzk_h2d": (_i32, [_vp, _vp, _vp, _u64]),
    "bzk_d2h": (_i32, [_vp, _vp, _vp, _u64]),
    "bzk_prof_enable": (_i32, [_vp, _i32]),
    "bzk_prof_reset": (_i32, [_vp]),
    "bzk_prof_query": (_i32, [_vp, C.c_char_p, C.POINTER(_u64), C.POINTER(C.c_double)]),
    "bzk_prof_dump": (_i32, [_vp, _vp, _u64]),
    "bzk_poseidon_batch": (_i32, [_vp, _vp, _u32, _u64, _vp]),
    "bzk_poseidon_batch_dev": (_i32, [_vp, _vp, _u32, _u64, _vp]),
    "bzk_merkle4_root": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "bzk_merkle4_root_dev": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "bzk_ntt": (_i32, [_vp, _vp, _u32, _i32, _i32]),
    "bzk_ntt_dev": (_i32, [_vp, _vp, _u32, _i32, _i32]),
    "bzk_msm_g1": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_g1_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_g2": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_g2_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_window_count": (_u32, [_u64]),
    "bzk_msm_g1_windows_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "bzk_msm_g2_windows_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "bzk_g1_sum": (_i32, [_vp, _u32, _vp]),
    "bzk_g2_sum": (_i32, [_vp, _u32, _vp]),
    "bzk_g1_synth_bases_dev": (_i32, [_vp, _u64, _u64, _u64, _vp]),
    "bzk_g2_synth_bases_dev": (_i32, [_vp, _u64, _u64, _u64, _vp]),
}


class BzkError(RuntimeError):
    pass


def load_library():
    """Loads libbzk.so.  Raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BzkError(f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(x):
    """bytes / bytearray / ctypes buffer / int (device pointer) / torch tensor -> void*"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, (bytes, bytearray)):
        return C.cast(C.c_char_p(bytes(x)), C.c_void_p)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.cast(x, C.c_void_p)


class Bzk:
    """One context = one GPU + one HIP stream (pass torch's stream handle to share it)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.bzk_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != 0:
            raise BzkError(f"bzk_ctx_create failed: {self.lib.bzk_strerror(st).decode()} (no CPU fallback exists)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.bzk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st: int, what: str):
        if st != 0:
            raise BzkError(f"{what}: {self.lib.bzk_strerror(st).decode()} [{self.lib.bzk_last_error(self.h).decode()}]")

    def sync(self):
        self._ck(self.lib.bzk_sync(self.h), "sync")

    # ---- profiling
    def prof_enable(self, on=True):
        self._ck(self.lib.bzk_prof_enable(self.h, int(on)), "prof_enable")

    def prof_reset(self):
        self._ck(self.lib.bzk_prof_reset(self.h), "prof_reset")

    def prof_query(self, name: str):
        n, ms = C.c_uint64(), C.c_double()
        self._ck(self.lib.bzk_prof_query(self.h, name.encode(), C.byref(n), C.byref(ms)), "prof_query")
        return n.value, ms.value

    def prof_dump(self) -> dict:
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.lib.bzk_prof_dump(self.h, buf, len(buf)), "prof_dump")
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out

    # ---- host-pointer (drop-in) forms
    def poseidon_batch(self, inp: bytes, arity: int) -> bytes:
        n = len(inp) // (32 * arity) if arity else 0
        out = C.create_string_buffer(max(32 * n, 1))
        self._ck(self.lib.bzk_poseidon_batch(self.h, _ptr(inp), arity, n, out), "poseidon_batch")
        return out.raw[: 32 * n]

    def merkle4_root(self, leaves: bytes, log4: int, want_nodes: bool = False):
        root = C.create_string_buffer(32)
        nn = (4 ** log4 - 1) // 3
        nodes = C.create_string_buffer(max(32 * nn, 1)) if want_nodes else None
        self._ck(self.lib.bzk_merkle4_root(self.h, _ptr(leaves), log4, root, nodes), "merkle4_root")
        return (root.raw, nodes.raw[: 32 * nn]) if want_nodes else root.raw

    def ntt(self, data: bytes, log_n: int, inverse=False, coset=False) -> bytes:
        buf = C.create_string_buffer(bytes(data), len(data))
        self._ck(self.lib.bzk_ntt(self.h, buf, log_n, int(inverse), int(coset)), "ntt")
        return buf.raw

    def msm_g1(self, bases: bytes, scalars: bytes, canonical=False) -> bytes:
        n = len(scalars) // 32
        out = C.create_string_buffer(97)
        self._ck(self.lib.bzk_msm_g1(self.h, _ptr(bases), _ptr(scalars), n, BZK_F_CANONICAL if canonical else 0, out), "msm_g1")
        return out.raw

    def msm_g2(self, bases: bytes, scalars: bytes, canonical=False) -> bytes:
        n = len(scalars) // 32
        out = C.create_string_buffer(193)
        self._ck(self.lib.bzk_msm_g2(self.h, _ptr(bases), _ptr(scalars), n, BZK_F_CANONICAL if canonical else 0, out), "msm_g2")
        return out.raw

    # ---- device-pointer forms (x = torch tensor / int device address)
    def poseidon_batch_dev(self, inp, arity: int, n: int, out):
        self._ck(self.lib.bzk_poseidon_batch_dev(self.h, _ptr(inp), arity, n, _ptr(out)), "poseidon_batch_dev")

    def merkle4_root_dev(self, leaves, log4: int, nodes=None) -> bytes:
        root = C.create_string_buffer(32)
        self._ck(self.lib.bzk_merkle4_root_dev(self.h, _ptr(leaves), log4, root, _ptr(nodes)), "merkle4_root_dev")
        return root.raw

    def ntt_dev(self, data, log_n: int, inverse=False, coset=False):
        self._ck(self.lib.bzk_ntt_dev(self.h, _ptr(data), log_n, int(inverse), int(coset)), "ntt_dev")

    def msm_g1_dev(self, bases, scalars, n: int, canonical=False) -> bytes:
        out = C.create_string_buffer(97)
        self._ck(self.lib.bzk_msm_g1_dev(self.h, _ptr(bases), _ptr(scalars), n, BZK_F_CANONICAL if canonical else 0, out), "msm_g1_dev")
        return out.raw

    def msm_g2_dev(self, bases, scalars, n: int, canonical=False) -> bytes:
        out = C.create_string_buffer(193)
        self._ck(self.lib.bzk_msm_g2_dev(self.h, _ptr(bases), _ptr(scalars), n, BZK_F_CANONICAL if canonical else 0, out), "msm_g2_dev")
        return out.raw

    def msm_window_count(self, n: int) -> int:
        return self.lib.bzk_msm_window_count(n)

    def msm_g1_windows_dev(self, bases, scalars, n: int, w0: int, w1: int, canonical=False) -> bytes:
        out = C.create_string_buffer(97)
        self._ck(self.lib.bzk_msm_g1_windows_dev(self.h, _ptr(bases), _ptr(scalars), n, BZK_F_CANONICAL if canonical else 0, w0, w1, out), "msm_g1_windows_dev")
        return out.raw

    def msm_g2_windows_dev(self, bases, scalars, n: int, w0: int, w1: int, canonical=False) -> bytes:
        out = C.create_string_buffer(193)
        self._ck(self.lib.bzk_msm_g2_windows_dev(self.h, _ptr(bases), _ptr(scalars), n, BZK_F_CANONICAL if canonical else 0, w0, w1, out), "msm_g2_windows_dev")
        return out.raw

    def g1_sum(self, packed: bytes) -> bytes:
        out = C.create_string_buffer(97)
        self._ck(self.lib.bzk_g1_sum(_ptr(packed), len(packed) // 97, out), "g1_sum")
        return out.raw

    def g2_sum(self, packed: bytes) -> bytes:
        out = C.create_string_buffer(193)
        self._ck(self.lib.bzk_g2_sum(_ptr(packed), len(packed) // 193, out), "g2_sum")
        return out.raw

    def g1_synth_bases_dev(self, seed: int, start: int, n: int, out):
        self._ck(self.lib.bzk_g1_synth_bases_dev(self.h, seed, start, n, _ptr(out)), "g1_synth_bases_dev")

    def g2_synth_bases_dev(self, seed: int, start: int, n: int, out):
        self._ck(self.lib.bzk_g2_synth_bases_dev(self.h, seed, start, n, _ptr(out)), "g2_synth_bases_dev")
