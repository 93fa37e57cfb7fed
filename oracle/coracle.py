"""ctypes binding of oracle/_build/liboracle.so (the C++ CPU restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("oracle.cpp", "field.hpp", "curve.hpp", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Csr(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("row_ptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("n_in", C.c_uint32), ("n_aux", C.c_uint32), ("log_m", C.c_uint32), ("n_a", C.c_uint32),
                ("n_b", C.c_uint32), ("vk", C.c_void_p), ("h", C.c_void_p), ("l", C.c_void_p), ("a", C.c_void_p),
                ("b_g1", C.c_void_p), ("b_g2", C.c_void_p), ("a_density", C.c_void_p), ("b_density", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_init()
    return _lib


def ncpu() -> int:
    return os.cpu_count() or 1


def _buf(b):
    if isinstance(b, (bytes, bytearray)):
        return (C.c_uint8 * len(b)).from_buffer_copy(b)
    return b


def fr_op(op: int, a: bytes, b: bytes | None = None) -> bytes:
    out = C.create_string_buffer(32)
    assert lib().orc_fr_op(op, a, b, out) == 0
    return out.raw


def fp_op(op: int, a: bytes, b: bytes | None = None) -> bytes:
    out = C.create_string_buffer(48)
    assert lib().orc_fp_op(op, a, b, out) == 0
    return out.raw


def poseidon_batch(inp: bytes, arity: int, nthreads: int = 1) -> bytes:
    n = len(inp) // (32 * arity)
    out = C.create_string_buffer(32 * n)
    assert lib().orc_poseidon_batch(inp, C.c_uint32(arity), C.c_uint64(n), out, nthreads) == 0
    return out.raw


def poseidon_params(t: int) -> bytes:
    n = lib().orc_poseidon_params(C.c_uint32(t), None, C.c_uint64(0))
    out = C.create_string_buffer(32 * n)
    assert lib().orc_poseidon_params(C.c_uint32(t), out, C.c_uint64(n)) == n
    return out.raw


def merkle4_root(leaves: bytes, log4: int, want_nodes: bool = False, nthreads: int = 1):
    root = C.create_string_buffer(32)
    nodes = C.create_string_buffer(32 * ((4 ** log4 - 1) // 3)) if want_nodes else None
    assert lib().orc_merkle4_root(leaves, C.c_uint32(log4), root, nodes, nthreads) == 0
    return (root.raw, nodes.raw) if want_nodes else root.raw


def ntt(data: bytes, log_n: int, inverse: bool = False, coset: bool = False, nthreads: int = 1) -> bytes:
    buf = C.create_string_buffer(bytes(data), len(data))
    assert lib().orc_ntt(buf, C.c_uint32(log_n), int(inverse), int(coset), nthreads) == 0
    return buf.raw


def msm_g1(bases: bytes, scalars: bytes, mont: bool = True, nthreads: int = 1, naive: bool = False) -> bytes:
    n = len(scalars) // 32
    assert len(bases) == 96 * n
    out = C.create_string_buffer(97)
    assert lib().orc_msm_g1(bases, scalars, C.c_uint64(n), int(mont), out, nthreads, int(naive)) == 0
    return out.raw


def msm_g1_np(bases, scalars, mont: bool = True, nthreads: int = 1) -> bytes:
    """msm_g1 over contiguous numpy uint8 arrays, without the copies `bytes` arguments cost (2^24 / 2^26-point checks)."""
    n = scalars.size // 32
    assert bases.size == 96 * n and bases.flags["C_CONTIGUOUS"] and scalars.flags["C_CONTIGUOUS"]
    out = C.create_string_buffer(97)
    assert lib().orc_msm_g1(C.c_void_p(bases.ctypes.data), C.c_void_p(scalars.ctypes.data), C.c_uint64(n), int(mont), out,
                            nthreads, 0) == 0
    return out.raw


def msm_g2(bases: bytes, scalars: bytes, mont: bool = True, nthreads: int = 1, naive: bool = False) -> bytes:
    n = len(scalars) // 32
    assert len(bases) == 192 * n
    out = C.create_string_buffer(193)
    assert lib().orc_msm_g2(bases, scalars, C.c_uint64(n), int(mont), out, nthreads, int(naive)) == 0
    return out.raw


def g1_bases(seed: int, start: int, n: int, nthreads: int = 1) -> bytes:
    out = C.create_string_buffer(96 * n)
    assert lib().orc_g1_bases(C.c_uint64(seed), C.c_uint64(start), C.c_uint64(n), out, nthreads) == 0
    return out.raw


def g2_bases(seed: int, start: int, n: int, nthreads: int = 1) -> bytes:
    out = C.create_string_buffer(192 * n)
    assert lib().orc_g2_bases(C.c_uint64(seed), C.c_uint64(start), C.c_uint64(n), out, nthreads) == 0
    return out.raw


def g1_generator() -> bytes:
    out = C.create_string_buffer(97)
    lib().orc_g1_generator(out)
    return out.raw


def g2_generator() -> bytes:
    out = C.create_string_buffer(193)
    lib().orc_g2_generator(out)
    return out.raw


def g1_mul(p97: bytes, k_canon32: bytes) -> bytes:
    out = C.create_string_buffer(97)
    lib().orc_g1_mul(p97, k_canon32, out)
    return out.raw


def g2_mul(p193: bytes, k_canon32: bytes) -> bytes:
    out = C.create_string_buffer(193)
    lib().orc_g2_mul(p193, k_canon32, out)
    return out.raw


def g1_add(a: bytes, b: bytes) -> bytes:
    out = C.create_string_buffer(97)
    lib().orc_g1_add(a, b, out)
    return out.raw


def g2_add(a: bytes, b: bytes) -> bytes:
    out = C.create_string_buffer(193)
    lib().orc_g2_add(a, b, out)
    return out.raw


def g1_on_curve(p97: bytes) -> bool:
    return bool(lib().orc_g1_on_curve(p97))


def g2_on_curve(p193: bytes) -> bool:
    return bool(lib().orc_g2_on_curve(p193))


class CsrHolder:
    """Keeps numpy-free byte buffers alive behind an orc_csr struct."""

    def __init__(self, n_rows: int, row_ptr, col, val: bytes):
        import array
        self.rp = (C.c_uint32 * (n_rows + 1))(*row_ptr)
        self.col = (C.c_uint32 * max(1, len(col)))(*col)
        self.val = C.create_string_buffer(bytes(val), max(1, len(val)))
        self.s = Csr(n_rows, C.cast(self.rp, C.c_void_p), C.cast(self.col, C.c_void_p), C.cast(self.val, C.c_void_p))
        self.nnz = len(col)

    def ref(self):
        return C.byref(self.s)


def r1cs_eval(A: CsrHolder, B: CsrHolder, Cm: CsrHolder, z: bytes, nthreads: int = 1):
    n_rows = A.s.n_rows
    az, bz, cz = (C.create_string_buffer(32 * n_rows) for _ in range(3))
    assert lib().orc_r1cs_eval(A.ref(), B.ref(), Cm.ref(), z, C.c_uint64(len(z) // 32), az, bz, cz, nthreads) == 0
    return az.raw, bz.raw, cz.raw


def r1cs_density(M: CsrHolder, n_vars: int) -> bytes:
    d = C.create_string_buffer(n_vars)
    lib().orc_r1cs_density(M.ref(), C.c_uint64(n_vars), d)
    return d.raw


def groth16_h(az: bytes, bz: bytes, cz: bytes, log_m: int, nthreads: int = 1) -> bytes:
    n_rows = len(az) // 32
    out = C.create_string_buffer(32 * ((1 << log_m) - 1))
    assert lib().orc_groth16_h(az, bz, cz, C.c_uint64(n_rows), C.c_uint32(log_m), out, nthreads) == 0
    return out.raw


def groth16_setup(A, B, Cm, n_in, n_aux, log_m, toxic: bytes, nthreads: int = 1) -> dict:
    nv = n_in + n_aux
    a_d, b_d = r1cs_density(A, nv), r1cs_density(B, nv)
    n_a, n_b = sum(a_d), sum(b_d)
    m = 1 << log_m
    bufs = {k: C.create_string_buffer(max(1, sz)) for k, sz in dict(
        vk=870, ic=97 * n_in, h=96 * (m - 1), l=96 * n_aux, a=96 * n_a, b_g1=96 * n_b, b_g2=192 * n_b).items()}
    rc = lib().orc_groth16_setup(A.ref(), B.ref(), Cm.ref(), C.c_uint32(n_in), C.c_uint32(n_aux), C.c_uint32(log_m),
                                 toxic, a_d, b_d, bufs["vk"], bufs["ic"], bufs["h"], bufs["l"], bufs["a"],
                                 bufs["b_g1"], bufs["b_g2"], nthreads)
    assert rc == 0, rc
    out = {k: v.raw[: {"vk": 870, "ic": 97 * n_in, "h": 96 * (m - 1), "l": 96 * n_aux, "a": 96 * n_a,
                       "b_g1": 96 * n_b, "b_g2": 192 * n_b}[k]] for k, v in bufs.items()}
    out.update(n_in=n_in, n_aux=n_aux, log_m=log_m, n_a=n_a, n_b=n_b, a_density=a_d, b_density=b_d)
    return out


def groth16_prove(params: dict, z: bytes, az: bytes, bz: bytes, cz: bytes, r: bytes, s: bytes, nthreads: int = 1) -> bytes:
    keep = {k: C.create_string_buffer(bytes(params[k]), max(1, len(params[k])))
            for k in ("vk", "h", "l", "a", "b_g1", "b_g2", "a_density", "b_density")}
    P = Params(params["n_in"], params["n_aux"], params["log_m"], params["n_a"], params["n_b"],
               *[C.cast(keep[k], C.c_void_p) for k in ("vk", "h", "l", "a", "b_g1", "b_g2", "a_density", "b_density")])
    out = C.create_string_buffer(387)
    rc = lib().orc_groth16_prove(C.byref(P), z, az, bz, cz, C.c_uint64(len(az) // 32), r, s, out, nthreads)
    assert rc == 0, rc
    return out.raw
