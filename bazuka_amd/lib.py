"""ctypes binding of libbzk.so - mirrors include/bzk.h one to one."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BZK_LIBBZK: another build of the same library (A/B runs of compile-time switches: the gitignored pattern bazuka_amd/libbzk.so.* travels to the GPU box)
LIB_PATH = os.environ.get("BZK_LIBBZK") or os.path.join(_HERE, "libbzk.so")

BZK_F_CANONICAL = 1
BZK_F_DEDUP = 2
BZK_F_THROUGHPUT = 4


def _flags(canonical=False, dedup=False, throughput=False) -> int:
    return (BZK_F_CANONICAL if canonical else 0) | (BZK_F_DEDUP if dedup else 0) | (BZK_F_THROUGHPUT if throughput else 0)


_lib = None

# name -> (restype, argtypes); must list EVERY symbol declared in include/bzk.h
_vp, _u8p, _u32, _u64, _i32 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int32
SIGNATURES = {
    "bzk_ctx_create": (_i32, [_i32, _vp, C.POINTER(_vp)]),
    "bzk_ctx_destroy": (None, [_vp]),
    "bzk_sync": (_i32, [_vp]),
    "bzk_strerror": (C.c_char_p, [_i32]),
    "bzk_last_error": (C.c_char_p, [_vp]),
    "bzk_last_refusal": (_i32, [_vp]),
    "bzk_abi_version": (_u32, []),
    "bzk_dev_alloc": (_i32, [_vp, _u64, C.POINTER(_vp)]),
    "bzk_dev_free": (_i32, [_vp, _vp]),
    "bzk_ctx_trim": (_i32, [_vp, C.POINTER(_u64)]),
    "bzk_h2d": (_i32, [_vp, _vp, _vp, _u64]),
    "bzk_d2h": (_i32, [_vp, _vp, _vp, _u64]),
    "bzk_prof_enable": (_i32, [_vp, _i32]),
    "bzk_prof_filter": (_i32, [_vp, C.c_char_p]),
    "bzk_prof_reset": (_i32, [_vp]),
    "bzk_prof_query": (_i32, [_vp, C.c_char_p, C.POINTER(_u64), C.POINTER(C.c_double)]),
    "bzk_prof_dump": (_i32, [_vp, _vp, _u64]),
    "bzk_poseidon_batch": (_i32, [_vp, _vp, _u32, _u64, _vp]),
    "bzk_poseidon_batch_dev": (_i32, [_vp, _vp, _u32, _u64, _vp]),
    "bzk_merkle4_root": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "bzk_merkle4_root_dev": (_i32, [_vp, _vp, _u32, _vp, _vp]),
    "bzk_state_compress": (_i32, [_vp, _vp, _u64, _vp, _vp, _vp, _u64, _vp, C.POINTER(_u64)]),
    "bzk_state_compress_bincode": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp]),
    "bzk_state_model_default": (_i32, [_vp, _u64, _vp]),
    "bzk_state_create": (_i32, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "bzk_state_free": (None, [_vp]),
    "bzk_state_update": (_i32, [_vp, _vp, _vp, _vp, _u64, _u64, _vp, C.POINTER(_u64), _vp]),
    "bzk_state_update_bincode": (_i32, [_vp, _vp, _u64, _u64, _vp]),
    "bzk_state_root": (_i32, [_vp, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "bzk_state_get": (_i32, [_vp, _vp, _vp, _u64, _vp]),
    "bzk_state_prove": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp, C.POINTER(_u32)]),
    "bzk_state_stats": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
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
    "bzk_params_load": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "bzk_params_free": (None, [_vp, _vp]),
    "bzk_params_slot": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "bzk_params_h_table": (_i32, [_vp, _vp, _i32]),
    "bzk_groth16_prove": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "bzk_groth16_prove_r1cs": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "bzk_r1cs_stage": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "bzk_staged_wait": (_i32, [_vp]),
    "bzk_staged_free": (None, [_vp]),
    "bzk_staged_read": (_i32, [_vp, _i32, _vp, _u64, C.POINTER(_u64)]),
    "bzk_groth16_prove_staged": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "bzk_bellman_params_info": (_i32, [_vp, _u64, C.POINTER(_u64)]),
    "bzk_bellman_params_decode": (_i32, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32]),
    "bzk_bellman_params_encode": (_i32, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _u64, C.POINTER(_u64)]),
    "bzk_params_load_bellman": (_i32, [_vp, _vp, _u64, _u32, _u32, _vp, _vp, C.POINTER(_vp), _vp, _u64]),
    "bzk_groth16_verify": (_i32, [_vp, _u64, _vp, _u32, _vp]),
    "bzk_params_read": (_i32, [_vp, _vp, _i32, _vp, _u64, C.POINTER(_u64)]),
    "bzk_groth16_setup": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _vp, C.POINTER(_vp), _vp, _u64]),
    "bzk_groth16_h_dev": (_i32, [_vp, _vp, _vp, _vp, _u32]),
    "bzk_mpn_create": (_i32, [_u32, _u32, C.POINTER(_vp)]),
    "bzk_mpn_destroy": (None, [_vp]),
    "bzk_mpn_set_height": (_i32, [_vp, _u64]),
    "bzk_mpn_set_threads": (_i32, [_vp, _i32]),
    "bzk_host_default_threads": (_i32, []),
    "bzk_mpn_set_device": (_i32, [_vp, _vp]),
    "bzk_mpn_add_account": (_i32, [_vp, _u64, _vp, _u32, _vp, _u64, _vp]),
    "bzk_mpn_add_key": (_i32, [_vp, _u64, _vp, _u32]),
    "bzk_mpn_root": (_i32, [_vp, _vp]),
    "bzk_mpn_push_tx": (_i32, [_vp, _u64, _u64, _vp, _u64, _vp, _u64]),
    "bzk_tree4_create": (_i32, [_vp, _u32, _vp, _vp, C.POINTER(_vp)]),
    "bzk_tree4_free": (None, [_vp, _vp]),
    "bzk_tree4_root": (_i32, [_vp, _vp, _vp]),
    "bzk_tree4_update": (_i32, [_vp, _vp, _vp, _vp, _u64]),
    "bzk_tree4_prove": (_i32, [_vp, _vp, _vp, _u64, _vp]),
    "bzk_tree4_node": (_i32, [_vp, _vp, _u32, _u64, _vp]),
    "bzk_mpn_state_compress_dev": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp]),
    "bzk_mpn_tree_create": (_i32, [_vp, _u32, _u32, _u64, C.POINTER(_vp)]),
    "bzk_mpn_tree_free": (None, [_vp, _vp]),
    "bzk_mpn_tree_root": (_i32, [_vp, _vp, _vp]),
    "bzk_mpn_tree_accounts": (_u64, [_vp]),
    "bzk_mpn_tree_set_accounts": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64]),
    "bzk_mpn_tree_get_accounts": (_i32, [_vp, _vp, _vp, _u64, _vp]),
    "bzk_mpn_tree_prove": (_i32, [_vp, _vp, _vp, _u64, _vp]),
    "bzk_mpn_tree_prove_token": (_i32, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "bzk_mpn_push_deposit": (_i32, [_vp, _u64, _vp, _u64]),
    "bzk_mpn_push_withdraw": (_i32, [_vp, _u64, _vp, _u64, _vp, _u64, _vp]),
    "bzk_mpn_deposit_synthesize": (_i32, [_vp, _u32, _vp, _i32, C.POINTER(_vp)]),
    "bzk_mpn_withdraw_synthesize": (_i32, [_vp, _u32, _vp, _i32, C.POINTER(_vp)]),
    "bzk_mpn_circuit_empty": (_i32, [_i32, _u32, _u32, _u32, _vp, _u64, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "bzk_mpn_update_synthesize": (_i32, [_vp, _u32, _vp, _vp, _i32, C.POINTER(_vp)]),
    "bzk_mpn_update_empty": (_i32, [_u32, _u32, _u32, _vp, _u64, _vp, _vp, _vp, _vp, _i32, C.POINTER(_vp)]),
    "bzk_r1cs_info": (_i32, [_vp, C.POINTER(_u64)]),
    "bzk_r1cs_defer_info": (_i32, [_vp, C.POINTER(_u64)]),
    "bzk_r1cs_fill_host": (_i32, [_vp]),
    "bzk_r1cs_defer_schedule_info": (_i32, [_vp, C.POINTER(_u64)]),
    "bzk_mpn_set_defer": (_i32, [_vp, _i32]),
    "bzk_r1cs_data": (_vp, [_vp, _i32, C.POINTER(_u64)]),
    "bzk_r1cs_free": (None, [_vp]),
    "bzk_host_poseidon": (_i32, [_vp, _u32, _vp]),
    "bzk_host_sha3_256": (_i32, [_vp, _u64, _vp]),
    "bzk_host_jubjub_keys": (_i32, [_vp, _u32, _vp]),
    "bzk_host_jubjub_sign": (_i32, [_vp, _vp, _vp]),
    "bzk_host_jubjub_verify": (_i32, [_vp, _vp, _vp]),
    "bzk_msm_g1_table_build": (_i32, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "bzk_msm_g2_table_build": (_i32, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "bzk_msm_g1_table_build_c": (_i32, [_vp, _vp, _u64, _u32, C.POINTER(_vp)]),
    "bzk_msm_g1_table_build_levels": (_i32, [_vp, _vp, _u64, _u32, C.POINTER(_vp)]),
    "bzk_msm_g2_table_build_levels": (_i32, [_vp, _vp, _u64, _u32, C.POINTER(_vp)]),
    "bzk_msm_table_levels": (_u32, [_vp]),
    "bzk_msm_g1_table_run_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_g2_table_run_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_g1_table_windows_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "bzk_msm_g2_table_windows_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "bzk_msm_table_window_count": (_u32, [_vp]),
    "bzk_msm_table_free": (None, [_vp, _vp]),
    "bzk_mpn_work_decode": (_i32, [_vp, _u64, _u32, C.POINTER(_vp), C.POINTER(_u64)]),
    "bzk_mpn_work_last_error": (C.c_char_p, []),
    "bzk_mpn_work_free": (None, [_vp]),
    "bzk_mpn_work_info": (_i32, [_vp, C.POINTER(_u64)]),
    "bzk_mpn_work_scalars": (_i32, [_vp, _vp]),
    "bzk_mpn_work_vk": (_i32, [_vp, _i32, _vp, _u64, C.POINTER(_u64)]),
    "bzk_mpn_work_commitment": (_i32, [_vp, _vp, _vp]),
    "bzk_mpn_work_verify": (_i32, [_vp, _vp, _vp]),
    "bzk_mpn_work_synthesize": (_i32, [_vp, _vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "bzk_mpn_work_encode": (_i32, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "bzk_mpn_make_work": (_i32, [_vp, _i32, _vp, _u64, C.POINTER(_vp)]),
    "bzk_host_scalar_new": (_i32, [_vp, _u32, _vp]),
    "bzk_zkproof_encode": (_i32, [_vp, _vp]),
    "bzk_zkproof_decode": (_i32, [_vp, _u64, _vp]),
    "bzk_msm_g1_bases_load_dev": (_i32, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "bzk_msm_g2_bases_load_dev": (_i32, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "bzk_msm_bases_free": (None, [_vp, _vp]),
    "bzk_msm_bases_size": (_u64, [_vp]),
    "bzk_msm_bases_info": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_i32), C.POINTER(_u64)]),
    "bzk_msm_g1_bases_run_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_g2_bases_run_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_msm_g1_bases_windows_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "bzk_msm_g2_bases_windows_dev": (_i32, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp]),
    "bzk_mg_unique_id": (_i32, [_vp]),
    "bzk_mg_probe": (_i32, [_i32]),
    "bzk_mg_create": (_i32, [C.POINTER(_i32), _i32, _u32, C.POINTER(_vp)]),
    "bzk_mg_create_rank": (_i32, [_i32, _i32, _i32, _vp, _u32, C.POINTER(_vp)]),
    "bzk_mg_destroy": (None, [_vp]),
    "bzk_mg_world": (_i32, [_vp]),
    "bzk_mg_local": (_i32, [_vp]),
    "bzk_mg_rank": (_i32, [_vp]),
    "bzk_mg_exchange": (_u32, [_vp]),
    "bzk_mg_ctx": (_vp, [_vp, _i32]),
    "bzk_mg_stats": (_i32, [_vp, _i32, C.POINTER(C.c_double)]),
    "bzk_mg_last_error": (C.c_char_p, [_vp]),
    "bzk_mg_bases_g1_load": (_i32, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "bzk_mg_bases_g2_load": (_i32, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "bzk_mg_bases_g1_load_dev": (_i32, [_vp, C.POINTER(_vp), _u64, C.POINTER(_vp)]),
    "bzk_mg_bases_g2_load_dev": (_i32, [_vp, C.POINTER(_vp), _u64, C.POINTER(_vp)]),
    "bzk_mg_bases_free": (None, [_vp, _vp]),
    "bzk_mg_msm_g1": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_mg_msm_g2": (_i32, [_vp, _vp, _vp, _u64, _u32, _vp]),
    "bzk_mg_msm_g1_dev": (_i32, [_vp, _vp, C.POINTER(_vp), _u64, _u32, _vp]),
    "bzk_mg_msm_g2_dev": (_i32, [_vp, _vp, C.POINTER(_vp), _u64, _u32, _vp]),
    "bzk_mg_params_load": (_i32, [_vp, _vp, _u32, C.POINTER(_vp)]),
    "bzk_mg_params_free": (None, [_vp, _vp]),
    "bzk_mg_params_slots": (_u32, [_vp]),
    "bzk_mg_params_stats": (_i32, [_vp, C.POINTER(_u64), _u32]),
    "bzk_mg_prove_submit": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    "bzk_mg_prove_wait": (_i32, [_vp, _vp, _u64]),
    "bzk_mg_prove": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "bzk_g1_synth_bases_dev": (_i32, [_vp, _u64, _u64, _u64, _vp]),
    "bzk_g2_synth_bases_dev": (_i32, [_vp, _u64, _u64, _u64, _vp]),
}


class ParamsDesc(C.Structure):
    _fields_ = [("n_in", _u32), ("n_aux", _u32), ("log_m", _u32), ("n_a", _u32), ("n_b", _u32), ("vk", _vp), ("h", _vp),
                ("l", _vp), ("a", _vp), ("b_g1", _vp), ("b_g2", _vp), ("a_density", _vp), ("b_density", _vp)]


class CsrDesc(C.Structure):
    _fields_ = [("n_rows", _u64), ("row_ptr", _vp), ("col", _vp), ("val", _vp)]


class Assignment(C.Structure):
    _fields_ = [("z", _vp), ("az", _vp), ("bz", _vp), ("cz", _vp), ("n_rows", _u64), ("n_vars", _u64)]


class WorkConfig(C.Structure):  # bzk_mpn_work_config
    _fields_ = [("log4_deposit_batch", C.c_uint8), ("log4_withdraw_batch", C.c_uint8), ("log4_update_batch", C.c_uint8),
                ("num_update_batches", _u64), ("num_deposit_batches", _u64), ("num_withdraw_batches", _u64),
                ("deposit_vk", _vp), ("deposit_vk_len", _u64), ("withdraw_vk", _vp), ("withdraw_vk_len", _u64),
                ("update_vk", _vp), ("update_vk_len", _u64), ("new_root_state_size", _u64)]


class BzkError(RuntimeError):
    """status: the C ABI's int32 (BZK_E_*; None where the failure is the binding's own)"""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


BZK_E_UNSAT = -4


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
    if isinstance(x, bytes):
        return C.cast(C.c_char_p(x), C.c_void_p)  # the object's own buffer (caller keeps x alive across the call)
    if isinstance(x, bytearray):
        return C.cast((C.c_char * len(x)).from_buffer(x), C.c_void_p)
    if isinstance(x, C.Array):
        return C.cast(C.addressof(x), C.c_void_p)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.cast(x, C.c_void_p)


class Bzk:
    """One context = one GPU + one HIP stream (pass torch's stream handle to share it)."""

    def __init__(self, device: int = 0, stream: int | None = None, handle=None):
        self.lib = load_library()
        self.borrowed = handle is not None
        if handle is not None:  # a context owned by someone else (bzk_mg_ctx): never destroyed from here
            self.h = C.c_void_p(handle) if isinstance(handle, int) else handle
            self.device = device
            return
        h = C.c_void_p()
        st = self.lib.bzk_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != 0:
            raise BzkError(f"bzk_ctx_create failed: {self.lib.bzk_strerror(st).decode()} (no CPU fallback exists)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            if not self.borrowed:
                self.lib.bzk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st: int, what: str):
        if st != 0:
            raise BzkError(f"{what}: {self.lib.bzk_strerror(st).decode()} [{self.lib.bzk_last_error(self.h).decode()}]", st)

    def sync(self):
        self._ck(self.lib.bzk_sync(self.h), "sync")

    def trim(self) -> int:
        """hand the grow-only call workspace back to the device (after a one-off large call); returns the bytes released"""
        n = C.c_uint64(0)
        self._ck(self.lib.bzk_ctx_trim(self.h, C.byref(n)), "ctx_trim")
        return n.value

    def last_refusal(self) -> int:
        """BZK_REFUSE_*: which of the reference's errors the last refused bzk_state_* call stands for"""
        return self.lib.bzk_last_refusal(self.h)

    # ---- profiling
    def prof_filter(self, substr: str | None):
        """event pairs only around launches whose label contains `substr` (None: every launch)"""
        self._ck(self.lib.bzk_prof_filter(self.h, substr.encode() if substr else None), "prof_filter")

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

    def msm_g1(self, bases: bytes, scalars: bytes, canonical=False, dedup=False, throughput=False) -> bytes:
        n = len(scalars) // 32
        out = C.create_string_buffer(97)
        self._ck(self.lib.bzk_msm_g1(self.h, _ptr(bases), _ptr(scalars), n, _flags(canonical, dedup, throughput), out), "msm_g1")
        return out.raw

    def msm_g2(self, bases: bytes, scalars: bytes, canonical=False, dedup=False, throughput=False) -> bytes:
        n = len(scalars) // 32
        out = C.create_string_buffer(193)
        self._ck(self.lib.bzk_msm_g2(self.h, _ptr(bases), _ptr(scalars), n, _flags(canonical, dedup, throughput), out), "msm_g2")
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

    def msm_g1_dev(self, bases, scalars, n: int, canonical=False, dedup=False, throughput=False) -> bytes:
        out = C.create_string_buffer(97)
        self._ck(self.lib.bzk_msm_g1_dev(self.h, _ptr(bases), _ptr(scalars), n, _flags(canonical, dedup, throughput), out), "msm_g1_dev")
        return out.raw

    def msm_g2_dev(self, bases, scalars, n: int, canonical=False, dedup=False, throughput=False) -> bytes:
        out = C.create_string_buffer(193)
        self._ck(self.lib.bzk_msm_g2_dev(self.h, _ptr(bases), _ptr(scalars), n, _flags(canonical, dedup, throughput), out), "msm_g2_dev")
        return out.raw

    # ---- device-resident 4-ary tree
    def tree4_create(self, log4: int, leaves_dev=None, default_leaf: bytes = bytes(32)):
        h = C.c_void_p()
        self._ck(self.lib.bzk_tree4_create(self.h, log4, _ptr(leaves_dev), _ptr(default_leaf), C.byref(h)), "tree4_create")
        return h

    def tree4_free(self, tree):
        self.lib.bzk_tree4_free(self.h, tree)

    def tree4_root(self, tree) -> bytes:
        out = C.create_string_buffer(32)
        self._ck(self.lib.bzk_tree4_root(self.h, tree, out), "tree4_root")
        return out.raw

    def tree4_update(self, tree, indices, leaves: bytes):
        arr = (C.c_uint64 * len(indices))(*indices)
        self._ck(self.lib.bzk_tree4_update(self.h, tree, arr, _ptr(leaves), len(indices)), "tree4_update")

    def tree4_prove(self, tree, indices, log4: int) -> bytes:
        arr = (C.c_uint64 * len(indices))(*indices)
        out = C.create_string_buffer(max(1, len(indices) * log4 * 96))
        self._ck(self.lib.bzk_tree4_prove(self.h, tree, arr, len(indices), out), "tree4_prove")
        return out.raw[: len(indices) * log4 * 96]

    def tree4_node(self, tree, depth: int, index: int) -> bytes:
        out = C.create_string_buffer(32)
        self._ck(self.lib.bzk_tree4_node(self.h, tree, depth, index, out), "tree4_node")
        return out.raw

    def msm_window_count(self, n: int) -> int:
        return self.lib.bzk_msm_window_count(n)

    def msm_g1_windows_dev(self, bases, scalars, n: int, w0: int, w1: int, canonical=False, dedup=False, throughput=False) -> bytes:
        out = C.create_string_buffer(97)
        flags = _flags(canonical, dedup, throughput)
        self._ck(self.lib.bzk_msm_g1_windows_dev(self.h, _ptr(bases), _ptr(scalars), n, flags, w0, w1, out), "msm_g1_windows_dev")
        return out.raw

    def msm_g2_windows_dev(self, bases, scalars, n: int, w0: int, w1: int, canonical=False, dedup=False, throughput=False) -> bytes:
        out = C.create_string_buffer(193)
        flags = _flags(canonical, dedup, throughput)
        self._ck(self.lib.bzk_msm_g2_windows_dev(self.h, _ptr(bases), _ptr(scalars), n, flags, w0, w1, out), "msm_g2_windows_dev")
        return out.raw

    # ---- resident base sets
    def msm_bases_load_dev(self, bases, n: int, g2=False):
        """a static point set converted once into the internal form and kept in HBM (include/bzk.h: resident base sets)"""
        h = C.c_void_p()
        fn = self.lib.bzk_msm_g2_bases_load_dev if g2 else self.lib.bzk_msm_g1_bases_load_dev
        self._ck(fn(self.h, _ptr(bases), n, C.byref(h)), "msm_bases_load_dev")
        return h

    def msm_bases_info(self, handle) -> dict:
        n, forms, nbytes = C.c_uint64(0), C.c_int32(0), C.c_uint64(0)
        self._ck(self.lib.bzk_msm_bases_info(handle, C.byref(n), C.byref(forms), C.byref(nbytes)), "msm_bases_info")
        return {"n": n.value, "forms": forms.value, "device_bytes": nbytes.value}

    def msm_bases_free(self, handle):
        self.lib.bzk_msm_bases_free(self.h, handle)

    def msm_bases_run_dev(self, handle, scalars, n: int, g2=False, canonical=False, dedup=False, throughput=False) -> bytes:
        out = C.create_string_buffer(193 if g2 else 97)
        fn = self.lib.bzk_msm_g2_bases_run_dev if g2 else self.lib.bzk_msm_g1_bases_run_dev
        self._ck(fn(self.h, handle, _ptr(scalars), n, _flags(canonical, dedup, throughput), out), "msm_bases_run_dev")
        return out.raw

    def msm_bases_windows_dev(self, handle, scalars, n: int, w0: int, w1: int, g2=False, canonical=False, dedup=False) -> bytes:
        out = C.create_string_buffer(193 if g2 else 97)
        fn = self.lib.bzk_msm_g2_bases_windows_dev if g2 else self.lib.bzk_msm_g1_bases_windows_dev
        self._ck(fn(self.h, handle, _ptr(scalars), n, _flags(canonical, dedup), w0, w1, out), "msm_bases_windows_dev")
        return out.raw

    def msm_table_build_c(self, bases, n: int, c: int):
        """full G1 table with an explicit window size"""
        h = C.c_void_p()
        self._ck(self.lib.bzk_msm_g1_table_build_c(self.h, _ptr(bases), n, c, C.byref(h)), "msm_table_build_c")
        return h

    def msm_table_build(self, bases, n: int, g2=False, levels: int = 0):
        """levels = 0: full table (one level per window); levels = L: folded table (see include/bzk.h)"""
        h = C.c_void_p()
        if levels:
            fn = self.lib.bzk_msm_g2_table_build_levels if g2 else self.lib.bzk_msm_g1_table_build_levels
            self._ck(fn(self.h, _ptr(bases), n, levels, C.byref(h)), "msm_table_build_levels")
        else:
            fn = self.lib.bzk_msm_g2_table_build if g2 else self.lib.bzk_msm_g1_table_build
            self._ck(fn(self.h, _ptr(bases), n, C.byref(h)), "msm_table_build")
        return h

    def msm_table_levels(self, table) -> int:
        return self.lib.bzk_msm_table_levels(table)

    def msm_table_run_dev(self, table, scalars, n: int, g2=False, canonical=False, throughput=False) -> bytes:
        out = C.create_string_buffer(193 if g2 else 97)
        fn = self.lib.bzk_msm_g2_table_run_dev if g2 else self.lib.bzk_msm_g1_table_run_dev
        self._ck(fn(self.h, table, _ptr(scalars), n, _flags(canonical, False, throughput), out), "msm_table_run")
        return out.raw

    def msm_table_windows_dev(self, table, scalars, n: int, w0: int, w1: int, g2=False, canonical=False) -> bytes:
        out = C.create_string_buffer(193 if g2 else 97)
        fn = self.lib.bzk_msm_g2_table_windows_dev if g2 else self.lib.bzk_msm_g1_table_windows_dev
        self._ck(fn(self.h, table, _ptr(scalars), n, BZK_F_CANONICAL if canonical else 0, w0, w1, out), "msm_table_windows")
        return out.raw

    def msm_table_window_count(self, table) -> int:
        return self.lib.bzk_msm_table_window_count(table)

    def msm_table_free(self, table):
        self.lib.bzk_msm_table_free(self.h, table)

    def g1_sum(self, packed: bytes) -> bytes:
        out = C.create_string_buffer(97)
        self._ck(self.lib.bzk_g1_sum(_ptr(packed), len(packed) // 97, out), "g1_sum")
        return out.raw

    def g2_sum(self, packed: bytes) -> bytes:
        out = C.create_string_buffer(193)
        self._ck(self.lib.bzk_g2_sum(_ptr(packed), len(packed) // 193, out), "g2_sum")
        return out.raw

    # ---- general `ZkStateModel::compress`
    def state_compress(self, model_bincode: bytes, pairs):
        """pairs: iterable of (locator tuple of ints, 32-byte Montgomery scalar) -> (state_hash 32 B, state_size)"""
        pairs = list(pairs)
        off, loc = [0], []
        for l, _ in pairs:
            loc.extend(l)
            off.append(len(loc))
        o = (_u64 * len(off))(*off)
        lo = (_u64 * max(1, len(loc)))(*loc)
        vals = b"".join(v for _, v in pairs)
        out, size = C.create_string_buffer(32), _u64()
        self._ck(self.lib.bzk_state_compress(self.h, _ptr(model_bincode), len(model_bincode), o, lo, _ptr(vals), len(pairs), out, C.byref(size)),
                 "state_compress")
        return out.raw, size.value

    def state_compress_bincode(self, model_bincode: bytes, pairs_bincode: bytes) -> bytes:
        out = C.create_string_buffer(40)
        self._ck(self.lib.bzk_state_compress_bincode(self.h, _ptr(model_bincode), len(model_bincode), _ptr(pairs_bincode), len(pairs_bincode), out),
                 "state_compress_bincode")
        return out.raw

    # ---- device-resident MPN account state
    def mpn_state_compress_dev(self, log4_tree: int, log4_token_tree: int, cells, tokens) -> bytes:
        """root of a dense MPN-shaped state in device memory: cells = 4^L x 4 scalars, tokens = 4^L x 4^T x 2 scalars"""
        out = C.create_string_buffer(32)
        self._ck(self.lib.bzk_mpn_state_compress_dev(self.h, log4_tree, log4_token_tree, _ptr(cells), _ptr(tokens), out), "mpn_state_compress_dev")
        return out.raw

    def mpn_tree_create(self, log4_tree: int, log4_token_tree: int, capacity: int):
        h = C.c_void_p()
        self._ck(self.lib.bzk_mpn_tree_create(self.h, log4_tree, log4_token_tree, capacity, C.byref(h)), "mpn_tree_create")
        return h

    def mpn_tree_free(self, tree):
        self.lib.bzk_mpn_tree_free(self.h, tree)

    def mpn_tree_root(self, tree) -> bytes:
        out = C.create_string_buffer(32)
        self._ck(self.lib.bzk_mpn_tree_root(self.h, tree, out), "mpn_tree_root")
        return out.raw

    def mpn_tree_set_accounts(self, tree, accounts):
        """accounts: list of (index, (nonce, wnonce, x, y) as 4 x 32 B Montgomery, {token slot: (token_id 32 B, balance 32 B)})"""
        n = len(accounts)
        idx = (_u64 * max(1, n))(*[a[0] for a in accounts])
        cells = b"".join(b"".join(a[1]) for a in accounts)
        off, tix, tv = [0], [], []
        for a in accounts:
            for slot, (tid, bal) in a[2].items():
                tix.append(slot)
                tv.append(tid + bal)
            off.append(len(tix))
        offs = (_u64 * (n + 1))(*off)
        tixs = (_u64 * max(1, len(tix)))(*tix)
        tvb = b"".join(tv)
        self._ck(self.lib.bzk_mpn_tree_set_accounts(self.h, tree, idx, _ptr(cells), offs, tixs, _ptr(tvb), n), "mpn_tree_set_accounts")

    def mpn_tree_get_accounts(self, tree, indices, log4_token_tree: int) -> list:
        """per account: dict(cells = 4 scalars, tokens_root, tokens = {slot: (token_id, balance)} for non-zero token ids)"""
        n, ts = len(indices), 4 ** log4_token_tree
        rec = 5 + 2 * ts
        out = C.create_string_buffer(max(1, n * rec * 32))
        self._ck(self.lib.bzk_mpn_tree_get_accounts(self.h, tree, (_u64 * max(1, n))(*indices), n, out), "mpn_tree_get_accounts")
        res = []
        raw = out.raw  # ONE copy of the buffer (`.raw` copies on every access)
        for a in range(n):
            sc = [raw[32 * (a * rec + j):32 * (a * rec + j + 1)] for j in range(rec)]
            toks = {i: (sc[5 + 2 * i], sc[6 + 2 * i]) for i in range(ts) if sc[5 + 2 * i] != bytes(32)}
            res.append({"cells": sc[:4], "tokens_root": sc[4], "tokens": toks})
        return res

    def mpn_tree_prove(self, tree, indices, log4_tree: int) -> bytes:
        n = len(indices)
        out = C.create_string_buffer(max(1, n * log4_tree * 96))
        self._ck(self.lib.bzk_mpn_tree_prove(self.h, tree, (_u64 * max(1, n))(*indices), n, out), "mpn_tree_prove")
        return out.raw[: n * log4_tree * 96]

    def mpn_tree_prove_token(self, tree, accounts, slots, log4_token_tree: int) -> bytes:
        n = len(accounts)
        out = C.create_string_buffer(max(1, n * log4_token_tree * 96))
        self._ck(self.lib.bzk_mpn_tree_prove_token(self.h, tree, (_u64 * max(1, n))(*accounts), (_u64 * max(1, n))(*slots), n, out),
                 "mpn_tree_prove_token")
        return out.raw[: n * log4_token_tree * 96]

    # ---- Groth16
    def params_load(self, params: dict):
        """params: dict with n_in, n_aux, log_m, n_a, n_b and byte strings vk(870), h, l, a, b_g1, b_g2,
        a_density, b_density (the layout oracle.coracle.groth16_setup produces)."""
        keep = {k: C.create_string_buffer(bytes(params[k]), max(1, len(params[k])))
                for k in ("vk", "h", "l", "a", "b_g1", "b_g2", "a_density", "b_density")}
        d = ParamsDesc(params["n_in"], params["n_aux"], params["log_m"], params["n_a"], params["n_b"],
                       *[C.cast(keep[k], C.c_void_p) for k in ("vk", "h", "l", "a", "b_g1", "b_g2", "a_density", "b_density")])
        h = C.c_void_p()
        self._ck(self.lib.bzk_params_load(self.h, C.byref(d), C.byref(h)), "params_load")
        return h

    def params_load_bellman(self, blob: bytes, n_in: int, n_aux: int, a_density: bytes, b_density: bytes):
        """bellman `Parameters::write` bytes + the circuit shape's density maps -> (params handle, bincode Groth16VerifyingKey)"""
        h = C.c_void_p()
        vk = C.create_string_buffer(878 + 97 * n_in)
        self._ck(self.lib.bzk_params_load_bellman(self.h, _ptr(blob), len(blob), n_in, n_aux, _ptr(a_density), _ptr(b_density), C.byref(h),
                                                  vk, len(vk)), "params_load_bellman")
        return h, vk.raw

    def params_free(self, ph):
        self.lib.bzk_params_free(self.h, ph)

    def params_slot(self, ph):
        """another prover slot over the same device-resident CRS (own scratch): one ctx + one slot per concurrent prover thread"""
        h = C.c_void_p()
        self._ck(self.lib.bzk_params_slot(self.h, ph, C.byref(h)), "params_slot")
        return h

    def params_h_table(self, ph, on: bool):
        self._ck(self.lib.bzk_params_h_table(self.h, ph, int(on)), "params_h_table")

    def groth16_prove(self, ph, z, az, bz, cz, r: bytes, s: bytes) -> bytes:
        """z / az / bz / cz: bytes or zero-copy ctypes views (R1cs.raw) - passed by address, never copied here."""
        keep = [x if not isinstance(x, bytearray) else bytes(x) for x in (z, az, bz, cz)]
        if not (len(az) == len(bz) == len(cz)) or len(az) % 32 or len(z) % 32:
            raise BzkError(f"groth16_prove: z / az / bz / cz lengths {len(z)}, {len(az)}, {len(bz)}, {len(cz)}")
        asg = Assignment(*[_ptr(x) for x in keep], len(az) // 32, len(z) // 32)  # libbzk checks n_vars against the params
        out = C.create_string_buffer(387)
        self._ck(self.lib.bzk_groth16_prove(self.h, ph, C.byref(asg), _ptr(r), _ptr(s), out), "groth16_prove")
        return out.raw

    def groth16_prove_r1cs(self, ph, r1cs, r: bytes, s: bytes) -> bytes:
        """the same over an R1cs of the host generator; an instance synthesized with MpnWorld.set_defer(True) is completed on the device first"""
        out = C.create_string_buffer(387)
        self._ck(self.lib.bzk_groth16_prove_r1cs(self.h, ph, r1cs.h, _ptr(r), _ptr(s), out), "groth16_prove_r1cs")
        return out.raw

    def r1cs_stage(self, r1cs):
        """the instance's arrays into HBM on THIS context's stream + its deferred-value program behind them; returns at once (a staged handle).
        Keep `r1cs` alive until staged_wait / a prove call on the handle has returned."""
        h = C.c_void_p()
        self._ck(self.lib.bzk_r1cs_stage(self.h, r1cs.h, C.byref(h)), "r1cs_stage")
        return h

    def staged_wait(self, staged):
        self._ck(self.lib.bzk_staged_wait(staged), "staged_wait")

    def staged_free(self, staged):
        self.lib.bzk_staged_free(staged)

    def staged_read(self, staged, which: int) -> bytes:
        """one staged array (0 z, 1 az, 2 bz, 3 cz) as the DEVICE left it after the uploads and the deferred-value program"""
        n = _u64()
        self._ck(self.lib.bzk_staged_read(staged, which, None, 0, C.byref(n)), "staged_read")
        buf = (C.c_uint8 * n.value)()
        self._ck(self.lib.bzk_staged_read(staged, which, buf, n.value, None), "staged_read")
        return bytes(buf)

    def groth16_prove_staged(self, ph, staged, r: bytes, s: bytes) -> bytes:
        out = C.create_string_buffer(387)
        self._ck(self.lib.bzk_groth16_prove_staged(self.h, ph, staged, _ptr(r), _ptr(s), out), "groth16_prove_staged")
        return out.raw

    def params_read(self, ph, which: int) -> bytes:
        n = _u64()
        self._ck(self.lib.bzk_params_read(self.h, ph, which, None, 0, C.byref(n)), "params_read")
        buf = C.create_string_buffer(max(1, n.value))
        self._ck(self.lib.bzk_params_read(self.h, ph, which, buf, n.value, None), "params_read")
        return buf.raw[: n.value]

    def groth16_setup(self, csr_abc, n_in: int, n_aux: int, toxic: bytes):
        """csr_abc: three (n_rows, row_ptr_bytes(u32), col_bytes(u32), val_bytes) tuples.  Returns (params handle,
        vk bincode bytes = Groth16VerifyingKey)."""
        keep, descs = [], []
        for n_rows, rp, col, val in csr_abc:
            # bytes are copied into stable buffers; ctypes arrays (R1cs.raw: zero-copy views of the generator's own
            # arrays - the only option once a matrix passes 2 GiB) are passed by address
            bufs = [x if isinstance(x, C.Array) else C.create_string_buffer(bytes(x), max(1, len(x))) for x in (rp, col, val)]
            keep.append(bufs)
            descs.append(CsrDesc(n_rows, *[C.cast(C.addressof(b), C.c_void_p) for b in bufs]))
        h = C.c_void_p()
        vk = C.create_string_buffer(878 + 97 * n_in)
        self._ck(self.lib.bzk_groth16_setup(self.h, C.byref(descs[0]), C.byref(descs[1]), C.byref(descs[2]), n_in, n_aux,
                                            _ptr(toxic), C.byref(h), vk, len(vk)), "groth16_setup")
        return h, vk.raw

    def groth16_h_dev(self, a, b, c, log_m: int):
        self._ck(self.lib.bzk_groth16_h_dev(self.h, _ptr(a), _ptr(b), _ptr(c), log_m), "groth16_h_dev")

    def g1_synth_bases_dev(self, seed: int, start: int, n: int, out):
        # test / bench helper: synchronised, so that a caller reading `out` through another stream (torch) cannot race the kernel
        self._ck(self.lib.bzk_g1_synth_bases_dev(self.h, seed, start, n, _ptr(out)), "g1_synth_bases_dev")
        self.sync()

    def g2_synth_bases_dev(self, seed: int, start: int, n: int, out):
        self._ck(self.lib.bzk_g2_synth_bases_dev(self.h, seed, start, n, _ptr(out)), "g2_synth_bases_dev")
        self.sync()


# --------------------------------------------------------------------------------------------------
# host-side (CPU, C++) MPN witness / R1CS generator - no GPU needed
# --------------------------------------------------------------------------------------------------
MG_X_AUTO, MG_X_HOST, MG_X_PEER, MG_X_RCCL = 0, 1, 2, 3
MG_X_NAMES = {MG_X_HOST: "host", MG_X_PEER: "peer", MG_X_RCCL: "rccl"}


def mg_probe(device: int) -> int:
    """bit 0: the device is a usable gfx950; bit 1: librccl loads with everything the RCCL exchange needs (bzk_mg_probe)"""
    st = load_library().bzk_mg_probe(device)
    if st < 0:
        raise BzkError(f"bzk_mg_probe: {load_library().bzk_strerror(st).decode()}")
    return st


def mg_unique_id() -> bytes:
    """group id of a process-per-GPU deployment: rank 0 draws it, the host hands it to the other ranks (bzk_mg_unique_id)"""
    lib = load_library()
    out = C.create_string_buffer(128)
    st = lib.bzk_mg_unique_id(out)
    if st != 0:
        raise BzkError(f"bzk_mg_unique_id: {lib.bzk_strerror(st).decode()}")
    return out.raw


class Mg:
    """A device group behind the C ABI (include/bzk.h row (e)): window-sharded MSMs and a proof pool over 1..8 GPUs.
    Mg(devices=[0, 1, ...])                 one process drives the devices
    Mg(device=d, rank=r, world=N, uid=...)  one process per GPU (uid from mg_unique_id() on rank 0)"""

    def __init__(self, devices=None, device=None, rank=None, world=None, uid=None, exchange=MG_X_AUTO):
        self.lib = load_library()
        h = C.c_void_p()
        if devices is not None:
            arr = (_i32 * len(devices))(*devices)
            st = self.lib.bzk_mg_create(arr, len(devices), exchange, C.byref(h))
        else:
            st = self.lib.bzk_mg_create_rank(device, rank, world, _ptr(uid), exchange, C.byref(h))
        if st != 0:
            raise BzkError(f"bzk_mg_create: {self.lib.bzk_strerror(st).decode()}")
        self.h = h
        self.world, self.local, self.rank = self.lib.bzk_mg_world(h), self.lib.bzk_mg_local(h), self.lib.bzk_mg_rank(h)
        self.exchange = MG_X_NAMES.get(self.lib.bzk_mg_exchange(h), "?")

    def close(self):
        if getattr(self, "h", None):
            self.lib.bzk_mg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, what):
        if st != 0:
            raise BzkError(f"{what}: {self.lib.bzk_strerror(st).decode()} [{self.lib.bzk_mg_last_error(self.h).decode()}]")

    def _ptrs(self, xs):
        assert len(xs) == self.local, "one device pointer per local device"
        return (_vp * self.local)(*[_ptr(x) for x in xs])

    def stats(self, reset: bool = False) -> dict:
        """where this rank's window-sharded calls spent their time (bzk_mg_stats)"""
        v = (C.c_double * 8)()
        self._ck(self.lib.bzk_mg_stats(self.h, int(reset), v), "mg_stats")
        return {"calls": int(v[0]), "local_ms": v[1], "exchange_ms": v[2], "peer_wait_ms": v[3], "combine_ms": v[4], "create_s": v[5],
                "comm_init_s": v[6]}

    def ctx_handle(self, i: int):
        return self.lib.bzk_mg_ctx(self.h, i)

    def bases_load(self, bases_host: bytes, n: int, g2=False):
        h = C.c_void_p()
        fn = self.lib.bzk_mg_bases_g2_load if g2 else self.lib.bzk_mg_bases_g1_load
        self._ck(fn(self.h, _ptr(bases_host), n, C.byref(h)), "mg_bases_load")
        return h

    def bases_load_dev(self, bases_dev, n: int, g2=False):
        h = C.c_void_p()
        fn = self.lib.bzk_mg_bases_g2_load_dev if g2 else self.lib.bzk_mg_bases_g1_load_dev
        self._ck(fn(self.h, self._ptrs(bases_dev), n, C.byref(h)), "mg_bases_load_dev")
        return h

    def bases_free(self, handle):
        self.lib.bzk_mg_bases_free(self.h, handle)

    def msm(self, bases, scalars_host: bytes, n: int, g2=False, canonical=False, dedup=False, throughput=False) -> bytes:
        out = C.create_string_buffer(193 if g2 else 97)
        fn = self.lib.bzk_mg_msm_g2 if g2 else self.lib.bzk_mg_msm_g1
        self._ck(fn(self.h, bases, _ptr(scalars_host), n, _flags(canonical, dedup, throughput), out), "mg_msm")
        return out.raw

    def msm_dev(self, bases, scalars_dev, n: int, g2=False, canonical=False, dedup=False, throughput=False) -> bytes:
        out = C.create_string_buffer(193 if g2 else 97)
        fn = self.lib.bzk_mg_msm_g2_dev if g2 else self.lib.bzk_mg_msm_g1_dev
        self._ck(fn(self.h, bases, self._ptrs(scalars_dev), n, _flags(canonical, dedup, throughput), out), "mg_msm_dev")
        return out.raw

    # ---- proof pool
    def params_load(self, params: dict, slots_per_device: int = 2):
        keep = {k: C.create_string_buffer(bytes(params[k]), max(1, len(params[k])))
                for k in ("vk", "h", "l", "a", "b_g1", "b_g2", "a_density", "b_density")}
        d = ParamsDesc(params["n_in"], params["n_aux"], params["log_m"], params["n_a"], params["n_b"],
                       *[C.cast(keep[k], C.c_void_p) for k in ("vk", "h", "l", "a", "b_g1", "b_g2", "a_density", "b_density")])
        h = C.c_void_p()
        self._ck(self.lib.bzk_mg_params_load(self.h, C.byref(d), slots_per_device, C.byref(h)), "mg_params_load")
        return h

    def params_free(self, ph):
        self.lib.bzk_mg_params_free(self.h, ph)

    def params_stats(self, ph):
        n = self.lib.bzk_mg_params_slots(ph)
        arr = (_u64 * max(1, n))()
        self._ck(self.lib.bzk_mg_params_stats(ph, arr, n), "mg_params_stats")
        return list(arr)[:n]

    def prove_submit(self, ph, z, az, bz, cz, r: bytes, s: bytes):
        """returns a pending-proof record; keep it (it owns the buffers) until prove_wait"""
        asg = Assignment(*[_ptr(x) for x in (z, az, bz, cz)], len(az) // 32, len(z) // 32)
        out = C.create_string_buffer(387)
        t = _u64()
        self._ck(self.lib.bzk_mg_prove_submit(self.h, ph, C.byref(asg), _ptr(r), _ptr(s), out, C.byref(t)), "mg_prove_submit")
        return {"ticket": t.value, "out": out, "keep": (asg, z, az, bz, cz, r, s), "ph": ph}

    def prove_wait(self, pending) -> bytes:
        self._ck(self.lib.bzk_mg_prove_wait(self.h, pending["ph"], pending["ticket"]), "mg_prove_wait")
        return pending["out"].raw


def _st(st, what):
    if st != 0:
        raise BzkError(f"{what}: {load_library().bzk_strerror(st).decode()}")


class R1cs:
    """A synthesized circuit instance (assignment + optional CSR matrices)."""
    VIEWS = {"z": 0, "az": 1, "bz": 2, "cz": 3, "a_density": 4, "b_density": 5, "valA": 6, "valB": 7, "valC": 8,
             "colA": 9, "colB": 10, "colC": 11, "rpA": 12, "rpB": 13, "rpC": 14}

    def __init__(self, handle):
        self.lib = load_library()
        self.h = handle
        info = (_u64 * 9)()
        _st(self.lib.bzk_r1cs_info(self.h, info), "r1cs_info")
        (self.n_in, self.n_aux, self.n_constraints, self.nnzA, self.nnzB, self.nnzC, first_bad, self.accepted,
         self.rejected) = [int(x) for x in info]
        self.first_unsatisfied = first_bad - 1
        self.satisfied = first_bad == 0

    DEFER_FIELDS = ("deferred", "n_tx", "n_ops", "n_regs", "n_inputs", "n_levels", "hole_aux", "hole_con", "filled", "flags")

    def defer_info(self) -> dict:
        """what a synthesis with MpnWorld.set_defer(True) left to the device (bzk_r1cs_defer_info)"""
        info = (_u64 * 10)()
        _st(self.lib.bzk_r1cs_defer_info(self.h, info), "r1cs_defer_info")
        return dict(zip(self.DEFER_FIELDS, [int(x) for x in info]))

    def defer_schedule_info(self) -> dict:
        """the one-launch device schedule of this instance's deferred-value program, checked on the host (bzk_r1cs_defer_schedule_info)"""
        info = (_u64 * 6)()
        _st(self.lib.bzk_r1cs_defer_schedule_info(self.h, info), "r1cs_defer_schedule_info")
        return dict(zip(("stages", "segments", "hash_ops", "fill_ops", "largest_segment", "violations"), [int(x) for x in info]))

    def fill_host(self) -> dict:
        """runs the instance's deferred-value program on the CPU (same ops as the device): the views are complete afterwards"""
        _st(self.lib.bzk_r1cs_fill_host(self.h), "r1cs_fill_host")
        return self.defer_info()

    def view(self, name: str) -> bytes:
        """a COPY of the array as bytes (tests); the prover path uses raw()"""
        n = _u64()
        p = self.lib.bzk_r1cs_data(self.h, self.VIEWS[name], C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def raw(self, name: str):
        """zero-copy ctypes view of the array inside the generator's (pinned) buffer; valid while this object lives"""
        n = _u64()
        p = self.lib.bzk_r1cs_data(self.h, self.VIEWS[name], C.byref(n))
        arr = (C.c_char * max(1, n.value)).from_address(p) if n.value else (C.c_char * 1)()
        arr._owner = self
        return arr

    def free(self):
        if self.h:
            self.lib.bzk_r1cs_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MpnWorld:
    def __init__(self, log4_tree: int, log4_token_tree: int):
        self.lib = load_library()
        h = C.c_void_p()
        _st(self.lib.bzk_mpn_create(log4_tree, log4_token_tree, C.byref(h)), "mpn_create")
        self.h = h

    def close(self):
        if self.h:
            self.lib.bzk_mpn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_height(self, height: int):
        _st(self.lib.bzk_mpn_set_height(self.h, height), "set_height")

    def set_threads(self, n: int):
        _st(self.lib.bzk_mpn_set_threads(self.h, n), "set_threads")

    def set_defer(self, on: bool = True):
        """witness-only update instances leave the hash-dependent values to the device (Bzk.groth16_prove_r1cs) or to R1cs.fill_host"""
        _st(self.lib.bzk_mpn_set_defer(self.h, 1 if on else 0), "set_defer")

    def set_device(self, ctx):
        """ctx: a Bzk context (kept alive by this object) whose GPU batches the builders' Merkle hashing; None: host path"""
        self._dev = ctx
        _st(self.lib.bzk_mpn_set_device(self.h, ctx.h if ctx is not None else None), "set_device")

    def add_account(self, index: int, seed: bytes, token_id: bytes, balance: int) -> bytes:
        out = C.create_string_buffer(64)
        _st(self.lib.bzk_mpn_add_account(self.h, index, _ptr(seed), len(seed), _ptr(token_id), balance, out), "add_account")
        return out.raw

    def add_key(self, index: int, seed: bytes):
        _st(self.lib.bzk_mpn_add_key(self.h, index, _ptr(seed), len(seed)), "add_key")

    def root(self) -> bytes:
        out = C.create_string_buffer(32)
        _st(self.lib.bzk_mpn_root(self.h, out), "root")
        return out.raw

    def push_tx(self, src: int, dst: int, token_id: bytes, amount: int, fee_token: bytes, fee: int):
        _st(self.lib.bzk_mpn_push_tx(self.h, src, dst, _ptr(token_id), amount, _ptr(fee_token), fee), "push_tx")

    def push_deposit(self, key_index: int, token_id: bytes, amount: int):
        _st(self.lib.bzk_mpn_push_deposit(self.h, key_index, _ptr(token_id), amount), "push_deposit")

    def push_withdraw(self, account: int, token_id: bytes, amount: int, fee_token: bytes, fee: int, fingerprint: bytes | None = None):
        """fingerprint None: derived from a synthetic L1 payment as the wallet does (the withdrawal can go on the wire)"""
        _st(self.lib.bzk_mpn_push_withdraw(self.h, account, _ptr(token_id), amount, _ptr(fee_token), fee, _ptr(fingerprint)), "push_withdraw")

    def make_work(self, kind: int, vks, reward: int, log4_batches=(1, 1, 1), num_batches=(1, 1, 1), state_size: int = 0) -> "MpnWork":
        """validator side (`prepare_works`): one work from the queued transactions.  kind 0 deposit / 1 withdraw / 2 update;
        vks = (deposit, withdraw, update) bincode Groth16VerifyingKey; log4_batches = (deposit, withdraw, update);
        num_batches = (update, deposit, withdraw) counts of MpnConfig."""
        keep = [C.create_string_buffer(bytes(v), len(v)) for v in vks]
        cfg = WorkConfig(log4_batches[0], log4_batches[1], log4_batches[2], num_batches[0], num_batches[1], num_batches[2],
                         C.cast(keep[0], _vp), len(vks[0]), C.cast(keep[1], _vp), len(vks[1]), C.cast(keep[2], _vp), len(vks[2]),
                         state_size)
        h = C.c_void_p()
        _st(self.lib.bzk_mpn_make_work(self.h, kind, C.byref(cfg), reward, C.byref(h)), "make_work")
        return MpnWork(h)

    def deposit_synthesize(self, log4_batch: int, commitment: bytes, record_matrices=False) -> R1cs:
        h = C.c_void_p()
        _st(self.lib.bzk_mpn_deposit_synthesize(self.h, log4_batch, _ptr(commitment), int(record_matrices), C.byref(h)), "deposit_synthesize")
        return R1cs(h)

    def withdraw_synthesize(self, log4_batch: int, commitment: bytes, record_matrices=False) -> R1cs:
        h = C.c_void_p()
        _st(self.lib.bzk_mpn_withdraw_synthesize(self.h, log4_batch, _ptr(commitment), int(record_matrices), C.byref(h)), "withdraw_synthesize")
        return R1cs(h)

    def update_synthesize(self, log4_batch: int, commitment: bytes, fee_token: bytes, record_matrices=False) -> R1cs:
        h = C.c_void_p()
        _st(self.lib.bzk_mpn_update_synthesize(self.h, log4_batch, _ptr(commitment), _ptr(fee_token), int(record_matrices),
                                               C.byref(h)), "update_synthesize")
        return R1cs(h)


class MpnWork:
    """One `MpnWork` (src/mpn/mod.rs:263-270): decoded from bincode bytes (worker side) or made from a world (validator side)."""
    KINDS = ("deposit", "withdraw", "update")

    def __init__(self, handle, consumed: int = 0):
        self.lib = load_library()
        self.h = handle
        self.consumed = consumed
        info = (_u64 * 12)()
        _st(self.lib.bzk_mpn_work_info(self.h, info), "work_info")
        (self.kind, self.log4_tree, self.log4_token_tree, self.log4_batch, self.n_transitions, self.height, self.reward,
         self.new_root_size, self.num_update_batches, self.num_deposit_batches, self.num_withdraw_batches, self.vk_len) = [int(x) for x in info]
        sc = C.create_string_buffer(160)
        _st(self.lib.bzk_mpn_work_scalars(self.h, sc), "work_scalars")
        self.state, self.aux_data, self.next_state, self.new_root_hash, self.contract_id = [sc.raw[32 * i:32 * i + 32] for i in range(5)]

    @classmethod
    def decode(cls, data, flags: int = 0, offset: int = 0) -> "MpnWork":
        """decodes the work at data[offset:] (bytes or a ctypes buffer) without slicing - a response holds several works"""
        lib = load_library()
        h, used = C.c_void_p(), _u64()
        if offset < 0 or offset > len(data):
            raise BzkError("work_decode: offset out of range")
        base = _ptr(data)
        at = C.c_void_p((base.value or 0) + offset) if offset else base
        st = lib.bzk_mpn_work_decode(at, len(data) - offset, flags, C.byref(h), C.byref(used))
        if st != 0:
            raise BzkError(f"work_decode: {lib.bzk_strerror(st).decode()} [{lib.bzk_mpn_work_last_error().decode()}]")
        return cls(h, used.value)

    def free(self):
        if self.h:
            self.lib.bzk_mpn_work_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def vk(self, which: int = -1) -> bytes:
        n = _u64()
        _st(self.lib.bzk_mpn_work_vk(self.h, which, None, 0, C.byref(n)), "work_vk")
        buf = C.create_string_buffer(max(1, n.value))
        _st(self.lib.bzk_mpn_work_vk(self.h, which, buf, n.value, None), "work_vk")
        return buf.raw[: n.value]

    def commitment(self, prover_pub: bytes) -> bytes:
        out = C.create_string_buffer(32)
        _st(self.lib.bzk_mpn_work_commitment(self.h, _ptr(prover_pub), out), "work_commitment")
        return out.raw

    def verify(self, prover_pub: bytes, proof387: bytes) -> bool:
        """`MpnWork::verify` (src/mpn/mod.rs:281-295), on the host"""
        st = self.lib.bzk_mpn_work_verify(self.h, _ptr(prover_pub), _ptr(proof387))
        if st < 0:
            raise BzkError(f"work_verify: {self.lib.bzk_strerror(st).decode()}")
        return st == 1

    def encode(self) -> bytes:
        n = _u64()
        _st(self.lib.bzk_mpn_work_encode(self.h, None, 0, C.byref(n)), "work_encode")
        buf = C.create_string_buffer(max(1, n.value))
        _st(self.lib.bzk_mpn_work_encode(self.h, buf, n.value, None), "work_encode")
        return buf.raw[: n.value]

    def synthesize(self, prover_pub: bytes, fee_token: bytes | None = None, threads: int = 0, record_matrices=False, defer=False) -> R1cs:
        """defer: BZK_SYNTH_DEFER - a work's hash-dependent values (all three kinds: update, deposit, withdraw) are left to the device (Bzk.groth16_prove_r1cs) / R1cs.fill_host"""
        h = C.c_void_p()
        _st(self.lib.bzk_mpn_work_synthesize(self.h, _ptr(prover_pub), _ptr(fee_token), threads, 2 if defer else int(record_matrices), C.byref(h)),
            "work_synthesize")
        return R1cs(h)


def bellman_params_decode(blob: bytes, threads: int = 0) -> dict:
    """bellman `Parameters::write` bytes -> dict(vk (870 packed), ic (97 B each), h, l, a, b_g1 (raw 96 B), b_g2 (raw 192 B), counts)"""
    lib = load_library()
    info = (_u64 * 7)()
    _st(lib.bzk_bellman_params_info(_ptr(blob), len(blob), info), "bellman_params_info")
    n_ic, n_h, n_l, n_a, n_b1, n_b2, used = [int(x) for x in info]
    bufs = {k: C.create_string_buffer(max(1, sz)) for k, sz in dict(vk=870, ic=97 * n_ic, h=96 * n_h, l=96 * n_l, a=96 * n_a, b_g1=96 * n_b1,
                                                                     b_g2=192 * n_b2).items()}
    _st(lib.bzk_bellman_params_decode(_ptr(blob), len(blob), bufs["vk"], bufs["ic"], bufs["h"], bufs["l"], bufs["a"], bufs["b_g1"],
                                      bufs["b_g2"], threads), "bellman_params_decode")
    sizes = dict(vk=870, ic=97 * n_ic, h=96 * n_h, l=96 * n_l, a=96 * n_a, b_g1=96 * n_b1, b_g2=192 * n_b2)
    out = {k: v.raw[: sizes[k]] for k, v in bufs.items()}
    out.update(n_ic=n_ic, n_h=n_h, n_l=n_l, n_a=n_a, n_b_g1=n_b1, n_b_g2=n_b2, consumed=used)
    return out


def bellman_params_encode(vk870: bytes, ic: bytes, h: bytes, l: bytes, a: bytes, b_g1: bytes, b_g2: bytes) -> bytes:
    lib = load_library()
    n = _u64()
    args = (_ptr(vk870), _ptr(ic), len(ic) // 97, _ptr(h), len(h) // 96, _ptr(l), len(l) // 96, _ptr(a), len(a) // 96, _ptr(b_g1), _ptr(b_g2),
            len(b_g1) // 96)
    _st(lib.bzk_bellman_params_encode(*args, None, 0, C.byref(n)), "bellman_params_encode")
    buf = C.create_string_buffer(n.value)
    _st(lib.bzk_bellman_params_encode(*args, buf, n.value, None), "bellman_params_encode")
    return buf.raw


def host_default_threads() -> int:
    """the host generator's default thread count (visible CPUs capped by the cgroup CPU quota; BZK_HOST_THREADS overrides)"""
    return int(load_library().bzk_host_default_threads())


def groth16_verify(vk_bincode: bytes, inputs: bytes, proof387: bytes) -> bool:
    """host-side `groth16_verify` (src/zk/groth16/mod.rs:67-121): inputs = n x 32 B Montgomery scalars"""
    st = load_library().bzk_groth16_verify(_ptr(vk_bincode), len(vk_bincode), _ptr(inputs), len(inputs) // 32, _ptr(proof387))
    if st < 0:
        raise BzkError(f"groth16_verify: {load_library().bzk_strerror(st).decode()}")
    return st == 1


def zkproof_encode(proof387: bytes) -> bytes:
    out = C.create_string_buffer(391)
    _st(load_library().bzk_zkproof_encode(_ptr(proof387), out), "zkproof_encode")
    return out.raw


def zkproof_decode(data: bytes) -> bytes:
    out = C.create_string_buffer(387)
    _st(load_library().bzk_zkproof_decode(_ptr(data), len(data), out), "zkproof_decode")
    return out.raw


def mpn_update_empty(L, T, B, commitment, height, state, aux, next_state, fee_token, record_matrices=False) -> R1cs:
    h = C.c_void_p()
    _st(load_library().bzk_mpn_update_empty(L, T, B, _ptr(commitment), height, _ptr(state), _ptr(aux), _ptr(next_state),
                                            _ptr(fee_token), int(record_matrices), C.byref(h)), "update_empty")
    return R1cs(h)


def mpn_circuit_empty(kind, L, T, B, commitment, height, state, aux, next_state, record_matrices=False) -> R1cs:
    """kind: 0 deposit, 1 withdraw"""
    h = C.c_void_p()
    _st(load_library().bzk_mpn_circuit_empty(kind, L, T, B, _ptr(commitment), height, _ptr(state), _ptr(aux), _ptr(next_state),
                                             int(record_matrices), C.byref(h)), "circuit_empty")
    return R1cs(h)


def _csr(locators):
    off, loc = [0], []
    for l in locators:
        loc.extend(int(x) for x in l)
        off.append(len(loc))
    return (_u64 * len(off))(*off), (_u64 * max(1, len(loc)))(*loc)


class DeviceState:
    """`KvStoreStateManager` for one contract with the values resident on the device (bzk_state_*): update / root / get / prove"""

    def __init__(self, ctx: "Bzk", model_bincode: bytes):
        self.ctx, self.lib, self.h = ctx, ctx.lib, _vp()
        ctx._ck(self.lib.bzk_state_create(ctx.h, _ptr(model_bincode), len(model_bincode), C.byref(self.h)), "state_create")

    def close(self):
        if self.h:
            self.lib.bzk_state_free(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            import sys
            if not sys.is_finalizing():    # at interpreter exit the HIP runtime may be gone already; the process frees the memory
                self.close()
        except Exception:
            pass

    def update(self, pairs, target_height: int, want_rollback=False):
        """pairs: iterable of (locator, 32-byte Montgomery scalar; zero = remove) -> (state_hash, state_size[, previous values])"""
        pairs = list(pairs)
        o, lo = _csr(l for l, _ in pairs)
        vals = b"".join(v for _, v in pairs)
        out, size = C.create_string_buffer(32), _u64()
        prev = C.create_string_buffer(max(1, 32 * len(pairs))) if want_rollback else None
        self.ctx._ck(self.lib.bzk_state_update(self.h, o, lo, _ptr(vals), len(pairs), target_height, out, C.byref(size), prev), "state_update")
        if want_rollback:
            return out.raw, size.value, [prev.raw[32 * i:32 * i + 32] for i in range(len(pairs))]
        return out.raw, size.value

    def update_bincode(self, delta_bincode: bytes, target_height: int) -> bytes:
        out = C.create_string_buffer(40)
        self.ctx._ck(self.lib.bzk_state_update_bincode(self.h, _ptr(delta_bincode), len(delta_bincode), target_height, out), "state_update_bincode")
        return out.raw

    def root(self):
        """-> (state_hash, state_size, height)"""
        out, size, height = C.create_string_buffer(32), _u64(), _u64()
        self.ctx._ck(self.lib.bzk_state_root(self.h, out, C.byref(size), C.byref(height)), "state_root")
        return out.raw, size.value, height.value

    def get(self, locators) -> list:
        locators = list(locators)
        o, lo = _csr(locators)
        out = C.create_string_buffer(max(1, 32 * len(locators)))
        self.ctx._ck(self.lib.bzk_state_get(self.h, o, lo, len(locators), out), "state_get")
        return [out.raw[32 * i:32 * i + 32] for i in range(len(locators))]

    def prove(self, tree_loc, indices) -> list:
        """-> per index: log4_size levels (leaf level first) of three 32-byte siblings"""
        tree_loc, indices = [int(x) for x in tree_loc], [int(x) for x in indices]
        tl = (_u64 * max(1, len(tree_loc)))(*tree_loc)
        ix = (_u64 * max(1, len(indices)))(*indices)
        log4 = _u32()
        self.ctx._ck(self.lib.bzk_state_prove(self.h, tl, len(tree_loc), None, 0, None, C.byref(log4)), "state_prove")  # the depth
        out = C.create_string_buffer(max(1, len(indices) * log4.value * 96))
        self.ctx._ck(self.lib.bzk_state_prove(self.h, tl, len(tree_loc), ix, len(indices), out, C.byref(log4)), "state_prove")
        L, raw = log4.value, out.raw
        return [[[raw[((i * L + d) * 3 + j) * 32:((i * L + d) * 3 + j) * 32 + 32] for j in range(3)] for d in range(L)] for i in range(len(indices))]

    def stats(self) -> dict:
        a, b, c = _u64(), _u64(), _u64()
        self.ctx._ck(self.lib.bzk_state_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), "state_stats")
        return {"slots": a.value, "device_bytes": b.value, "keys": c.value}


def state_model_default(model_bincode: bytes) -> bytes:
    """`ZkStateModel::compress_default` (host)"""
    out = C.create_string_buffer(32)
    _st(load_library().bzk_state_model_default(_ptr(model_bincode), len(model_bincode), out), "state_model_default")
    return out.raw


def host_poseidon(inp: bytes) -> bytes:
    out = C.create_string_buffer(32)
    _st(load_library().bzk_host_poseidon(_ptr(inp), len(inp) // 32, out), "host_poseidon")
    return out.raw


def host_scalar_new(le_bytes: bytes) -> bytes:
    """`ZkScalar::new`: little-endian integer mod r -> 32 Montgomery bytes"""
    out = C.create_string_buffer(32)
    _st(load_library().bzk_host_scalar_new(_ptr(le_bytes) if le_bytes else None, len(le_bytes), out), "host_scalar_new")
    return out.raw


def host_sha3_256(b: bytes) -> bytes:
    out = C.create_string_buffer(32)
    _st(load_library().bzk_host_sha3_256(_ptr(b) if b else None, len(b), out), "host_sha3")
    return out.raw


def host_jubjub_keys(seed: bytes) -> bytes:
    out = C.create_string_buffer(128)
    _st(load_library().bzk_host_jubjub_keys(_ptr(seed), len(seed), out), "jubjub_keys")
    return out.raw


def host_jubjub_sign(key: bytes, msg: bytes) -> bytes:
    out = C.create_string_buffer(96)
    _st(load_library().bzk_host_jubjub_sign(_ptr(key), _ptr(msg), out), "jubjub_sign")
    return out.raw


def host_jubjub_verify(pub_xy: bytes, msg: bytes, sig: bytes) -> bool:
    r = load_library().bzk_host_jubjub_verify(_ptr(pub_xy), _ptr(msg), _ptr(sig))
    if r < 0:
        raise BzkError("jubjub_verify: bad argument")
    return bool(r)
