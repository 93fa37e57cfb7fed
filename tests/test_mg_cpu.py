"""CPU suite: the device-group entry points (bzk_mg_*) validate their arguments and, like every other device entry, fail loudly
without a gfx950 device - there is no CPU fallback behind them; a group id can still be drawn (128 bytes)."""
import ctypes as C

import pytest


def test_mg_argument_validation_and_no_cpu_fallback():
    import torch
    from bazuka_amd import lib as L
    lib = L.load_library()
    h = C.c_void_p()
    one = (C.c_int32 * 1)(0)
    assert lib.bzk_mg_create(None, 1, 0, C.byref(h)) == -1
    assert lib.bzk_mg_create(one, 0, 0, C.byref(h)) == -1
    assert lib.bzk_mg_create(one, 1, 9, C.byref(h)) == -1          # unknown transport
    uid = L.mg_unique_id()
    assert len(uid) == 128 and len(set(uid)) > 8
    assert lib.bzk_mg_create_rank(0, 2, 2, uid, 0, C.byref(h)) == -1   # rank outside the world
    assert lib.bzk_mg_create_rank(0, 0, 1, None, 0, C.byref(h)) == -1
    assert lib.bzk_mg_world(None) == 0 and lib.bzk_mg_local(None) == 0 and lib.bzk_mg_ctx(None, 0) is None
    if torch.cuda.is_available():
        pytest.skip("GPU present: the loud-failure half is for hosts without one")
    assert lib.bzk_mg_create(one, 1, 0, C.byref(h)) == -3 and not h.value   # BZK_E_DEVICE, no handle
    with pytest.raises(L.BzkError):
        L.Mg(devices=[0])
