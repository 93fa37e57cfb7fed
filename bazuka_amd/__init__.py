"""bazuka_amd - MI355X-native Groth16 hot path for Bazuka's MPN rollup.

The product is the C-ABI shared library ``libbzk.so`` (include/bzk.h) built from hand-written HIP
for gfx950 under ``bazuka_amd/csrc``.  This Python package is only the thin ctypes driver used by
tests/ and bench.py (device memory and process-group plumbing come from PyTorch-ROCm).  There is
no CPU fallback anywhere in this package: if the library or a gfx950 device is missing, calls fail.
"""
from .lib import Bzk, BzkError, DeviceState, Mg, load_library, mg_probe, mg_unique_id, LIB_PATH  # noqa: F401
