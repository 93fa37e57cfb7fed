"""bazuka_amd - MI355X-native Groth16 hot path for Bazuka's MPN rollup.

The product is the C-ABI shared library ``libbzk.so`` (include/bzk.h) built from hand-written HIP
for gfx950 under ``bazuka_amd/csrc``.  This Python package is only the thin ctypes driver used by
tests/ and bench.py (device memory and process-group plumbing come from PyTorch-ROCm).  There is
no CPU fallback anywhere in this package: if the library or a gfx950 device is missing, calls fail.
"""
import os as _os

# A prover keeps 4 slots x 4 streams busy; HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
# queue run one behind the other.  16 queues: pipelined proofs/s +4 - 6 %, four MSMs in flight 1.14x -> 1.19x of one at a time
# (profiles/r06_run1_msms_in_flight_and_hw_queues.txt, r06_run2...).  The HIP runtime reads the variable when it initialises - on the process's first HIP
# call - so it is set here, at import, unless the host chose a value itself (the native worker does the same in its main()).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .lib import Bzk, BzkError, DeviceState, Mg, load_library, mg_probe, mg_unique_id, LIB_PATH  # noqa: F401
