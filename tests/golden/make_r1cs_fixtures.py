"""Generates tests/golden/r1cs_sha256.json: per scenario (tests/r1cs_scenarios.py) the variable / constraint counts and the
sha256 of each of the 15 arrays of a synthesized instance (z, A.z, B.z, C.z, densities, CSR val / col / row_ptr of A, B, C)
AS THE INDEPENDENT PYTHON RESTATEMENT of the reference's circuits computes them (oracle/pycircuit.py) from the work's
bincode bytes parsed by tests/bincode_ref.py.  The product never feeds these hashes: it only supplies the scenario's
`MpnWork` bytes (whose sha256 is recorded too, so a drifting witness builder is noticed).

    python tests/golden/make_r1cs_fixtures.py        # ~1 min (pure Python, 903 037 constraints for the largest)
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import bincode_ref as B  # noqa: E402
import r1cs_scenarios as S  # noqa: E402
from oracle import pycircuit as pc  # noqa: E402


def fixture(name):
    blob = S.make_work(name)
    work = B.decode(B.MpnWork, blob)
    com = B.encode(B.Address, S.PROVER) + B.encode(B.U64, work["reward"])
    commitment = int.from_bytes(hashlib.sha3_256(com).digest(), "little") % pc.R_MOD  # src/mpn/mod.rs:283-285
    cs = pc.circuit_of_work(work, commitment, 1, lambda p: B.encode(B.ContractWithdraw, p))
    assert pc.first_unsatisfied(cs) == -1
    views = pc.all_views(cs)
    return {"work_sha256": hashlib.sha256(blob).hexdigest(), "n_in": cs.n_in, "n_aux": cs.n_aux,
            "n_constraints": len(cs.A) + cs.n_in, "sha256": {k: hashlib.sha256(v).hexdigest() for k, v in views.items()}}


if __name__ == "__main__":
    out = {}
    for name in S.SCENARIOS:
        out[name] = fixture(name)
        print(name, out[name]["n_aux"], out[name]["n_constraints"], flush=True)
    json.dump(out, open(os.path.join(HERE, "r1cs_sha256.json"), "w"), indent=1, sort_keys=True)
