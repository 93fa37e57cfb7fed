"""Generates tests/golden/r1cs_sha256.json: per scenario (tests/r1cs_scenarios.py) the variable / constraint counts and the
sha256 of each of the 15 arrays of a synthesized instance (z, A.z, B.z, C.z, densities, CSR val / col / row_ptr of A, B, C)
AS THE INDEPENDENT PYTHON RESTATEMENT of the reference's circuits computes them (oracle/pycircuit.py) from the work's
bincode bytes parsed by tests/bincode_ref.py.  The product never feeds these hashes: it only supplies the scenario's
`MpnWork` bytes (whose sha256 is recorded too, so a drifting witness builder is noticed).

    python tests/golden/make_r1cs_fixtures.py                    # the small scenarios, ~1 min (903 037 constraints the largest)
    python tests/golden/make_r1cs_fixtures.py deposit_15_3_3 ...  # named scenarios only, merged into the existing file

The production shapes (r1cs_scenarios.PRODUCTION: 2^21 / 2^22 / 2^24 domains, up to 14.4 M constraints) do not fit in
memory as Python lists; they go through pycircuit.StreamingConstraintSystem, which folds every constraint into the running
sha256 states as it is enforced (tests/test_pycircuit_cpu.py checks the streaming and the stored form agree on the small
scenarios).  One-off cost: ~2 / ~3 / ~20 minutes of pure Python.
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


def fixture(name, streaming=None):
    blob = S.make_work(name)
    work = B.decode(B.MpnWork, blob)
    com = B.encode(B.Address, S.PROVER) + B.encode(B.U64, work["reward"])
    commitment = int.from_bytes(hashlib.sha3_256(com).digest(), "little") % pc.R_MOD  # src/mpn/mod.rs:283-285
    if streaming if streaming is not None else name in S.PRODUCTION:
        cs = pc.StreamingConstraintSystem(6)
        pc.circuit_of_work(work, commitment, 1, lambda p: B.encode(B.ContractWithdraw, p), cs=cs)
        n_in, n_aux, n_rows, unsat, sha = cs.finish()
        assert unsat == -1, unsat
        return {"work_sha256": hashlib.sha256(blob).hexdigest(), "n_in": n_in, "n_aux": n_aux, "n_constraints": n_rows,
                "n_enabled": len(work["data"][1]), "sha256": sha}
    cs = pc.circuit_of_work(work, commitment, 1, lambda p: B.encode(B.ContractWithdraw, p))
    assert pc.first_unsatisfied(cs) == -1
    views = pc.all_views(cs)
    return {"work_sha256": hashlib.sha256(blob).hexdigest(), "n_in": cs.n_in, "n_aux": cs.n_aux,
            "n_constraints": len(cs.A) + cs.n_in, "sha256": {k: hashlib.sha256(v).hexdigest() for k, v in views.items()}}


if __name__ == "__main__":
    path = os.path.join(HERE, "r1cs_sha256.json")
    names = sys.argv[1:] or [n for n in S.SCENARIOS if n not in S.PRODUCTION]
    out = json.load(open(path)) if sys.argv[1:] and os.path.exists(path) else {}
    for name in names:
        out[name] = fixture(name)
        print(name, out[name]["n_aux"], out[name]["n_constraints"], flush=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
