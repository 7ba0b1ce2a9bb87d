"""Regenerates tests/golden/hs15_kkt.json.

The reference cannot be executed in the build container (no Julia), so this
fixture is NOT a capture of reference output: it is the solution of the HS15
`test_kkt_system` protocol (reference lib/MadNLPTests/src/MadNLPTests.jl:53-110)
computed by the oracle restatement, accepted only because it satisfies the
reference's own acceptance identity mul!(y, kkt, solve_kkt!(kkt, ones)) == ones
to 1e-13 for all three KKT formulations, and because pr_diag / condensed K equal
the closed-form values derived by hand from the reference source in SURVEY.md 8(c)2:
pr_diag = [0.999, 1, 0.999, 0.999], K = diag(3.998, 201).
Run from the repo root:  python tests/golden/make_hs15_golden.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_oracle_golden import _make, run_test_kkt_system  # noqa: E402

if __name__ == "__main__":
    sols = [run_test_kkt_system(_make(k), k == "sparse_condensed")
            for k in ("sparse_condensed", "dense_condensed", "dense")]
    out = {
        "_source": "see tests/golden/make_hs15_golden.py",
        "pr_diag": [0.999, 1.0, 0.999, 0.999],
        "K_condensed_diag": [3.998, 201.0],
        # NOT reference output: computed by the oracle restatement (see the header); pins HIP == oracle
        "oracle_solve_kkt_ones": [float(v) for v in sols[1]],
    }
    json.dump(out, open(os.path.join(os.path.dirname(__file__), "hs15_kkt.json"), "w"), indent=1)
