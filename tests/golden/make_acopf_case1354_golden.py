"""Regenerates tests/golden/acopf_case1354_oracle.json: the polar AC-OPF NLP on the case1354pegase-sized synthetic grid
(n = 11 192, m = 16 646; `madnlp_jl_amd.problems.ACOPFModel`) solved by the host IPM mirror on the ORACLE back-end
(numpy assembly `oracle/sparse_condensed.py` + LAPACK Bunch-Kaufman `oracle/lapack_cpu.py`), with the presets of
SparseCondensedKKTSystem (reference src/IPM/options.jl:146-147,160,226) and tol = 1e-6.

NOT reference output (no Julia in the build container): it is the restatement's own run, the checker of the bench line's
`end_to_end_ipm` record and of `tests/test_acopf.py::test_device_resident_case1354_run_matches_the_oracle_golden`.
Acceptance level = the reference's CPU == GPU comparison (lib/MadNLPGPU/test/densekkt_rocm.jl:31-37: same status, objective
and solution to a loose tolerance).

Run from the repo root (several minutes of host time: ~30 dsytrf at N = 11 192):
    python tests/golden/make_acopf_case1354_golden.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main(case="case1354pegase", out_name="acopf_case1354_oracle.json"):
    from madnlp_jl_amd.ipm import IPMOptions, MadNLPSolver
    from madnlp_jl_amd.problems import ACOPFModel
    from tests.test_ipm_oracle import oracle_factory
    nlp = ACOPFModel(case)
    o = IPMOptions(tol=1e-6)
    o.relax_equality, o.dual_initialization = True, "zero"
    s = MadNLPSolver(nlp, oracle_factory("sparse_condensed", nlp), o, sparse=True)
    t0 = time.perf_counter()
    status = s.solve()
    wall = time.perf_counter() - t0
    x = s.x[:nlp.n]
    c = nlp.cons(x)
    pg = nlp.S["pg"]
    last = s.history[-1]
    out = {
        "_source": "tests/golden/make_acopf_case1354_golden.py (oracle back-end: numpy assembly + scipy/OpenBLAS dsytrf; NOT reference output)",
        "case": case, "n": int(nlp.n), "m": int(nlp.m), "tol": 1e-6,
        "status": status, "iterations": int(s.cnt.k), "factorizations": int(s.cnt.factorization_cnt),
        "backsolves": int(s.cnt.backsolve_cnt), "objective": float(s.obj_val),
        "inf_pr": float(last.inf_pr), "inf_du": float(last.inf_du), "inf_compl": float(last.inf_compl),
        "max_constraint_violation": float(max(0.0, (nlp.lcon - c).max(), (c - nlp.ucon).max())),
        "pg": [float(v) for v in s.x[pg]],
        "history": [{"k": int(h.k), "inf_pr": float(h.inf_pr), "inf_du": float(h.inf_du), "inf_compl": float(h.inf_compl),
                     "mu": float(h.mu), "del_w": float(h.del_w)} for h in s.history],
        "host_wall_s": wall, "host_cpus": os.cpu_count(),
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), out_name)
    json.dump(out, open(path, "w"), indent=1)
    print(f"{path}: {status}, {out['iterations']} it / {out['factorizations']} fact / {out['backsolves']} solves, "
          f"obj {out['objective']:.10g}, {wall:.0f} s")


if __name__ == "__main__":
    main(*sys.argv[1:3])
