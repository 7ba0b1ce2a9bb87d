"""CPU side of the C4 (case9241pegase shape, N = 85 568) comparison: the 58.6 GB dense factor and ~209 TFLOP per
factorization do not fit a bounded CPU run, so LAPACK dsytrf/dpotrf (scipy/OpenBLAS, the routines MadNLP's
LapackCPUSolver calls) are timed at N = 11 192 and N = 22 384 on this host and extrapolated with N^3 at the
rate of the LARGEST measured size (labelled as an extrapolation).  usage: python tools/cpu_c4_table.py out.json"""
import json
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "16")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys
import time

import numpy as np
from scipy.linalg import lapack
from threadpoolctl import threadpool_limits

try:
    import psutil
    PHYS = psutil.cpu_count(logical=False) or os.cpu_count()
except Exception:
    PHYS = os.cpu_count()

N4 = 85568


def spd(n, seed=0):
    rng = np.random.default_rng(seed)
    R = rng.standard_normal((n, 64))
    A = R @ R.T
    A[np.diag_indices(n)] += n
    return np.asfortranarray(A)


def time_one(A, routine, threads):
    with threadpool_limits(limits=threads, user_api="blas"):
        B = A.copy(order="F")
        t0 = time.perf_counter()
        if routine == "dsytrf":
            lw = int(lapack.dsytrf_lwork(A.shape[0], lower=1)[0])
            lapack.dsytrf(B, lower=1, lwork=lw, overwrite_a=1)
        else:
            lapack.dpotrf(B, lower=1, clean=0, overwrite_a=1)
        return time.perf_counter() - t0


def main():
    out = {"host_physical_cores": PHYS, "host_logical_cpus": os.cpu_count(), "rows": [], "N_target": N4}
    few = max(2, min(16, PHYS // 2))
    A1 = spd(11192)
    with threadpool_limits(limits=few, user_api="blas"):
        lapack.dpotrf(spd(2048), lower=1)  # spin the pool up
    plan1 = [("dsytrf", 1), ("dsytrf", few), ("dpotrf", 1), ("dpotrf", few), ("dpotrf", PHYS)]
    for routine, thr in plan1:
        t = time_one(A1, routine, thr)
        out["rows"].append({"N": 11192, "routine": routine, "threads": thr, "seconds": t, "gflops": 11192 ** 3 / 3 / t / 1e9})
    del A1
    A2 = spd(22384)
    for routine, thr in [("dsytrf", few), ("dpotrf", few), ("dpotrf", PHYS)]:
        t = time_one(A2, routine, thr)
        out["rows"].append({"N": 22384, "routine": routine, "threads": thr, "seconds": t, "gflops": 22384 ** 3 / 3 / t / 1e9})
    # extrapolation at the rate of the largest size measured per (routine, threads)
    ext = []
    for routine in ("dsytrf", "dpotrf"):
        for thr in sorted({r["threads"] for r in out["rows"] if r["routine"] == routine}):
            rs = [r for r in out["rows"] if r["routine"] == routine and r["threads"] == thr]
            big = max(rs, key=lambda r: r["N"])
            ext.append({"routine": routine, "threads": thr, "from_N": big["N"], "gflops_assumed": big["gflops"],
                        "seconds_extrapolated_N85568": N4 ** 3 / 3 / (big["gflops"] * 1e9), "label": "N^3 extrapolation"})
    out["extrapolation"] = ext
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
