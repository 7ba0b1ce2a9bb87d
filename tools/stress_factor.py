"""Stress the factorization paths against LAPACK on ill-conditioned matrices of many orders (round 2: added after a
compiler problem made the persistent panel kernel return wrong pivots for ONE matrix in the whole suite).
usage: python tools/stress_factor.py [count] [--fuse ROWS]"""
import os
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "8")  # the pool's boxes cap the process at 16 CPUs: idle BLAS pools must not spin on 256
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402


def spd(rng, N, decades):
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    w = 10.0 ** rng.uniform(-decades / 2, decades / 2, N)
    A = (Q * w) @ Q.T
    return np.asfortranarray((A + A.T) / 2)


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
    fuse = int(sys.argv[sys.argv.index("--fuse") + 1]) if "--fuse" in sys.argv else None
    ctx = mj.HipContext(0)
    rng = np.random.default_rng(2026)
    bad = 0
    for it in range(count):
        N = int(rng.choice([65, 129, 200, 257, 400, 577, 640, 900, 1100, 1500, 2100, 3000, 5000]))
        if it >= count - 2:
            N = 6000
        alg = [mj.CHOLESKY, mj.LDL][it % 2]
        A = spd(rng, N, 6)
        if alg == mj.LDL and it % 4 == 1:  # quasi-definite saddle: factorable without pivoting
            n1 = N // 2
            A[n1:, n1:] *= -1.0
        ob = int(rng.choice([256, 512]))
        M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=alg, outer_block=ob))
        if N > 2560 or it % 3 == 0:
            M.set_option("single_rows", 0)  # exercise the look-ahead schedule on small systems too
        if fuse is not None:
            M.set_option("pp_fuse_rows", fuse)
        M.factorize()
        b = rng.standard_normal(N)
        x = M.solve_linear_system(b.copy())
        res = np.abs(A @ x - b).max() / (np.abs(A).max() * np.abs(x).max() + np.abs(b).max())
        xr = sla.solve(A, b, assume_a="sym")
        res_ref = np.abs(A @ xr - b).max() / (np.abs(A).max() * np.abs(xr).max() + np.abs(b).max())
        ok = M.info == 0 and res <= 1e-12 and res <= 100 * res_ref + 1e-15
        exp_in = (N, 0, 0) if not (alg == mj.LDL and it % 4 == 1) else (N // 2, 0, N - N // 2)
        ok = ok and M.inertia() == exp_in
        bad += not ok
        print(f"{it:3d} N={N:5d} {alg:9s} ob={ob} algo={M.get_stat('panel_algo'):.0f} info={M.info} res={res:.1e} ref={res_ref:.1e} "
              f"inertia={M.inertia()} {'ok' if ok else 'FAIL'}", flush=True)
        M.close()
    print("FAILURES:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
