"""Pivoted tier, multi-workgroup panels against one-workgroup panels on the same matrix: where the permutation, the
1x1 / 2x2 pattern and D first differ, and both reconstruction errors.  usage: python tools/bk_compare.py [N] [random|saddle]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import madnlp_jl_amd as mj  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 700
kind = sys.argv[2] if len(sys.argv) > 2 else "random"
rng = np.random.default_rng(N)
if kind == "random":
    S = rng.standard_normal((N, N)); A = (S + S.T) / 2
else:
    n1 = 2 * N // 3
    H = rng.standard_normal((n1, n1)); H = (H + H.T) / 2
    J = rng.standard_normal((N - n1, n1))
    A = np.zeros((N, N)); A[:n1, :n1] = H; A[n1:, :n1] = J; A[:n1, n1:] = J.T
A = np.asfortranarray(A)
ctx = mj.HipContext(0)
out = []
for wgs in (0, 1):
    M = mj.HipLinearSolver(A, ctx=ctx, opt=mj.HipSolverOptions(lapack_algorithm=mj.BUNCHKAUFMAN))
    M.set_option("bk_panel_wgs", wgs)
    M.factorize()
    inertia = M.inertia()
    active, count, perm, doff = M.bk_info()
    Lg, D = M.get_factor()
    Lu = np.tril(Lg, -1) + np.eye(N)
    Dm = np.diag(D)
    for k in np.nonzero(doff)[0]:
        Dm[k + 1, k] = Dm[k, k + 1] = doff[k]
    Af = np.tril(A) + np.tril(A, -1).T
    R = Lu @ Dm @ Lu.T - Af[np.ix_(perm, perm)]
    bad_rows = np.nonzero(np.abs(R).max(axis=1) > 1e-9)[0]
    print(f"bk_panel_wgs={wgs}: multi {M.get_stat('bk_panel_multi')} fallbacks {M.get_stat('bk_mw_fallbacks')} inertia {inertia} "
          f"reconstruction error {np.abs(R).max():.2e}; rows with error: {len(bad_rows)} first {bad_rows[:8]}")
    out.append((perm.copy(), doff.copy(), D.copy(), Lu))
    M.close()
(p0, o0, D0, L0), (p1, o1, D1, L1) = out
dp = np.nonzero(p0 != p1)[0]
do = np.nonzero((o0 != 0) != (o1 != 0))[0]
dd = np.nonzero(np.abs(D0 - D1) > 1e-9 * np.abs(D1).max())[0]
dl = np.nonzero(np.abs(L0 - L1).max(axis=0) > 1e-9)[0]
print("first differing perm index", dp[:5], "2x2 pattern", do[:5], "D", dd[:5], "L column", dl[:5])
if len(dp):
    k = dp[0]
    print("around it: perm multi", p0[max(0, k - 2):k + 4], "single", p1[max(0, k - 2):k + 4], "D multi", D0[max(0, k - 2):k + 4], "single", D1[max(0, k - 2):k + 4])
