"""HS15 data exactly as the reference's `test_kkt_system` sees it
(`lib/MadNLPTests/src/Instances/hs15.jl:48-103`, `lib/MadNLPTests/src/MadNLPTests.jl:53-110`)
-- TEST INFRASTRUCTURE ONLY.  0-based indices."""
from __future__ import annotations

import numpy as np

N, M = 2, 2
# Both constraints are inequalities (lcon=[1,0], ucon=[Inf,Inf]); stacked (x, s) bounds:
# x1 <= 0.5 (ub), s1 >= 1, s2 >= 0 (lb).   ind_lb=[3,4], ind_ub=[1] 1-based.
IND_INEQ = np.array([0, 1])
IND_EQ = np.array([], dtype=np.int64)
IND_LB = np.array([2, 3])
IND_UB = np.array([0])
JAC_I = np.array([0, 0, 1, 1])
JAC_J = np.array([0, 1, 0, 1])
HESS_I = np.array([0, 1, 1])
HESS_J = np.array([0, 0, 1])


def jac_coord(x):
    return np.array([x[1], x[0], 1.0, 2 * x[1]])


def jac_dense(x):
    return np.array([[x[1], x[0]], [1.0, 2 * x[1]]], order="F")


def hess_coord(x, y, obj_weight=1.0):
    H = np.array([obj_weight * (-400.0 * x[1] + 1200.0 * x[0] ** 2 + 2.0),
                  obj_weight * (-400.0 * x[0]),
                  obj_weight * 200.0])
    H[1] += y[0] * 1.0
    H[2] += y[1] * 2.0
    return H


def hess_dense(x, y, obj_weight=1.0):
    h = hess_coord(x, y, obj_weight)
    return np.array([[h[0], h[1]], [h[1], h[2]]], order="F")
