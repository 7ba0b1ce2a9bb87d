"""Oracle restatement of the regular-phase IPM reductions of reference `src/IPM/kernels.jl:263-388,675-695`
(TEST INFRASTRUCTURE ONLY).  Loops are restated as numpy reductions; the `_r` arguments are the gathered views the
reference passes (`x_lr = x[ind_lb]` ...)."""
from __future__ import annotations

import numpy as np

INF = float("inf")


def get_varphi(obj_val, x_lr, xl_r, xu_r, x_ur, mu):
    """`get_varphi` / `_get_varphi` `kernels.jl:263-283`."""
    dl, du = x_lr - xl_r, xu_r - x_ur
    if (dl < 0).any() or (du < 0).any():
        return INF
    with np.errstate(divide="ignore"):
        return obj_val + (-mu * np.log(dl)).sum() + (-mu * np.log(du)).sum()


def get_inf_du(f, zl, zu, jacl, sd):
    """`kernels.jl:285-291`."""
    return np.abs(f - zl + zu + jacl).max(initial=0.0) / sd


def get_inf_compl(x_lr, xl_r, zl_r, xu_r, x_ur, zu_r, mu, sc):
    """`kernels.jl:293-303`."""
    a = np.abs((x_lr - xl_r) * zl_r - mu).max(initial=0.0)
    b = np.abs((xu_r - x_ur) * zu_r - mu).max(initial=0.0)
    return max(a, b) / sc


def get_average_complementarity(x_lr, xl_r, zl_r, x_ur, xu_r, zu_r):
    """`kernels.jl:305-313`."""
    n = len(x_lr) + len(x_ur)
    if n == 0:
        return 0.0
    cc_lb = np.dot(x_lr, zl_r) - np.dot(xl_r, zl_r)
    cc_ub = np.dot(xu_r, zu_r) - np.dot(x_ur, zu_r)
    return (cc_lb + cc_ub) / n


def get_min_complementarity(x_lr, xl_r, zl_r, x_ur, xu_r, zu_r):
    """`kernels.jl:322-332`."""
    a = ((x_lr - xl_r) * zl_r).min(initial=INF)
    b = ((xu_r - x_ur) * zu_r).min(initial=INF)
    return min(a, b)


def get_varphi_d(f, x, xl, xu, dx, mu):
    """`kernels.jl:341-354`."""
    return float(((f - mu / (x - xl) + mu / (xu - x)) * dx).sum())


def get_alpha_max(x, xl, xu, dx, tau):
    """`kernels.jl:356-371`."""
    a = 1.0
    neg, pos = dx < 0, dx > 0
    with np.errstate(over="ignore"):  # unbounded sides: (-x + xu) * tau / dx overflows to +Inf, as in the reference
        if neg.any():
            a = min(a, ((-x[neg] + xl[neg]) * tau / dx[neg]).min())
        if pos.any():
            a = min(a, ((-x[pos] + xu[pos]) * tau / dx[pos]).min())
    return a


def get_alpha_z(zl_r, zu_r, dzl, dzu, tau):
    """`kernels.jl:373-388`."""
    a = 1.0
    nl, nu = dzl < 0, dzu < 0
    if nl.any():
        a = min(a, ((-zl_r[nl]) * tau / dzl[nl]).min())
    if nu.any():
        a = min(a, ((-zu_r[nu]) * tau / dzu[nu]).min())
    return a


def get_rel_search_norm(x, dx):
    """`kernels.jl:675-682`."""
    return (np.abs(dx) / (1.0 + np.abs(x))).max(initial=0.0)


def get_sd(l, zl_r, zu_r, s_max):
    """`kernels.jl:684-689`."""
    return max(s_max, (np.abs(l).sum() + np.abs(zl_r).sum() + np.abs(zu_r).sum()) / max(1, len(l) + len(zl_r) + len(zu_r))) / s_max


def get_sc(zl_r, zu_r, s_max):
    """`kernels.jl:690-695`."""
    return max(s_max, (np.abs(zl_r).sum() + np.abs(zu_r).sum()) / max(1, len(zl_r) + len(zu_r))) / s_max


# ---- elementwise pieces of the regular phase ------------------------------------------------------------------------
def set_aug_rhs(f, zl, zu, jacl, c, x_lr, xl_r, zl_r, xu_r, x_ur, zu_r, mu):
    """`set_aug_rhs!` `kernels.jl:113-131`: returns (px, py, pzl, pzu)."""
    return -f + zl - zu - jacl, -c, (xl_r - x_lr) * zl_r + mu, (xu_r - x_ur) * zu_r - mu


def dual_inf_perturbation(px, ind_llb, ind_uub, mu, kappa_d):
    """`kernels.jl:818-823` (in place)."""
    px[ind_llb] -= mu * kappa_d
    px[ind_uub] += mu * kappa_d


def adjust_boundary(x, xl, xu, ind_lb, ind_ub, mu):
    """`adjust_boundary!` `kernels.jl:656-673` on full-length xl / xu (in place)."""
    eps = np.finfo(np.float64).eps
    c1, c2 = eps * mu, eps ** 0.75
    x_lr, xl_r = x[ind_lb], xl[ind_lb]
    xl[ind_lb] = np.where(x_lr - xl_r < c1, xl_r - c2 * np.maximum(1.0, np.abs(x_lr)), xl_r)
    x_ur, xu_r = x[ind_ub], xu[ind_ub]
    xu[ind_ub] = np.where(xu_r - x_ur < c1, xu_r + c2 * np.maximum(1.0, np.abs(x_ur)), xu_r)


def reset_bound_dual(z, x1, x2, mu, kappa_sigma):
    """`reset_bound_dual!(z, x1, x2, mu, kappa_sigma)` `kernels.jl:788-801` (in place, full-length)."""
    with np.errstate(divide="ignore"):
        d = x1 - x2
        z[:] = np.maximum(np.minimum(z, (kappa_sigma * mu) / d), (mu / kappa_sigma) / d)
