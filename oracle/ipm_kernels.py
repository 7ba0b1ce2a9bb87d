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


# ---- restoration phase (robust restorer) -----------------------------------------------------------------------------
# reference `src/IPM/kernels.jl:390-636` (GPU twins `lib/MadNLPGPU/src/IPM/kernels.jl:117-462`); pp / nn / zp / zn are the
# m-vectors of the RobustRestorer (`src/IPM/restoration.jl:1-37`), x / xl / xu / D_R / x_ref / f_R full primal length.
def _barrier_terms(d, mu):
    """`d < 0 ? Inf : mu * log(d)` elementwise (`kernels.jl:555-568`)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(d < 0, INF, mu * np.log(np.where(d < 0, 1.0, d)))


def get_obj_val_R(p, n, D_R, x, x_ref, rho, zeta):
    """`kernels.jl:390-407`."""
    return float((rho * (p + n)).sum() + (zeta / 2 * D_R ** 2 * (x - x_ref) ** 2).sum())


def get_theta_R(c, p, n):
    """`kernels.jl:411-421`."""
    return float(np.abs(c - p + n).sum())


def get_inf_pr_R(c, p, n):
    """`kernels.jl:423-433`."""
    return float(np.abs(c - p + n).max(initial=0.0))


def get_inf_du_R(f_R, l, zl, zu, jacl, zp, zn, rho, sd):
    """`kernels.jl:435-454`."""
    a = np.abs(f_R - zl + zu + jacl).max(initial=0.0)
    b = np.abs(rho - l - zp).max(initial=0.0)
    c = np.abs(rho + l - zn).max(initial=0.0)
    return float(max(a, b, c) / sd)


def get_inf_compl_R(x_lr, xl_r, zl_r, xu_r, x_ur, zu_r, pp, zp, nn, zn, mu_R, sc):
    """`kernels.jl:456-484`."""
    v = [np.abs((x_lr - xl_r) * zl_r - mu_R).max(initial=0.0), np.abs((xu_r - x_ur) * zu_r - mu_R).max(initial=0.0),
         np.abs(pp * zp - mu_R).max(initial=0.0), np.abs(nn * zn - mu_R).max(initial=0.0)]
    return float(max(v) / sc)


def get_alpha_max_R(x, xl, xu, dx, pp, dpp, nn, dnn, tau_R):
    """`kernels.jl:486-515`."""
    a = get_alpha_max(x, xl, xu, dx, tau_R)
    for v, dv in ((pp, dpp), (nn, dnn)):
        neg = dv < 0
        if neg.any():
            a = min(a, (-v[neg] * tau_R / dv[neg]).min())
    return float(a)


def get_alpha_z_R(zl_r, zu_r, dzl, dzu, zp, dzp, zn, dzn, tau_R):
    """`kernels.jl:517-542`."""
    a = 1.0
    for z, dz in ((zl_r, dzl), (zu_r, dzu), (zp, dzp), (zn, dzn)):
        neg = dz < 0
        if neg.any():
            a = min(a, (-z[neg] * tau_R / dz[neg]).min())
    return float(a)


def get_varphi_R(obj_val, x_lr, xl_r, xu_r, x_ur, pp, nn, mu_R):
    """`kernels.jl:544-570`."""
    return float(obj_val - (_barrier_terms(x_lr - xl_r, mu_R).sum() + _barrier_terms(xu_r - x_ur, mu_R).sum()
                            + _barrier_terms(pp, mu_R).sum() + _barrier_terms(nn, mu_R).sum()))


def get_F(c, f, zl, zu, jacl, x_lr, xl_r, zl_r, xu_r, x_ur, zu_r, mu):
    """`kernels.jl:572-610`.  The upper-bound term is restated as the reference computes it: `(xu_r - xu_r) * zu_r - mu`
    (`:606`, the same expression in the GPU twin `:407-410`), i.e. |0 * zu_r - mu| when the guard holds."""
    F1 = np.abs(c).sum()
    F2 = np.abs(f - zl + zu + jacl).sum()
    F3 = np.where((x_lr >= xl_r) & (zl_r >= 0), np.abs((x_lr - xl_r) * zl_r - mu), INF).sum()
    with np.errstate(invalid="ignore"):
        F4 = np.where((xu_r >= x_ur) & (zu_r >= 0), np.abs((xu_r - xu_r) * zu_r - mu), INF).sum()
    return float(F1 + F2 + F3 + F4)


def get_varphi_d_R(f_R, x, xl, xu, dx, pp, nn, dpp, dnn, mu_R, rho):
    """`kernels.jl:612-636`."""
    return float(((f_R - mu_R / (x - xl) + mu_R / (xu - x)) * dx).sum() + ((rho - mu_R / pp) * dpp).sum()
                 + ((rho - mu_R / nn) * dnn).sum())


def populate_RR_nn(c, mu, rho):
    """`populate_RR_nn!` `kernels.jl:825-829`: returns nn."""
    t = (mu - rho * c) / (2 * rho)
    return t + np.sqrt(t ** 2 + mu * c / (2 * rho))


def initialize_robust_restorer(x, c, zl_r, zu_r, mu, rho):
    """The vector part of `initialize_robust_restorer!` `src/IPM/restoration.jl:39-76`: returns
    (x_ref, D_R, mu_R, nn, pp, zp, zn, zl_r, zu_r)."""
    x_ref = x.copy()
    with np.errstate(divide="ignore"):
        D_R = np.minimum(1.0, 1.0 / np.abs(x_ref))
    mu_R = max(mu, np.abs(c).max(initial=0.0))
    nn = populate_RR_nn(c, mu_R, rho)
    pp = c + nn
    return x_ref, D_R, mu_R, nn, pp, mu_R / pp, mu_R / nn, np.minimum(rho, zl_r), np.minimum(rho, zu_r)


def set_f_RR(D_R, x, x_ref, zeta):
    """`set_f_RR!` `kernels.jl:106-110`: returns f_R."""
    return zeta * D_R ** 2 * (x - x_ref)


def set_aug_RR(x_lr, xl_r, zl_r, xu_r, x_ur, zu_r, D_R, pp, zp, nn, zn, zeta, primal_reg, dual_reg, ind_lb, ind_ub):
    """`set_aug_RR!` `kernels.jl:72-87` followed by `_set_aug_diagonal!` `:22-27`: returns
    (reg, du_diag, l_diag, u_diag, l_lower, u_lower, pr_diag)."""
    reg = primal_reg + zeta * D_R ** 2
    du_diag = -dual_reg - pp / zp - nn / zn
    l_diag, u_diag = xl_r - x_lr, x_ur - xu_r
    pr_diag = reg.copy()
    np.subtract.at(pr_diag, ind_lb, zl_r / l_diag)
    np.subtract.at(pr_diag, ind_ub, zu_r / u_diag)
    return reg, du_diag, l_diag, u_diag, zl_r.copy(), zu_r.copy(), pr_diag


def set_aug_rhs_RR(f_R, zl, zu, jacl, c, y, pp, nn, zp, zn, x_lr, xl_r, zl_r, xu_r, x_ur, zu_r, mu_R, rho):
    """`set_aug_rhs_RR!` `kernels.jl:133-158`: returns (px, py, pzl, pzu)."""
    px = -f_R + zl - zu - jacl
    py = -c + pp - nn + (mu_R - (rho - y) * pp) / zp - (mu_R - (rho + y) * nn) / zn
    return px, py, (xl_r - x_lr) * zl_r + mu_R, (xu_r - x_ur) * zu_r - mu_R


def finish_aug_solve_RR(l, dl, pp, nn, zp, zn, mu_R, rho):
    """`finish_aug_solve_RR!` `kernels.jl:251-257`: returns (dpp, dnn, dzp, dzn)."""
    dzp = rho - l - dl - zp
    dzn = rho + l + dl - zn
    dpp = -pp + mu_R / zp - (pp / zp) * dzp
    dnn = -nn + mu_R / zn - (nn / zn) * dzn
    return dpp, dnn, dzp, dzn


def reset_bound_dual_1(z, x, mu, kappa_sigma):
    """`reset_bound_dual!(z, x, mu, kappa_sigma)` `kernels.jl:775-786` (in place)."""
    with np.errstate(divide="ignore"):
        z[:] = np.maximum(np.minimum(z, (kappa_sigma * mu) / x), (mu / kappa_sigma) / x)


def set_initial_bounds(xl, xu, tol):
    """`set_initial_bounds!` `kernels.jl:206-218` (in place)."""
    if tol > 0:
        xl[:] = xl - np.maximum(1.0, np.abs(xl)) * tol
        xu[:] = xu + np.maximum(1.0, np.abs(xu)) * tol


def set_initial_rhs(f, zl, zu):
    """`set_initial_rhs!` `kernels.jl:220-230`: px (the other blocks are zero)."""
    return -f + zl - zu


def set_g_ifr(f, x, xl, xu, jacl, mu):
    """`set_g_ifr!` `kernels.jl:242-248`."""
    with np.errstate(divide="ignore"):
        return f - mu / (x - xl) + mu / (xu - x) + jacl


def initialize_variables(x, xl, xu, bound_push, bound_fac):
    """`initialize_variables!` / `_initialize_variables!` `kernels.jl:638-654`: returns the pushed x."""
    out = x.copy()
    for i in range(len(x)):
        l, u, v = xl[i], xu[i], x[i]
        if l != -INF and u != INF:
            out[i] = min(u - min(bound_push * max(1.0, abs(u)), bound_fac * (u - l)),
                         max(l + min(bound_push * max(1.0, abs(l)), bound_fac * (u - l)), v))
        elif l != -INF and u == INF:
            out[i] = max(l + bound_push * max(1.0, abs(l)), v)
        elif l == -INF and u != INF:
            out[i] = min(u - bound_push * max(1.0, abs(u)), v)
    return out
