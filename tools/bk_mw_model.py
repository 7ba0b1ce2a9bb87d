"""Host model of the multi-workgroup Bunch-Kaufman panel (csrc/bk.hip, bkp_panel_mw_kernel): the same data flow, one
Python object per workgroup that only sees its own rows, the replicated panel table and the hop messages.

The panel works on VIRTUAL positions: a workgroup owns the rows [p0 + 256 g, p0 + 256 (g + 1)) of the matrix as it is
stored when the panel starts and nothing moves while the panel is factored -- an interchange only changes `pos[row]` and
the replicated table `prow[i]` (row at panel position p0 + i).  Per column there is one all-to-all hop (message A:
local column maximum, the pivot row's diagonal entry, the working column at the rows of the next two panel positions)
and, when the 1x1 test fails, a second one (message B: local maximum of the partner column, its entries at the pivot
row, the partner row and the next two panel positions).  The entries W[q, c] of the next pivot row that were produced
after the last hop travel in those messages; the older ones are read from the global W (written a hop earlier).  At the
panel's end the rows are written out in position order and `rowof[pos]` tells the permutation kernels what moved.

The pivoting rule is dsytf2's (alpha = (1 + sqrt(17)) / 8): this model is checked against a plain implementation with
physical interchanges (tests/test_bk_multi_cpu.py) -- same pivots, same factors.
"""
import numpy as np

ALPHA = (1.0 + np.sqrt(17.0)) / 8.0
NB = 64
T = 256          # rows per workgroup


def a0(F, r, q):
    """entry (r, q) of the symmetric matrix whose lower triangle is stored in F"""
    return F[r, q] if r >= q else F[q, r]


class WG:
    def __init__(self, g, p0, Np, F):
        self.g, self.p0, self.Np, self.F = g, p0, Np, F
        self.rows = np.arange(p0 + T * g, min(Np, p0 + T * (g + 1)))
        self.pos = self.rows.copy()                      # position of every owned row
        self.active = np.ones(len(self.rows), bool)      # not eliminated
        self.LW = np.zeros((len(self.rows), NB + 1))     # L entries of the panel's columns (LDS in the kernel)
        self.Wown = np.zeros((len(self.rows), NB + 1))   # W entries of the owned rows (= what it stores to the global W)
        self.prow = np.arange(p0, p0 + NB + 2)           # replicated: row at panel position p0 + i
        self.wq_cache = {}                               # rows of W kept in LDS: {row: vector}

    def owns(self, r):
        return self.rows[0] <= r <= self.rows[-1] if len(self.rows) else False

    def li(self, r):
        return r - self.rows[0]


def panel(F, p0, Np, state):
    """factor one panel of the matrix stored in F (lower triangle, position order); returns what the kernel leaves in
    global memory: kb, LWp / Wp (position order), rowof, the D entries, and the number of hops"""
    nrow = Np - p0
    G = (nrow + T - 1) // T
    wgs = [WG(g, p0, Np, F) for g in range(G)]
    Wglob = np.zeros((Np, NB + 1))                      # virtual order: W[row, c]
    owner = lambda r: wgs[(r - p0) // T]
    dvec, doff, ptype = state["dvec"], state["doff"], state["ptype"]
    kb, hops = 0, 0
    wq_new = {}                                          # (replicated) newest W entries of candidate rows, from the messages

    def wrow(wg, q, kb):
        """W[q, 0:kb] as workgroup wg sees it: global entries older than a hop + the newest ones from the messages"""
        v = Wglob[q, :kb].copy()
        for c, val in wq_new.get(q, {}).items():
            if c < kb:
                v[c] = val
        return v

    while kb < NB - 1 and p0 + kb < Np:
        k = p0 + kb
        q = int(wgs[0].prow[kb])
        n1 = int(wgs[0].prow[kb + 1]) if k + 1 < Np else -1
        n2 = int(wgs[0].prow[kb + 2]) if k + 2 < Np else -1
        # ---- phase A (every workgroup): the pivot column on the owned active rows, local maximum, message A
        msgA, wk_all = [], {}
        for wg in wgs:
            wq = wrow(wg, q, kb)
            wk = np.array([a0(F, r, q) for r in wg.rows]) - wg.LW[:, :kb] @ wq if len(wg.rows) else np.zeros(0)
            wk_all[wg.g] = wk
            best = (-1.0, 1 << 30, -1, 0.0)              # (|v|, pos, row, v): largest |v|, smallest position among ties
            for i, r in enumerate(wg.rows):
                if wg.active[i] and r != q:
                    a = abs(wk[i])
                    if not a <= np.finfo(float).max:
                        a = np.finfo(float).max
                    if a > best[0] or (a == best[0] and wg.pos[i] < best[1]):
                        best = (a, int(wg.pos[i]), int(r), wk[i])
            m = {"max": best, "akk": wk[wg.li(q)] if wg.owns(q) else None,
                 "n1": wk[wg.li(n1)] if n1 >= 0 and wg.owns(n1) else None,
                 "n2": wk[wg.li(n2)] if n2 >= 0 and wg.owns(n2) else None}
            msgA.append(m)
        hops += 1
        # (the owners' stores of the previous column's W entries were complete before they posted message A: visible now)
        for rows, c, vals in state.pop("visible", []):
            Wglob[rows, c] = vals
        # ---- every workgroup reads all messages (replicated decision)
        colmax, imax, rmax, p21 = -1.0, 1 << 30, -1, 0.0
        for m in msgA:
            a, ps, r, v = m["max"]
            if a > colmax or (a == colmax and ps < imax):
                colmax, imax, rmax, p21 = a, ps, r, v
        if colmax < 0.0:
            colmax = 0.0
        akk = msgA[owner(q).g]["akk"]
        wk_n1 = msgA[owner(n1).g]["n1"] if n1 >= 0 else 0.0
        wk_n2 = msgA[owner(n2).g]["n2"] if n2 >= 0 else 0.0
        absakk = abs(akk)
        kstep, swap_to, zero, use_wn = 1, -1, False, False
        fmax = np.finfo(float).max
        wn_all, msgB = {}, None
        if not (max(absakk, colmax) > 0.0) or not (absakk <= fmax) or colmax >= fmax:
            zero = True
        elif absakk < ALPHA * colmax:
            # ---- phase B: the partner column (row rmax), message B
            msgB = []
            for wg in wgs:
                wq = wrow(wg, rmax, kb)
                wn = np.array([a0(F, r, rmax) for r in wg.rows]) - wg.LW[:, :kb] @ wq if len(wg.rows) else np.zeros(0)
                wn_all[wg.g] = wn
                lm = -1.0
                for i, r in enumerate(wg.rows):
                    if wg.active[i] and r != rmax:
                        lm = max(lm, abs(wn[i]))
                msgB.append({"rowmax": lm, "imax": wn[wg.li(rmax)] if wg.owns(rmax) else None,
                             "q": wn[wg.li(q)] if wg.owns(q) else None,
                             "n1": wn[wg.li(n1)] if n1 >= 0 and wg.owns(n1) else None,
                             "n2": wn[wg.li(n2)] if n2 >= 0 and wg.owns(n2) else None})
            hops += 1
            rowmax = max(m["rowmax"] for m in msgB)
            p22 = msgB[owner(rmax).g]["imax"]
            if absakk >= ALPHA * colmax * (colmax / rowmax):
                pass
            elif abs(p22) >= ALPHA * rowmax:
                swap_to, use_wn = k, True                # 1x1 pivot on the partner's diagonal entry
            else:
                swap_to, kstep = k + 1, 2
        wn_q = msgB[owner(q).g]["q"] if msgB else 0.0
        wn_n1 = msgB[owner(n1).g]["n1"] if msgB and n1 >= 0 else 0.0
        wn_n2 = msgB[owner(n2).g]["n2"] if msgB and n2 >= 0 else 0.0
        # ---- interchange of positions swap_to <-> imax (replicated table; the owners update pos[])
        if swap_to >= 0 and swap_to != imax:
            a_row = int(wgs[0].prow[swap_to - p0])       # the row that leaves position swap_to
            for wg in wgs:
                wg.prow[swap_to - p0] = rmax
                if imax - p0 < len(wg.prow):
                    wg.prow[imax - p0] = a_row
                if wg.owns(a_row):
                    wg.pos[wg.li(a_row)] = imax
                if wg.owns(rmax):
                    wg.pos[wg.li(rmax)] = swap_to
        # ---- eliminate (every workgroup on its rows); the newest W entries of the candidate rows from the messages
        newest = {}
        if zero:
            dvec[k], doff[k], ptype[k] = 0.0, 0.0, 1
            if state["info"] == 0:
                state["info"] = k + 1
            for wg in wgs:
                wg.LW[:, kb] = 0.0; wg.Wown[:, kb] = 0.0
                if wg.owns(q):
                    wg.active[wg.li(q)] = False
            for r in (n1, n2):
                if r >= 0:
                    newest.setdefault(r, {})[kb] = 0.0
        elif kstep == 1:
            pr = rmax if use_wn else q
            d = p22 if use_wn else akk
            dvec[k], doff[k], ptype[k] = d, 0.0, 1
            for wg in wgs:
                w = wn_all[wg.g] if use_wn else wk_all[wg.g]
                for i, r in enumerate(wg.rows):
                    if wg.active[i] and r != pr:
                        wg.LW[i, kb] = w[i] * (1.0 / d)
                    if wg.active[i]:
                        wg.Wown[i, kb] = w[i]
                if wg.owns(pr):
                    wg.active[wg.li(pr)] = False
            src = {n1: wn_n1, n2: wn_n2, q: wn_q} if use_wn else {n1: wk_n1, n2: wk_n2}
            for r, v in src.items():
                if r >= 0:
                    newest.setdefault(r, {})[kb] = v
        else:
            p11 = akk
            d11, d22 = p22 / p21, p11 / p21
            tt = 1.0 / (d11 * d22 - 1.0) / p21
            dvec[k], dvec[k + 1], doff[k], doff[k + 1], ptype[k], ptype[k + 1] = p11, p22, p21, 0.0, 2, 3
            for wg in wgs:
                wk, wn = wk_all[wg.g], wn_all[wg.g]
                for i, r in enumerate(wg.rows):
                    if wg.active[i] and r != q and r != rmax:
                        wg.LW[i, kb] = tt * (d11 * wk[i] - wn[i])
                        wg.LW[i, kb + 1] = tt * (d22 * wn[i] - wk[i])
                    if wg.active[i]:
                        wg.Wown[i, kb] = wk[i]; wg.Wown[i, kb + 1] = wn[i]
                for r in (q, rmax):
                    if wg.owns(r):
                        wg.active[wg.li(r)] = False
            for r, v0, v1 in ((n1, wk_n1, wn_n1), (n2, wk_n2, wn_n2)):
                if r >= 0:
                    newest.setdefault(r, {})[kb] = v0
                    newest[r][kb + 1] = v1
        # the owners' stores to the global W become visible with the NEXT hop (message A of the next column): the next
        # column's phase A must not read them (the candidates' entries travel in the messages), its phase B may
        state["visible"] = [(wg.rows.copy(), c, wg.Wown[:, c].copy()) for wg in wgs for c in range(kb, kb + kstep)]
        wq_new = newest
        kb += kstep
    for rows, c, vals in state.pop("visible", []):
        Wglob[rows, c] = vals
    # ---- panel end: rows in position order
    LWp = np.zeros((Np, NB + 8)); Wp = np.zeros((Np, NB + 8))
    rowof = np.arange(Np)
    for wg in wgs:
        for i, r in enumerate(wg.rows):
            ps = int(wg.pos[i])
            rowof[ps] = r
            ncol = min(kb, ps - p0) if ps < p0 + kb else kb   # an eliminated row keeps the columns in front of its own
            LWp[ps, :ncol] = wg.LW[i, :ncol]
            if ps >= p0 + kb:
                Wp[ps, :kb] = wg.Wown[i, :kb]
    return kb, LWp, Wp, rowof, hops


def factor(A):
    """P A P' = L D L' of the symmetric matrix A (full storage) with the multi-workgroup panel model + the permutation
    and trailing-update steps the other kernels do.  Returns L (unit lower), dvec, doff, ptype, perm, info, hops."""
    N = A.shape[0]
    F = np.tril(A).astype(float)
    Lfull = np.zeros((N, N))
    perm = np.arange(N)
    state = {"dvec": np.zeros(N), "doff": np.zeros(N), "ptype": np.ones(N, int), "info": 0}
    p0, hops = 0, 0
    while p0 < N:
        kb, LWp, Wp, rowof, h = panel(F, p0, N, state)
        hops += h
        # symmetric permutation of the trailing matrix, the same row permutation of the previous columns and of perm
        sym = F + np.tril(F, -1).T
        sym = sym[np.ix_(rowof, rowof)]
        F = np.tril(sym)
        Lfull[p0:, :p0] = Lfull[rowof[p0:], :p0]
        perm = perm[rowof]
        # the panel's columns of L, then the trailing update
        for c in range(kb):
            Lfull[p0 + c + 1:, p0 + c] = LWp[p0 + c + 1:, c]
        r0 = p0 + kb
        upd = LWp[r0:, :kb] @ Wp[r0:, :kb].T
        F[r0:, r0:] -= np.tril(upd)
        p0 = r0
    L = Lfull + np.eye(N)
    for k in range(N):
        if state["ptype"][k] == 2:
            L[k + 1, k] = 0.0
    return L, state["dvec"], state["doff"], state["ptype"], perm, state["info"], hops


def factor_plain(A):
    """dsytf2-style unblocked reference with physical interchanges applied to the previous columns as well"""
    N = A.shape[0]
    S = A.astype(float).copy()
    L = np.eye(N)
    dvec, doff, ptype = np.zeros(N), np.zeros(N), np.ones(N, int)
    perm = np.arange(N)
    info, k = 0, 0
    fmax = np.finfo(float).max

    def swap(i, j):
        S[[i, j], :] = S[[j, i], :]; S[:, [i, j]] = S[:, [j, i]]
        L[[i, j], :k] = L[[j, i], :k]
        perm[[i, j]] = perm[[j, i]]
    while k < N:
        absakk = abs(S[k, k])
        if k + 1 < N:
            col = np.abs(S[k + 1:, k]); col[~(col <= fmax)] = fmax
            imax = k + 1 + int(np.argmax(col)); colmax = col[imax - k - 1]
        else:
            imax, colmax = k, 0.0
        kstep, kp = 1, k
        if not (max(absakk, colmax) > 0.0) or not (absakk <= fmax) or colmax >= fmax:
            if info == 0:
                info = k + 1
            dvec[k] = 0.0; k += 1
            continue
        if absakk < ALPHA * colmax:
            row = np.abs(S[k:, imax]).copy(); row[imax - k] = -1.0
            rowmax = row.max()
            if absakk >= ALPHA * colmax * (colmax / rowmax):
                pass
            elif abs(S[imax, imax]) >= ALPHA * rowmax:
                kp = imax
            else:
                kp, kstep = imax, 2
        kk = k + kstep - 1
        if kp != kk:
            swap(kk, kp)
        if kstep == 1:
            d = S[k, k]
            l = S[k + 1:, k] * (1.0 / d)
            S[k + 1:, k + 1:] -= np.outer(l, S[k + 1:, k])
            L[k + 1:, k] = l
            dvec[k] = d
        else:
            p11, p21, p22 = S[k, k], S[k + 1, k], S[k + 1, k + 1]
            d11, d22 = p22 / p21, p11 / p21
            tt = 1.0 / (d11 * d22 - 1.0) / p21
            w1, w2 = S[k + 2:, k].copy(), S[k + 2:, k + 1].copy()
            l1, l2 = tt * (d11 * w1 - w2), tt * (d22 * w2 - w1)
            S[k + 2:, k + 2:] -= np.outer(l1, w1) + np.outer(l2, w2)
            L[k + 2:, k], L[k + 2:, k + 1] = l1, l2
            dvec[k], dvec[k + 1], doff[k], ptype[k], ptype[k + 1] = p11, p22, p21, 2, 3
        k += kstep
    return L, dvec, doff, ptype, perm, info


def reconstruct(L, dvec, doff, ptype):
    N = len(dvec)
    D = np.diag(dvec)
    for k in range(N):
        if ptype[k] == 2:
            D[k + 1, k] = D[k, k + 1] = doff[k]
    return L @ D @ L.T
