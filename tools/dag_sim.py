"""Discrete-event model of the task-DAG factorization schedule (csrc/dag.hip + the persistent pivot chain), used to compare
task ORDERS on the CPU before spending GPU time: 720 workgroup slots pop the task list in order and block on the same
progress words as the kernel (front per row, per-tile chunk order, band-tile flags); chunk compute shares the machine
(processor sharing with a per-workgroup ceiling), the tile-closing tasks and the chain steps are latency-bound.

usage: python tools/dag_sim.py [ntile] [order ...]      orders: ready (the shipped one), col, mix:<beta>, ...
Calibration constants at the top were set from the traces of round 3 (profiles/r03_dag_timeline_C3.md)."""
import heapq
import sys

BAND, FINAL, FIRST = 1, 2, 4
PRIO = lambda k: k["I"]   # noqa: E731  (dynamic mode: priority among the ready heads of the rows; smaller = first)
PRIO_FIRST = True

# ---- calibration (microseconds)
SLOTS = 720            # 240 CUs x 3 workgroups
CUS = 240
T_K_ALONE = 21.0       # one 128x128x128 k-step of a workgroup that has its CU for itself
T_K_CU = 18.0          # the same when the CU is saturated (3 workgroups): per-CU time per k-step
T_TASK = 6.0           # pop + epilogue read-modify-write of the tile
T_CLOSE_K = 22.0       # the closing task's single k-step on an idle CU (operands were just written by other workgroups) ...
T_FINAL = 22.0         # ... and its finalization (two triangular solves against the diagonal blocks); both stretch with the
CLOSE_LOAD = 0.5       # load of the CU: x (1 - CLOSE_LOAD + CLOSE_LOAD * slowdown) -> 42 + 31 us mid-run (measured), ~50 us in the tail
T_STEP = 98.0          # one strip-column of the chain, diagonal blocks published at +50 / +98
T_BANDROW = 30.0       # band rows below the diagonal ones finish this long after the last diagonal block


def build(ntile, chunk=8, band_tiles=8, order="ready"):
    ts = []

    def add_tile(I, J, K, band):
        body = K if band else max(0, K - 1)
        cuts = [body]
        e = body
        ln = 1
        while ln < chunk and e > 0:
            e = max(0, e - ln)
            cuts.append(e)
            ln *= 2
        first = 1 + (I * 5 + J * 3) % chunk
        while e > first:
            e = max(first, e - chunk)
            cuts.append(e)
        if e > 0:
            cuts.append(0)
        q = 0
        for c in range(len(cuts) - 1, 0, -1):
            kb, ke = cuts[c], cuts[c - 1]
            last = band and ke == K
            ts.append(dict(ready=ke, cls=0 if band else 2, J=J, I=I, flags=(BAND if band else 0) | (FINAL if last else 0) | (FIRST if q == 0 else 0),
                           q=q, kb=kb, ke=ke, need=(J if band else J)))
            q += 1
        if not band:
            ts.append(dict(ready=K, cls=1, J=J, I=I, flags=FINAL | (FIRST if q == 0 else 0), q=q, kb=body, ke=K, need=J))

    for Jt in range(ntile):
        Js = Jt // 2
        for I in range(2 * Js + band_tiles, ntile):
            add_tile(I, Jt, Jt, False)
        if 2 * Js - 2 > 0:
            for I in range(max(2 * Js, Jt), min(2 * Js + band_tiles, ntile)):
                add_tile(I, Jt, 2 * Js - 2, True)
    ts.sort(key=order_key(order, ntile, band_tiles))
    return ts


def order_key(order, ntile, band_tiles):
    if order == "ready":
        return lambda t: (t["ready"], t["cls"], t["J"], t["I"])
    if order == "col":      # left-looking by tile column: everything of column J when the chain gets there
        return lambda t: (t["J"], t["cls"] != 0, t["q"], t["I"])
    if order.startswith("mix:"):   # between the moment a chunk becomes ready and the moment its tile is due
        beta = float(order[4:])
        return lambda t: (t["ready"] + beta * (t["J"] - t["ready"]) if t["cls"] == 2 else t["ready"], t["cls"], t["J"], t["I"])
    if order.startswith("row:"):   # deferral grows with the distance of the row from the band
        beta = float(order[4:])

        def key(t):
            due = t["J"] if t["cls"] != 2 else min(t["J"], t["ready"] + beta * max(0, t["I"] - band_tiles - t["ready"]))
            return (max(t["ready"], due), t["cls"], t["J"], t["I"])
        return key
    if order.startswith("dyn"):
        return lambda t: (t["ready"], t["cls"], t["J"], t["I"])
    if order.startswith("three:"):
        return lambda t: (t["ready"], t["cls"], t["J"], t["I"])
    if order.startswith("two:"):   # handled by the caller (queue assignment), order inside a queue: by readiness
        return lambda t: (t["ready"], t["cls"], t["J"], t["I"])
    raise SystemExit("unknown order " + order)


def assign_queues(ts, order):
    """two:<H>:<nU>  urgent queue = closing tasks, band tiles and the chunks of rows within H tile rows of the chain
    position that makes them ready; nU of the 720 slots serve it first."""
    if order.startswith("three:"):   # three:<nBand>:<nClose>  band tiles | closing tasks | chunks, each with its own slots
        _, nb, nc = order.split(":")
        for k in ts:
            k["queue"] = k["cls"]
        return [int(nb), int(nc), SLOTS - int(nb) - int(nc)]
    if not order.startswith("two:"):
        return None
    _, H, nU = order.split(":")
    H, nU = int(H), int(nU)
    for k in ts:
        k["queue"] = 0 if (k["cls"] != 2 or k["I"] - k["ready"] <= H) else 1
    return [nU, SLOTS - nU]


def simulate(ntile, ts, band_tiles=8, verbose=False, pools=None, dyn=None):
    """pools: None = one queue, every slot pops it in order.  Otherwise a list of slot counts, one per queue id (task key
    "queue"): a slot pops its home queue in order and helps the next non-empty queue when its own is exhausted."""
    nsc = (ntile + 1) // 2
    if pools is None:
        pools = [SLOTS]
    nq = len(pools)
    queues = [[i for i, k in enumerate(ts) if k.get("queue", 0) == q] for q in range(nq)]
    if dyn is not None:
        # dyn = key function ordering the per-row lists: a free slot takes the first READY head in row order (deadline
        # priority), never blocks on a task that is not ready
        rowq = [sorted([i for i, k in enumerate(ts) if k["I"] == r], key=lambda i: dyn(ts[i])) for r in range(ntile)]
        rowpos = [0] * ntile
        queues = [[]]
    qpos = [0] * nq
    freeq = list(pools)
    rowdone = [0] * ntile            # tile columns of row I that are final (the kernel's front words)
    tprog = {}                       # (I, J) -> chunks applied
    af = set()                       # band tiles accumulated
    wait_row = [[] for _ in range(ntile)]
    wait_tile = {}
    real = []                        # (time, seq, kind, payload)
    comp = []                        # (finish_V, seq, task index)
    seq = [0]
    t = 0.0
    V = 0.0
    ncomp = 0
    qhead = 0
    free = SLOTS
    busy_area = 0.0
    chain = dict(next=0, busy_until=0.0, started=[None] * nsc, d3=[None] * nsc)
    stall = [0.0] * nsc
    done_tasks = 0

    def push_real(tt, kind, payload):
        seq[0] += 1
        heapq.heappush(real, (tt, seq[0], kind, payload))

    def rate():
        return 0.0 if ncomp == 0 else min(1.0 / T_K_ALONE, CUS / (T_K_CU * ncomp))

    def stretch():
        r = rate()
        slow = 1.0 if r == 0 else (1.0 / r) / T_K_ALONE
        return 1.0 - CLOSE_LOAD + CLOSE_LOAD * slow

    def deps_ok(i):
        k = ts[i]
        I, J, ke = k["I"], k["J"], k["ke"]
        if k["kb"] < ke:
            if rowdone[I] < ke:
                return ("row", I, ke)
            if rowdone[J] < ke:
                return ("row", J, ke)
        return None

    def try_start(i):
        nonlocal ncomp
        d = deps_ok(i)
        if d is not None:
            wait_row[d[1]].append((d[2], i, "start"))
            return
        k = ts[i]
        nk = k["ke"] - k["kb"]
        if k["cls"] == 1:           # closing task: latency-bound k-step, then epilogue order, then diagonal blocks, then finalize
            push_real(t + (T_CLOSE_K * stretch() if nk else 0.0) + T_TASK, "closed_k", i)
        else:
            seq[0] += 1
            heapq.heappush(comp, (V + nk, seq[0], i))
            ncomp += 1

    def after_compute(i):
        k = ts[i]
        if not (k["flags"] & FIRST) and tprog.get((k["I"], k["J"]), 0) < k["q"]:
            wait_tile.setdefault((k["I"], k["J"]), []).append((k["q"], i))
            return
        if k["cls"] == 1:
            if rowdone[k["J"]] < k["J"] + 1:          # diagonal blocks of column J
                wait_row[k["J"]].append((k["J"] + 1, i, "final"))
                return
            push_real(t + T_FINAL * stretch(), "done", i)
        else:
            push_real(t + T_TASK, "done", i)

    def set_rowdone(r, c):
        if c <= rowdone[r]:
            return
        rowdone[r] = c
        w = wait_row[r]
        if w:
            keep = []
            for need, i, what in w:
                if need <= c:
                    (try_start if what == "start" else after_compute)(i)
                else:
                    keep.append((need, i, what))
            wait_row[r] = keep
        chain_poke()

    def chain_ready(Js):
        if 2 * Js - 2 > 0:
            for I in (2 * Js, 2 * Js + 1):
                for J in (2 * Js, 2 * Js + 1):
                    if I < ntile and J < ntile and I >= J and (I, J) not in af:
                        return False
        return True

    def chain_poke():
        Js = chain["next"]
        if Js >= nsc or chain["started"][Js] is not None or t < chain["busy_until"]:
            return
        if Js > 0 and chain["d3"][Js - 1] is None:
            return
        if not chain_ready(Js):
            return
        chain["started"][Js] = t
        stall[Js] = t - chain["busy_until"]
        chain["busy_until"] = t + T_STEP
        push_real(t + T_STEP * 50.0 / 98.0, "diag", (Js, 0))
        push_real(t + T_STEP, "diag", (Js, 1))

    def band_rows_poke(Js):
        # rows 2Js+2 .. of the band finish strip-column Js once their band tiles are accumulated and they are final up to it
        d3 = chain["d3"][Js]
        if d3 is None:
            return
        for r in range(2 * Js + 2, min(2 * Js + band_tiles, ntile)):
            if rowdone[r] >= 2 * Js + 2 or (Js, r) in pending_band:
                continue
            ok = rowdone[r] >= 2 * Js
            if ok and 2 * Js - 2 > 0:
                ok = (r, 2 * Js) in af and (r, 2 * Js + 1) in af
            if ok:
                pending_band.add((Js, r))
                push_real(max(t, d3) + T_BANDROW, "bandrow", (Js, r))

    pending_band = set()
    home = {}

    # main loop
    while True:
        # pop tasks while slots are free
        while dyn is not None and freeq[0] > 0:
            got = None
            best = None
            for r in range(ntile):
                if rowpos[r] < len(rowq[r]):
                    i = rowq[r][rowpos[r]]
                    k = ts[i]
                    need_i = k["ke"] if k["cls"] != 1 else k["J"]
                    need_j = k["ke"] if k["cls"] != 1 else k["J"] + 1
                    if rowdone[k["I"]] >= need_i and rowdone[k["J"]] >= need_j:
                        pr = PRIO(k)
                        if best is None or pr < best:
                            best, got = pr, (r, i)
                        if PRIO_FIRST:
                            break
            if got is None:
                break
            rowpos[got[0]] += 1
            freeq[0] -= 1
            home[got[1]] = 0
            try_start(got[1])
        for q in range(nq if dyn is None else 0):
            while freeq[q] > 0:
                src = q if qpos[q] < len(queues[q]) else next((z for z in range(nq) if qpos[z] < len(queues[z])), None)
                if src is None:
                    break
                freeq[q] -= 1
                i = queues[src][qpos[src]]
                qpos[src] += 1
                home[i] = q
                try_start(i)
        chain_poke()
        r = rate()
        tr = real[0][0] if real else None
        tv = t + (comp[0][0] - V) / r if comp and r > 0 else None
        if tr is None and tv is None:
            break
        if tv is not None and (tr is None or tv <= tr):
            dt = tv - t
            busy_area += ncomp * dt
            V += r * dt
            t = tv
            _, _, i = heapq.heappop(comp)
            ncomp -= 1
            after_compute(i)
        else:
            dt = tr - t
            busy_area += ncomp * dt
            V += r * dt
            t = tr
            _, _, kind, p = heapq.heappop(real)
            if kind == "closed_k":
                after_compute(p)
            elif kind == "done":
                k = ts[p]
                done_tasks += 1
                freeq[home[p]] += 1
                if k["cls"] == 1:
                    set_rowdone(k["I"], k["J"] + 1)
                    for Js in range(max(0, (k["J"] - 1) // 2), min(nsc, k["J"] // 2 + 2)):
                        band_rows_poke(Js)
                elif k["flags"] & FINAL:
                    af.add((k["I"], k["J"]))
                    band_rows_poke(k["J"] // 2)
                else:
                    tprog[(k["I"], k["J"])] = k["q"] + 1
                    w = wait_tile.pop((k["I"], k["J"]), None)
                    if w:
                        for q, i in w:
                            if q <= k["q"] + 1:
                                after_compute(i)
                            else:
                                wait_tile.setdefault((k["I"], k["J"]), []).append((q, i))
            elif kind == "diag":
                Js, h = p
                row = 2 * Js + h
                if row < ntile:
                    set_rowdone(row, row + 1 if h == 0 else 2 * Js + 2)
                if h == 1:
                    if 2 * Js < ntile:
                        set_rowdone(2 * Js, 2 * Js + 2)
                    chain["d3"][Js] = t
                    chain["next"] = Js + 1
                    band_rows_poke(Js)
            elif kind == "bandrow":
                Js, r_ = p
                set_rowdone(r_, 2 * Js + 2)
                band_rows_poke(Js + 1)
        chain_poke()
    assert done_tasks == len(ts), (done_tasks, len(ts), qpos)
    d3 = chain["d3"]
    if verbose:
        prev = 0.0
        for Js in range(nsc):
            print(f"Js={Js:2d} D3 {d3[Js]:8.0f} (+{d3[Js]-prev:5.0f}) stall {stall[Js]:5.0f}")
            prev = d3[Js]
    ksteps = sum(k["ke"] - k["kb"] for k in ts)
    return dict(total_us=t, chain_end_us=d3[-1], slot_busy=busy_area / (t * SLOTS), ksteps=ksteps, ntasks=len(ts),
                chain_stall_us=sum(stall))


if __name__ == "__main__":
    # (module-level PRIO / PRIO_FIRST are rebound per order below)
    ntile = int(sys.argv[1]) if len(sys.argv) > 1 else 88
    global_dummy = None
    orders = sys.argv[2:] or ["ready", "col", "mix:0.5"]
    for o in orders:
        ts = build(ntile, order=o if not o.startswith("dyn") else "ready")
        dyn = (lambda k: (k["ready"], k["cls"], k["J"])) if o.startswith("dyn") else None
        if o.startswith("dyn:"):
            g = float(o[4:])
            PRIO_FIRST = False
            PRIO = lambda k, g=g: (k["cls"] == 2, k["J"] - g * k["I"])  # noqa: E731
        if o.startswith("dynrow"):      # closing tasks and band tiles first, chunks by row (the row the chain needs next)
            PRIO_FIRST = False
            w = float(o[7:]) if len(o) > 7 else 0.0
            PRIO = lambda k, w=w: (k["cls"] == 2, k["I"] + w * k["J"])  # noqa: E731
        r = simulate(ntile, ts, verbose=len(orders) == 1, pools=assign_queues(ts, o), dyn=dyn)
        print(f"{o:12s} total {r['total_us']/1e3:7.3f} ms  chain end {r['chain_end_us']/1e3:7.3f} ms  chain stalled {r['chain_stall_us']/1e3:6.3f} ms  "
              f"compute share of slot-time {r['slot_busy']:.3f}  ({r['ntasks']} tasks, {r['ksteps']} k-steps)")
