"""Host-side mirror of the task list of the task-DAG schedule, for the trace tools."""

def dag_tasks(ntile, chunk, band_tiles, js2=None, taper0=1):
    """Python mirror of dag_build_tasks (csrc/dag.hip): list of (ready, cls, J, I, flags, q, kbeg, kend), in queue order."""
    BAND, FINAL, FIRST = 1, 2, 4
    ts = []

    def add_tile(I, J, K, band):
        body = K if band else max(0, K - 1)
        cuts = [body]
        e = body
        ln = taper0
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
            ts.append((ke, 0 if band else 2, J, I, (BAND if band else 0) | (FINAL if last else 0) | (FIRST if q == 0 else 0), q, kb, ke))
            q += 1
        if not band:
            ts.append((K, 1, J, I, FINAL | (FIRST if q == 0 else 0), q, body, K))

    if js2 is None:
        js2 = (ntile + 1) // 2
    for Jt in range(ntile):
        Js = Jt // 2
        bt = ntile if Js >= js2 else band_tiles
        for I in range(2 * Js + bt, ntile):
            add_tile(I, Jt, Jt, False)
        if 2 * Js - 2 > 0:
            for I in range(max(2 * Js, Jt), min(2 * Js + bt, ntile)):
                add_tile(I, Jt, 2 * Js - 2, True)
    ts.sort(key=lambda t: (t[0], t[1], t[2], t[3]))
    return ts
