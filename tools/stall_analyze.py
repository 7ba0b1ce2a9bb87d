"""Offline reading of a state file of tools/stall_hunt.py (a diagnostic build with -DMNK_DIAG_BULK_DBG=1): which bulk
workgroup held which task in which wait, and which of the long waits are roots (their condition is already true in the
progress words the time-out left, or nobody holds the task that would make it true).
usage: python tools/stall_analyze.py gpurun_out/stall2/state_0.json [chunk band taper0]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dag_tasks import dag_tasks  # noqa: E402

r = json.load(open(sys.argv[1]))
chunk, band, taper0 = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 16, 2)
Np, nt = r["Np"], r["ntile"]
ts = dag_tasks(nt, chunk, band // 2, js2=nt // 2 + nt % 2, taper0=taper0)
assert len(ts) == int(r["dag_ntasks"]), (len(ts), r["dag_ntasks"])
fl = np.array(r["flags"])
qctr = fl[0]
front = fl[2:2 + Np // 64]
af = fl[2 + Np // 64:2 + Np // 64 + nt * nt].reshape(nt, nt)
tprog = fl[2 + Np // 64 + nt * nt:2 + Np // 64 + 2 * nt * nt].reshape(nt, nt)
b = np.array(r["bulk"])
act = b[b[:, 6] > 0]
holder = {int(x[0]): i for i, x in enumerate(act)}
print(f"queue counter {qctr} of {len(ts)}; {len(act)} workgroups; chain: {[c for c in r['chain'] if c[0]]}")
# which task publishes what
closing = {}   # (I, J) -> position of the closing / band-final task
chunkpos = {}  # (I, J, q) -> position
for pos, (ready, cls, J, I, flags, q, kb, ke) in enumerate(ts):
    chunkpos[(I, J, q)] = pos
    if flags & 2:
        closing[(I, J)] = pos


def describe(pos):
    ready, cls, J, I, flags, q, kb, ke = ts[pos]
    return f"pos {pos} tile ({I},{J}) flags {flags} q {q} k [{kb},{ke})"


rows = []
for x in act:
    t, stage = int(x[0]), int(x[1])
    if stage not in (2, 3):
        continue
    ready, cls, J, I, flags, q, kb, ke = ts[t]
    if stage == 3:
        need = ("tprog", I, J, int(x[3]))
        now = int(tprog[I, J])
        ok_now = now >= int(x[3])
        pub = chunkpos.get((I, J, int(x[3]) - 1))
    else:
        s0, s2 = int(x[2]) & 0xffff, int(x[2]) >> 16
        c = int(x[3])
        seen = [int(v) for v in x[8:12]]
        words = [s0, s0 + 1, s2, s2 + 1]
        lag = [w for w, v in zip(words, seen) if v <= c]
        need = ("front", lag, c)
        now = [int(front[w]) for w in words]
        ok_now = min(now) > c
        pub = closing.get((lag[0] // 2, c)) if lag else None
    rows.append((t, stage, describe(t), need, "seen", [int(v) for v in x[8:12]] if stage == 2 else int(x[4]), "polls>>18", int(x[5]), "now", now,
                 "TRUE NOW" if ok_now else "", "publisher", describe(pub) if pub is not None else None,
                 "held by a workgroup in stage %d" % act[holder[pub]][1] if pub in holder else ("not held, popped" if pub is not None and pub < qctr else "not popped")))
rows.sort()
for row in rows[: int(os.environ.get("ROWS", "40"))]:
    print(*row)
pubs = {}
for row in rows:
    pubs.setdefault((row[11], row[12]), 0)
    pubs[(row[11], row[12])] += 1
print("\nwaited-for publishers:")
for k, v in sorted(pubs.items(), key=lambda kv: -kv[1]):
    print(v, k)
