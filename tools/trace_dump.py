"""Dump the kernel trace of a rocprofv3 rocpd database as CSV (name, start_ns, end_ns, queue/stream ids) for offline
timeline analysis.  usage: python tools/trace_dump.py results.db out.csv [max_rows]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols, file=sys.stderr)
want = [c for c in ("name", "start", "end", "queue_id", "stream_id", "grid_x", "grid_size_x", "workgroup_x", "workgroup_size_x") if c in cols]
rows = cur.execute(f"select {','.join(want)} from kernels order by start").fetchall()
mx = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
t0 = rows[0][1] if rows else 0
with open(sys.argv[2], "w") as f:
    f.write(",".join(want) + "\n")
    for r in rows[:mx]:
        r = list(r)
        r[0] = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("mnk::", "").replace(",", ";")
        r[1] -= t0
        r[2] -= t0
        f.write(",".join(str(x) for x in r) + "\n")
