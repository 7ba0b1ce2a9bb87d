"""Per-outer-step summary of a factorization kernel trace (rocpd sqlite): duration of the (a)/(b)
updates on the update stream and of the panel stream's busy span.  usage: trace_steps.py db [which]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id,grid_x,workgroup_x from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "fill_lower" in r[0] or "copy_lower_kernel" in r[0]]
ends = [i for i, r in enumerate(rows) if "linv256_kernel" in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
i0 = starts[which]
i1 = [e for e in ends if e > i0][0] + 1
seg = rows[i0:i1]
t0 = seg[0][1]
print("factorization span %.1f us, %d kernels" % ((seg[-1][2] - t0) / 1e3, len(seg)))
qs = sorted({r[3] for r in seg})
print("queues", qs)
big = [r for r in seg if "gemm" in r[0] and ("queue" in r[0] or "ILi2ELi2ELi4ELi2" in r[0] or "<2, 2, 4, 2" in r[0] or "<2, 2, 2, 2" in r[0])]
for r in seg:
    n = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("mnk::", "")
    d = (r[2] - r[1]) / 1e3
    if ("queue" in n) or ("gemm_nt_kernel<2, 2, 4, 2" in n and d > 60) or ("gemm_nt_kernel<2, 2, 2, 2" in n):
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} {d:8.1f} q{r[3]} g{r[4]//r[5]:5d} {n[:40]}")
