"""Summarize a rocprofv3 rocpd SQLite database (kernel-trace): per-kernel count / total / avg /
min / max duration, like `--stats` CSV output.  usage: python tools/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("mnk::", "")
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1e30, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {100*a[1]/tot:.1f} |")
    text = "\n".join(lines)
    # span of every factorize! call (densify kernel -> end of the last inverse kernel): the figure that has to agree
    # with the HIP-event duration bench.py reports as ms_per_factorize
    order = sorted(rows, key=lambda r: r[1])
    # (start: the kernel that densifies the matrix into the factor buffer; the zero-fill may run in the background long
    # before it.  end: the LAST inverse launch before the next start -- the task-DAG schedule inverts in two launches)
    starts = [i for i, r in enumerate(order) if "scatter_csc_kernel" in r[0] or "copy_lower_kernel" in r[0]]
    ends = [i for i, r in enumerate(order) if "linv_tri_kernel" in r[0] or "linv256_mfma_kernel" in r[0]]
    spans = []
    for n, i0 in enumerate(starts):
        nxt = starts[n + 1] if n + 1 < len(starts) else len(order)
        e = [x for x in ends if i0 < x < nxt]
        if e:
            spans.append((order[e[-1]][2] - order[i0][1]) / 1e6)
    if len(spans) > 1:
        sp = spans[1:]
        text += ("\n\nSpan of one `factorize!` call in this trace (`scatter_csc_kernel` to the end of the last inverse kernel `linv256_mfma_kernel`, "
                 "the two look-ahead streams overlapping inside): mean %.3f ms, min %.3f ms over %d calls "
                 "(to be compared with `ms_per_factorize` of the same run's bench.py JSON line)." % (sum(sp) / len(sp), min(sp), len(sp)))
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
