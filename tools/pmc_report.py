"""Turn rocprofv3 PMC databases (rocpd sqlite, one counter set per pass) into the tables kept under profiles/.
usage:
  pmc_report.py traffic <fetch.db> <write.db> <out.md> <out.json> "<config text>"
  pmc_report.py mfma <mfma.db> <out.md> "<config text>"
Counters are aggregated per kernel over the LAST factorization in the trace (from the densify kernel to
the last inverse kernel).  FETCH_SIZE / WRITE_SIZE are KiB; gfx950 counts half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so read bytes = 2 x FETCH_SIZE x 1024."""
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("mnk::", "")
    return n[:60]


def load(dbfile):
    db = sqlite3.connect(dbfile)
    rows = db.execute("select dispatch_id, name, start, end, counter_name, counter_value from pmc_events order by start").fetchall()
    disp = {}
    for d, name, s, e, c, v in rows:
        r = disp.setdefault(d, {"name": name, "start": s, "end": e, "c": {}})
        r["c"][c] = r["c"].get(c, 0.0) + v
    order = sorted(disp.values(), key=lambda r: r["start"])
    starts = [i for i, r in enumerate(order) if "scatter_csc_kernel" in r["name"] or "copy_lower_kernel" in r["name"]]
    ends = [i for i, r in enumerate(order) if "linv_tri_kernel" in r["name"] or "linv256_mfma_kernel" in r["name"]]
    i0 = starts[-1]
    i1 = [e for e in ends if e > i0][-1] + 1
    seg = order[i0:i1]
    # the zero-fill of the buffer belongs to the call even when it ran in the background before it
    fills = [r for r in order[:i0] if "fill_lower" in r["name"]]
    return (fills[-1:] if fills else []) + seg


def per_kernel(seg, counters):
    agg = {}
    for r in seg:
        a = agg.setdefault(short(r["name"]), {"n": 0, "t": 0.0, **{c: 0.0 for c in counters}})
        a["n"] += 1
        a["t"] += (r["end"] - r["start"]) / 1e6
        for c in counters:
            a[c] += r["c"].get(c, 0.0)
    return agg


def traffic(fetch_db, write_db, out_md, out_json, cfg):
    f = per_kernel(load(fetch_db), ["FETCH_SIZE"])
    w = per_kernel(load(write_db), ["WRITE_SIZE"])
    lines = [f"# PMC traffic, one factorize! call ({cfg}), MI355X", "",
             "Two separate passes (`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, counters only,",
             "as MI355X_MICROARCH.md prescribes). FETCH_SIZE/WRITE_SIZE are KiB; gfx950 counts half of the bytes of wide",
             "coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is used as reported.", "",
             "| kernel | launches | FETCH_SIZE KiB | read GB (x2 corrected) | WRITE_SIZE KiB | write GB |", "|---|---|---|---|---|---|"]
    rd = wr = 0.0
    for k in sorted(f, key=lambda k: -f[k]["FETCH_SIZE"]):
        fk = f[k]["FETCH_SIZE"]
        wk = w.get(k, {"WRITE_SIZE": 0.0})["WRITE_SIZE"]
        rd += 2 * fk * 1024
        wr += wk * 1024
        if fk > 1000 or wk > 1000:
            lines.append(f"| {k} | {f[k]['n']} | {fk:.0f} | {2*fk*1024/1e9:.2f} | {wk:.0f} | {wk*1024/1e9:.2f} |")
    lines += ["", f"**Total per factorize! call: read {rd/1e9:.1f} GB + write {wr/1e9:.1f} GB = {(rd+wr)/1e9:.1f} GB**"]
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump({"config": cfg, "read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr, "source": out_md}, open(out_json, "w"))
    print("\n".join(lines))


def mfma(db, out_md, cfg):
    cs = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    a = per_kernel(load(db), cs)
    lines = [f"# MFMA utilisation per kernel, one factorize! call ({cfg}), MI355X", "",
             "`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`. SQ_VALU_MFMA_BUSY_CYCLES is summed",
             "over the SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: busy fraction = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8).", "",
             "| kernel | launches | total ms | MFMA busy cycles | GUI_ACTIVE (8 XCDs) | MFMA busy / SIMD-cycles |", "|---|---|---|---|---|---|"]
    for k in sorted(a, key=lambda k: -a[k]["t"]):
        r = a[k]
        if r["t"] < 0.05:
            continue
        gui = r["GRBM_GUI_ACTIVE"]
        frac = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui / 8) if gui > 0 else 0.0
        lines.append(f"| {k} | {r['n']} | {r['t']:.2f} | {r['SQ_VALU_MFMA_BUSY_CYCLES']:.3e} | {gui:.3e} | {frac:.2f} |")
    lines += ["", "Kernels on the two look-ahead streams run concurrently on disjoint CU sets (64 / 192 CUs), so a kernel that owns",
              "192 CUs can reach at most 0.75 here; GUI_ACTIVE counts wall cycles of the dispatch."]
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:7])
    else:
        mfma(*sys.argv[2:5])
