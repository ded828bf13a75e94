#!/usr/bin/env python3
"""Timeline of the LAST bundle-adjustment step in a rocprofv3 rocpd .db: kernels in start order with the idle gap
before each (gaps >= --gap us are printed), plus busy/idle totals between the last two k_classify dispatches.
    python tools/rocpd_timeline.py gpurun_out/prof/x_results.db [--gap 3] [--back K]
"""
import re
import sqlite3
import sys


def main(path: str, gap_us: float, back: int = 0) -> None:
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    rows = [(re.sub(r"\(.*", "", re.sub(r"^void ", "", n)), s, e) for n, s, e in rows]
    marks = [i for i, r in enumerate(rows) if "k_classify" in r[0]]
    if back:
        marks = marks[:-back]      # --back K: the step K before the last one (bench.py's last steps carry stage events)
    if len(marks) < 2:
        print("need two k_classify dispatches"); return
    seg = rows[marks[-2] + 1: marks[-1] + 1]
    t0 = seg[0][1]
    busy = sum(e - s for _, s, e in seg)
    span = seg[-1][2] - rows[marks[-2]][2]
    print(f"# last step: {len(seg)} dispatches, span {span/1e3:.1f} us, kernel time {busy/1e3:.1f} us, idle {100*(1-busy/span):.1f} %")
    agg = {}
    prev_end = rows[marks[-2]][2]
    for n, s, e in seg:
        g = (s - prev_end) / 1e3
        key = n.split("ENS_")[0][-40:]
        a = agg.setdefault(n, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(g, 0.0)
        if g >= gap_us:
            print(f"  gap {g:8.1f} us before {n[:60]} at +{(s-t0)/1e3:.1f} us")
        prev_end = max(prev_end, e)
    print(f"{'kernel':64s} {'calls':>6s} {'busy_us':>10s} {'gap_before_us':>14s}")
    for n, (c, b, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:64]:64s} {c:6d} {b:10.1f} {g:14.1f}")


if __name__ == "__main__":
    gap = 3.0
    if "--gap" in sys.argv:
        gap = float(sys.argv[sys.argv.index("--gap") + 1])
    back = int(sys.argv[sys.argv.index("--back") + 1]) if "--back" in sys.argv else 0
    main(sys.argv[1], gap, back)
