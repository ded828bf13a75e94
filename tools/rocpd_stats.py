#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd .db file.

rocprofv3 on ROCm 7.2 writes a SQLite "rocpd" database by default; this prints the same table its
--stats CSV would hold, so the summary can be committed under profiles/.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/r01_xxx_kernel_stats.txt]
"""
import re
import sqlite3
import sys


def main(path: str) -> None:
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"{'kernel':70s} {'calls':>8s} {'total_ms':>11s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, n, tot, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)
        print(f"{short[:70]:70s} {n:8d} {tot/1e6:11.3f} {avg/1e3:11.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
