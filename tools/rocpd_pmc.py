#!/usr/bin/env python3
"""Per-kernel sums of a PMC counter from a rocprofv3 rocpd .db (one --pmc pass per database)."""
import re
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    pe, pi, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    rows = cur.execute(f"select s.kernel_name, i.name, count(*), sum(e.value) from {pe} e join {pi} i on e.pmc_id = i.id "
                       f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.name order by 4 desc").fetchall()
    print(f"# source: {path}")
    print(f"{'kernel':64s} {'counter':12s} {'dispatches':>10s} {'sum':>16s} {'per_dispatch':>14s}")
    for name, ctr, n, v in rows:
        short = re.sub(r"\(.*", "", name)
        print(f"{short[:64]:64s} {ctr:12s} {n:10d} {v:16.1f} {v/n:14.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
