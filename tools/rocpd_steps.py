#!/usr/bin/env python3
"""Per-launch durations of the factorisation kernels in dispatch order (last factorisation of a chol_test run).
    python tools/rocpd_steps.py x_results.db"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
rows = [((re.search(r"k_[a-z0-9_]+", n) or re.search(r".*", n)).group(0)[:24], s, e, g) for n, s, e, g in rows]
last = max(i for i, r in enumerate(rows) if "bsolve" in r[0])
first = max(i for i, r in enumerate(rows[:last]) if "potrf_diag" in r[0])       # a factorisation opens with k_potrf_diag
seg = rows[first:last + 1]
prev = seg[0][1]
out = []
for n, s, e, g in seg:
    out.append(f"{n[2:10]}[{g // 256}]:{(e-s)/1e3:.1f}(+{(s-prev)/1e3:.1f})"); prev = e
print(" ".join(out))
print("span us", (seg[-1][2] - seg[0][1]) / 1e3)
