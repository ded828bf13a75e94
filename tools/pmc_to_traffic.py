#!/usr/bin/env python3
"""HBM traffic of one dense factorisation from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over tools/_bin/chol_test.

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes, i.e. reports half of a coalesced read stream -- calibrated here on the same run: the two 289.5 MB
device-to-device copies of S report FETCH = 0.5 x and WRITE = 1.0 x their size.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Usage: pmc_to_traffic.py <fetch.db> <write.db> <n_factorizations_in_run> > profiles/chol_traffic.json
"""
import json
import sqlite3
import sys


def sums(path):
    con = sqlite3.connect(path); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))
    pe, kd, ks = t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    rows = cur.execute(f"select s.kernel_name, sum(e.value) from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name").fetchall()
    return {k: v for k, v in rows}


def main():
    f, w, reps = sums(sys.argv[1]), sums(sys.argv[2]), int(sys.argv[3])
    chol = lambda d: sum(v for k, v in d.items() if "mage" in k and "set_scalar" not in k)
    copy_f = sum(v for k, v in f.items() if "copyBuffer" in k); copy_w = sum(v for k, v in w.items() if "copyBuffer" in k)
    fetch_kb, write_kb = chol(f) / reps, chol(w) / reps
    out = {"fetch_size_kb_raw_per_factorization": fetch_kb, "write_size_kb_per_factorization": write_kb,
           "bytes_per_factorization": (2 * fetch_kb + write_kb) * 1024,
           "calibration": {"copy_fetch_kb": copy_f, "copy_write_kb": copy_w, "fetch_to_write_ratio_on_copies": copy_f / copy_w if copy_w else None},
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B; ratio above ~0.5 confirms it on this run)",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/_bin/chol_test 6016", "n_pad": 6016}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
