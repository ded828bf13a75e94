#!/usr/bin/env python3
"""HBM traffic of the HBM-bound stages of one LM iteration (linearise | Schur build | back-substitution + trial error) from two
rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over `python bench.py --no-cpu-baseline`.

Per kernel: calls, FETCH_SIZE and WRITE_SIZE per call, bytes per call = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte requests at 64 bytes), the same one
tools/pmc_to_traffic.py calibrates on the factorisation run's own copies.  Per stage: the sum over its kernels of bytes per call x
calls per LM iteration (every stage kernel runs once per iteration with one trial; the bench seeds lambda so that it does).
Usage: pmc_stage_traffic.py <fetch.db> <write.db> > profiles/ba_stage_traffic.json
"""
import json
import re
import sqlite3
import sys

STAGES = {
    "linearize": ["k_small_linearize", "k_linearize_lm", "k_linearize_cam", "k_error"],       # k_error: only its current-estimate launches (separate path)
    "schur_build": ["k_zero_lower", "k_zero_skyline", "k_lm_invert", "k_schur_prepare", "k_schur_block", "k_schur_block_compact", "k_schur_stream", "k_schur_rhs"],
    "backsubst_and_trial_error": ["k_backsub", "k_pose_update", "k_error"],
}


def sums(path):
    con = sqlite3.connect(path); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))      # noqa: E731
    pe, kd, ks = t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    rows = cur.execute(f"select s.kernel_name, count(distinct d.id), sum(e.value) from {pe} e join {kd} d on e.event_id = d.event_id "
                       f"join {ks} s on d.kernel_id = s.id group by s.kernel_name").fetchall()
    out = {}
    for name, n, v in rows:
        m = re.search(r"\d+(k_[a-z0-9_]+?)(?:I[Lb]|E)", name)
        short = m.group(1) if m else name
        c = out.setdefault(short, [0, 0.0])
        c[0] += n; c[1] += v
    return out


def main():
    f, w = sums(sys.argv[1]), sums(sys.argv[2])
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        calls = f.get(k, w.get(k))[0]
        fk, wk = f.get(k, [0, 0.0])[1] / calls, w.get(k, [0, 0.0])[1] / calls
        kernels[k] = {"calls": calls, "fetch_size_kb_raw_per_call": fk, "write_size_kb_per_call": wk, "hbm_bytes_per_call": (2 * fk + wk) * 1024}
    n_iter = kernels.get("k_backsub", kernels.get("k_schur_block", {"calls": 1}))["calls"]
    stages = {}
    for st, names in STAGES.items():
        b = 0.0
        for k in names:
            if k not in kernels:
                continue
            share = 1.0
            if k == "k_error":      # fused linearisation (k_small_linearize): every k_error launch evaluates a trial; separate path: every other one
                share = (1.0 if st == "backsubst_and_trial_error" else 0.0) if "k_small_linearize" in kernels else 0.5
            b += share * kernels[k]["hbm_bytes_per_call"] * kernels[k]["calls"] / n_iter
        stages[st] = b
    print(json.dumps({"lm_iterations_in_run": n_iter, "stages_hbm_bytes_per_iteration": stages, "total_hbm_bytes_per_iteration": sum(stages.values()),
                      "kernels": kernels,
                      "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts 128-B requests as 64 B; calibrated in chol_traffic.json)",
                      "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python bench.py --no-cpu-baseline"}, indent=1))


if __name__ == "__main__":
    main()
