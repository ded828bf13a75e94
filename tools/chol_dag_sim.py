#!/usr/bin/env python3
"""Discrete-event model of the tiled dense Cholesky (chol_kernels.hip) as a task graph -- host only, no GPU.

What it is for: the right-looking schedule pays max(chain, bulk) PER COLUMN; before building a
persistent launch in which the chain runs ahead of the bulk, price it.  Task costs are the round-4 measurements
(profiles/r04_chol_links.txt, r04_bench_kernel_stats.txt):

    potrf(k)         19.2 us on the chain workgroup (+ 1.1 us tile -> LDS, + 1.6 us factor -> memory / flag)
    strip(i,k,s)     one 16-row strip of tile (i,k): 6.5 us from cold operands; PHASED against potrf(k) it may start once block
                     columns 0-3 are published (9 us into the tile) and ends no earlier than 3.8 us after the tile is factored
    update(i,j,k,h)  half tile (128 x 64) by one four-wavefront team: 21.5 us per round of 512 slots (two teams per compute unit)
    diag(k+1,p)      the split update of the next diagonal tile: 9 tasks of 3.5 us, + 1.1 us gather by the chain workgroup
    hand-off         a dependent task on another compute unit sees its predecessor 1.5 us after it ended (write-through store
                     + flag + poll: MI355X_MICROARCH.md price list, handoff-flag under load)

Model 1 ("today") replays the 47-column right-looking schedule launch by launch with the same costs and must reproduce the
measured 2.47 ms; model 2 is list scheduling of the whole graph on 512 team slots with a priority rule and a look-ahead limit.

    python tools/chol_dag_sim.py [--nt 47] [--table]
"""
import argparse
import heapq
import json
import sys

P_POTRF = 19.2      # in-tile factorisation
P_LOAD = 1.1        # tile -> LDS
P_STORE = 1.6       # factor / flag -> memory
P_STRIP = 6.5       # one strip, cold
P_STRIP_TAIL = 3.8  # phased strip: ends this long after the tile is factored
P_STRIP_EARLY = 9.0 # phased strip: may start this long after potrf(k) started (block columns 0-3 published)
P_HALF = 21.5       # half-tile update, per round of all slots
P_DIAG = 3.5        # one of the 9 split tasks of the next diagonal tile
P_GATHER = 1.1
P_HOP = 1.5         # visibility of a predecessor on another compute unit
P_BOUNDARY = 3.5    # kernel boundary of today's schedule (launch end -> next launch's workgroup 0)
P_RAMP = 5.0        # per-launch ramp of today's update-bound launches
P_PANEL = 10.4      # today's separate panel-solve launch
P_BSOLVE = 151.0    # backward substitution (persistent launch of its own, unchanged)
P_FIRST = 24.0      # k_potrf_diag of column 0


def today(nt, slots=512):
    """The round-4 schedule, launch by launch: 19 update-bound columns (half-tile kernel + panel-solve launch), then chain-bound columns."""
    chain = P_DIAG + P_GATHER + P_LOAD + P_POTRF + P_STORE + P_STRIP_TAIL + P_BOUNDARY      # one chain-bound column
    t = P_FIRST + P_PANEL
    rows = []
    for k in range(nt - 1):
        m = nt - 1 - k
        tiles = m * (m + 1) // 2
        if tiles >= 400:
            rounds = 2.0 * (tiles - 1) / slots
            full, part = int(rounds), rounds - int(rounds)
            # a last round that fills at most half of the slots goes in quarters: half as long
            bulk = P_RAMP + full * P_HALF + (0 if part == 0 else (P_HALF * 0.5 if part <= 0.5 else P_HALF))
            dur = max(bulk, P_DIAG + P_GATHER + P_LOAD + P_POTRF + P_STORE) + P_PANEL
            kind = "bulk"
        else:
            # quarter tiles, one workgroup per unit: the update itself
            upd = P_RAMP + 4.0 * (tiles - 1) / (slots // 2) * (P_HALF * 0.5 * 0.55)
            dur = max(chain, upd)
            kind = "chain"
        rows.append((k, kind, dur))
        t += dur
    return t + P_BSOLVE, rows


class Sim:
    """List scheduling of the task graph: `slots` worker teams + one chain workgroup; ready tasks served by priority."""

    def __init__(self, nt, slots=510, lookahead=None, priority="column", chain_speedup=0.0, half=P_HALF, reserve=0, fuse=1, wj=1.0, wk=0.0, hop=P_HOP, split=1, split_by="panel"):
        self.nt, self.slots, self.lookahead, self.priority = nt, slots - 2 * reserve, lookahead, priority
        self.potrf = P_POTRF - chain_speedup
        self.half = half
        self.fuse = fuse
        self.wj, self.wk, self.hop, self.split = wj, wk, hop, split
        self.split_by = split_by

    def nsplit(self, i, j, k):
        """Team tasks per tile update: halves while there are tiles enough to fill the machine, quarters / eighths behind (rule: self.split)."""
        m = self.nt - 1 - k if self.split_by == "panel" else self.nt - j
        tiles = m * (m + 1) // 2
        if self.split == 1:
            return 2
        if tiles * 2 >= self.split:
            return 2
        if tiles * 4 >= self.split:
            return 4
        return 8

    def run(self):
        nt = self.nt
        # state
        upd_done = {}            # (i, j) -> [count of finished half tasks per panel k]
        upd_time = {}            # (i, j, k) -> time both halves finished
        strip_cnt, strip_time = {}, {}
        potrf_start, potrf_end = [None] * nt, [None] * nt
        diag_cnt, diag_time = [0] * nt, [0.0] * nt
        panel_left = [0] * nt    # unfinished update half-tasks of panel k (for the look-ahead limit)
        for k in range(nt - 1):
            m = nt - 1 - k
            panel_left[k] = sum(self.nsplit(i, j, k) for i in range(k + 1, nt) for j in range(k + 1, i + 1)) - self.nsplit(k + 1, k + 1, k)
        ready = []               # heap of (priority key, seq, task)
        events = []              # heap of (time, seq, kind, payload)
        seq = [0]
        free = self.slots
        now = 0.0
        busy_area = 0.0
        trace = []

        def push_ready(task, t_ready):
            seq[0] += 1
            heapq.heappush(events, (t_ready, seq[0], "ready", task))

        def prio(task):
            kind = task[0]
            if self.priority == "fifo":
                return (0, seq[0])
            # earliest deadline first: tile (i, j) must have absorbed panel k by the time the chain reaches column j, less the
            # j - 1 - k sequential updates still ahead of it: deadline ~ wj * j + wk * k
            if kind == "D":
                return (self.wj * task[1] + self.wk * (task[1] - 1) - 2e6, -1, 0)      # the chain itself
            if kind == "S":
                _, i, k, s = task
                return ((self.wj + self.wk) * k - 1e6, 0, i)                             # strips of column k: before any update
            _, i, j, k, h = task
            return (self.wj * j + self.wk * k, 1, i, k)

        def release_strips(i, k):
            # tile (i, k) has received all its k updates: its strips may run (phased against potrf(k))
            t0 = max(upd_time.get((i, k, k - 1), 0.0) + (P_HOP if k > 0 else 0.0), 0.0)
            for s in range(8):
                push_ready(("S", i, k, s), t0)

        def maybe_release_update(i, j, k, t):
            # update(i, j, k) needs strips (i, k), (j, k) complete and update(i, j, k - 1) complete
            if strip_cnt.get((i, k), 0) < 8 or strip_cnt.get((j, k), 0) < 8:
                return
            if k > 0 and (i, j, k - 1) not in upd_time:
                return
            if (i, j, k, "released") in upd_time:
                return
            upd_time[(i, j, k, "released")] = t
            t0 = max(strip_time[(i, k)], strip_time[(j, k)], upd_time.get((i, j, k - 1), 0.0)) + P_HOP
            if i == j and j == k + 1:
                for p in range(9):
                    push_ready(("D", j, p), t0)
            else:
                for h in range(self.nsplit(i, j, k)):
                    push_ready(("U", i, j, k, h), t0)

        def start_potrf(k, t):
            potrf_start[k] = t + P_LOAD
            potrf_end[k] = potrf_start[k] + self.potrf
            seq[0] += 1
            heapq.heappush(events, (potrf_end[k] + P_STORE, seq[0], "potrf_done", k))
            # phased strips of column k wake up
            seq[0] += 1
            heapq.heappush(events, (potrf_start[k] + P_STRIP_EARLY, seq[0], "potrf_early", k))

        potrf_early = [False] * nt
        waiting_strips = {}      # k -> list of strip tasks ready but potrf(k) not yet at its early mark
        chain_wait = {}          # look-ahead: potrf(k) deferred until panel k-1-L is drained

        start_potrf(0, 0.0)
        for i in range(1, nt):
            release_strips(i, 0)
        finish = 0.0
        while events or ready:
            # start as many ready tasks as there are free slots at `now`
            while ready and free > 0:
                _, _, task = heapq.heappop(ready)
                kind = task[0]
                if kind == "S":
                    _, i, k, s = task
                    dur = max(P_STRIP, potrf_end[k] + P_STRIP_TAIL - now)
                elif kind == "D":
                    dur = P_DIAG
                else:
                    ns = self.nsplit(task[1], task[2], task[3])
                    dur = self.half * 2.0 / ns * (1.0 if ns <= 4 else 1.25)
                free -= 1
                busy_area += dur
                seq[0] += 1
                heapq.heappush(events, (now + dur, seq[0], "done", task))
            if not events:
                break
            t, _, ev, payload = heapq.heappop(events)
            now = t
            if ev == "ready":
                task = payload
                if task[0] == "S" and not potrf_early[task[2]]:
                    waiting_strips.setdefault(task[2], []).append(task)
                else:
                    heapq.heappush(ready, (prio(task), seq[0], task))
            elif ev == "potrf_early":
                k = payload
                potrf_early[k] = True
                for task in waiting_strips.pop(k, []):
                    seq[0] += 1
                    heapq.heappush(ready, (prio(task), seq[0], task))
            elif ev == "potrf_done":
                finish = max(finish, now)
            elif ev == "done":
                free += 1
                task = payload
                kind = task[0]
                if kind == "S":
                    _, i, k, s = task
                    strip_cnt[(i, k)] = strip_cnt.get((i, k), 0) + 1
                    if strip_cnt[(i, k)] == 8:
                        strip_time[(i, k)] = now
                        # updates by panel k touching row / column i
                        for j in range(k + 1, i + 1):
                            maybe_release_update(i, j, k, now)
                        for i2 in range(i, nt):
                            maybe_release_update(i2, i, k, now)
                elif kind == "D":
                    _, j, p = task
                    diag_cnt[j] += 1
                    if diag_cnt[j] == 9:
                        diag_time[j] = now
                        upd_time[(j, j, j - 1)] = now
                        t0 = now + P_HOP + P_GATHER
                        L = self.lookahead
                        if L is not None and j - 1 - L >= 0 and panel_left[j - 1 - L] > 0:
                            chain_wait[j - 1 - L] = (j, t0)
                        else:
                            start_potrf(j, max(t0, potrf_end[j - 1] + P_STORE))
                else:
                    _, i, j, k, h = task
                    c = upd_done.get((i, j, k), 0) + 1
                    upd_done[(i, j, k)] = c
                    panel_left[k] -= 1
                    if panel_left[k] == 0 and k in chain_wait:
                        jj, t0 = chain_wait.pop(k)
                        start_potrf(jj, max(t0, now, potrf_end[jj - 1] + P_STORE))
                    if c == self.nsplit(i, j, k):
                        upd_time[(i, j, k)] = now
                        if j == k + 1:
                            # tile (i, j) is complete: its strips
                            release_strips(i, j)
                        else:
                            maybe_release_update(i, j, k + 1, now)
        makespan = max(finish, now)
        return {"makespan_us": makespan, "busy_frac": busy_area / (makespan * self.slots),
                "potrf_end": potrf_end, "chain_us": sum(1 for _ in potrf_end) * 0.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=47)
    ap.add_argument("--table", action="store_true")
    a = ap.parse_args()
    t_today, rows = today(a.nt)
    print(f"model 1, today's schedule (nt = {a.nt}): {t_today / 1000:.3f} ms   (measured: 2.476 ms at nt = 47)")
    if a.table:
        for k, kind, dur in rows:
            print(f"   column {k:2d} {kind:5s} {dur:6.1f} us")
    out = {"today_ms": t_today / 1000, "variants": []}

    def show(label, **kw):
        r = Sim(a.nt, **kw).run()
        total = (r["makespan_us"] + P_BSOLVE + P_BOUNDARY) / 1000
        frac = (a.nt * 128 - 28) ** 3 / 3 / (total * 1e-3) / 78.6e12 if a.nt == 47 else float("nan")
        print(f"   {label:58s} factor {r['makespan_us'] / 1000:.3f} ms  + solve = {total:.3f} ms   slots busy {r['busy_frac']:.2f}   frac {frac:.3f}")
        out["variants"].append({"label": label, "factor_ms": r["makespan_us"] / 1000, "total_ms": total, "slots_busy": r["busy_frac"], "frac": frac})
        return r

    print("model 2, ONE persistent launch, 510 worker teams + the chain workgroup:")
    show("(i) unlimited look-ahead, column priority")
    show("    unlimited look-ahead, FIFO (release order)", priority="fifo")
    for L in (0, 1, 2, 3, 4, 8):
        show(f"(ii) look-ahead depth {L}", lookahead=L)
    for n in (4, 8, 16, 32):
        show(f"(iii) {n} compute units reserved (idle for the bulk)", reserve=n)
    print("   sensitivity:")
    show("chain 5 us shorter per column (strip + diagonal update fused into the chain WG)", chain_speedup=5.0)
    show("chain 8 us shorter per column", chain_speedup=8.0)
    show("half-tile task 19 us (C kept over two panels)", half=19.0)
    show("both", chain_speedup=8.0, half=19.0)
    json.dump(out, open("/dev/stdout", "w") if False else sys.stderr, indent=None) if False else None
    return out


if __name__ == "__main__":
    main()
