#!/usr/bin/env python3
"""Where is the GPU idle inside a bench step?  Reads a rocprofv3 rocpd database (kernel trace of `bench.py --steps K`),
splits the kernel timeline into steps (a step starts with the first colsum/stats kernel after a knn/cross kernel has
been seen), and prints per step: span, union of busy time over all streams, idle time, and the idle gaps grouped by the
(kernel before, kernel after) pair.  Usage: step_timeline.py <results.db> [first_kernel_substring]"""
import collections
import sqlite3
import sys


def short(name):
    return name.replace("void ", "").replace("am::", "").split("(")[0].split("<")[0][-44:]


def main():
    con = sqlite3.connect(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else "colsum_partial"
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    steps, cur, seen_big = [], [], False
    for name, s, e in rows:
        if first in name and seen_big:
            steps.append(cur)
            cur, seen_big = [], False
        if "cross_wide" in name or "prdc_cross" in name or "cross_fast" in name:
            seen_big = True
        cur.append((name, s, e))
    steps.append(cur)
    for i, st in enumerate(steps):
        if len(st) < 10:
            continue
        t0, t1 = st[0][1], max(e for _, _, e in st)
        busy, idle = 0, collections.Counter()
        horizon, last = st[0][1], st[0][0]
        for name, s, e in st:
            if s > horizon:
                idle[(short(last), short(name))] += s - horizon
            if e > horizon:
                busy += e - max(s, horizon)
                horizon, last = e, name
        total_idle = sum(idle.values())
        print(f"step {i}: {len(st)} kernels, span {(t1 - t0) / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, idle {total_idle / 1e6:.3f} ms")
        for (a, b), v in idle.most_common(14):
            print(f"    {v / 1e3:8.1f} us   {a} -> {b}")
    # gap between steps = host work after the last D2H of one step until the first kernel of the next
    for a, b in zip(steps, steps[1:]):
        if len(a) >= 10 and len(b) >= 10:
            print(f"between steps: {(b[0][1] - max(e for _, _, e in a)) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
