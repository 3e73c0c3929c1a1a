#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (SQLite) outputs: per-kernel time stats and per-kernel
mean PMC counter values.  Usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def short(name, n=70):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def summarize(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f"## {path}")
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>11s} {'min_us':>11s} {'max_us':>11s} {'%':>6s}")
    for name, c, tot, avg, mn, mx in rows[:25]:
        print(f"{short(name):72s} {c:6d} {tot / 1e6:10.3f} {avg / 1e3:11.2f} {mn / 1e3:11.2f} {mx / 1e3:11.2f} {100 * tot / total:6.2f}")
    try:
        pm = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                         "group by kernel_name, counter_name").fetchall()
    except sqlite3.Error as e:
        pm = []
        print("no counters:", e)
    if pm:
        print(f"\n{'kernel':72s} {'counter':28s} {'n':>5s} {'mean/dispatch':>16s}")
        keep = {r[0] for r in rows[:4]}
        for name, ctr, c, avg, tot in sorted(pm):
            if name in keep and name.startswith('void am::') or name.startswith('am::') and name in keep:
                print(f"{short(name):72s} {ctr:28s} {c:5d} {avg:16.1f}")
    print()


if __name__ == "__main__":
    for p in sys.argv[1:]:
        summarize(p)
