#!/usr/bin/env python3
"""GPU idle gaps between consecutive kernels in a rocprofv3 rocpd trace (development aid)."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
gaps = []
busy = 0
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    busy += e0 - s0
    gaps.append((s1 - e0, n0.split("(")[0][-40:], n1.split("(")[0][-40:]))
total = rows[-1][2] - rows[0][1]
print(f"span {total / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {sum(g[0] for g in gaps if g[0] > 0) / 1e6:.2f} ms over {len(rows)} kernels")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"{g[0] / 1e3:9.1f} us   {g[1]:42s} -> {g[2]}")
