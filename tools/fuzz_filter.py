#!/usr/bin/env python3
"""Randomised parity fuzz of the f16 filter + exact verification paths against the exact f32 kernels (GPU, subprocesses:
the library reads its path knobs once per process).  Usage: tools/fuzz_filter.py [n_cases] [seed]"""
import os
import random
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    rows = rnd.choice([int(v) for v in os.environ["FUZZ_ROWS"].split(",")] if os.environ.get("FUZZ_ROWS") else
                      [6200, 8200, 9001, 12000, 20000, 33000, 34567, 40000, 47001, 52000, 65536, 81000, 100000])
    dim = rnd.choice([int(v) for v in os.environ["FUZZ_DIMS"].split(",")] if os.environ.get("FUZZ_DIMS") else
                     [64, 67, 96, 128, 130, 200, 256, 257, 384, 512])
    k = rnd.choice([1, 3, 5, 8, 10])
    data = rnd.choice(os.environ.get("FUZZ_DATA", "randn,clustered,scales,lowrank,unit,dups,silence,hub").split(","))
    rows2 = rnd.choice([rows, rows, max(600, rows // 7), min(100000, rows * 2), 3001])      # candidate rows (membership only)
    base = dict(os.environ, AB_ROWS=str(rows), AB_ROWS2=str(rows2), AB_DIM=str(dim), AB_K=str(k), AB_REPS="1", AB_DATA=data,
                AB_SEED=str(rnd.randrange(1000)), AB_WANT_MIN=str(rnd.randrange(2)))
    line = f"case {case}: rows={rows}/{rows2} dim={dim} k={k} data={data} seed={base['AB_SEED']} want_min={base['AB_WANT_MIN']}"
    for tool, pattern, off in (("ab_knn.py", r"radii sha1 ([0-9a-f]+)", {"AM_KNN_FAST": "0", "AM_HIP_LIBRARY": "dev"}),
                               ("ab_cross.py", r"sha1 ([0-9a-f]+)", {"AM_PRDC_FAST": "0", "AM_KNN_FAST": "0", "AM_HIP_LIBRARY": "dev"})):
        outs = []
        for extra in (off, {}):
            res = subprocess.run([sys.executable, os.path.join(root, "tools", tool)], env=dict(base, **extra),
                                 capture_output=True, text=True, timeout=900)
            m = re.search(pattern, res.stdout)
            outs.append(m.group(1) if m else "ERROR:" + (res.stderr or res.stdout)[-200:])
        ok = outs[0] == outs[1] and not outs[0].startswith("ERROR")
        bad += not ok
        line += f" | {tool}: {'ok' if ok else 'MISMATCH ' + str(outs)}"
    print(line, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
