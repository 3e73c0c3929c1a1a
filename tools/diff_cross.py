#!/usr/bin/env python3
"""Development aid: run am_prdc_counts_f32 with the filter-and-verify path on and off (separate processes, the
switch is read once per process) and report every difference in the three outputs."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from audio_metrics_amd import hip_ops as ops
n, m, d, k = (int(os.environ.get(v, dflt)) for v, dflt in (("AB_ROWS", "100000"), ("AB_COLS", "0"), ("AB_DIM", "512"), ("AB_K", "5")))
m = m or n
gen = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, d, generator=gen, device="cuda")
y = torch.randn(m, d, generator=gen, device="cuda") * 1.05 + 0.05
kind = os.environ.get("AB_KIND", "randn")
if kind == "unit":
    x = (x + 0.5) / (x + 0.5).norm(dim=1, keepdim=True); y = (y + 0.55) / (y + 0.55).norm(dim=1, keepdim=True)
elif kind == "dup":
    x[n // 2:] = x[: n - n // 2]; y[: m // 3] = x[: m // 3]
elif kind == "self":
    y = x.clone()
rx, ry = ops.knn_radii(x, k), ops.knn_radii(y, k)
col, rany, rcov, rmin = ops.prdc_counts(x, y, rx, ry, True)
col2, rany2, rcov2 = ops.prdc_counts(x, y, rx, ry)
torch.cuda.synchronize()
np.savez(sys.argv[1], col=col.cpu().numpy(), rany=rany.cpu().numpy(), rcov=rcov.cpu().numpy(), rmin=rmin.cpu().numpy(),
         col2=col2.cpu().numpy(), rany2=rany2.cpu().numpy(), rcov2=rcov2.cpu().numpy())
''' % ROOT

out = {}
for fast in ("0", "1"):
    fp = f"/tmp/diff_cross_{fast}.npz"
    r = subprocess.run([sys.executable, "-c", CHILD, fp], env=dict(os.environ, AM_PRDC_FAST=fast, AM_HIP_LIBRARY="dev"), capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stdout, r.stderr)
        sys.exit(1)
    sys.stderr.write("".join(l + "\n" for l in r.stderr.splitlines() if "cross_fast" in l))
    out[fast] = np.load(fp)
a, b = out["0"], out["1"]
bad = 0
for key in ("col", "rany", "rcov", "rmin", "col2", "rany2", "rcov2"):
    diff = np.flatnonzero(a[key] != b[key])
    bad += len(diff)
    print(f"{key}: {len(diff)} differences", diff[:8], a[key][diff[:8]], b[key][diff[:8]])
for k3 in ("col", "rany", "rcov"):                     # with / without the optional row minimum: same flags and counts
    if not np.array_equal(a[k3], a[k3 + "2"]) or not np.array_equal(b[k3], b[k3 + "2"]):
        bad += 1
        print("want_min on/off differ in", k3)
print("sums exact", int(a["col"].sum()), int(a["rany"].sum()), "fast", int(b["col"].sum()), int(b["rany"].sum()))
print("IDENTICAL" if bad == 0 else "MISMATCH")
sys.exit(0 if bad == 0 else 1)
