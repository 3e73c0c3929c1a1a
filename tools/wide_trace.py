#!/usr/bin/env python3
"""Where does a stage of the wide f16 filter kernels spend its cycles?  (A/B build only: AM_HIP_LIBRARY=dev.)
Runs one membership-filter call (AB_WHICH=cross) or one k-NN sweep (AB_WHICH=knn) with AM_WIDE_TRACE=1 and averages the
s_memtime stamps of waves 0 / 4 of 64 workgroups (AM_WIDE_TRACE_B0 = the first of them; the k-NN sweep runs its windows in
descending order, so a later block offset shows the epilogues under tighter bounds):
  0 stage start  1 after 8 MFMA + 4 DMA pieces  2 after 16 MFMA + 8 pieces  3 after 32 MFMA  4 after the epilogue
  5 after s_waitcnt vmcnt(0)   (next 0: after the barrier)"""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ["AM_HIP_LIBRARY"] = "dev"
os.environ["AM_WIDE_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
from audio_metrics_amd import _lib, hip_ops as ops  # noqa: E402

n, d, k = (int(os.environ.get(key, dflt)) for key, dflt in (("AB_ROWS", "100000"), ("AB_DIM", "512"), ("AB_K", "5")))
which = os.environ.get("AB_WHICH", "cross")
ref, cand = (torch.as_tensor(a).cuda() for a in gi.bench_pair("randn", n, d))
r_ref, r_cand = ops.knn_radii(ref, k), ops.knn_radii(cand, k)
if which == "cross":
    ops.prdc_counts(ref, cand, r_ref, r_cand)
torch.cuda.synchronize()
lib = _lib.load()
lib.am_wide_trace_read.restype = ctypes.c_int
buf = np.zeros(64 * 2 * 96 * 6, dtype=np.uint64)
assert lib.am_wide_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size)) == 0
tr = buf.reshape(64, 2, 96, 6).astype(np.int64)
ok = (tr[:, :, :, 0] > 0) & (tr[:, :, :, 5] > 0)
for wv in (0, 1):
    t = tr[:, wv]
    valid = ok[:, wv]
    valid[:, -1] = False
    nxt = np.roll(t[:, :, 0], -1, axis=1)
    seg = [t[:, :, i + 1] - t[:, :, i] for i in range(5)] + [nxt - t[:, :, 5]]
    names = ["mm c0 + 4 pieces", "mm c1 + 4 pieces", "mm c2, c3", "epilogue", "wait vmcnt(0)", "barrier"]
    for label, sel in (("stages without epilogue", valid & (np.arange(96)[None, :] % 8 != 7)), ("last stage of a tile", valid & (np.arange(96)[None, :] % 8 == 7))):
        if sel.sum() == 0:
            continue
        parts = [float(s[sel].mean()) for s in seg]
        print(f"wave {wv * 4} {label:26s} total {sum(parts):7.0f} cycles: " + "  ".join(f"{nm} {p:6.0f}" for nm, p in zip(names, parts)))
        if "last" in label:                                    # the tile epilogue: this wave's own time, and until the barrier lets go
            own, held = seg[3][sel], (seg[3] + seg[4] + seg[5])[sel]
            q = lambda a: " / ".join(f"{int(np.percentile(a, p))}" for p in (10, 50, 90, 100))
            print(f"       epilogue p10/p50/p90/max {q(own)}   epilogue + wait + barrier {q(held)}   ({int(sel.sum())} tiles, b0={os.environ.get('AM_WIDE_TRACE_B0', '0')})")
