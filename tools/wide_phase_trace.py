#!/usr/bin/env python3
"""Where does a phase of the wide f16 filter kernels (csrc/wide_phased.h) spend its cycles?
Needs the stamp build of the library (-DAM_DEV_KNOBS -DAM_WIDE_STAMPS, see tools/build_variant.sh) named by
AM_HIP_LIBRARY.  Runs one membership-filter call (AB_WHICH=cross) or one k-NN sweep (AB_WHICH=knn) with
AM_WIDE_TRACE=1 and averages the s_memtime stamps of waves 0 / 4 of the first 64 workgroups over the first 96 phases:
  0 phase start  1 fragment reads + DMA issued  2 after s_waitcnt vmcnt(8)  3 after the first barrier
  4 eight MFMAs issued  5 after the closing barrier (and the epilogue in the last phase of a tile)"""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ.setdefault("AM_HIP_LIBRARY", "libaudio_metrics_hip_trace.so")
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
nk = max(d // 64, 1)
phase = np.arange(96)
if os.environ.get("AB_POINT"):
    # single-stamp builds (-DAM_WIDE_STAMPS=k, k = 1..4): offset of stamp k from the phase start and the phase length
    k = int(os.environ["AB_POINT"])
    what = {1: "reads + DMA issued", 2: "after vmcnt(8)", 3: "after barrier 1", 4: "8 MFMAs issued"}[k]
    for grp in (0, 1):
        s0, sk = tr[:, grp, :, 0], tr[:, grp, :, 1]
        good = (s0[:, :-1] > 0) & (sk[:, :-1] > 0) & (s0[:, 1:] > 0)
        off, length = (sk - s0)[:, :-1], s0[:, 1:] - s0[:, :-1]
        line = []
        for p in range(4):
            sel = good & ((phase[:-1] % 4 == p) & (phase[:-1] // 4 % nk == nk - 2))[None, :]    # second-to-last slab: steady state
            line.append(f"phase {p + 1}: {off[sel].mean():5.0f} of {length[sel].mean():5.0f}")
        print(f"waves {grp * 4}-{grp * 4 + 3} {what:20s} " + "   ".join(line))
    sys.exit(0)
if not (tr[:, :, :, 5] > 0).any():
    # light build (-DAM_WIDE_STAMPS=1): phase starts only.  Waves 0-3 start phase p when the barrier closing their phase
    # p-1 releases (= the first barrier of waves 4-7's phase p-1); waves 4-7 start phase p when the first barrier of waves
    # 0-3's phase p releases.  a[p] -> b[p]: waves 0-3 read / issue, waves 4-7 multiply; b[p] -> a[p+1]: the reverse.
    a, b = tr[:, 0, :, 0], tr[:, 1, :, 0]
    good = (a[:, :-1] > 0) & (b[:, :-1] > 0) & (a[:, 1:] > 0)
    first, second = (b - a)[:, :-1], a[:, 1:] - b[:, :-1]
    for p in range(4):
        for label, tile_end in (("", False), (" (last slab of a tile)", True)):
            sel = good & ((phase[:-1] % 4 == p) & ((phase[:-1] // 4 % nk == nk - 1) == tile_end))[None, :]
            if sel.sum():
                print(f"phase {p + 1}{label:24s}: waves 0-3 read/issue | 4-7 multiply {first[sel].mean():6.0f}   waves 0-3 multiply | 4-7 read/issue "
                      f"{second[sel].mean():6.0f} cycles")
    slab = a[:, 4:] - a[:, :-4]
    sel = (a[:, 4:] > 0) & (a[:, :-4] > 0) & ((phase[:-4] % 4 == 0) & (phase[:-4] // 4 % nk < nk - 1))[None, :]
    print(f"k-slab (4 phases, no epilogue inside): {slab[sel].mean():7.0f} cycles; tile of {nk} slabs: "
          f"{(a[:, 4 * nk:] - a[:, :-4 * nk])[(a[:, 4 * nk:] > 0) & (a[:, :-4 * nk] > 0)].mean():8.0f}")
    sys.exit(0)
ok = (tr[:, :, :, 0] > 0) & (tr[:, :, :, 5] > 0)
names = ["reads + DMA issue", "wait vmcnt(8)", "barrier 1", "8 MFMA", "barrier 2 (+ epilogue)"]
for grp in (0, 1):
    t = tr[:, grp]
    valid = ok[:, grp]
    seg = [t[:, :, i + 1] - t[:, :, i] for i in range(5)]
    phase = np.arange(96)[None, :]
    for label, sel in [(f"phase {p + 1}", valid & (phase % 4 == p) & (phase // 4 % nk != nk - 1)) for p in range(4)] + \
                      [("phase 4, tile end", valid & (phase % 4 == 3) & (phase // 4 % nk == nk - 1))]:
        if sel.sum() == 0:
            continue
        parts = [float(s[sel].mean()) for s in seg]
        print(f"waves {grp * 4}-{grp * 4 + 3} {label:18s} total {sum(parts):7.0f} cycles: " + "  ".join(f"{nm} {p:6.0f}" for nm, p in zip(names, parts)))
    slab = t[:, 4:, 0] - t[:, :-4, 0]
    sel = valid[:, 4:] & valid[:, :-4] & (phase[:, :-4] % 4 == 0) & (phase[:, :-4] // 4 % nk != nk - 1)
    if sel.sum():
        print(f"waves {grp * 4}-{grp * 4 + 3} k-slab (4 phases, no epilogue): {float(slab[sel].mean()):7.0f} cycles")
if os.environ.get("AB_DUMP"):
    wg = int(os.environ["AB_DUMP"])
    base = tr[wg, 0, 8, 0]
    for ph in range(8, 20):
        for grp in (0, 1):
            print(f"wg {wg} phase {ph} waves {grp * 4}-{grp * 4 + 3}: " + " ".join(f"{int(v - base):6d}" for v in tr[wg, grp, ph]))
