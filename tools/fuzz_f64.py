#!/usr/bin/env python3
"""Randomised check of the float64 filter routes (csrc/pairwise_fast.h: knn_fast_select64_kernel, cross_verify_regions64_kernel)
against the general f64 kernels, on the data families of the float32 fuzzers (tools/route_probe.py: randn, unit, clustered,
scales, lowrank, dups, silence, hub, sparse, tiny, huge) with a float64 perturbation on top, so that no row is exactly
representable in float32.  Radii: within the rounding of two summation orders; membership counts and flags: exactly equal.
Usage: tools/fuzz_f64.py [n_cases] [seed]"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import route_probe  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402



def run_case(fam, rows, rows2, dim, k, seed, factor=1.0):
    """One case: (radii ok, counts ok, text).  factor: both sets multiplied by it AFTER the float64 perturbation (ADVICE r5:
    1e-30 ... 1e30 - past float32's range the filter route must hand the call to the general kernels, not lose memberships)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = route_probe.make(fam, rows, dim, seed).double()
    y = route_probe.make(fam, rows2, dim, seed + 100).double()
    x = x * (1.0 + 1e-9 * torch.randn(x.shape, generator=g, device="cuda", dtype=torch.float64)) * factor
    y = y * (1.0 + 1e-9 * torch.randn(y.shape, generator=g, device="cuda", dtype=torch.float64)) * factor
    ops.filter_stats_read("cuda:0")
    r = ops.knn_radii(x, k)
    s_knn = ops.filter_stats_read("cuda:0")
    # the widened error bound of the float64 filter route, measured on every pair its selection evaluated (am_filter_stats
    # slot 9, round 6): |f16 value of the float32-rounded rows - float64 value| / ((fast_c + 2^-19) (|x|^2 + G)) must stay <= 1
    ratio = s_knn["bound_ratio_max"] if s_knn.get("bound_pairs", 0) > 0 else None
    general = ops.knn_radii(x, k, columns=x.clone())
    scale = float(torch.linalg.norm(x, dim=1).max())
    err = (r - general).abs()
    bound = 1e-12 * scale + 4e-16 * scale * scale / general.clamp_min(1e-300)
    ok_r = bool((err <= bound).all()) or bool(torch.isinf(general).all())
    ok_ratio = ratio is None or 0.0 <= ratio <= 1.0
    noisy = 0
    if not ok_r:
        # Near-duplicate rows: the general kernel (like torch.cdist's matmul form in the reference) evaluates |x|^2 + |y|^2 - 2<x, y>,
        # whose rounding noise (~D 2^-53 |x|^2) swamps a squared distance of 1e-18 |x|^2; the filter route sums squared differences.
        # Rows outside the bound are recomputed here from differences, one row at a time: the filter route must agree with THAT.
        rows_bad = torch.nonzero(err > bound).flatten()[:64]
        ok_r = True
        for i in rows_bad.tolist():
            d = (x - x[i]).square().sum(1).sqrt()
            want = torch.kthvalue(d, k + 1).values
            ok_r = ok_r and bool((r[i] - want).abs() <= 1e-9 * want + 1e-300)
        noisy = int(rows_bad.numel())
    ok_r = ok_r and ok_ratio
    r2 = ops.knn_radii(y, k)
    ops.filter_stats_read("cuda:0")
    got = ops.prdc_counts(x, y, r, r2)
    s_cnt = ops.filter_stats_read("cuda:0")
    want = ops.prdc_counts(x, y, r, r2, want_min=True)
    ok_c = all(torch.equal(a, b) for a, b in zip(got, want[:3]))
    recount = 0
    if not ok_c:
        # the general kernel's matmul form cancels catastrophically between near-parallel rows of almost equal length (one-hot rows:
        # (a - b)^2 from a^2 + b^2 - 2ab, relative error ~1e-11), as torch.cdist's does in the reference: where the two routes differ,
        # the columns / rows in question are recounted from squared differences and the FILTER route must equal that
        ok_c = True
        cols = torch.nonzero(got[0] != want[0]).flatten()[:16]
        for j in cols.tolist():
            dist = (x - y[j]).square().sum(1).sqrt()
            ok_c = ok_c and int((dist < r).sum()) == int(got[0][j])
        rows_d = torch.nonzero((got[1] != want[1]) | (got[2] != want[2])).flatten()[:16]
        for i in rows_d.tolist():
            dist = (y - x[i]).square().sum(1).sqrt()
            ok_c = ok_c and bool((dist < r2).any()) == bool(got[1][i]) and bool((dist < r[i]).any()) == bool(got[2][i])
        recount = int(cols.numel() + rows_d.numel())
    line = (f"{fam} rows={rows}/{rows2} dim={dim} k={k} seed={seed} factor={factor:g} | radii {'ok' if ok_r else 'MISMATCH'} "
            f"(filter route {s_knn['knn_calls']}, fallback rows {s_knn['knn_fallback_rows']}, worst err/bound {float((err / bound).max()):.2e}, "
            f"measured |a - t| / filter bound {'-' if ratio is None else format(ratio, '.3f')} over {s_knn.get('bound_pairs', 0)} pairs"
            f"{', ' + str(noisy) + ' near-duplicate rows checked against a difference-form recomputation' if noisy else ''}) | "
            f"counts {'ok' if ok_c else 'MISMATCH'} (filter route {s_cnt['prdc_calls']}, fallback {s_cnt['prdc_fallback_calls']}"
            f"{', ' + str(recount) + ' entries where the general kernel differs recounted from differences' if recount else ''})")
    return ok_r, ok_c, line


FAMILIES = ["randn", "unit", "clustered", "scales", "lowrank", "dups", "silence", "hub", "sparse", "tiny", "huge"]


def draw_case(rnd, rows_choices=(16500, 20000, 24001, 33000, 40000, 52000)):
    fam = rnd.choice(FAMILIES)
    rows = rnd.choice(list(rows_choices))
    rows2 = rnd.choice([rows, rows, max(9000, rows // 2), 16400])
    dim = rnd.choice([8, 16, 33, 64, 67, 96, 128, 200])
    k = rnd.choice([1, 3, 5, 10])
    return fam, rows, rows2, dim, k, rnd.randrange(1000)


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ops.filter_stats_enable("cuda:0", True)
    bad = 0
    for case in range(n_cases):
        ok_r, ok_c, line = run_case(*draw_case(rnd))
        print(f"case {case}: {line}", flush=True)
        bad += (not ok_r) + (not ok_c)
    print(f"mismatches: {bad}")
    sys.exit(1 if bad else 0)
