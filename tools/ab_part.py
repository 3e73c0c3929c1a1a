#!/usr/bin/env python3
"""Time one rank's share of the partitioned symmetric k-NN at the BASELINE size (single GPU emulation).
AB_DUPS=<rows>: that many rows are one and the same vector (AB_UNIT=1: unit-norm rows; else a low-norm vector among randn rows,
every row's nearest neighbour)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops  # noqa: E402

n, d, k = 100000, 512, 5
gen = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, d, device="cuda", generator=gen)
dups, unit = int(os.environ.get("AB_DUPS", "0")), os.environ.get("AB_UNIT", "0") == "1"
if unit:
    x += 0.5
if dups:
    x[torch.randperm(n, generator=gen, device="cuda")[:dups]] = x[0] * (1.0 if unit else 0.1)
if unit:
    x /= x.norm(dim=1, keepdim=True)


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


for g in [int(v) for v in os.environ.get("AB_WORLDS", "2,4,8").split(",")]:
    rows = n // g
    tb = t(lambda: ops.knn_bounds(x, k, 0, rows))
    bounds = torch.cat([ops.knn_bounds(x, k, p * rows, rows) for p in range(g)])
    prep = ops.prepare(x) if os.environ.get("AB_PREPARED", "0") == "1" else None
    extra = {} if prep is None else {"prepared": prep}
    tp = t(lambda: ops.knn_sym_part(x, k, int(os.environ.get("AB_PART", "0")), g, bounds, **extra))
    if os.environ.get("AB_ONLY_PART", "0") == "1":                 # nothing but this rank's share, with the library's kernel clock
        ops.kernel_clock_enable(True)
        ops.filter_stats_enable("cuda:0", True)
        line = []
        for part in sorted({0, g // 2, g - 1}):
            ops.knn_sym_part(x, k, part, g, bounds, **extra)
            torch.cuda.synchronize()
            for kid in (0, 2):
                ops.kernel_clock_read(kid)
            ops.filter_stats_read("cuda:0")
            tpp = t(lambda: ops.knn_sym_part(x, k, part, g, bounds, **extra))
            (c0, sweep), (c2, verify) = ops.kernel_clock_read(0), ops.kernel_clock_read(2)
            st = ops.filter_stats_read("cuda:0")
            line.append(f"part {part}: {tpp:.2f} ms (sweep {sweep / max(c0, 1):.2f}, verify {verify / max(c2, 1):.2f}, queued {st['knn_queued'] / max(c0, 1):.0f})")
        print(f"world={g}: " + "  ".join(line))
        ops.kernel_clock_enable(False)
        continue
    lists = torch.stack([ops.knn_sym_part(x, k, p, g, bounds) for p in range(g)])
    tf = t(lambda: ops.knn_lists_finish(lists, x, k))
    tg = t(lambda: ops.knn_radii(x[:rows], k, columns=x))
    flagged = int(torch.isnan(lists[:, :, 0]).any(dim=0).sum())
    assert torch.equal(ops.knn_lists_finish(lists, x, k), ops.knn_radii(x, k))
    print(f"duplicates {dups} ({'unit-norm' if unit else 'randn'}) flagged rows {flagged} world={g}: bounds {tb:.2f} ms + part {tp:.2f} ms + finish {tf:.2f} ms = {tb + tp + tf:.2f} ms   (general shard kernel {tg:.2f} ms)")
