#!/usr/bin/env python3
"""One GPU emulating rank 0 of a W-rank evaluate (2 x 100k x 512, k = 5): every call rank 0 makes, with the other
ranks' contributions precomputed outside the timed region and the collectives left out.  Gives the per-rank compute +
host time that bounds the multi-GPU step from below (development aid; the real multi-GPU number is bench.py --gpus N)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from audio_metrics_amd.metrics.kd import subset_indices  # noqa: E402

n, d, k = int(os.environ.get("AB_ROWS", "100000")), 512, 5
dev = torch.device("cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
ref = torch.randn(n, d, generator=gen, device="cuda")
cand = torch.randn(n, d, generator=gen, device="cuda") * 1.05 + 0.05

for world in [int(w) for w in os.environ.get("AB_WORLDS", "1,2,4,8").split(",")]:
    rows = n // world
    pre = {}
    for name, x in (("ref", ref), ("cand", cand)):
        bounds = torch.cat([ops.knn_bounds(x, k, p * rows, rows) for p in range(world)])
        lists = torch.stack([ops.knn_sym_part(x, k, p, world, bounds) for p in range(world)])
        pre[name] = (bounds, lists, ops.knn_lists_finish(lists, x, k))
    mean_r, mean_c = ref.double().mean(0), cand.double().mean(0)

    def step():
        # the order of distributed.evaluate_sharded: statistics of the own rows, Frechet solve enqueued on a side stream,
        # both sets prepared once (norms, maxima, f16 copy), radii, membership counts of the own reference rows, KD share
        for x, mean in ((ref, mean_r), (cand, mean_c)):
            ops.colsum(x[:rows])
            ops.scatter(x[:rows], mean)
        cov_r = ops.scatter(ref, mean_r) / (n - 1)              # (stand-ins for the all-reduced covariances; subtracted below)
        cov_c = ops.scatter(cand, mean_c) / (n - 1)
        job = ops.frechet_async(mean_c, cov_c, mean_r, cov_r)
        prep = {"ref": ops.prepare(ref), "cand": ops.prepare(cand)}
        radii = {}
        for name, x in (("ref", ref), ("cand", cand)):
            bounds, lists, _ = pre[name]
            ops.knn_bounds(x, k, 0, rows, prepared=prep[name])                      # own rows (the all-gathered result is `bounds`)
            ops.knn_sym_part(x, k, 0, world, bounds, prepared=prep[name])           # own share (the all-gathered result is `lists`)
            radii[name] = ops.knn_lists_finish(lists, x, k)
        col, rany, rcov = ops.prdc_counts(ref[:rows], cand, radii["ref"][:rows], radii["cand"],
                                          prepared_ref=prep["ref"].rows(0, rows), prepared_cand=prep["cand"])
        tot = ops.prdc_reduce(col, rany, rcov)
        idx1, idx2 = subset_indices(n, n, 100, 1000, 1234)
        part = ops.kd_poly(cand, ref, ops.upload_host_array(idx1[0::world], dev), ops.upload_host_array(idx2[0::world], dev),
                           1.0 / d, 1, 3)
        return job.result()["fd"], part.cpu(), int(tot[0])

    step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    # (the two full-size scatter calls stand in for the all-reduced covariance and are not part of a rank's work)
    t0 = time.perf_counter()
    for _ in range(3):
        ops.scatter(ref, mean_r); ops.scatter(cand, mean_c)
    torch.cuda.synchronize()
    extra = (time.perf_counter() - t0) / 3 * 1e3
    print(f"world={world}: rank-0 step {min(ts) - extra:.2f} ms (median {sorted(ts)[len(ts) // 2] - extra:.2f}) without collectives", flush=True)
