#!/usr/bin/env python3
"""Which data-dependent route of the filter path a (family, shape) takes - read from am_filter_stats - and whether its
outputs equal the exact kernels' bit for bit, all in ONE process with the SHIPPED library: the exact k-NN values come from
the general entry point (columns = a copy of the set: am_knn_path == 0), the exact membership counts from reference-row
chunks small enough for am_prdc_path == 0.  Feeds tests/test_gpu_routes.py.  Usage: tools/route_probe.py [case ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab_data import make  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402


def shared_clusters(n, d, seed, offset):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    centers = torch.randn(50, d, generator=gen, device="cuda")
    gen2 = torch.Generator(device="cuda").manual_seed(seed + offset)
    lab = torch.randint(0, 50, (n,), generator=gen2, device="cuda")
    return centers[lab] + 1e-3 * torch.randn(n, d, generator=gen2, device="cuda")


def exact_radii(x, k):
    assert ops.knn_path(x.shape[0], x.shape[0], x.shape[1], k, self_distance=False) == 0
    return ops.knn_radii(x, k, columns=x.clone())


def exact_counts(ref, cand, r_ref, r_cand, want_min=False):
    nr, nc, d = ref.shape[0], cand.shape[0], ref.shape[1]
    step = nr
    while ops.prdc_path(step, nc, d) != 0:
        step = (step + 1) // 2
    cols, anys, covs, mins = 0, [], [], []
    for lo in range(0, nr, step):
        out = ops.prdc_counts(ref[lo:lo + step], cand, r_ref[lo:lo + step], r_cand, want_min=want_min)
        cols = cols + out[0]
        anys.append(out[1])
        covs.append(out[2])
        if want_min:
            mins.append(out[3])
    return (cols, torch.cat(anys), torch.cat(covs)) + ((torch.cat(mins),) if want_min else ())


CASES = {
    # name: (family, rows, rows2, dim, k, seed)
    "randn_20k_128": ("randn", 20000, 20000, 128, 5, 1),
    "unit_33k_192": ("unit", 33000, 9000, 192, 10, 2),
    "clustered_20k_512": ("clustered", 20000, 20000, 512, 5, 3),
    "shared_20k_512": ("shared", 20000, 20000, 512, 5, 4),
    "silence_40k_256": ("silence", 40000, 3001, 256, 5, 5),
    "hub_30k_128": ("hub", 30000, 30000, 128, 3, 6),
    "dups_12k_67": ("dups", 12000, 12000, 67, 8, 7),
    "scales_34567_130": ("scales", 34567, 4938, 130, 1, 8),
    "lowrank_20k_257": ("lowrank", 20000, 20000, 257, 5, 9),
    "randn_8200_512": ("randn", 8200, 8200, 512, 10, 10),
    "sparse_16400_64": ("sparse", 16400, 16400, 64, 5, 11),
    "tiny_20k_96": ("tiny", 20000, 20000, 96, 5, 12),
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    ops.filter_stats_enable("cuda:0", True)
    for name in names:
        fam, n, n2, d, k, seed = CASES[name]
        if fam == "shared":
            x, y = shared_clusters(n, d, seed, 1), shared_clusters(n2, d, seed, 2)
        else:
            x, y = make(fam, n, d, seed), make(fam, n2, d, seed + 100)
        ops.filter_stats_read("cuda:0")
        t0 = time.perf_counter()
        r = ops.knn_radii(x, k)
        torch.cuda.synchronize()
        t_knn = time.perf_counter() - t0
        s_knn = ops.filter_stats_read("cuda:0")
        r2 = ops.knn_radii(y, k)
        ops.filter_stats_read("cuda:0")
        want_min = seed % 2 == 1
        t0 = time.perf_counter()
        got = ops.prdc_counts(x, y, r, r2, want_min=want_min)
        torch.cuda.synchronize()
        t_cross = time.perf_counter() - t0
        s_cross = ops.filter_stats_read("cuda:0")
        ok_r = torch.equal(r.view(torch.int32), exact_radii(x, k).view(torch.int32))
        want = exact_counts(x, y, r, r2, want_min)
        ok_c = all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
                   for a, b in zip(got, want))
        print(f"{name}: paths {ops.knn_path(n, n, d, k)}/{ops.prdc_path(n, n2, d)} radii {'ok' if ok_r else 'MISMATCH'} counts {'ok' if ok_c else 'MISMATCH'} | "
              f"knn {t_knn * 1e3:.1f} ms queued {s_knn['knn_queued']} spilled {s_knn['knn_spilled']} verified {s_knn['knn_verified_pairs']} "
              f"fallback_rows {s_knn['knn_fallback_rows']} bound {s_knn['bound_ratio_max']:.3f} on {s_knn['bound_pairs']} | "
              f"cross {t_cross * 1e3:.1f} ms queued {s_cross['prdc_queued']} overflow {s_cross['prdc_overflow_queue']} "
              f"fallback {s_cross['prdc_fallback_calls']}", flush=True)
