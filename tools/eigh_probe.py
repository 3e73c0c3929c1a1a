#!/usr/bin/env python3
"""Times am_eigh_sym_f64 (the PCA projection's eigensolver) on Gram matrices of random data and checks the result against
torch.linalg.eigh.  AB_DIM (default 512), AB_ROWS (rows of the data the Gram matrix comes from, default 4096)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_metrics_amd import hip_ops as ops  # noqa: E402

d, n = int(os.environ.get("AB_DIM", "512")), int(os.environ.get("AB_ROWS", "4096"))
dev = torch.device("cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
for kind in ("randn", "decaying", "rank_deficient"):
    x = torch.randn(n if kind != "rank_deficient" else d // 3, d, generator=gen, device=dev, dtype=torch.float64)
    if kind == "decaying":
        x = x * torch.logspace(0, -5, d, device=dev, dtype=torch.float64)
    x = x - x.mean(0)
    a = x.T @ x
    ops.eigh_descending(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        evals, evecs = ops.eigh_descending(a)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    want = torch.linalg.eigvalsh(a).flip(0)
    err = float((evals - want).abs().max() / want.abs().max())
    resid = float((evecs @ a - evals[:, None] * evecs).norm() / a.norm())
    orth = float((evecs @ evecs.T - torch.eye(d, device=dev, dtype=torch.float64)).abs().max())
    print(f"{kind:15s} D={d}: {ms:7.2f} ms per solve | max |lambda - eigvalsh| / lambda_max = {err:.2e} | residual {resid:.2e} | orthogonality {orth:.2e}", flush=True)
