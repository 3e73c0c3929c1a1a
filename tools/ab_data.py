"""Synthetic inputs for the A/B tools: well-behaved and adversarial embedding sets (device tensors)."""
import torch


def make(kind, n, d, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n, d, generator=gen, device="cuda")
    if kind == "randn":
        return x
    if kind == "clustered":            # tight clusters: d2 dominated by cancellation noise, clamps to 0, ties
        centers = torch.randn(50, d, generator=gen, device="cuda")
        lab = torch.randint(0, 50, (n,), generator=gen, device="cuda")
        return centers[lab] + 1e-3 * x
    if kind == "scales":               # row norms spread over four orders of magnitude, a few all-zero rows
        s = 10.0 ** (torch.rand(n, 1, generator=gen, device="cuda") * 4 - 2)
        x = x * s
        x[::997] = 0.0
        return x
    if kind == "dups":                 # every row has an exact duplicate: zero radii, exact ties
        h = n // 2
        return torch.cat([x[:h], x[:n - h]])
    if kind in ("silence", "hub"):     # a block of IDENTICAL rows (0.2 - 12 % of the set): silent windows embed to one vector;
        m = max(40, int(n * [0.002, 0.01, 0.04, 0.12][seed % 4]))     # "hub": a low-norm one, every row's nearest neighbour
        if kind == "silence":
            x = x + 0.5
        x[torch.randperm(n, generator=gen, device="cuda")[:m]] = x[0] * (1.0 if kind == "silence" else 0.1)
        return x / x.norm(dim=1, keepdim=True) if kind == "silence" else x
    if kind == "lowrank":              # a 4-dimensional subspace plus tiny noise, large common offset
        basis = torch.randn(4, d, generator=gen, device="cuda")
        return torch.randn(n, 4, generator=gen, device="cuda") @ basis + 1e-4 * x + 3.0
    if kind == "unit":                 # CLAP-like: positive offset, L2-normalised rows
        x = x + 0.5
        return x / x.norm(dim=1, keepdim=True)
    if kind == "tiny":                 # all magnitudes near 1e-18: squared norms close to the f32 underflow range
        return x * 1e-18
    if kind == "huge":                 # magnitudes near 1e15: squared norms ~1e33 (below f32 overflow), products large
        return x * 1e15
    if kind == "const":                # every row identical: all distances are rounding noise, every tie exact
        return x[:1].expand(n, d).contiguous()
    if kind == "sparse":               # one-hot-like rows: most products are exact zeros, many exactly equal distances
        idx = torch.randint(0, d, (n,), generator=gen, device="cuda")
        out = torch.zeros(n, d, device="cuda")
        out[torch.arange(n, device="cuda"), idx] = 1.0 + 0.01 * torch.rand(n, generator=gen, device="cuda")
        return out
    raise ValueError(kind)
