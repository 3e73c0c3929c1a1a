import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops
n, d, k = 20000, 128, 5
gen = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, d, generator=gen, device="cuda")
y = torch.randn(n, d, generator=gen, device="cuda") * 1.05 + 0.05
rx, ry = ops.knn_radii(x, k), ops.knn_radii(y, k)
col, rany, rmin = ops.prdc_counts(x, y, rx, ry)
a0 = np.load("/tmp/diff_cross_0.npz")
bad = np.flatnonzero(a0["rany"] != rany.cpu().numpy())
print("bad rows", len(bad), bad[:10])
nx, ny = (x.double() ** 2).sum(1), (y.double() ** 2).sum(1)
C = 2.0 ** -7 + 2.0 ** -10 + 2.0 ** -12
rnmax = nx.max()
T = ry.double() ** 2
for i in bad[:6]:
    t = nx[i] + ny - 2 * (y.double() @ x[i].double())
    xb, yb = x[i].bfloat16().double(), y.bfloat16().double()
    a = nx[i] + ny - 2 * (yb @ xb)
    inside = torch.nonzero(t < T).flatten()
    E = C * (rnmax + ny)
    print("row", int(i), "inside cols", inside.tolist()[:6], "t-T", (t - T)[inside][:6].tolist(), "a-(T+E)", (a - (T + E))[inside][:6].tolist(),
          "a-(T-E)", (a - (T - E))[inside][:6].tolist(), "err/eps", ((a - t).abs() / (C * (nx[i] + ny)))[inside][:6].tolist())
