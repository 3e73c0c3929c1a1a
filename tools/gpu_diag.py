#!/usr/bin/env python3
"""First-contact diagnostics on a real MI355X: per-op parity numbers (printed,
not asserted) and rough timings.  Development aid; the pass/fail versions live
in tests/test_gpu_*.py."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs as gi  # noqa: E402
import oracle  # noqa: E402
from oracle import exact  # noqa: E402
import audio_metrics_amd as am  # noqa: E402
from audio_metrics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
G = {k: np.load(os.path.join(ROOT, "tests", "golden", k + ".npz")) for k in ("stats", "fad", "kd", "prdc", "apa")}


def section(name):
    print(f"\n=== {name}", flush=True)


def guarded(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()
    sys.stdout.flush()


def t_prdc():
    section("PRDC vs exact C model (bit-exact) and reference goldens")
    for name, (kind, seed, nr, nc, d, k) in gi.PRDC_CASES.items():
        ref, cand = gi.pair(kind, seed, nr, nc, d)
        R, C = torch.as_tensor(ref).to(dev), torch.as_tensor(cand).to(dev)
        rr = ops.knn_radii(R, k)
        rc = ops.knn_radii(C, k)
        col, rany, rcov, rmin = ops.prdc_counts(R, C, rr, rc, want_min=True)
        tot = ops.prdc_reduce(col, rany, rcov).cpu().numpy()
        _, aux = exact.prdc(ref, cand, k)
        rr_, rc_ = rr.cpu().numpy(), rc.cpu().numpy()
        print(f"{name:26s} radii bitdiff ref {int((rr_.view(np.uint32) != aux['r_ref'].view(np.uint32)).sum())}"
              f" cand {int((rc_.view(np.uint32) != aux['r_cand'].view(np.uint32)).sum())}"
              f" | col diff {int((col.cpu().numpy() != aux['col_count']).sum())}"
              f" any diff {int((rany.cpu().numpy() != aux['row_any']).sum())}"
              f" min bitdiff {int((rmin.cpu().numpy().view(np.uint32) != aux['row_min'].view(np.uint32)).sum())}"
              f" | max rel radii vs golden {np.max(np.abs(rr_ - G['prdc'][name + '/r_ref']) / G['prdc'][name + '/r_ref']):.2e}"
              f" | totals {tot.tolist()} golden P {float(G['prdc'][name + '/precision']):.4f} -> {tot[0] / nc:.4f}")


def t_stats():
    section("stats vs oracle / goldens")
    for name, (seed, d, splits) in gi.STATS_CASES.items():
        x = gi.randn(seed, sum(splits), d, 1.3, 0.2)
        X = torch.as_tensor(x).to(dev)
        mean, cov = ops.stats(X)
        o = oracle.OracleData(False).add(torch.as_tensor(x))
        x64 = x.astype(np.float64)
        c64 = np.cov(x64.T) if len(x) > 1 else np.zeros((d, d))
        cm = cov.cpu().numpy()
        print(f"{name:22s} oneshot: mean err vs f64 {np.abs(mean.cpu().numpy() - x64.mean(0)).max():.2e}"
              f" cov relF vs f64 {np.linalg.norm(cm - c64) / max(np.linalg.norm(c64), 1e-300):.2e}"
              f" (oracle f32 relF {np.linalg.norm(o.cov.numpy() - c64) / max(np.linalg.norm(c64), 1e-300):.2e})"
              f" asym {np.abs(cm - cm.T).max():.1e}")
        # incremental adds through the device Chan merge
        s = 0
        n_acc = None
        for b in splits:
            m2, c2 = ops.stats(X[s:s + b])
            if n_acc is None:
                n_acc, m_acc, c_acc = b, m2, c2
            else:
                m_acc, c_acc = ops.stats_merge(n_acc, m_acc, c_acc, b, m2, c2, inplace=True)
                n_acc += b
            s += b
        gm = G["stats"][name + "/mean"]
        print(f"{'':22s} chan:    mean err vs golden {np.abs(m_acc.cpu().numpy() - gm).max():.2e}"
              f" cov trace rel {abs(np.trace(c_acc.cpu().numpy()) - float(G['stats'][name + '/cov_trace'])) / max(abs(float(G['stats'][name + '/cov_trace'])), 1e-300):.2e}"
              f" block max abs {np.abs(c_acc.cpu().numpy()[:16, :16] - G['stats'][name + '/cov_block']).max():.2e}")


def t_fad():
    section("FAD vs goldens")
    for name, (kind, seed, nr, nc, d) in gi.FAD_CASES.items():
        ref, cand = gi.pair(kind, seed, nr, nc, d)
        t0 = time.time()
        ma, ca = ops.stats(torch.as_tensor(cand).to(dev))
        mb, cb = ops.stats(torch.as_tensor(ref).to(dev))
        r = ops.frechet(ma, ca, mb, cb)
        torch.cuda.synchronize()
        g = float(G["fad"][name + "/fad"])
        print(f"{name:22s} fd {r['fd']:.10g} golden {g:.10g} rel {abs(r['fd'] - g) / abs(g):.2e} | tr_sqrt rel "
              f"{abs(r['tr_sqrt'] - float(G['fad'][name + '/tr_sqrt'])) / abs(float(G['fad'][name + '/tr_sqrt'])):.2e}"
              f" iters {r['iters']} resid {r['resid']:.2e}  [{time.time() - t0:.3f}s]")


def t_kd():
    section("KD vs goldens")
    for name, (kind, seed, n1, n2, d) in gi.KD_CASES.items():
        f2, f1 = gi.pair(kind, seed, n2, n1, d)
        i1, i2 = oracle.draw_subsets(n1, n2)
        out = ops.kd_poly(torch.as_tensor(f1).to(dev), torch.as_tensor(f2).to(dev), torch.as_tensor(i1).to(dev),
                          torch.as_tensor(i2).to(dev), 1.0 / d, 1.0, 3).cpu().numpy()
        g = G["kd"][name + "/mmds"]
        print(f"{name:18s} mean {out.mean():.9g} golden {float(G['kd'][name + '/mean']):.9g}"
              f" rel {abs(out.mean() - float(G['kd'][name + '/mean'])) / abs(float(G['kd'][name + '/mean'])):.2e}"
              f" | per-subset max abs {np.abs(out - g).max():.2e} max rel {np.max(np.abs(out - g) / np.abs(g)):.2e}")


def t_perf():
    section("rough timings")
    for n in (20000, 100000):
        d = 512
        x = torch.randn(n, d, device=dev)
        y = torch.randn(n, d, device=dev) * 1.05 + 0.05

        def timeit(fn, reps=2):
            fn()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.time() - t0) / reps
        ts = timeit(lambda: ops.stats(x))
        print(f"N={n}: stats {ts * 1e3:.3f} ms ({2 * n * d * d / ts / 1e12:.1f} TF)")
        ma, ca = ops.stats(x)
        mb, cb = ops.stats(y)
        tf = timeit(lambda: ops.frechet(ma, ca, mb, cb))
        print(f"N={n}: frechet {tf * 1e3:.3f} ms")
        i1, i2 = oracle.draw_subsets(n, n)
        I1, I2 = torch.as_tensor(i1).to(dev), torch.as_tensor(i2).to(dev)
        tk = timeit(lambda: ops.kd_poly(y, x, I1, I2, 1.0 / d, 1.0, 3))
        print(f"N={n}: kd {tk * 1e3:.3f} ms ({100 * 3 * 2 * 1000 * 1000 * d / tk / 1e12:.1f} TF algorithmic)")
        for k in (5, 10):
            tr = timeit(lambda: ops.knn_radii(x, k), reps=1)
            print(f"N={n}: knn_radii k={k} {tr * 1e3:.2f} ms ({2 * n * n * d / tr / 1e12:.1f} TF)")
        rr, rc = ops.knn_radii(x, 5), ops.knn_radii(y, 5)
        tc = timeit(lambda: ops.prdc_counts(x, y, rr, rc), reps=1)
        print(f"N={n}: prdc_counts {tc * 1e3:.2f} ms ({2 * n * n * d / tc / 1e12:.1f} TF)")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), am._lib.load().am_version())
    which = sys.argv[1:] or ["prdc", "stats", "fad", "kd", "perf"]
    for w in which:
        guarded(globals()["t_" + w])
