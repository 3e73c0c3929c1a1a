#!/usr/bin/env python3
"""Where does the f16 filter path start to pay - on data that is NOT randn?  

For every (data family, width) one child process per MODE - `filter` (the f16 filter sweeps forced on from the first row),
`exact` (forced off) and `shipped` (the library's own thresholds: 6144 / 8192 / 16384 rows for D >= 256 / >= 128 / >= 32 and
2^24 pairs for the membership counts) - times a cold PRDC (radii of both sets + membership counts) at 4 000 ... 20 000 rows,
k = 5 and 10, and prints a sha1 of radii, counts and flags: the three modes must agree bit for bit, and `shipped` should sit
on the faster of the other two.  Uses the -DAM_DEV_KNOBS build (the shipped library reads no environment).

    python tools/threshold_sweep.py            # parent: runs all children, prints the table
"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = os.environ.get("TS_KINDS", "randn,unit,clustered").split(",")
DIMS = [int(v) for v in os.environ.get("TS_DIMS", "64,128,512").split(",")]
ROWS = [int(v) for v in os.environ.get("TS_ROWS", "4000,6144,8192,12000,16384,20000").split(",")]
KS = [int(v) for v in os.environ.get("TS_KS", "5,10").split(",")]
MODES = {"filter": {"AM_KNN_FAST_MIN_ROWS": "1", "AM_FAST_MIN_PAIRS_LOG2": "1", "AM_KNN_SYM_MIN_ROWS": "1024"},
         "exact": {"AM_KNN_FAST": "0", "AM_PRDC_FAST": "0"}, "shipped": {}}


def child(kind, dim):
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ab_data import make
    from audio_metrics_amd import hip_ops as ops
    ops.filter_stats_enable("cuda:0", True)
    for n in ROWS:
        if kind.startswith("shared"):                         # both sets around the SAME 50 clusters; "shared:0.03" sets their spread
            spread = float(kind.split(":")[1]) if ":" in kind else 1e-3
            gen = torch.Generator(device="cuda").manual_seed(5)
            centres = torch.randn(50, dim, generator=gen, device="cuda")
            ref = centres[torch.randint(0, 50, (n,), generator=gen, device="cuda")] + spread * torch.randn(n, dim, generator=gen, device="cuda")
            cand = centres[torch.randint(0, 50, (n,), generator=gen, device="cuda")] + spread * torch.randn(n, dim, generator=gen, device="cuda")
        else:
            ref, cand = make(kind, n, dim, 11), make(kind, n, dim, 12)
        if kind == "randn":
            cand = cand * 1.05 + 0.05
        for k in KS:
            def run():
                r_ref, r_cand = ops.knn_radii(ref, k), ops.knn_radii(cand, k)
                return (r_ref, r_cand) + tuple(ops.prdc_counts(ref, cand, r_ref, r_cand))
            out = run()
            torch.cuda.synchronize()
            ops.filter_stats_read("cuda:0")
            reps = 8
            t0 = time.perf_counter()
            for _ in range(reps):
                out = run()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            st = ops.filter_stats_read("cuda:0")
            digest = hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in out)).hexdigest()[:10]
            print(f"ROW {kind} {dim} {n} {k} {ms:.3f} {digest} {ops.knn_path(n, n, dim, k)} {ops.prdc_path(n, n, dim)} "
                  f"{st['knn_queued'] / (2 * reps * n):.1f} {st['prdc_queued'] / reps:.0f} {st['knn_fallback_rows'] / reps:.0f}", flush=True)


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    table = {}
    for kind in KINDS:
        for dim in DIMS:
            for mode, knobs in MODES.items():
                env = dict(os.environ, AM_HIP_LIBRARY="dev", **knobs)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, str(dim)], env=env, capture_output=True,
                                   text=True, timeout=600)
                if r.returncode != 0:
                    print(f"# {kind} D={dim} {mode}: child failed\n{r.stderr[-800:]}", flush=True)
                    continue
                for line in r.stdout.splitlines():
                    if line.startswith("ROW "):
                        _, kd, d, n, k, ms, digest, kp, pp, qpr, pq, fb = line.split()
                        table.setdefault((kd, int(d), int(n), int(k)), {})[mode] = (float(ms), digest, kp, pp, qpr, pq, fb)
    print(f"{'data':10s} {'D':>4s} {'rows':>6s} {'k':>3s} | {'filter ms':>9s} {'exact ms':>9s} {'shipped ms':>10s} (paths) | queued/row  membership queue  fallback rows | verdict")
    wrong = 0
    for (kd, d, n, k), modes in sorted(table.items()):
        if len(modes) < 3:
            continue
        f, e, s = modes["filter"], modes["exact"], modes["shipped"]
        same = f[1] == e[1] == s[1]
        best = min(f[0], e[0])
        ok = s[0] <= 1.10 * best + 0.03                                   # within 10 % (+ 30 us of timing noise) of the better form
        wrong += (not ok) or (not same)
        print(f"{kd:10s} {d:4d} {n:6d} {k:3d} | {f[0]:9.3f} {e[0]:9.3f} {s[0]:10.3f} ({s[2]}/{s[3]})  | {f[4]:>10s} {f[5]:>16s} {f[6]:>14s} | "
              f"{'bits equal' if same else 'BITS DIFFER'}, {'ok' if ok else 'threshold misplaced: shipped is %.0f %% above the better form' % (100 * (s[0] / best - 1))}",
              flush=True)
    print("rows where the shipped thresholds pick the slower form or the bits differ:", wrong)


if __name__ == "__main__":
    main()
