#!/usr/bin/env python3
"""Headline benchmark: evaluate() embeddings/sec for FAD + KD + PRDC on two sets of
100k CLAP-512 (f32) embeddings already resident in HBM (BASELINE.json metric,
configs[2]), on N GPUs of one node.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks (child process: torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config e2e          # BASELINE configs[4] shape: audio -> embedder -> APA + FAD (see run_e2e)

A step is one COLD evaluation: both sets' statistics, the Frechet distance, the
100 x 1000 kernel-distance subsets, both sets' k-NN radii and the membership
counts are all recomputed (nothing is cached between steps).  With N > 1 the rows
of both sets are sharded over the ranks (strong scaling: the problem is fixed) and
the stats / gathered embeddings / radii / counts go through RCCL collectives.
Rank 0 prints ONE JSON line.

Inputs come from numpy's PCG64 (tests/golden/inputs.py: bench_pair), so the CPU oracle can reproduce them:
tests/golden/bench_prdc.npz holds oracle.prdc_blocked's values for exactly these sets and the line's `result` is
checked against it (`result_check`).

Beside the contract's keys the line carries `roofline` / `other_tile_kernel` / `cpu_baseline`, `variants`, `filter`, `warm`,
`first_call`, and (round 6) `process_cold` - with one GPU: ONE evaluate() of a fresh process, started as a child before this
process touches the GPU (--no-process-cold skips it) - and, with several ranks, `c_entry_ms` beside `python_schedule_ms`: both
exchange schedules timed back to back (the one the flags chose is `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

METRIC = "evaluate() embeddings/sec (FAD+KD+PRDC), 2×100k CLAP-512 sets, 1/2/4/8 GPUs"
F32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: bf16/f16 dense peak (v_mfma_f32_32x32x16_f16)


def cpu_baseline(ref, cand, k, prdc_rows=(10000, 20000, 40000)):
    """The CPU oracle (a port of the reference's torch/numpy calls, oracle/) timed on this host.  stats + FAD + KD run at
    the full size; PRDC materialises N x N matrices in the reference (164 GB at 100k), so it is timed on row subsamples
    (10k / 20k / 40k), the growth exponent is fitted, and the full-size time is the 40k point scaled by
    (N / 40k)^2."""
    import oracle
    # LAPACK geev (the reference's eigvals) and small-block cdist degrade badly with hundreds of
    # threads (measured on the 256-core GPU host: eigvals 477 s); cap at 32 and report the count used.
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref, cand = ref.cpu(), cand.cpu()
    n = len(ref)
    t0 = time.perf_counter()
    a = oracle.OracleData(False).add(cand)
    b = oracle.OracleData(False).add(ref)
    t_stats = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.frechet_distance(a, b)
    t_fad = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.kid_from_features(cand, ref)
    t_kd = time.perf_counter() - t0
    points = []
    for m in prdc_rows:
        m = min(m, n)
        t0 = time.perf_counter()
        oracle.prdc_blocked(ref[:m], cand[:m], k, block=2048)
        points.append((m, time.perf_counter() - t0))
        if m == n:
            break
    m_last, t_last = points[-1]
    t_prdc = t_last * (n / m_last) ** 2
    if len(points) > 1:
        lx, ly = np.log([p[0] for p in points]), np.log([p[1] for p in points])
        exponent = float(np.polyfit(lx, ly, 1)[0])
    else:
        exponent = 2.0
    total = t_stats + t_fad + t_kd + t_prdc
    pts = ", ".join(f"{m}: {t:.2f}s" for m, t in points)
    return {
        "value": 2 * n / total, "unit": "embeddings/s", "cores": torch.get_num_threads(), "kind": "port",
        "prdc_points_s": {str(m): t for m, t in points}, "prdc_fitted_exponent": exponent, "prdc_extrapolated_s": t_prdc,
        "sample": (f"oracle/ (torch-CPU port of the reference): stats {t_stats:.2f}s + FAD {t_fad:.2f}s + KD {t_kd:.2f}s "
                   f"at full 2x{n}x{ref.shape[1]}; PRDC(k={k}) timed on 2 x {{{pts}}} rows (fitted growth N^{exponent:.2f}) "
                   f"and the {m_last}-row point scaled by (N/{m_last})^2 to {t_prdc:.0f}s because the reference's N x N "
                   f"matrices do not fit host RAM at 100k"),
    }


def warm_evaluate(am, ref, cand, k, steps):
    """SURVEY 8(d) 'warm' figure (never `value`): the reference side is cached the way a second
    AudioMetrics.evaluate() finds it (statistics and radii_k kept on the reference object, data.py:60-66), so a step
    recomputes only the candidate's statistics and radii, the cross counts, KD and FAD."""
    reference = am.AudioMetricsData(True)
    reference.add(ref)
    reference.get_radii(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from audio_metrics_amd import hip_ops as ops
    from audio_metrics_amd.distributed import evaluate_single
    for _ in range(steps):
        candidate = am.AudioMetricsData(True)
        candidate.add(cand)
        # what AudioMetrics.evaluate() issues for these two sets (audio_metrics.py: _run_fused): one library call
        res = evaluate_single(reference.embeddings, candidate.embeddings, ("fad", "kd", "prdc"), k, ops,
                              given_ref={"mean": reference.mean, "cov": reference.cov, "radii": reference.get_radii(k)},
                              given_cand={"mean": candidate.mean, "cov": candidate.cov})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": (len(ref) + len(cand)) / dt, "unit": "embeddings/s", "ms_per_step": dt * 1e3, "steps": steps,
            "what": "reference statistics and radii cached (second evaluate() against the same reference)", "result": res}


def check_against_fixture(result, kind, n, d, k, suffix=""):
    """`result` against the fixture for the same numpy-seeded sets (tests/golden/bench_prdc.npz, written by
    tests/golden/make_goldens.py bench in the build container): FAD and KD are the reference's OWN frechet_distance /
    kernel_distance outputs on these inputs, the four PRDC values come from oracle.prdc_blocked (row blocks of the
    reference's torch calls - its N x N formulation needs 164 GB at 100k rows)."""
    path = os.path.join(ROOT, "tests", "golden", "bench_prdc.npz")
    if d != 512:
        kind = f"{kind}_d{d}{suffix}"               # the narrower sets' keys carry the width (make_goldens.py gen_bench), float64 ones "_f64"
    tag = f"{kind}_k{k}"
    if not (os.path.exists(path) and n == 100000):
        return None
    g = np.load(path, allow_pickle=False)
    diffs, ok = {}, True
    if f"{tag}/precision" in g.files and "precision" in result:
        for key in ("precision", "recall", "density", "coverage"):
            want = float(g[f"{tag}/{key}"])
            diffs[key] = abs(result[key] - want)
            # single distances differ in their last f32 bit between the two arithmetic orders: a few of the 1e10 strict
            # comparisons flip.  1e-4 relative (north star) with a floor of five rows of the 100k.
            ok = ok and diffs[key] <= max(1e-4 * abs(want), 5.0 / n)
    if f"{kind}/fad" in g.files:
        for key, floor in (("fad", 0.0), ("kernel_distance_mean", 5e-7), ("kernel_distance_std", 5e-7)):
            if key in result:
                want = float(g[f"{kind}/{key}"])
                diffs[key] = abs(result[key] - want)
                ok = ok and diffs[key] <= max(1e-4 * abs(want), floor)      # 5e-7: f32 noise floor of the reference's own KD
    if not diffs:
        return None
    return {"fixture": f"tests/golden/bench_prdc.npz:{tag} + {kind}/fad,kd", "abs_diff": diffs, "ok": bool(ok)}


def stream_add(am, x, batch, store, reps=1):
    """the rows of `x` fed to AudioMetricsData.add() in `batch`-row device batches, the way the
    embedding pipeline does (reference data.py:37-47, 68-94; embed.py:231-236).  One kernel launch per add (am_stats_push_f32)."""
    n = x.shape[0]
    times = []
    for _ in range(reps + 1):                                  # first round = warm-up (allocator, code objects)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        data = am.AudioMetricsData(store)
        for lo in range(0, n, batch):
            data.add(x[lo:lo + batch])
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"embeddings_per_s": n / best, "ms": best * 1e3, "adds": (n + batch - 1) // batch, "us_per_add": best / ((n + batch - 1) // batch) * 1e6,
            "rows": n, "batch": batch, "store_embeddings": bool(store)}


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` invoked plainly (no launcher in the environment): start the N ranks as a CHILD process -
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` - before this process has touched the
    GPU, let the child write rank 0's JSON line to our stdout, and exit with its code.  (A child, never an exec: replacing
    a process that has initialised the GPU takes the box down on this pool; this one has not - `import torch` does not -
    but a child is right either way.)  The reference's multi-GPU entry needs no launcher either (util/gpu_parallel.py:79-118)."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC is the only form the host driver supports
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n_ranks, 1))))
    # --standalone: torchrun runs its own c10d rendezvous on a port IT binds (a port found here by bind-then-close could be
    # taken by somebody else before the child binds it again)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n_ranks), os.path.abspath(__file__), *sys.argv[1:]]
    # the child gets its own session: a SIGTERM / SIGINT that ends THIS process (tools/run_round_checks.sh wraps every call in
    # `timeout`) is passed on to the whole group - torchrun and its N ranks - instead of leaving them on the GPU until the
    # collective timeout
    child = subprocess.Popen(cmd, env=env, start_new_session=True)

    def pass_on(signum, frame):
        try:
            os.killpg(child.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        try:
            child.wait(timeout=20)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        sys.exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, pass_on)
    return child.wait()


def library_stamp():
    """sha256 over the sources / headers / flags the LOADED library was built from (lib/*.so.stamp.json)."""
    from audio_metrics_amd import _build, _lib
    try:
        with open(_build._stamp_path(_lib.library_path())) as f:
            return json.load(f).get("sources_sha256")
    except (OSError, ValueError):
        return None


def recorded_traffic(kernel, stamp):
    """(bytes per launch or None, provenance) of `kernel` from profiles/traffic.json: PMC counters (FETCH_SIZE x 2 +
    WRITE_SIZE, separate --pmc passes) cannot be collected inside an unprofiled bench run, so the figure is a RECORD of the
    profile named in the table - valid only for the library it was measured on: when the loaded library was built from
    other sources than the profiled one, `traffic` is null and the provenance says so."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, {"table": None}
    entry = table.get("kernels", {}).get(kernel)
    if entry is None:
        return None, {"table": "profiles/traffic.json", "kernel": kernel, "state": "no record for this kernel"}
    same = stamp is not None and entry.get("sources_sha256") == stamp
    source = {"table": "profiles/traffic.json", "profile": entry.get("profile"), "measured_on_sources_sha256": entry.get("sources_sha256"),
              "loaded_sources_sha256": stamp, "l2_hit": entry.get("l2_hit"),
              "state": "recorded on this library" if same else "STALE: the loaded library was built from other sources than the "
                       "profiled one - re-run tools/profile_bench.sh + tools/update_traffic.py"}
    return (entry.get("bytes_per_launch") if same else None), source


def first_call(step, fence, samples=3):
    """A step that finds NOTHING from an earlier one but the loaded code objects: the KD index table is drawn and uploaded
    again, the evaluate workspace and every other block come fresh from hipMalloc (torch.cuda.empty_cache()).  Never
    `value`; reported beside it so that what `value` leaves out is a number, not a sentence."""
    from audio_metrics_amd import hip_ops as ops
    from audio_metrics_amd.metrics import kd
    times = []
    for _ in range(samples):
        kd._DEVICE_TABLES.clear()
        ops.release_workspaces()
        fence()
        torch.cuda.empty_cache()
        fence()
        t0 = time.perf_counter()
        step()
        fence()
        times.append((time.perf_counter() - t0) * 1e3)
    return {"ms": times, "ms_min": min(times), "ms_mean": sum(times) / len(times),
            "what": "one cold evaluate() after clearing the KD index-table cache, the cached evaluate workspace and torch's "
                    "caching allocator (hipMalloc of ~1.2 GB + table draw and upload are inside)"}


def process_cold_child(args):
    """`bench.py --process-cold-child` (started by the parent BEFORE it touches the GPU): what a user's process pays for ONE
    evaluate() - VERDICT r5 next-6.  Everything from interpreter start to the floats on the host, split by where it goes;
    the sets are generated on the device (no upload).  Prints one JSON line."""
    t_proc = float(os.environ.get("AM_BENCH_T0", "0")) or None        # the parent's clock right before it started this process
    # (bench.py imports torch at module level: interpreter start + `import torch` + numpy lie between t_proc and this line)
    marks = [("t0", time.perf_counter())]
    startup_ms = (time.time() - t_proc) * 1e3 if t_proc else None
    import torch as th
    th.cuda.set_device(0)
    th.zeros(1, device="cuda").add_(1)
    th.cuda.synchronize()
    marks.append(("hip_runtime_and_first_torch_kernel", time.perf_counter()))
    import audio_metrics_amd as am
    from audio_metrics_amd import hip_ops as ops
    from audio_metrics_amd import distributed
    from audio_metrics_amd.metrics import kd
    am._lib.load()
    marks.append(("import_package_and_dlopen_library", time.perf_counter()))
    n, d, k = args.rows, args.dim, args.nearest_k
    g = th.Generator(device="cuda").manual_seed(1234)
    ref = th.randn(n, d, device="cuda", generator=g)
    cand = th.randn(n, d, device="cuda", generator=g) * 1.05 + 0.05
    th.cuda.synchronize()
    marks.append(("sets_generated_on_device", time.perf_counter()))

    def one():
        t0 = time.perf_counter()
        res = distributed.evaluate_single(ref, cand, ("fad", "kd", "prdc"), k, ops)
        th.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, res

    first_ms, res = one()                                # code objects, workspace hipMalloc, KD table, the step
    kd._DEVICE_TABLES.clear()
    ops.release_workspaces()
    th.cuda.synchronize()
    th.cuda.empty_cache()
    realloc_ms, _ = one()                                # workspace + table again, code objects resident
    warm_ms = min(one()[0] for _ in range(3))
    phases = {name: (t - marks[i][1]) * 1e3 for i, (name, t) in enumerate(marks[1:])}
    out = {"process_cold_ms": first_ms, "above_warm_step_ms": first_ms - warm_ms, "warm_step_ms": warm_ms,
           "split_ms": {"first_launch_code_objects": first_ms - realloc_ms, "workspace_hipMalloc_and_kd_table": realloc_ms - warm_ms,
                        "step": warm_ms},
           "before_the_call_ms": dict({"interpreter_start_and_import_torch": startup_ms}, **phases),
           "interpreter_start_to_first_result_ms": ((time.time() - t_proc) * 1e3 if t_proc else None),
           "what": "a FRESH process: import torch, HIP runtime, import the package + dlopen the library, sets generated on the "
                   "device, then ONE evaluate() (am_evaluate_f32) timed to the floats on the host; the split comes from a "
                   "second call after releasing the workspace / KD table / allocator cache and a third, warm one",
           "fad": res["fad"], "precision": res["precision"]}
    print(json.dumps(out), flush=True)


def process_cold(argv_tail):
    """Runs process_cold_child in a child process and returns its record (None + reason on failure).  Called before the parent
    initialises the GPU; the child has the device to itself."""
    import subprocess
    env = dict(os.environ, AM_BENCH_T0=repr(time.time()))
    cmd = [sys.executable, os.path.abspath(__file__), "--process-cold-child", *argv_tail]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        return {"error": "child timed out"}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    return {"error": "no record", "returncode": r.returncode, "stderr_tail": r.stderr[-400:]}


def timed_steps(step, fence, steps, warmup):
    for _ in range(warmup):
        result = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        result = step()
    fence()
    return time.perf_counter() - t0, result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("evaluate", "e2e"), default="evaluate")
    ap.add_argument("--rows", type=int, default=100000, help="rows per set (default: the BASELINE config)")
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--nearest-k", type=int, default=5)
    ap.add_argument("--data", choices=("randn", "clap"), default="randn",
                    help="randn: SURVEY 8(d) C3 sets; clap: unit-norm rows with offsets 0.5 / 0.55 (CLAP-shaped)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-process-cold", action="store_true", help="skip the fresh-process time-to-first-result child")
    ap.add_argument("--process-cold-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pairs", type=int, default=2000, help="--config e2e: audio pairs per side")
    ap.add_argument("--c-entry", action="store_true",
                    help="N > 1: every rank's step is ONE library call (am_evaluate_sharded_f32 with hooks over the process "
                         "group) instead of the Python exchange schedule; same results")
    ap.add_argument("--bulk-communicator", action="store_true",
                    help="row gathers on a second RCCL communicator (distributed.enable_bulk_communicator; default: one)")
    args = ap.parse_args()
    if args.process_cold_child:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        process_cold_child(args)
        return
    cold = None
    # (never under a profiler: its preloaded library has initialised the GPU in THIS process already, and starting another
    # program from a process that holds the GPU is what this pool forbids)
    profiled = any(key.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for key in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if args.gpus == 1 and args.config == "evaluate" and not args.no_process_cold and "WORLD_SIZE" not in os.environ and not profiled:
        # a fresh process's ONE evaluate(), measured in a child BEFORE this process initialises the GPU (the child has the device
        # to itself; a child, never an exec)
        cold = process_cold(["--rows", str(args.rows), "--dim", str(args.dim), "--nearest-k", str(args.nearest_k)])

    # dmabuf IPC is the only form the host driver supports (see the environment notes): must be in the environment BEFORE the
    # first HIP call of the process initialises the runtime; harmless when already set
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))                # plain `python bench.py --gpus N`: the ranks run in a child
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # Test hooks (tests/test_gpu_distributed.py runs the N=2 launch on a 1-GPU box): AM_BENCH_DEVICE pins every
    # rank to one device, AM_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on the same GPU.
    device_index = int(os.environ.get("AM_BENCH_DEVICE", local_rank))
    backend = os.environ.get("AM_BENCH_BACKEND", "nccl")
    if torch.cuda.device_count() <= device_index:
        sys.exit(f"rank {rank}: cuda:{device_index} does not exist ({torch.cuda.device_count()} GPUs visible) - "
                 f"--gpus {args.gpus} needs one GPU per rank")
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        # a collective that does not complete (a rank died, a fabric problem) becomes an error after ten minutes instead of a
        # hang that only the caller's own time limit ends
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("AM_BENCH_COLLECTIVE_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    import audio_metrics_amd as am
    from audio_metrics_amd import hip_ops as ops
    from audio_metrics_amd.distributed import enable_bulk_communicator, evaluate_sharded, shard_bounds, warm_up_communicators
    am._lib.load()                                       # no HIP library -> fail here, loudly
    n_ranks_seen = 1
    if world > 1:
        if args.bulk_communicator:
            enable_bulk_communicator()
        # every communicator exists and has carried a collective before anything is timed (also with --warmup 0); the
        # all-reduce of ones is what the line reports as n_ranks_seen
        n_ranks_seen = warm_up_communicators(dev)
        if n_ranks_seen != world:
            sys.exit(f"the process group connected {n_ranks_seen} ranks, the launcher promised {world}")
    import inputs as gi

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == "e2e":
        run_e2e(args, am, dev, world, rank, fence, n_ranks_seen)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    n, d, k = args.rows, args.dim, args.nearest_k
    lo, hi = shard_bounds(n, world, rank)

    def make_sets(kind):
        ref_h, cand_h = gi.bench_pair(kind, n, d)        # same seed on every rank: identical full sets
        return torch.as_tensor(ref_h).to(dev), torch.as_tensor(cand_h).to(dev)

    ref, cand = make_sets(args.data)
    ref_l, cand_l = ref[lo:hi], cand[lo:hi]              # this rank's row shard (as its embedder would produce)

    # the sharding rule is fixed (shard_bounds), so every rank knows every shard's size: no count exchange per step
    shard_counts = ([shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)],) * 2

    def step():
        return evaluate_sharded(ref_l, cand_l, metrics=("fad", "kd", "prdc"), nearest_k=k, shard_counts=shard_counts,
                                c_entry=args.c_entry and world > 1)

    # ---- the timed region: K plain steps, nothing else (no event brackets, no statistics kernels)
    for _ in range(args.warmup):
        result = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- with several ranks: the OTHER schedule timed the same way, so that one driver run says which is faster
    # (VERDICT r5 next-5b): `value` stays the schedule the flags chose, the line carries both as python_schedule_ms / c_entry_ms
    other_schedule_ms = other_schedule_note = None
    if world > 1:
        def other_step():
            return evaluate_sharded(ref_l, cand_l, metrics=("fad", "kd", "prdc"), nearest_k=k, shard_counts=shard_counts,
                                    c_entry=not args.c_entry)
        try:
            for _ in range(max(1, args.warmup)):
                other_result = other_step()
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                other_result = other_step()
            fence()
            t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            other_schedule_ms = float(t.item()) / args.steps * 1e3
            agree = all(abs(other_result[key] - result[key]) <= 1e-9 * max(1.0, abs(result[key])) for key in result)
            other_schedule_note = "same values as the timed schedule" if agree else f"VALUES DIFFER: {other_result} vs {result}"
        except Exception as exc:                         # the line with `value` must survive a failure of the comparison run
            other_schedule_note = f"the other schedule failed: {type(exc).__name__}: {exc}"[:400]

    # ---- a second, UNTIMED pass of the same step with the clocks on, all HIP events on the stream the kernels run on:
    # KernelTimer brackets each C-ABI entry point from the host side (with one rank the timed step is ONE entry point,
    # am_evaluate_f32; the same chain issued entry point by entry point - fused=False - gives the per-entry figures);
    # the library's kernel clock brackets the two tile kernels themselves (the durations `rocprofv3 --kernel-trace
    # --stats` reports for them); the filter statistics count what the f16 filters left for the exact kernels.
    probe_steps = max(2, min(args.steps, 5))
    ops.kernel_clock_enable(True)
    ops.filter_stats_enable(dev, True)
    for kid in (ops.KERNEL_KNN, ops.KERNEL_PRDC_CROSS, ops.KERNEL_KNN_VERIFY, ops.KERNEL_PRDC_VERIFY):
        ops.kernel_clock_read(kid)
    with ops.KernelTimer() as timer:
        for _ in range(probe_steps):
            probe_result = evaluate_sharded(ref_l, cand_l, metrics=("fad", "kd", "prdc"), nearest_k=k, fused=False)
        fence()
    kern = timer.summary()
    clocks = {name: ops.kernel_clock_read(kid) for name, kid in
              (("knn", ops.KERNEL_KNN), ("cross", ops.KERNEL_PRDC_CROSS), ("knn_verify", ops.KERNEL_KNN_VERIFY),
               ("cross_verify", ops.KERNEL_PRDC_VERIFY))}
    ops.kernel_clock_enable(False)
    filter_stats = ops.filter_stats_read(dev)
    assert probe_result == result or world > 1, (probe_result, result)       # same kernels, same values
    fence()
    t0 = time.perf_counter()
    for _ in range(probe_steps):
        evaluate_sharded(ref_l, cand_l, metrics=("fad", "kd", "prdc"), nearest_k=k, fused=False)
    fence()
    unfused_ms = (time.perf_counter() - t0) / probe_steps * 1e3

    # ---- several ranks: what every collective of one step moved and how long the compute stream stood still for it
    # (distributed.exchange_log_*), and the bandwidth of ONE all-gather of a whole set with nothing else running - the
    # number that tells a mesh (every peer's share over its own xGMI link at once) from a ring (one link's worth), which is
    # what decides whether the two 205 MB gathers hide under the sweeps at 8 ranks (tools/scale_model.py predicts both)
    exchange = None
    if world > 1:
        from audio_metrics_amd import distributed as dmod
        fence()
        dmod.exchange_log_begin()
        evaluate_sharded(ref_l, cand_l, metrics=("fad", "kd", "prdc"), nearest_k=k, shard_counts=shard_counts)
        records = dmod.exchange_log_end()
        fence()
        gathered = torch.empty((n, d), dtype=ref_l.dtype, device=dev)
        equal = all(c == shard_counts[0][0] for c in shard_counts[0])
        gather_ms = None
        if equal:                                            # (unequal shards travel padded: measured inside the step only)
            for _ in range(2):
                fence()
                t1 = time.perf_counter()
                dmod._all_gather_into(gathered, ref_l.contiguous(), world, None)
                fence()
                gather_ms = (time.perf_counter() - t1) * 1e3
        t = torch.tensor([sum(r["exposed_ms"] for r in records), gather_ms or 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        received = (world - 1) / world * n * d * ref_l.element_size()
        exchange = {"collectives": records, "exchange_exposed_ms": float(t[0].item()),
                    "gather_ms": float(t[1].item()) if gather_ms else None,
                    "gather_GBps": (received / (float(t[1].item()) * 1e-3) / 1e9) if gather_ms else None,
                    "gather_note": f"one all-gather of a whole {n} x {d} set alone on the fabric (max over ranks): bytes a rank "
                                   "RECEIVES / time.  7 peers x ~76 GB/s per direction: ~370 GB/s if RCCL drives every link at once "
                                   "(mesh), ~60 GB/s through one link (ring)",
                    "exposed_note": "per collective of one step on rank 0: HIP events on the compute stream around the point where "
                                    "it waits for the exchange (exposed_ms; gloo: host time), bytes; exchange_exposed_ms = their "
                                    "sum, max over ranks",
                    "backend": backend}

    # ---- other workloads SURVEY 8(d) asks for beside the headline one (never `value`): the filter kernels' speed depends
    # on how many pairs their error bound cannot decide, i.e. on the data and on k
    variants = {}
    if world == 1 and not args.no_variants:
        # BASELINE configs[1]: FAD + KD only (no PRDC), cold
        dt, vres = timed_steps(lambda: evaluate_sharded(ref, cand, metrics=("fad", "kd")), fence, 10, 2)
        variants["fad_kd_only"] = {"ms_per_step": dt / 10 * 1e3, "embeddings_per_s": 10 * 2 * n / dt, "result": vres,
                                   "workload": f"BASELINE.json configs[1]: FAD+KD cold evaluate() of 2x{n}x{d}, no PRDC",
                                   "result_check": check_against_fixture(vres, args.data, n, d, k)}
        # the streaming add() path the pipeline drives (24.4 k emb/s without the store, 0.7-1.8 k with it)
        variants["stream_add_32"] = {"store": stream_add(am, ref, 32, True), "no_store": stream_add(am, ref, 32, False),
                                     "reference_cpu_embeddings_per_s": {"no_store": 24400, "store": [700, 1800]},
                                     "what": "AudioMetricsData.add() fed 32-row device batches of the reference set"}
        if d == 512:
            # VGGish's width (reference embedders/vggish.py:5-33: 128-d; BASELINE configs[0]'s shape) at the headline row count:
            # the tile kernels' MFMA work shrinks with D, their epilogues do not
            vr, vc = (torch.as_tensor(a).to(dev) for a in gi.bench_pair("randn", n, 128))
            ops.filter_stats_read(dev)
            dt, vres = timed_steps(lambda: evaluate_sharded(vr, vc, metrics=("fad", "kd", "prdc"), nearest_k=k), fence, 5, 2)
            stats = ops.filter_stats_read(dev)
            variants["vggish_128"] = {"ms_per_step": dt / 5 * 1e3, "embeddings_per_s": 5 * 2 * n / dt,
                                      "workload": f"FAD+KD+PRDC(k={k}) cold evaluate() of 2x{n}x128 randn sets (VGGish width)",
                                      "filter": per_step(stats, 7), "result": vres,
                                      "knn_path": ops.knn_path(n, n, 128, k), "prdc_path": ops.prdc_path(n, n, 128),
                                      "result_check": check_against_fixture(vres, "randn", n, 128, k)}
            del vr, vc
            # float64 rows of width 64 - what n_pca = 64 hands on (reference projection.py:20-21; every stage then runs in float64,
            # data.py:39-44, prdc.py:12-13, kd.py:112-116): the *_f64 entry points, the large sets through the f16 filter with an f64
            # evaluation of the undecided pairs; checked against the reference's own float64 values / the blocked oracle in float64
            vr, vc = (torch.as_tensor(a).to(dev) for a in gi.pair64("randn", gi.BENCH_SEED, n, n, 64))
            ops.filter_stats_read(dev)
            dt, vres = timed_steps(lambda: evaluate_sharded(vr, vc, metrics=("fad", "kd", "prdc"), nearest_k=k), fence, 5, 2)
            stats = ops.filter_stats_read(dev)
            variants["pca64_f64"] = {"ms_per_step": dt / 5 * 1e3, "embeddings_per_s": 5 * 2 * n / dt, "dtype": "f64",
                                     "filter": per_step(stats, 7),      # incl. bound_ratio_max: the float64 route's widened bound, measured
                                     "workload": f"FAD+KD+PRDC(k={k}) cold evaluate() of 2x{n}x64 FLOAT64 randn sets (the shape n_pca = 64 leaves)",
                                     "result": vres, "result_check": check_against_fixture(vres, "randn", n, 64, k, "_f64")}
            del vr, vc
        for name, kind, vk in (("clap_shaped_k5", "clap", 5), ("randn_k10", "randn", 10), ("clap_shaped_k10", "clap", 10)):
            if (kind, vk) == (args.data, k):
                continue
            vr, vc = (ref, cand) if kind == args.data else make_sets(kind)
            ops.filter_stats_read(dev)
            dt, vres = timed_steps(lambda: evaluate_sharded(vr, vc, metrics=("fad", "kd", "prdc"), nearest_k=vk), fence, 3, 1)
            stats = ops.filter_stats_read(dev)
            variants[name] = {"ms_per_step": dt / 3 * 1e3, "embeddings_per_s": 3 * 2 * n / dt,
                              "ms_note": "timed with the filter-statistics kernels on (they add ~0.05 ms per step)",
                              "filter": per_step(stats, 4), "result": vres,
                              "knn_path": ops.knn_path(n, n, d, vk), "prdc_path": ops.prdc_path(n, n, d),
                              "result_check": check_against_fixture(vres, kind, n, d, vk)}
            del vr, vc
    ops.filter_stats_enable(dev, False)

    if rank == 0:
        rows_local = hi - lo
        # Which form of the two PRDC tile kernels ran (am_knn_path / am_prdc_path: 0 exact general, 1 exact symmetric,
        # 2 / 3 f16 filter sweep (128 / 256-row engine) + exact f32 verification of the undecided pairs).
        def entry(*names):                        # the plain or the prepared-set form of an entry point, whichever ran
            return next((n for n in names if n in kern), names[0])

        part_form = "am_knn_sym_part_f32" in kern or "am_knn_sym_part_prepared_f32" in kern
        knn_entry = (entry("am_knn_sym_part_prepared_f32", "am_knn_sym_part_f32") if part_form
                     else entry("am_knn_radii_prepared_f32", "am_knn_radii_f32"))
        cross_entry = entry("am_prdc_counts_prepared_f32", "am_prdc_counts_f32")
        knn_path = ops.knn_path(n, n, d, k) if (world == 1 or part_form) else 0
        cross_path = ops.prdc_path(rows_local, n, d)
        engine = {1: "pstat", 2: "pstat64"}.get(ops.filter_engine(d), "wide")   # path 3: operand-stationary (512 / 2 x 256 threads) or streamed
        knn_kernel = {0: "knn_partial_kernel", 1: "knn_sym_kernel", 2: "knn_fast_kernel", 3: f"knn_{engine}_kernel"}[knn_path]
        cross_kernel = {0: "prdc_cross_kernel", 2: "cross_fast_kernel", 3: f"cross_{engine}_kernel"}[cross_path]
        peak_of = {0: F32_MFMA_PEAK_TFLOPS, 1: F32_MFMA_PEAK_TFLOPS, 2: F16_MFMA_PEAK_TFLOPS, 3: F16_MFMA_PEAK_TFLOPS}
        mfma_of = {0: "v_mfma_f32_32x32x2_f32", 1: "v_mfma_f32_32x32x2_f32", 2: "v_mfma_f32_32x32x16_f16",
                   3: "v_mfma_f32_32x32x16_f16"}

        def per_launch(name, entry):
            launches, total = clocks[name]
            if launches:
                return total / launches, launches / probe_steps
            calls, ms = kern[entry]
            return ms / calls, calls / probe_steps

        knn_ms, knn_lps = per_launch("knn", knn_entry)
        cross_ms, cross_lps = per_launch("cross", cross_entry)
        kcalls, kms = kern[knn_entry]
        ccalls, cms = kern[cross_entry]
        # ALGORITHMIC work of one launch (one dot product per (row, column) pair, no symmetry credit):
        # 2 * rows_of_this_rank * N * D flop.  The symmetric forms (paths 1, 2, 3) multiply a cyclic half of the tile
        # pairs - self distances are bitwise symmetric - so they EXECUTE about half of it.
        flop_alg = 2.0 * rows_local * n * d
        t_tiles = (n + 255) // 256 if knn_path == 3 else (n + 127) // 128
        knn_exec = ((t_tiles // 2 + 1) / t_tiles) if knn_path in (1, 2, 3) else 1.0
        stamp = library_stamp()

        def roof(kernel, path, ms, lps, exec_frac, entry, entry_ms):
            peak = peak_of[path]
            algorithmic = flop_alg / (ms * 1e-3) / 1e12
            executed = algorithmic * exec_frac
            # `frac` is the MATRIX-PIPE figure: flops the kernel really issues / time / peak.  The algorithmic rate (SURVEY
            # 8(d) counts every (row, column) pair) is kept beside it; for the symmetric sweep it is about twice as high.
            traffic, traffic_source = recorded_traffic(kernel, stamp) if world == 1 else (None, {"state": "recorded for one rank only"})
            return {"bound": "mfma", "achieved": executed, "peak": peak, "unit": "TFLOP/s", "frac": executed / peak,
                    "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": kernel, "mfma": mfma_of[path], "entry_point": entry, "launch_ms": ms, "launches_per_step": lps,
                    "entry_ms": entry_ms, "executed_flop_per_launch": flop_alg * exec_frac,
                    "algorithmic_flop_per_launch": flop_alg, "algorithmic_tflops": algorithmic,
                    "algorithmic_frac": algorithmic / peak}

        knn_roof = roof(knn_kernel, knn_path, knn_ms, knn_lps, knn_exec, knn_entry, kms / kcalls)
        cross_roof = roof(cross_kernel, cross_path, cross_ms, cross_lps, 1.0, cross_entry, cms / ccalls)
        # dominant kernel = the one with the larger share of the step
        main, other = (knn_roof, cross_roof) if knn_ms * knn_lps >= cross_ms * cross_lps else (cross_roof, knn_roof)
        main["note"] = (
            "achieved = flops the kernel EXECUTES per launch (the symmetric k-NN sweep multiplies the cyclic half of the "
            "tile pairs) / launch_ms (hipEvents around the kernel inside the library, on its stream; compare rocprofv3's "
            "average for it); peak = dense MFMA peak of the instruction the kernel issues; algorithmic_* counts every "
            "(row, column) pair as SURVEY 8(d) does.  Path 2/3 kernels are FILTERS: they evaluate every pair on the f16 "
            "matrix cores with a proven error bound and queue the few pairs the bound cannot decide; those are "
            "re-evaluated with the exact f32 fmaf chain (verify kernels, listed under other_kernels), so the outputs are "
            "bit-identical to the exact f32 kernels'.")
        if knn_path == 3 or cross_path == 3:
            # QUOTED from an earlier microbenchmark run, NOT measured by this command: a loop of nothing but this MFMA on
            # random operands (tools/ubench/pstat.hip, ablation 7; tools/ubench/pskew.hip part 1 in round 6: 128-140 cycles per
            # 4 MFMAs of a SIMD at 1.58-1.78 GHz)
            main["sustained_mfma_only_quoted"] = {"tflops": 1720.0, "frac_of_peak": 0.69, "measured_in_this_run": False,
                                                  "source": "profiles/r5/ubench_pstat_abl.txt (round 5, another box); profiles/r6/ubench_pskew_v3.txt",
                                                  "note": "a constant quoted for context: what the f16 matrix pipe sustained on this pool at "
                                                          "the clock its power draw allowed; frac above is priced against the nominal peak"}
        verify = {}
        for name, label in (("knn_verify", "knn_fast_verify_kernel"), ("cross_verify", "cross_verify_kernel")):
            launches, total = clocks[name]
            if launches:
                verify[label] = {"launch_ms": total / launches, "launches_per_step": launches / probe_steps,
                                 "bound": "hbm/L2 gather: two 4*D-byte rows per surviving pair, one fmaf chain per lane"}
        # the whole step against the matrix peak (SURVEY 8(d) algorithmic work: statistics of both sets, the KD Gram blocks and
        # the three pairwise passes, no symmetry credit; the Frechet solve - 10 iterations x 8e8 f64 flop - is left out)
        step_flop = 2 * 2.0 * n * d * d + 100 * 3 * 2.0 * 1000 * 1000 * d + 3 * 2.0 * n * n * d
        step_rate = step_flop / (elapsed / args.steps) / 1e12
        roofline_step = {"algorithmic_flop_per_step": step_flop, "algorithmic_tflops": step_rate,
                         "peak": F16_MFMA_PEAK_TFLOPS, "frac": step_rate / F16_MFMA_PEAK_TFLOPS, "frac_of_f32_mfma_peak": step_rate / F32_MFMA_PEAK_TFLOPS,
                         "note": "algorithmic flop of one step (2*2*N*D^2 + S*3*2*m^2*D + 3*2*N^2*D) / ms_per_step, all ranks together, "
                                 "against the dense f16 MFMA peak of ONE GPU x n_gpus; the tile kernels are f16 filters with exact "
                                 "f32 verification, which is how the rate can exceed the f32 MFMA peak"}
        if world > 1:
            roofline_step["peak"] = F16_MFMA_PEAK_TFLOPS * world
            roofline_step["frac"] = step_rate / (F16_MFMA_PEAK_TFLOPS * world)
            roofline_step["frac_of_f32_mfma_peak"] = step_rate / (F32_MFMA_PEAK_TFLOPS * world)
        entry_sum = sum(tot for name, (c, tot) in kern.items() if name != "am_frechet_enqueue_f64") / probe_steps
        host_gap = {"ms_per_step": elapsed / args.steps * 1e3, "sum_main_stream_entry_ms": entry_sum,
                    "ms_per_step_minus_entries": elapsed / args.steps * 1e3 - entry_sum,
                    "unfused_ms_per_step": unfused_ms,
                    "what": ("timed step (one am_evaluate_f32 call + one read-back when n_gpus = 1) against the summed GPU time of "
                             "the main-stream entry points of the same chain; unfused_ms_per_step = the chain issued entry by "
                             "entry from Python with its three read-backs, untimed pass")}
        out = {
            "metric": METRIC,
            "value": args.steps * 2 * n / elapsed,
            "unit": "embeddings/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "library_sources_sha256": library_stamp(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"FAD+KD+PRDC(k={k}) cold evaluate() of 2x{n} CLAP-{d} f32 embedding sets resident in HBM "
                                   "(BASELINE.json configs[2])",
                       "n_ref": n, "n_cand": n, "dim": d, "nearest_k": k, "kd_subsets": 100, "kd_subset_size": 1000,
                       "inputs": ("numpy PCG64 seed %d: reference randn, candidate randn*1.05+0.05" % gi.BENCH_SEED) if args.data == "randn"
                                 else ("numpy PCG64 seed %d: unit-norm rows of randn+0.5 / randn+0.55 (CLAP-shaped)" % gi.BENCH_SEED),
                       "sharding": f"rows/{world}",
                       "schedule": ("am_evaluate_sharded_f32 (one library call per rank, collectives as hooks)" if args.c_entry and world > 1
                                    else "am_evaluate_f32 (one library call)" if world == 1
                                    else "distributed.evaluate_sharded (Python exchange schedule over the split entry points)"),
                       "arithmetic": "PRDC: results are the exact f32 values (bit-identical to the f32-MFMA kernels); the tile kernels "
                                     "pre-filter on f16 MFMA with f32 accumulation where am_knn_path/am_prdc_path >= 2 and every "
                                     "undecided pair is re-evaluated with the exact f32 fmaf chain.  KD (subsets of >= 512 rows, "
                                     "D >= 128, degree 3): every gathered row is split into two f16 planes (hi = rn16(x), lo = "
                                     "rn16(x - hi), ~22 significant bits) and the dot products are <hi,hi'> + <lo,hi'> + <hi,lo'> "
                                     "on the f16 MFMA with f32 accumulation: error of a dot product <= ~3 * 2^-22 |x||y| (one f32 "
                                     "rounding's worth; the reference forms them in f32 too), kernel values and all sums in f64.  "
                                     "Statistics: exact f32 products on the f32 MFMA, f64 accumulation.  FAD: f64 throughout.",
                       "persisting": "between steps nothing of the RESULT is cached (statistics, radii, counts, KD values and the "
                                     "Frechet solve are recomputed); what survives is allocation and setup: torch's caching "
                                     "allocator and the evaluate workspace (hip_ops._eval_workspace), the KD subset index table "
                                     "on the device (a pure function of (n1, n2, S, m, seed): metrics/kd.py device_subset_indices), "
                                     "code objects and - with several ranks - the communicators.  `first_call` times a step "
                                     "without them."},
            "roofline": main,
            "roofline_step": roofline_step,
            "other_tile_kernel": other,
            "other_kernels": verify,
            "filter": dict(per_step(filter_stats, probe_steps), knn_path=knn_path, prdc_path=cross_path,
                           note="per step: pairs the f16 filters queued / pairs evaluated exactly / rows or calls that fell "
                                "back to the exact f32 kernels (am_filter_stats_enable)"),
            "kernels_ms_per_call": {name: tot / c for name, (c, tot) in sorted(kern.items())},
            "kernels_calls_per_step": {name: c / probe_steps for name, (c, tot) in sorted(kern.items())},
            "kernels_note": ("from the UNTIMED second pass (fused=False: the same chain issued entry point by entry point): "
                             "event-to-event time of each C-ABI entry point on its own stream; am_frechet_enqueue_f64 only "
                             "enqueues the solve on a side stream (it runs UNDER the PRDC kernels), so it is left out of the sum"),
            "host_gap": host_gap,
            "result": result,
            "result_check": check_against_fixture(result, args.data, n, d, k),
            "variants": variants,
        }
        if exchange is not None:
            out["exchange"] = exchange
        if other_schedule_note is not None:
            out["schedules_check"] = other_schedule_note
        if other_schedule_ms is not None:
            mine = out["ms_per_step"]
            out["c_entry_ms"] = mine if args.c_entry else other_schedule_ms
            out["python_schedule_ms"] = other_schedule_ms if args.c_entry else mine
            out["schedules_note"] = ("both exchange schedules timed back to back on the same sets, K steps each, max over ranks: "
                                     "c_entry_ms = am_evaluate_sharded_f32 (one C call per rank, collectives as hooks over this "
                                     "process group - equal shares: one in-place all_gather_into_tensor per row gather), "
                                     "python_schedule_ms = distributed.evaluate_sharded over the split entry points; `value` is "
                                     "the one the flags chose (" + ("--c-entry" if args.c_entry else "default: the Python schedule") + ")")
        if cold is not None:
            out["process_cold"] = cold
        out["scale_model"] = scale_model_prediction(world, n, d, k, out["ms_per_step"])
        if world == 1:
            out["warm"] = warm_evaluate(am, ref, cand, k, max(1, min(args.steps, 3)))
            out["first_call"] = first_call(step, fence)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ref, cand, k)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def scale_model_prediction(world, n, d, k, measured_ms):
    """tools/scale_model.py's prediction for this rank count (one GPU emulating a rank's compute segments, the collectives
    simulated under two fabric assumptions) beside the measured step: one run of the driver's scaling bench then says which
    assumption holds.  None when no prediction for this workload is committed."""
    path = os.path.join(ROOT, "profiles", "scale_model.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    if [table.get("rows"), table.get("dim"), table.get("nearest_k")] != [n, d, k]:
        return None
    row = table.get("worlds", {}).get(str(world))
    if row is None:
        return None
    mesh, ring = row.get("f32_first/mesh", {}), row.get("f32_first/ring", {})
    return {"source": "profiles/scale_model.json (tools/scale_model.py on ONE GPU, library sources " + str(table.get("library", "?"))[:12] + ")",
            "predicted_ms_mesh": mesh.get("step_ms"), "predicted_ms_ring": ring.get("step_ms"),
            "predicted_exposed_ms_mesh": mesh.get("exposed_ms"), "predicted_exposed_ms_ring": ring.get("exposed_ms"),
            "measured_ms": measured_ms,
            "note": "mesh: every peer's share over its own xGMI link at once at 70 % of 76.5 GB/s; ring: the same volumes through "
                    "one link per direction at 80 %; compute segments timed on one GPU, collectives simulated in issue order"}


def per_step(stats, steps):
    """Counters per step; the MEASURED error-bound ratio (a maximum, not a count) is passed through."""
    return {key: (value if key == "bound_ratio_max" else value / steps) for key, value in stats.items()}


def run_e2e(args, am, dev, world, rank, fence, n_ranks_seen=1):
    """BASELINE configs[4] shape: (context, stem) audio pairs -> embedder forward on this rank's GPU -> device-side
    aggregation -> APA + FAD, one process per GPU with the statistics merged across ranks (distributed.merged_stats).
    The embedder is `SyntheticEmbedder` (LAION-CLAP and its checkpoint cannot be installed here) - a torch module with
    the embedder protocol; its forward is PyTorch-ROCm code and NOT part of the measured claim.  Reported: whole
    add_reference + evaluate wall time, and the aggregation + metric share (everything but host audio synthesis/mixing
    and the embedder forward)."""
    import random
    from audio_metrics_amd.embedders import SyntheticEmbedder
    sr, seconds = 48000, 5
    pairs = args.pairs
    lo, hi = pairs * rank // world, pairs * (rank + 1) // world
    from concurrent.futures import ThreadPoolExecutor
    synth_pool = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1), thread_name_prefix="bench-synth")
    serial = [0]

    def one_pair(index):
        # every pair from its own PCG64 stream (deterministic whatever the thread), generated in f32 directly
        return np.random.default_rng([1000 + rank, index]).standard_normal((sr * seconds, 2), dtype=np.float32)

    def audio(count):
        """`count` synthetic pairs, generated a few ahead on a thread pool so that the input generator - which is not part of
        the pipeline under test - does not dominate the wall time (it was 82 % of it)."""
        from collections import deque
        ahead = deque()
        for i in range(count):
            ahead.append(synth_pool.submit(one_pair, serial[0]))
            serial[0] += 1
            if len(ahead) >= 32:
                yield ahead.popleft().result()
        while ahead:
            yield ahead.popleft().result()

    clock = {"forward": 0.0}

    class Timed(SyntheticEmbedder):
        def forward(self, data, sr=None):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            out = super().forward(data, sr)
            torch.cuda.synchronize(dev)
            clock["forward"] += time.perf_counter() - t0
            return out

    from audio_metrics_amd import embed as am_embed
    am_embed.PROFILE = stage = {}                      # stage clock of the front end (embed.py)
    embedder = Timed(dim=512, sr=sr, device=dev)
    group = dist.group.WORLD if world > 1 else None
    # device_indices=None, the reference's default: every visible GPU inside one process (util/gpu_parallel.py:24-25); with one
    # process per GPU (process_group) this rank's device
    metric = am.AudioMetrics(metrics=["apa", "fad"], embedder=embedder, mix_function="P0", device_indices=None,
                             win_dur=float(seconds), process_group=group)
    random.seed(1234 + rank)
    host = {"t": 0.0}

    def timed_source(count):
        it = audio(count)
        while True:
            t0 = time.perf_counter()
            try:
                item = next(it)
            except StopIteration:
                return
            host["t"] += time.perf_counter() - t0
            yield item

    fence()
    t0 = time.perf_counter()
    metric.add_reference(timed_source(hi - lo))
    result = metric.evaluate(timed_source(hi - lo))
    fence()
    elapsed = time.perf_counter() - t0
    stages = ("mix_wait", "mix_cpu", "batch", "forward", "file")
    if world > 1:
        t = torch.tensor([elapsed, clock["forward"], host["t"]] + [stage.get(key, 0.0) for key in stages], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [float(v) for v in t.tolist()]
        elapsed, fwd, synth = vals[:3]
        stage = dict(zip(stages, vals[3:]))
    else:
        fwd, synth = clock["forward"], host["t"]
    if rank == 0:
        windows = 2 * pairs                       # reference + candidate pairs, one 5 s window each
        print(json.dumps({
            "metric": "APA+FAD end-to-end pairs/sec (audio -> embedder -> device aggregation -> metrics)",
            "value": windows / elapsed, "unit": "pairs/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": 1, "warmup": 0,
            "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"APA+FAD end-to-end, {pairs} reference + {pairs} candidate pairs of ({sr * seconds}, 2) f32 "
                                   "randn audio at 48 kHz (BASELINE.json configs[4]" + ("" if pairs >= 10000 else " shape, reduced pair count") + "), "
                                   "SyntheticEmbedder (NOT CLAP: laion_clap and its checkpoint are not installable here)",
                       "pairs_per_side": pairs, "sharding": f"pairs/{world}", "mix_function": "P0",
                       "embedder_devices": [str(d_) for d_ in metric._pool.devices]},
            "breakdown_s": {
                "total": elapsed, "host_audio_synthesis": synth,
                "host_mix_wait": stage.get("mix_wait", 0.0),                      # the consumer waiting for mixed windows
                "host_mix_cpu_summed_over_threads": stage.get("mix_cpu", 0.0),
                "host_batch_fill": stage.get("batch", 0.0),
                "embedder_forward_incl_h2d": fwd,
                "device_aggregate_enqueue": stage.get("file", 0.0),
                "metrics_and_rest": elapsed - synth - stage.get("mix_wait", 0.0) - stage.get("batch", 0.0)
                                    - stage.get("forward", fwd) - stage.get("file", 0.0),
                "mix_workers": am_embed.default_mix_workers(),
                "note": ("single consumer thread: host_mix_wait + host_batch_fill + embedder_forward + device_aggregate_enqueue + "
                         "metrics_and_rest + host_audio_synthesis = total; the mix functions themselves run on the worker threads")},
            "result": result}), flush=True)


if __name__ == "__main__":
    main()
