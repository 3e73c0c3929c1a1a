#!/usr/bin/env python3
"""Predicted 1/2/4/8-GPU step of bench.py (2 x N x 512, k = 5) from what CAN be measured on one GPU: a rank's compute
segments, timed with events while the other ranks' contributions are precomputed (tools/emulate_rank.py's scheme), plus a
model of the collectives between them.  Nothing here has run on more than one GPU - the model exists so that the first
real scaling curve has something to be compared with, and so that the exposed part of the exchange is a number.

Collective model (assumptions, stated in the output):
  * xGMI: 7 links per GPU, 153 GB/s per link bidirectional = 76.5 GB/s per direction (the figure this project was given for
    MI355X; MI355X_MICROARCH.md has no xGMI section).  `mesh`: an all-gather of B bytes in total delivers each peer's B / W
    over its own link, all links at once, at 70 % of the link rate; an all-reduce of S bytes moves 2 (W - 1) / W x S per
    rank over (W - 1) links.  `ring`: the same volumes through ONE link per direction at 80 % (what a ring-only RCCL
    schedule gives on a point-to-point fabric) - the pessimistic end.
  * every collective costs 25 us of launch / synchronisation on top (RCCL small-message latency on one node).
  * one communication stream, collectives in issue order (distributed.evaluate_sharded's order, one communicator); a
    collective starts when it has been issued, its input exists and the previous collective is done; compute waits where
    distributed.py waits.
Orders compared:
  * `f32_first` (distributed.py as it is): rows of the reference set gathered first (N x D x 4 bytes), the candidate rows
    behind the reference set's bounds exchange.
  * `f16_first` (not built): the prepared f16 copies + norms (N x D x 2) first, so that the filter
    sweeps start after half the bytes; the f32 rows - needed only by the exact verification and the kernel distance - behind.

    python tools/scale_model.py > profiles/scale_model.json        (on the GPU box; AB_ROWS=1000000 for configs[3]);
    bench.py --gpus N prints this file's prediction for N beside its measured step (`scale_model` in the line)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_metrics_amd import hip_ops as ops  # noqa: E402
from audio_metrics_amd.metrics.kd import subset_indices  # noqa: E402

LINK = 76.5e9            # bytes / s per direction per xGMI link
LAT = 25e-6              # s per collective


def measure(n, d, k, world, dev):
    """ms of each compute segment of rank 0 (events on the compute stream, best of 4)."""
    gen = torch.Generator(device="cuda").manual_seed(0)
    ref = torch.randn(n, d, generator=gen, device="cuda")
    cand = torch.randn(n, d, generator=gen, device="cuda") * 1.05 + 0.05
    if os.environ.get("AB_DATA", "randn") == "clap":              # CLAP-shaped: unit-norm rows with offsets 0.5 / 0.55
        ref, cand = ref + 0.5, (cand - 0.05) / 1.05 + 0.55
        ref, cand = ref / ref.norm(dim=1, keepdim=True), cand / cand.norm(dim=1, keepdim=True)
    rows = n // world
    pre = {}
    for name, x in (("ref", ref), ("cand", cand)):
        bounds = torch.cat([ops.knn_bounds(x, k, p * rows, rows if p < world - 1 else n - p * rows) for p in range(world)])
        lists = torch.stack([ops.knn_sym_part(x, k, p, world, bounds) for p in range(world)]) if world > 1 else None
        pre[name] = (bounds, lists)
    mean_r, mean_c = ref.double().mean(0), cand.double().mean(0)
    idx1, idx2 = subset_indices(n, n, 100, 1000, 1234)
    i1, i2 = ops.upload_host_array(idx1[0::world], dev), ops.upload_host_array(idx2[0::world], dev)
    segs = {}

    def seg(name, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        segs.setdefault(name, []).append((a, b))
        return out

    best = {}
    for rep in range(5):
        segs.clear()
        seg("colsum", lambda: (ops.colsum(ref[:rows]), ops.colsum(cand[:rows])))
        seg("scatter", lambda: (ops.scatter(ref[:rows], mean_r), ops.scatter(cand[:rows], mean_c)))
        prep = {}
        radii = {}
        for name, x in (("ref", ref), ("cand", cand)):
            prep[name] = seg(f"prepare_{name}", lambda: ops.prepare(x))
            seg(f"prepare_shard_{name}", lambda: ops.prepare(x[:rows]))                       # (f16_first prepares the own rows only)
            if world > 1:
                bounds, lists = pre[name]
                seg(f"bounds_{name}", lambda: ops.knn_bounds(x, k, 0, rows, prepared=prep[name]))
                ops.kernel_clock_enable(True)
                for kid in range(4):
                    ops.kernel_clock_read(kid)
                seg(f"part_{name}", lambda: ops.knn_sym_part(x, k, 0, world, bounds, prepared=prep[name]))
                torch.cuda.synchronize()
                sweep = ops.kernel_clock_read(ops.KERNEL_KNN)[1]
                ops.kernel_clock_enable(False)
                best[f"sweep_{name}"] = min(best.get(f"sweep_{name}", 1e9), sweep)
                radii[name] = seg(f"finish_{name}", lambda: ops.knn_lists_finish(lists, x, k))
            else:
                radii[name] = seg(f"part_{name}", lambda: ops.knn_radii(x, k, prepared=prep[name]))
        seg("counts", lambda: ops.prdc_reduce(*ops.prdc_counts(ref[:rows], cand, radii["ref"][:rows], radii["cand"],
                                                               prepared_ref=prep["ref"].rows(0, rows), prepared_cand=prep["cand"])))
        seg("kd", lambda: ops.kd_poly(cand, ref, i1, i2, 1.0 / d, 1, 3))
        torch.cuda.synchronize()
        if rep == 0:
            continue
        for name, evs in segs.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            best[name] = min(best.get(name, 1e9), ms)
    return best


def simulate(seg, n, d, k, world, order, fabric):
    """-> (step_ms, exposed_ms, bytes per rank inbound).  Two clocks: compute stream, communication stream."""
    if world == 1:
        total = sum(v for key, v in seg.items() if not key.startswith(("sweep_", "prepare_shard_")))
        return total, 0.0, 0
    eff, links = (0.7, world - 1) if fabric == "mesh" else (0.8, 1)

    def gather_s(total_bytes):
        return LAT + (total_bytes * (world - 1) / world) / (LINK * eff * links)

    def reduce_s(nbytes):
        return LAT + (2.0 * nbytes * (world - 1) / world) / (LINK * eff * links)

    t_comp, t_comm = 0.0, 0.0          # when each stream is free
    inbound = 0

    def compute(ms, after=0.0):
        nonlocal t_comp
        t_comp = max(t_comp, after) + ms * 1e-3
        return t_comp

    def collective(seconds, ready, nbytes=0):
        nonlocal t_comm, inbound
        t_comm = max(t_comm, ready) + seconds
        inbound += int(nbytes * (world - 1) / world)
        return t_comm

    rows_f32, rows_f16, norms = n * d * 4, n * d * 2, n * 4
    lists = n * (6 if k <= 5 else 11) * 4
    done_colsum = compute(seg["colsum"])
    c_means = collective(reduce_s(2 * d * 8), done_colsum, 2 * d * 8)
    part_verify = {s: max(seg[f"part_{s}"] - seg[f"sweep_{s}"], 0.0) for s in ("ref", "cand")}
    if order == "f32_first":
        g_ref = collective(gather_s(rows_f32), 0.0, rows_f32)                      # issued at once (input: the local rows)
        done_scatter = compute(seg["scatter"], c_means)
        c_cov = collective(reduce_s(2 * d * d * 8), done_scatter, 2 * d * d * 8)
        t = compute(seg["prepare_ref"], g_ref)
        t = compute(seg["bounds_ref"])
        c_b = collective(gather_s(norms), t, norms)
        g_cand = collective(gather_s(rows_f32), t, rows_f32)                       # issued behind the bounds exchange
        t = compute(seg["part_ref"], c_b)
        c_l = collective(gather_s(lists * world), t, lists * world)
        t = compute(seg["finish_ref"], c_l)
        t = compute(seg["prepare_cand"], g_cand)
        t = compute(seg["bounds_cand"])
        c_b = collective(gather_s(norms), t, norms)
        t = compute(seg["part_cand"], c_b)
        c_l = collective(gather_s(lists * world), t, lists * world)
        t = compute(seg["finish_cand"], c_l)
    else:
        done_scatter = compute(seg["scatter"], c_means)
        c_max = collective(reduce_s(32), 0.0, 32)                                  # agreed f16 scale: MAX of four values
        t = compute(seg["prepare_shard_ref"], c_max)
        h_ref = collective(gather_s(rows_f16 + norms), t, rows_f16 + norms)
        t = compute(seg["prepare_shard_cand"])
        c_cov = collective(reduce_s(2 * d * d * 8), max(done_scatter, t), 2 * d * d * 8)
        t = compute(seg["bounds_ref"], h_ref)
        c_b = collective(gather_s(norms), t, norms)
        h_cand = collective(gather_s(rows_f16 + norms), t, rows_f16 + norms)
        g_ref = collective(gather_s(rows_f32), t, rows_f32)
        g_cand = collective(gather_s(rows_f32), t, rows_f32)
        t = compute(seg["sweep_ref"], c_b)
        t = compute(part_verify["ref"], g_ref)                                     # the exact verification reads f32 rows
        c_l = collective(gather_s(lists * world), t, lists * world)
        t = compute(seg["bounds_cand"], h_cand)
        c_b2 = collective(gather_s(norms), t, norms)
        t = compute(seg["finish_ref"], c_l)
        t = compute(seg["sweep_cand"], c_b2)
        t = compute(part_verify["cand"], g_cand)
        c_l = collective(gather_s(lists * world), t, lists * world)
        t = compute(seg["finish_cand"], c_l)
    t = compute(seg["counts"])
    c_cnt = collective(reduce_s(n * 4 + 8), t, n * 4 + 8)
    t = compute(seg["kd"])
    c_kd = collective(reduce_s(800), t, 800)
    end = max(t, c_cnt, c_kd, c_cov) + 60e-6                                        # one read-back
    busy = sum(v for key, v in seg.items() if not key.startswith(("sweep_", "prepare_shard_"))) * 1e-3
    if order == "f16_first":
        busy += (seg["prepare_shard_ref"] + seg["prepare_shard_cand"] - seg["prepare_ref"] - seg["prepare_cand"]) * 1e-3
    return end * 1e3, (end - busy) * 1e3, inbound


def library_stamp():
    """sources_sha256 of the library the segments were timed on (what bench.py's roofline.traffic_source carries too)."""
    from audio_metrics_amd import _build, _lib
    try:
        with open(_build._stamp_path(_lib.library_path())) as fh:
            return json.load(fh).get("sources_sha256")
    except (OSError, ValueError):
        return None


def main():
    n, d, k = int(os.environ.get("AB_ROWS", "100000")), 512, int(os.environ.get("AB_K", "5"))
    dev = torch.device("cuda:0")
    out = {"workload": f"bench.py: FAD+KD+PRDC(k={k}) cold evaluate of 2 x {n} x {d} ({os.environ.get('AB_DATA', 'randn')})", "measured_on": "ONE MI355X (rank 0 emulated)",
           "assumptions": {"xgmi_link_GBps_per_direction": LINK / 1e9, "links_per_gpu": 7, "collective_latency_us": LAT * 1e6,
                           "mesh": "all links at once at 70 % of the link rate", "ring": "one link per direction at 80 %"},
           "rows": n, "dim": d, "nearest_k": k, "data": os.environ.get("AB_DATA", "randn"),
           "library": library_stamp(), "worlds": {}}
    for world in [int(w) for w in os.environ.get("AB_WORLDS", "1,2,4,8").split(",")]:
        seg = measure(n, d, k, world, dev)
        entry = {"compute_segments_ms": {key: round(v, 4) for key, v in sorted(seg.items())}}
        for order in ("f32_first", "f16_first"):
            for fabric in ("mesh", "ring"):
                step, exposed, inbound = simulate(seg, n, d, k, world, order, fabric)
                entry[f"{order}/{fabric}"] = {"step_ms": round(step, 3), "exposed_ms": round(exposed, 3),
                                              "inbound_MB_per_rank": round(inbound / 1e6, 1),
                                              "embeddings_per_s": round(2 * n / (step * 1e-3))}
        out["worlds"][str(world)] = entry
        print(f"world={world}: " + "  ".join(f"{key} {v['step_ms']:.2f} ms (exposed {v['exposed_ms']:.2f})" for key, v in entry.items()
                                             if isinstance(v, dict) and "step_ms" in v), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
