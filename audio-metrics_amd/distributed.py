"""Row-sharded evaluation across the GPUs of one node (one process per GPU,
torch.distributed over RCCL / xGMI).

Embedding sets are partitioned by contiguous sample-index ranges; every metric
is a reduction over rows plus at most one exchange step:

  stats / FAD   local column sums of BOTH sets -> one all-reduce -> global means;
                local centred scatters of both sets -> one all-reduce -> covariances;
                the 2 MB Newton-Schulz problem is then solved redundantly on every rank,
                on a side stream under the PRDC kernels.
  KD            all-gather of the embeddings (needed by PRDC anyway): the reference rows before the
                statistics kernels, the candidate rows behind the reference set's bounds exchange
                (one communicator: they travel under the reference set's sweep); subsets are dealt
                round-robin to ranks; the S partial results are summed.
  PRDC          each rank owns a row block of both sets against the gathered columns:
                radii stay local -> all-gather; column counts and the two local row totals
                travel in one int32 buffer -> one all-reduce.

All collectives carry exact integers or f64 partial sums whose reduction order
is fixed by the backend, and with world_size == 1 the same code is the
single-GPU ``evaluate``.  A rank may hold zero rows of a set (fewer samples than
ranks): it contributes zeros and takes part in every collective.  The local
compute object ``ops`` defaults to the HIP library (``hip_ops``); tests inject a
CPU oracle-backed stand-in to exercise the sharding and collective logic under gloo.
"""
import numpy as np
import torch
import torch.distributed as dist

from .metrics.kd import subset_indices, KID_SUBSETS, KID_SUBSET_SIZE, KID_DEGREE, KID_COEF0


def _plain_upload(array, dev):
    """Stand-in for hip_ops.upload_host_array when the compute object is a CPU stand-in (tests)."""
    return torch.as_tensor(np.ascontiguousarray(array))


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


# Test hook (tests/test_gpu_distributed.py): with True, a process group of ONE rank still issues every collective and takes
# the partitioned k-NN path.  A 1-GPU box cannot host two RCCL ranks, so this is how the RCCL branch of this module
# (all_gather_into_tensor, asynchronous work handles, the ordering of the communication stream against the compute and
# side streams) is executed before the first multi-GPU run.  Never set by the product.
COLLECTIVES_AT_WORLD_ONE = False
LAST_C_ENTRY_COLLECTIVES = None


def _alone(world):
    """No exchange needed: one rank (and the test hook is off)."""
    if world == 1 and COLLECTIVES_AT_WORLD_ONE and dist.is_available() and dist.is_initialized():
        return False
    return world == 1

# ---- optional exchange log (bench.py --gpus N): what every collective of an evaluate moved and how long the COMPUTE stream
# stood still for it.  Off by default (no events, no host clocks).  exchange_log_begin() ... exchange_log_end() around one
# evaluate_sharded() returns a list of records {name, kind, bytes, exposed_ms, host_ms}:
#   exposed_ms  HIP events on the compute stream right before / behind the point where it waits for the collective - the time
#               the exchange is NOT hidden under kernels (for a blocking backend - gloo - the host time is the whole story)
#   host_ms     wall time the issuing call and the wait kept the host
# The first real multi-GPU run then says by itself which exchange is exposed and by how much.
_EXCHANGE_LOG = None


def exchange_log_begin():
    global _EXCHANGE_LOG
    _EXCHANGE_LOG = []


def exchange_log_end():
    """The records of the evaluate since exchange_log_begin(); synchronises the device once to read the event pairs."""
    global _EXCHANGE_LOG
    log, _EXCHANGE_LOG = _EXCHANGE_LOG or [], None
    if any(r.get("_events") for r in log) and torch.cuda.is_available():
        torch.cuda.synchronize()
    out = []
    for r in log:
        ev = r.pop("_events", None)
        r["exposed_ms"] = float(ev[0].elapsed_time(ev[1])) if ev else r["host_ms"]
        out.append(r)
    return out


class _Exchange:
    """Context manager around the point where the current stream waits for a collective (or issues a blocking one)."""

    def __init__(self, name, kind, tensor, world):
        self.on = _EXCHANGE_LOG is not None
        if not self.on:
            return
        import time
        self.t0 = time.perf_counter()
        self.rec = {"name": name, "kind": kind, "bytes": int(tensor.numel() * tensor.element_size() * (world if kind == "all_gather" else 1))}
        self.cuda = tensor.is_cuda
        if self.cuda:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(tensor.device))
            self.dev = tensor.device

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if not self.on or exc[0] is not None:
            return False
        import time
        if self.cuda:
            self.e1.record(torch.cuda.current_stream(self.dev))
            self.rec["_events"] = (self.e0, self.e1)
        self.rec["host_ms"] = (time.perf_counter() - self.t0) * 1e3
        _EXCHANGE_LOG.append(self.rec)
        return False


def shard_bounds(n, world, rank):
    """Contiguous row range [lo, hi) of `rank` (rows_g = [g*N/G, (g+1)*N/G))."""
    return n * rank // world, n * (rank + 1) // world


def _all_reduce(t, world, group, name="all_reduce"):
    if not _alone(world):
        with _Exchange(name, "all_reduce", t, world):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _flat_gather_supported(group):
    """RCCL gathers straight into one tensor; gloo takes the list form (decided once from the backend, never by
    catching a failed collective: an exception on one rank must not turn into a different collective there).
    Composite backend strings ("cuda:nccl,cpu:gloo") count as RCCL for device tensors."""
    return "nccl" in str(dist.get_backend(group))


_BULK = {}


def enable_bulk_communicator(group=None):
    """OPTIONAL second communicator over the same ranks for the two big row gathers (205 MB per set at 2 x 100k x 512) -
    a setup-time COLLECTIVE: every process of the job must call it (``dist.new_group``), once, before the first evaluate;
    it is never created behind the caller's back.  Default: off.
    The collectives of one communicator run in issue order on one stream.  With a single communicator (the default)
    ``evaluate_sharded`` therefore issues the candidate rows' gather right BEHIND the reference set's bounds exchange, so
    that it travels under the reference set's sweep and only the reference set's list exchange queues behind it; with the
    second communicator the candidate rows start travelling a little earlier (under the statistics kernels and the
    bounds pre-pass).  The price of the second one is two RCCL communicators with collectives in flight on one device at
    the same time - safe only as long as both kernels can be resident together - which is why it has to be asked for.
    Only for the default group (a sub-group keeps one communicator)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    default = dist.group.WORLD
    if group is not None and group is not default:
        return None
    cached = _BULK.get("pair")
    if cached is None or cached[0] is not default:
        cached = (default, dist.new_group())
        _BULK["pair"] = cached
    return cached[1]


def disable_bulk_communicator():
    """Back to one communicator (collective in the sense that every rank must do the same)."""
    cached = _BULK.pop("pair", None)
    if cached is not None and dist.is_available() and dist.is_initialized():
        dist.destroy_process_group(cached[1])


def _bulk_group(group):
    """The second communicator when ``enable_bulk_communicator`` created one for this group, else None."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    default = dist.group.WORLD
    if group is not None and group is not default:
        return None
    cached = _BULK.get("pair")
    if cached is None or cached[0] is not default:
        return None
    return cached[1]


def warm_up_communicators(device, group=None):
    """One tiny collective on every communicator evaluate_sharded will use (every rank must call).  RCCL builds a
    communicator lazily at its first collective (hundreds of milliseconds): a caller that times its first evaluate -
    bench.py with --warmup 0 - pays that outside the timed region this way.  Returns the number of ranks the all-reduce
    saw (a sum of ones: what the launcher promised must be what RCCL connected)."""
    world, _ = _world(group)
    if _alone(world):
        return 1
    one = torch.ones(1, dtype=torch.float32, device=device)
    out = torch.empty(world, dtype=torch.float32, device=device)
    _all_gather_into(out, one, world, group)
    bulk = _bulk_group(group)
    if bulk is not None:
        _all_gather_into(out, one, world, bulk)
    _all_reduce(one, world, group)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return int(round(float(one.item())))


def _all_gather_into(out, local, world, group, async_op=False):
    if _flat_gather_supported(group):
        return dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=async_op)
    parts = list(out.view(world, *local.shape).unbind(0))
    return dist.all_gather(parts, local.contiguous(), group=group, async_op=async_op)


class _Gathered:
    """Row shards of unequal length being concatenated: start() issues the collective, rows() waits and trims."""

    def __init__(self, local, counts, world, group, async_op=True, name="all_gather"):
        self.counts, self.world, self.local, self.name = counts, world, local, name
        self.work = None
        self._rows = None
        self.alone = _alone(world)
        if self.alone:
            return
        width = tuple(local.shape[1:])
        cmax = max(counts)
        if local.shape[0] == cmax:
            pad = local
        else:
            pad = torch.zeros((cmax, *width), dtype=local.dtype, device=local.device)
            pad[:local.shape[0]] = local
        self.out = torch.empty((world * cmax, *width), dtype=local.dtype, device=local.device)
        if async_op:
            self.work = _all_gather_into(self.out, pad, world, group, async_op=True)
        else:
            with _Exchange(name, "all_gather", pad, world):
                self.work = _all_gather_into(self.out, pad, world, group, async_op=False)
        self._pad = pad                                   # stays alive until the collective has run

    def rows(self):
        if self.alone:
            return self.local
        if self._rows is not None:
            return self._rows
        if self.work is not None:
            with _Exchange(self.name, "all_gather", self._pad, self.world):      # (the wait: what of the gather is not hidden)
                self.work.wait()
            self.work = None
        cmax = max(self.counts)
        if all(c == cmax for c in self.counts):
            self._rows = self.out
        else:
            self._rows = torch.cat([self.out[r * cmax:r * cmax + self.counts[r]] for r in range(self.world)])
        return self._rows


def _all_gather_rows(local, counts, world, group, name="all_gather"):
    return _Gathered(local, counts, world, group, async_op=False, name=name).rows()


def global_count(n_local, device, group=None):
    """Total number of rows over the ranks (every rank must call)."""
    world, _ = _world(group)
    if _alone(world):
        return int(n_local)
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    _all_reduce(t, world, group)
    return int(t.item())


def local_rows(data, group=None):
    """This rank's stored rows of an ``AudioMetricsData`` as an [n, D] f32 matrix - possibly with ZERO rows: a rank that was
    fed no audio holds a set that never received a row (``embeddings is None``) and must still take part in every
    collective of the evaluation.  Collective itself when a group is given (every rank must call): the width D is agreed on
    with an all-reduce, and the one condition that is an error - a set that keeps no embeddings - is evaluated identically
    on every rank (it is a property of the configuration), so either all ranks raise or none does."""
    world, _ = _world(group)
    rows = None if data is None else data.embeddings
    stores = data is not None and bool(data.store_embeddings)
    if _alone(world):
        if rows is None:
            raise ValueError("the metric needs the stored embeddings of a set that kept none")
        return rows
    if data is not None:
        dev = data.device
    elif _flat_gather_supported(group):                    # RCCL reduces device tensors only
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    is64 = rows is not None and rows.dtype == torch.float64
    t = torch.tensor([0 if rows is None else rows.shape[1], 0 if stores else 1, 1 if is64 else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    width, someone_keeps_none, someone_is64 = (int(v) for v in t.cpu().tolist())
    if someone_keeps_none:
        raise ValueError("the metric needs the stored embeddings of a set that kept none")
    # the dtype is agreed on like the width: the shards travel through one all-gather (float64 anywhere -> float64 everywhere,
    # torch.cat's promotion of the reference's single store, data.py:68-72)
    dtype = torch.float64 if someone_is64 else torch.float32
    if rows is None:
        rows = torch.empty((0, width), dtype=dtype, device=dev)
    elif rows.dtype != dtype:
        rows = rows.to(dtype)
    return rows


def global_means_pair(ref_local, cand_local, n_ref, n_cand, ops, world, group):
    """Means of two row-sharded sets: local column sums of both in ONE all-reduce (8 KB)."""
    d = ref_local.shape[1]
    dev = ref_local.device

    def colsum(x):
        return ops.colsum(x) if x.shape[0] > 0 else torch.zeros(d, dtype=torch.float64, device=dev)

    sums = torch.stack((colsum(ref_local), colsum(cand_local)))                 # [2, D]
    _all_reduce(sums, world, group, "column_sums")
    return sums[0] / float(n_ref), sums[1] / float(n_cand)


def global_covariances_pair(ref_local, cand_local, mean_r, mean_c, n_ref, n_cand, ops, world, group):
    """Covariances of two row-sharded sets around the global means: local centred scatters of both in ONE all-reduce (4 MB)."""
    d = ref_local.shape[1]
    dev = ref_local.device

    def scatter(x, mean):
        return ops.scatter(x, mean) if x.shape[0] > 0 else torch.zeros((d, d), dtype=torch.float64, device=dev)

    sc = torch.stack((scatter(ref_local, mean_r), scatter(cand_local, mean_c)))  # [2, D, D]
    _all_reduce(sc, world, group, "scatters")
    return sc[0] / float(max(n_ref - 1, 1)), sc[1] / float(max(n_cand - 1, 1))


def global_stats_pair(ref_local, cand_local, n_ref, n_cand, ops, world, group):
    """((mean, cov) of the reference, (mean, cov) of the candidate) of two row-sharded sets: two all-reduces in all
    (one fused buffer per phase)."""
    mean_r, mean_c = global_means_pair(ref_local, cand_local, n_ref, n_cand, ops, world, group)
    cov_r, cov_c = global_covariances_pair(ref_local, cand_local, mean_r, mean_c, n_ref, n_cand, ops, world, group)
    return (mean_r, cov_r), (mean_c, cov_c)


def global_stats(local, group=None, ops=None):
    """(n, mean f64[D], unbiased covariance f64[D, D]) of ONE row-sharded set (every rank must call; a rank may hold zero
    rows): row count and column sums in one all-reduce, centred scatters in a second - the statistics IncrementalPCA's
    partial_fit needs of the union of the shards (projection.py:6-46 fits on the whole reference set)."""
    if ops is None:
        from . import hip_ops as ops
    world, _ = _world(group)
    d, dev = local.shape[1], local.device
    head = torch.zeros(d + 1, dtype=torch.float64, device=dev)
    if local.shape[0] > 0:
        head[:d] = ops.colsum(local)
        head[d] = float(local.shape[0])
    _all_reduce(head, world, group)
    n = int(round(float(head[d].item())))
    if n == 0:
        raise ValueError(f"empty embedding set over {world} ranks")
    mean = head[:d] / float(n)
    sc = ops.scatter(local, mean) if local.shape[0] > 0 else torch.zeros((d, d), dtype=torch.float64, device=dev)
    _all_reduce(sc, world, group)
    return n, mean, sc / float(max(n - 1, 1))


class _Stats:
    """(n, mean, cov) holder with the attributes the metric functions read."""

    def __init__(self, n, mean, cov):
        self.n, self.mean, self.cov = n, mean, cov
        self.embeddings, self.radii = None, {}

    def __len__(self):
        return self.n or 0


def merged_stats(data, group=None, ops=None):
    """Statistics of the union of every rank's ``AudioMetricsData`` (stats-only sets: FAD / APA operands): the
    (n, mean, cov) triples are all-gathered (2 MB per rank at D = 512) and Chan-merged in rank order on every
    rank - the reference's own merge (data.py:77-94), so the result does not depend on who holds which rows."""
    if ops is None:
        from . import hip_ops as ops
    world, _ = _world(group)
    if _alone(world):
        return data
    dev = data.device
    n_local = len(data)
    d_t = torch.tensor([0 if n_local == 0 else data.mean.numel()], dtype=torch.int64, device=dev)
    dist.all_reduce(d_t, op=dist.ReduceOp.MAX, group=group)
    d = int(d_t.item())
    if d == 0:
        return data
    packed = torch.zeros(1 + d + d * d, dtype=torch.float64, device=dev)
    if n_local:
        packed[0] = float(n_local)
        packed[1:1 + d] = data.mean
        cov = data.cov if tuple(data.cov.shape) == (d, d) else torch.zeros((d, d), dtype=torch.float64, device=dev)
        packed[1 + d:] = cov.reshape(-1)
    everyone = torch.empty((world, packed.numel()), dtype=torch.float64, device=dev)
    _all_gather_into(everyone.view(-1), packed, world, group)
    counts = [int(round(v)) for v in everyone[:, 0].cpu().tolist()]
    n, mean, cov = 0, None, None
    for r in range(world):
        if counts[r] == 0:
            continue
        m_r, c_r = everyone[r, 1:1 + d].clone(), everyone[r, 1 + d:].view(d, d).clone()
        if n == 0:
            n, mean, cov = counts[r], m_r, c_r
        else:
            mean, cov = ops.stats_merge(n, mean, cov, counts[r], m_r, c_r, inplace=True)
            n += counts[r]
    return _Stats(n or None, mean, cov)


def sharded_radii(local, full, counts, k, ops, world, rank, group, prepared=None, after_first_exchange=None):
    """k-NN radii of a row-sharded set whose gathered copy `full` every rank holds.
    after_first_exchange: called once, right behind the first collective this function issues (the bounds all-gather of the
    partitioned form; at the start otherwise) - where the caller queues a long transfer that may run under the sweep.
    Returns (radii of this rank's rows, radii of all rows).  Wide, large sets take the partitioned symmetric
    kernel (half the tile pairs; rank r owns a contiguous range of the 128-row blocks; per-row lists all-gathered
    and merged); otherwise every rank runs the general kernel on its row shard against all columns."""
    n, d = full.shape
    lo = sum(counts[:rank])
    hi = lo + counts[rank]
    if not _alone(world) and min(counts) > 0 and hasattr(ops, "knn_sym_part") and full.dtype != torch.float64 \
            and ops.knn_sym_eligible(n, d, k):
        extra = {} if prepared is None else {"prepared": prepared}
        bounds = _all_gather_rows(ops.knn_bounds(full, k, lo, counts[rank], **extra), counts, world, group, "knn_bounds")
        if after_first_exchange is not None:
            after_first_exchange()
        lists = ops.knn_sym_part(full, k, rank, world, bounds, **extra)
        all_lists = torch.empty((world, *lists.shape), dtype=lists.dtype, device=lists.device)
        with _Exchange("knn_lists", "all_gather", lists, world):
            _all_gather_into(all_lists.view(-1), lists.view(-1), world, group)
        r_full = ops.knn_lists_finish(all_lists, full, k)
        return r_full[lo:hi], r_full
    if after_first_exchange is not None:
        after_first_exchange()
    if not _alone(world) and full.dtype == torch.float64 and hasattr(ops, "knn_path") and ops.knn_path(n, n, d, k) == 3 and k <= 10:
        # float64 rows at the sizes of the f16 filter sweep: the whole set's radii on the filter route cost a rank less than its
        # share of the general f64 kernel would (100 000 x 64: 2.6 ms against 32 ms / world) - every rank computes them, nothing
        # is exchanged (a pure function of the shapes: the same branch on every rank)
        r_full = ops.knn_radii(full, k)
        return r_full[lo:hi], r_full
    if _alone(world) and prepared is not None:
        r_local = ops.knn_radii(full, k, prepared=prepared)
    elif local.shape[0] > 0:
        r_local = ops.knn_radii(local, k, columns=full)
    else:
        r_local = torch.empty(0, dtype=torch.float64 if full.dtype == torch.float64 else torch.float32, device=full.device)
    return r_local, _all_gather_rows(r_local, counts, world, group, "radii")


def result_from_record(head, mmds, metrics, n_ref, n_cand, nearest_k):
    """The evaluate() dict from am_evaluate_f32's record (audio_metrics.py:254-274 key order: fad, kernel distance, PRDC)."""
    result = {}
    if "fad" in metrics:
        result["fad"] = head[0]
    if "kd" in metrics:
        result["kernel_distance_mean"] = float(np.mean(mmds))
        result["kernel_distance_std"] = float(np.std(mmds))
    if "prdc" in metrics:
        n_prec, n_rec, sum_cnt, n_cov = (int(v) for v in head[5:9])
        result.update(precision=n_prec / n_cand, recall=n_rec / n_ref,
                      density=(1.0 / float(nearest_k)) * (sum_cnt / n_cand), coverage=n_cov / n_ref)
    return result


def evaluate_single(ref, cand, metrics, nearest_k, ops, kid_subsets=KID_SUBSETS, kid_subset_size=KID_SUBSET_SIZE, rng_seed=1234,
                    given_ref=None, given_cand=None):
    """One GPU: the whole chain is ONE library call (am_evaluate_f32) and one read-back.  given_*: see hip_ops.evaluate."""
    from .metrics.kd import device_subset_indices
    n_ref, n_cand = ref.shape[0], cand.shape[0]
    if n_ref == 0 or n_cand == 0:
        raise ValueError(f"empty embedding set: {n_ref} reference and {n_cand} candidate rows over 1 ranks")
    if "fad" in metrics:
        from .metrics.fad import warn_if_rank_deficient
        warn_if_rank_deficient(n_ref, n_cand, ref.shape[1], ref.dtype)
    idx1 = idx2 = None
    if "kd" in metrics:
        m = kid_subset_size
        if m >= min(n_ref, n_cand):
            m = max(1, min(n_ref, n_cand) // 2)
        idx1, idx2 = device_subset_indices(n_cand, n_ref, kid_subsets, m, rng_seed, ref.device)      # features_1 = candidate
    stats = None
    if "fad" in metrics:
        # the statistics are written where this function can reach them: a covariance product that needs more than the 32
        # Newton-Schulz iterations the chain enqueues is finished by the stand-alone solver on them
        d = ref.shape[1]
        given_ref, given_cand = dict(given_ref or {}), dict(given_cand or {})
        stats = []
        for given in (given_cand, given_ref):                               # (x = candidate, y = reference)
            if given.get("mean") is None or given.get("cov") is None:
                given.setdefault("mean_out", torch.empty(d, dtype=torch.float64, device=ref.device))
                given.setdefault("cov_out", torch.empty((d, d), dtype=torch.float64, device=ref.device))
                stats += [given["mean_out"], given["cov_out"]]
            else:
                stats += [given["mean"], given["cov"]]
    head, mmds = ops.evaluate(ref, cand, metrics, nearest_k, idx1, idx2, None, KID_COEF0, KID_DEGREE, given_ref, given_cand)
    if "fad" in metrics:
        code = int(head[4])
        if code == 4:
            from ._lib import HipLibraryError
            raise HipLibraryError("am_evaluate_f32: non-finite covariance product or trace in Newton-Schulz")
        if code == 0:
            head[0] = ops.frechet(*stats)["fd"]
    return result_from_record(head, mmds, metrics, n_ref, n_cand, nearest_k)


def _evaluate_sharded_c(ref_local, cand_local, ref_counts, cand_counts, metrics, nearest_k, group, ops, kid_subsets,
                        kid_subset_size, rng_seed):
    """The C-level form of evaluate_sharded: one am_evaluate_sharded_f32 call, one read-back."""
    from .collectives import TorchCollectives
    from .metrics.kd import device_subset_indices
    world, _ = _world(group)
    n_ref, n_cand = sum(ref_counts), sum(cand_counts)
    idx1 = idx2 = None
    if "kd" in metrics:
        m = kid_subset_size
        if m >= min(n_ref, n_cand):
            m = max(1, min(n_ref, n_cand) // 2)
        idx1, idx2 = device_subset_indices(n_cand, n_ref, kid_subsets, m, rng_seed, ref_local.device)     # features_1 = candidate
    coll = TorchCollectives(group)
    coll.force = COLLECTIVES_AT_WORLD_ONE              # (test hook: a group of one rank runs its collectives for real)
    global LAST_C_ENTRY_COLLECTIVES
    LAST_C_ENTRY_COLLECTIVES = coll                    # which torch.distributed calls the hooks became (tests, bench)
    head, mmds = ops.evaluate_sharded_c(ref_local, cand_local, ref_counts, cand_counts, metrics, coll, nearest_k, idx1, idx2,
                                        None, KID_COEF0, KID_DEGREE)
    if "fad" in metrics:
        code = int(head[4])
        if code == 4:
            from ._lib import HipLibraryError
            raise HipLibraryError("am_evaluate_sharded_f32: non-finite covariance product or trace in Newton-Schulz")
        if code == 0:          # (more than the 32 iterations the call enqueues: the same on every rank - finish on fresh statistics)
            (mu_r, cov_r), (mu_c, cov_c) = global_stats_pair(ref_local, cand_local, n_ref, n_cand, ops, world, group)
            head[0] = ops.frechet(mu_c, cov_c, mu_r, cov_r)["fd"]
    return result_from_record(head, mmds, metrics, n_ref, n_cand, nearest_k)


def evaluate_sharded(ref_local, cand_local, metrics=("fad", "kd", "prdc"), nearest_k=5, group=None, ops=None,
                     kid_subsets=KID_SUBSETS, kid_subset_size=KID_SUBSET_SIZE, rng_seed=1234, fused=True, shard_counts=None,
                     c_entry=False):
    """FAD / KD / PRDC of (candidate vs reference) from this rank's row shards.
    c_entry=True: the whole schedule below as ONE library call per rank (am_evaluate_sharded_f32, float32 rows) with the
    collectives handed in as hooks over this process group (collectives.TorchCollectives) - what a host that is not Python
    would call with am_rccl_collectives; same results.
    Returns the same keys as ``AudioMetrics.evaluate`` on every rank.  With one rank the whole chain is one library
    call (``evaluate_single``); fused=False keeps the entry points separate there too (per-entry timing in bench.py).
    shard_counts = ([reference rows of rank 0, 1, ...], [candidate rows ...]) when the caller knows how the rows are dealt
    (a fixed sharding rule): saves the count all-gather and its read-back, the only host synchronisation at the start of a
    step."""
    if ops is None:
        from . import hip_ops as ops
    world, rank = _world(group)
    dev = ref_local.device
    if ref_local.dtype == torch.float64 or cand_local.dtype == torch.float64:
        # one dtype for both sets, as numpy / torch promote (kd.py:115); the float32 one-call chain does not apply
        ref_local, cand_local, fused = ref_local.to(torch.float64), cand_local.to(torch.float64), False
    if _alone(world) and fused and hasattr(ops, "evaluate"):
        return evaluate_single(ref_local, cand_local, metrics, nearest_k, ops, kid_subsets, kid_subset_size, rng_seed)
    if shard_counts is not None:
        ref_counts, cand_counts = [int(c) for c in shard_counts[0]], [int(c) for c in shard_counts[1]]
        if len(ref_counts) != world or len(cand_counts) != world or ref_counts[rank] != ref_local.shape[0] \
                or cand_counts[rank] != cand_local.shape[0]:
            raise ValueError("shard_counts do not describe this process group's shards")
        allc = None
    elif not _alone(world):
        counts = torch.tensor([ref_local.shape[0], cand_local.shape[0]], dtype=torch.int64, device=dev)
        allc = torch.empty(world * 2, dtype=torch.int64, device=dev)
        _all_gather_into(allc, counts, world, group)
        allc = allc.view(world, 2).cpu().tolist()
    else:
        allc = [[ref_local.shape[0], cand_local.shape[0]]]
    if allc is not None:
        ref_counts, cand_counts = [c[0] for c in allc], [c[1] for c in allc]
    n_ref, n_cand = sum(ref_counts), sum(cand_counts)
    if n_ref == 0 or n_cand == 0:                          # the same error on every rank (all of them hold the totals)
        raise ValueError(f"empty embedding set: {n_ref} reference and {n_cand} candidate rows over {world} ranks")
    d = ref_local.shape[1]
    if "fad" in metrics:
        from .metrics.fad import warn_if_rank_deficient
        warn_if_rank_deficient(n_ref, n_cand, d, ref_local.dtype)
    if c_entry and ref_local.dtype != torch.float64 and hasattr(ops, "evaluate_sharded_c"):
        return _evaluate_sharded_c(ref_local, cand_local, ref_counts, cand_counts, metrics, nearest_k, group, ops, kid_subsets,
                                   kid_subset_size, rng_seed)

    # 1) the 8 KB column-sum all-reduce goes FIRST: the collectives of one communicator run in issue order on one
    #    communication stream, so anything issued behind a 205 MB gather waits for all of it - and the centred scatter
    #    kernels need the global means.  2) the reference rows' gather is started (asynchronously).  3) the scatter kernels
    #    run on the compute stream while it is in flight; their 4 MB all-reduce queues behind it, which is fine: only the
    #    Frechet solve on the side stream waits for it.  4) the candidate rows' gather: with PRDC it is issued right behind
    #    the reference set's bounds exchange (start_cand below) and travels under the reference set's sweep - issued here,
    #    on the same communicator, that 400 KB exchange would wait for all 205 MB of it; without PRDC (KD only), or on the
    #    optional second communicator (enable_bulk_communicator), it starts at once.
    need_full = ("kd" in metrics) or ("prdc" in metrics)
    means = None
    if "fad" in metrics:
        means = global_means_pair(ref_local, cand_local, n_ref, n_cand, ops, world, group)
    ref_g = cand_g = bulk = None
    if need_full:
        bulk = None if _alone(world) else _bulk_group(group)
        ref_g = _Gathered(ref_local, ref_counts, world, bulk if bulk is not None else group, name="reference_rows")

    def start_cand():
        nonlocal cand_g
        if cand_g is None:
            cand_g = _Gathered(cand_local, cand_counts, world, bulk if bulk is not None else group, name="candidate_rows")
        return cand_g

    if need_full and (bulk is not None or "prdc" not in metrics):
        start_cand()

    # statistics; the Frechet solve goes to a side stream right away (its stopping rule runs on the device, so
    # the host just enqueues it) and overlaps the PRDC chain issued next
    fad_job = None
    if "fad" in metrics:
        mu_r, mu_c = means
        cov_r, cov_c = global_covariances_pair(ref_local, cand_local, mu_r, mu_c, n_ref, n_cand, ops, world, group)
        starter = getattr(ops, "frechet_async", None)
        fad_job = starter(mu_c, cov_c, mu_r, cov_r) if starter is not None else None

    prdc_pending = None
    if "prdc" in metrics:
        k = nearest_k
        # norms, maxima and scaled f16 copies of the two gathered sets: once per evaluate, shared by the k-NN entry points
        # and the membership counts (the reference shard of this rank is a row range of the prepared reference set).
        # The reference set's radii are under way before the candidate rows are waited for: their gather overlaps the
        # reference set's sweep.
        prepare = getattr(ops, "prepare", None)
        ref_full = ref_g.rows()
        prep_r = prepare(ref_full) if prepare is not None else None
        r_ref_l, _ = sharded_radii(ref_local, ref_full, ref_counts, k, ops, world, rank, group, prep_r,
                                   after_first_exchange=start_cand)
        cand_full = start_cand().rows()
        prep_c = prepare(cand_full) if prepare is not None else None
        _, r_cand = sharded_radii(cand_local, cand_full, cand_counts, k, ops, world, rank, group, prep_c)
        packed = torch.zeros(n_cand + 2, dtype=torch.int32, device=dev)        # column counts | #rows any | #rows covered
        if ref_local.shape[0] > 0:
            lo = sum(ref_counts[:rank])
            extra = {} if prep_r is None else {"prepared_ref": prep_r.rows(lo, lo + ref_counts[rank]), "prepared_cand": prep_c}
            col, rany, rcov = ops.prdc_counts(ref_local, cand_full, r_ref_l, r_cand, **extra)
            local_tot = ops.prdc_reduce(col, rany, rcov)
            packed[:n_cand] = col
            packed[n_cand:] = torch.stack((local_tot[1], local_tot[3])).to(torch.int32)
        _all_reduce(packed, world, group, "membership_counts")
        none = torch.zeros(0, dtype=torch.uint8, device=dev)
        tot = ops.prdc_reduce(packed[:n_cand], none, none)                     # [#cols with count > 0, 0, sum of counts, 0]
        prdc_pending = (tot, packed[n_cand:], k)

    kd_pending = None
    if "kd" in metrics:
        m = kid_subset_size
        if m >= min(n_ref, n_cand):
            m = max(1, min(n_ref, n_cand) // 2)
        idx1, idx2 = subset_indices(n_cand, n_ref, kid_subsets, m, rng_seed)     # features_1 = candidate
        mmds = torch.zeros(kid_subsets, dtype=torch.float64, device=dev)
        ref_full, cand_full = ref_g.rows(), start_cand().rows()
        if rank < kid_subsets:                                 # this rank's subsets: rank, rank + world, ...
            upload = getattr(ops, "upload_host_array", None) or _plain_upload
            part = ops.kd_poly(cand_full, ref_full, upload(idx1[rank::world], dev), upload(idx2[rank::world], dev),
                               1.0 / d, KID_COEF0, KID_DEGREE)
            mmds[rank::world] = part
        _all_reduce(mmds, world, group, "kd_subsets")
        kd_pending = mmds

    # ONE read-back for everything that is already on the device: per-subset MMD^2 values, PRDC totals (the Frechet record
    # lives on the side stream and is read by its own job - five doubles - after the side stream has been waited for)
    pieces, layout = [], []
    if kd_pending is not None:
        pieces.append(kd_pending)
        layout.append(("kd", kd_pending.numel()))
    if prdc_pending is not None:
        tot, rows, k = prdc_pending
        pieces += [tot.to(torch.float64), rows.to(torch.float64)]          # exact: integer counts far below 2^53
        layout += [("tot", tot.numel()), ("rows", rows.numel())]
    host = torch.cat(pieces).cpu().numpy() if pieces else None
    got, at = {}, 0
    for name, count in layout:
        got[name] = host[at:at + count]
        at += count
    result = {}
    if fad_job is not None:
        result["fad"] = fad_job.result()["fd"]
    elif "fad" in metrics:
        result["fad"] = ops.frechet(mu_c, cov_c, mu_r, cov_r)["fd"]
    if kd_pending is not None:
        mm = got["kd"]
        result["kernel_distance_mean"] = float(np.mean(mm))
        result["kernel_distance_std"] = float(np.std(mm))
    if prdc_pending is not None:
        n_prec, sum_cnt = int(got["tot"][0]), int(got["tot"][2])
        n_rec, n_cov = int(got["rows"][0]), int(got["rows"][1])
        result.update(precision=n_prec / n_cand, recall=n_rec / n_ref,
                      density=(1.0 / float(prdc_pending[2])) * (sum_cnt / n_cand), coverage=n_cov / n_ref)
    return result
