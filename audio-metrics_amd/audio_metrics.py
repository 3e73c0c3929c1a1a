"""``AudioMetrics`` - the drop-in front end (``add_reference`` / ``evaluate`` / ``save_state`` / ``load_state``).

Public surface, result keys, state-file layout and error behaviour are those of the reference's class
(src/audio_metrics/audio_metrics.py:15-313); the organisation is this build's own:

  * the three reference sets and their PCA-projected shadows are described by one table (``REFERENCE_SETS``) that
    drives accumulation, reset, projection caching and (de)serialisation;
  * metrics are entries of a dispatch table evaluated in the reference's result-key order;
  * the embedder runs on every GPU of ``device_indices`` (``embed.EmbedderPool``), each GPU aggregating its own
    rows; with ``process_group`` (one process per GPU, torch.distributed over RCCL) every rank feeds its shard of
    the audio and the metrics are reduced across ranks by ``distributed.py``.
Everything below the embedder runs on the MI355X through the HIP library."""
from collections import namedtuple
from pathlib import Path

import torch

from .data import AudioMetricsData
from .embed import EmbedderPool, ItemCategory, embedding_pipeline
from .embedders import DEFAULT_EMBEDDER, EMBEDDERS
from .metrics.apa import apa, apa_compute_d_x_xp
from .metrics.fad import frechet_distance
from .metrics.kd import kernel_distance
from .metrics.prdc import prdc
from .mix_functions import resolve_mix_function
from .projection import IncrementalPCA

# attribute -> (category that feeds it, projection attribute it goes through, attribute of its projected shadow)
SetSpec = namedtuple("SetSpec", "category projection shadow")
REFERENCE_SETS = {
    "stem_reference": SetSpec(ItemCategory.stem, "stem_projection", "stem_reference_pca"),
    "mix_reference": SetSpec(ItemCategory.aligned, "mix_projection", "mix_reference_pca"),
    "mix_anti_reference": SetSpec(ItemCategory.misaligned, "mix_projection", "mix_anti_reference_pca"),
}
PROJECTIONS = ("stem_projection", "mix_projection")
PLAIN_STATE = ("metrics", "need_apa", "win_dur", "input_sr", "apa_d_x_xp")
ROW_METRICS = frozenset(("kd", "precision", "prdc"))          # metrics that need the stored rows (audio_metrics.py:17)
MAX_NEAREST_K = 10                                             # audio_metrics.py:263
FUSED_METRICS = ("fad", "kd", "prdc")                          # what one am_evaluate_f32 call covers (result-key order)

EMPTY_REFERENCE = ("The reference dataset is empty. This can have various causes:"
                   "  - You have not called AudioMetrics.add_reference()"
                   "  - You have called AudioMetrics.add_reference() with an empty dataset"
                   "  - The duration of your audio is shorter than `win_dur` ({win_dur}s)."
                   "    (You can specify your own `win_dur` when instantiating AudioMetrics)")


def _visible_devices(device_indices, one_process_per_gpu=False, embedder=None):
    """The GPUs the embedder replicas run on, the reference's rule (util/gpu_parallel.py:24-28, audio_metrics.py:276-279):
    ``None`` -> EVERY visible GPU; a non-empty sequence -> exactly those; an empty (falsy, not None) one -> no replica
    handler at all: the embedder runs where it lives.  The first device of the list is this object's home (statistics,
    stored rows, metric kernels): with ``None`` the thread's current device leads, the others follow in index order.
    One process per GPU (``process_group``): the launcher gave every rank ITS device, so ``None`` means that device only
    - all ranks replicating onto all GPUs would be world x world replicas."""
    if not torch.cuda.is_available():
        raise RuntimeError("No GPUs found, cannot compute audio metrics")            # gpu_parallel.py:27-28
    current = torch.cuda.current_device()
    if device_indices is None:
        if one_process_per_gpu:
            return [torch.device("cuda", current)]
        count = torch.cuda.device_count()
        if count <= 0:
            raise RuntimeError("No GPUs found, cannot use `gpu_parallel()`")
        return [torch.device("cuda", i) for i in [current] + [i for i in range(count) if i != current]]
    devices = [torch.device("cuda", int(i)) for i in device_indices]
    if not devices:
        home = None
        if embedder is not None and not isinstance(embedder, str):
            home = embedder.get_device()
        if home is None or home.type != "cuda":
            return [torch.device("cuda", current)]
        return [torch.device("cuda", current if home.index is None else home.index)]
    return devices


class _Sets:
    """What one evaluate() call works on: (candidate, reference[, anti-reference]) after projection."""
    __slots__ = ("stem_cand", "stem_ref", "mix_cand", "mix_ref", "mix_anti")

    def __init__(self):
        for name in self.__slots__:
            setattr(self, name, None)


class AudioMetrics:
    _need_embeddings = set(ROW_METRICS)

    def __init__(self, metrics=["apa", "fad"], n_pca=None, device_indices=None, embedder=None, mix_function=None,
                 win_dur=5.0, input_sr=None, process_group=None, replica_dealing="round_robin"):
        self._devices = _visible_devices(device_indices, process_group is not None, embedder)
        self.device = self._devices[0]                 # where statistics, stored rows and metric kernels live
        self._group = process_group
        self.metrics = metrics
        self.need_apa = "apa" in metrics
        self.win_dur = win_dur
        self.input_sr = input_sr
        for name in PROJECTIONS:
            setattr(self, name, None if n_pca is None else IncrementalPCA(n_components=n_pca, device=self.device))
        self.embedder = self.get_embedder(embedder) if embedder is None or isinstance(embedder, str) else embedder
        # resolved (and, for the library's own names, checked for its dependencies) here rather than at the first mix
        self.mix_function = self.get_mix_function(mix_function)
        # replica_dealing="free": batches go to whichever GPU is free, as in the reference (util/gpu_parallel.py:59-76);
        # the default deals them round-robin so that the stored row order - and the KD subsets - are reproducible
        self._pool = EmbedderPool(self.embedder, self._devices, dealing=replica_dealing)
        self.apa_d_x_xp = None
        for name, spec in REFERENCE_SETS.items():
            setattr(self, name, None)
            setattr(self, spec.shadow, None)
        self.reset_reference()

    # ------------------------------------------------------------ what this configuration needs
    @property
    def stems_mode(self):
        return any(m != "apa" for m in self.metrics)

    @property
    def store_mix_embeddings(self):
        return self.need_apa and self.mix_projection is not None

    @property
    def store_stem_embeddings(self):
        return self.stem_projection is not None or not ROW_METRICS.isdisjoint(self.metrics)

    def _active(self, name):
        return self.stems_mode if name == "stem_reference" else self.need_apa

    def _stored(self, name):
        return self.store_stem_embeddings if name == "stem_reference" else self.store_mix_embeddings

    def get_mix_function(self, mix_function):
        if callable(mix_function):
            return mix_function
        return resolve_mix_function(mix_function, needed=self.need_apa)

    def get_embedder(self, embedder):
        name = DEFAULT_EMBEDDER if embedder is None else embedder
        if name not in EMBEDDERS:
            raise ValueError(f"Unknown embedder {name}, must be one of {EMBEDDERS.keys()}")
        factory, kwargs = EMBEDDERS[name]
        return factory(**kwargs, device=self.device)

    # ------------------------------------------------------------ reference bookkeeping
    def reset_reference(self):
        """Empty reference sets for everything this configuration accumulates (audio_metrics.py:151-161)."""
        for name, spec in REFERENCE_SETS.items():
            if self._active(name):
                setattr(self, name, AudioMetricsData(self._stored(name), device=self.device))
                setattr(self, spec.shadow, None)
        if self.need_apa:
            self.apa_d_x_xp = None

    def assert_reference(self):
        for name in REFERENCE_SETS:
            if name != "mix_anti_reference" and self._active(name) and self._global_count(getattr(self, name)) == 0:
                raise ValueError(EMPTY_REFERENCE.format(win_dur=self.win_dur))

    def _embed(self, waveforms, apa_mode):
        return embedding_pipeline(
            waveforms, embedder=self.embedder, mix_function=self.mix_function, gpu_handler=self._pool,
            apa_mode=apa_mode if self.need_apa else None, stems_mode=self.stems_mode,
            store_mix_embeddings=self.store_mix_embeddings, store_stem_embeddings=self.store_stem_embeddings,
            win_dur=self.win_dur, input_sr=self.input_sr, device=self.device)

    def add_reference(self, reference):
        fresh = self._embed(reference, "reference")
        for name, spec in REFERENCE_SETS.items():
            part = fresh.get(spec.category)
            if part is None:
                continue
            if name != "mix_anti_reference":              # new rows outdate the projected shadows of their group
                for other, other_spec in REFERENCE_SETS.items():
                    if other_spec.projection == spec.projection:
                        setattr(self, other_spec.shadow, None)
            target = getattr(self, name)
            target += part
            if name == "stem_reference":
                target.recompute_stats()                    # audio_metrics.py:139
        # (the reference keeps a cached d_x_xp across add_reference calls, audio_metrics.py:251-252; so does this)

    # ------------------------------------------------------------ PCA projection (audio_metrics.py:163-209)
    def _through(self, projection, data, store):
        out = AudioMetricsData(store, device=self.device)
        if self._group is not None and (data is None or data.embeddings is None or data.embeddings.shape[0] == 0):
            return out                                      # a rank that was fed no audio projects nothing
        out.add(projection.transform(data.embeddings))
        return out

    def _fit(self, projection, data):
        """partial_fit on the whole set (projection.py:6-46 via audio_metrics.py:170,188).  One process per GPU: the set is
        the union of the ranks' shards - its statistics travel in two all-reduces and every rank runs the same fit."""
        if self._group is None:
            projection.partial_fit(data.embeddings)
            return
        from . import distributed
        rows = distributed.local_rows(data, self._group)
        projection.partial_fit(rows, batch_stats=distributed.global_stats(rows, self._group))

    def _projected(self, projection_name, names, candidate, store):
        """(references of `names`..., candidate) in the space of `projection_name`; the projected reference sets
        are cached in their shadow attributes, the projection is (incrementally) fitted on the FIRST set only."""
        projection = getattr(self, projection_name)
        if projection is None:
            return [getattr(self, n) for n in names] + [candidate]
        shadows = [REFERENCE_SETS[n].shadow for n in names]
        if getattr(self, shadows[0]) is None:
            self._fit(projection, getattr(self, names[0]))
            for n, shadow in zip(names, shadows):
                setattr(self, shadow, self._through(projection, getattr(self, n), store))
        return [getattr(self, s) for s in shadows] + [self._through(projection, candidate, store)]

    # ------------------------------------------------------------ evaluation
    def __call__(self, candidate):
        return self.evaluate(candidate)

    def evaluate(self, candidate):
        self.assert_reference()
        fresh = self._embed(candidate, "candidate")
        sets = _Sets()
        if self.stems_mode:
            cand = fresh.get(ItemCategory.stem)
            if cand is None or self._global_count(cand) == 0:
                raise ValueError("No stem candidate embeddings were computed")
            sets.stem_ref, sets.stem_cand = self._projected(
                "stem_projection", ["stem_reference"], cand, not ROW_METRICS.isdisjoint(self.metrics))
        if self.need_apa:
            cand = fresh.get(ItemCategory.aligned)
            if cand is None or self._global_count(cand) == 0:
                raise ValueError("No apa candidate embeddings were computed")
            sets.mix_ref, sets.mix_anti, sets.mix_cand = self._projected(
                "mix_projection", ["mix_reference", "mix_anti_reference"], cand, False)
        if self._group is not None:
            return self._evaluate_sharded(sets)
        fused = self._run_fused(sets)                  # None: the one-call form does not apply - every metric runs on its own
        result = dict(fused) if fused is not None else {}
        for key, run in METRIC_TABLE:
            if key in self.metrics and not (fused is not None and key in FUSED_METRICS):
                result.update(run(self, sets))
        return result

    def _run_fused(self, sets):
        """FAD / KD / PRDC of the stem sets as ONE library call and one read-back (am_evaluate_f32,
        audio_metrics.py:254-274) when at least two of them are asked for and both sets are in the plain device form; the
        statistics come from the sets (accumulated add by add), the reference's cached radii are reused and the radii
        computed here are cached on the sets exactly as get_radii() would (data.py:60-66).  None -> the per-metric runners."""
        wanted = [m for m in FUSED_METRICS if m in self.metrics]
        cand, ref = sets.stem_cand, sets.stem_ref
        if len(wanted) < 2 or cand is None or ref is None:
            return None
        rows = [m for m in wanted if m != "fad"]
        d = ref.mean.numel()
        for side in (ref, cand):
            if rows and (side.embeddings is None or side.embeddings.shape[1] != d):
                return None
            if rows and side.embeddings.dtype == torch.float64:      # float64 rows (PCA output, f64 embedders): the f64 entry points
                return None
            if tuple(side.cov.shape) != (d, d) or side.mean.device != self.device:
                return None
        from . import distributed, hip_ops
        k = max(1, min(MAX_NEAREST_K, len(ref), len(cand)))
        if not rows:
            return None
        given = []
        for side in (ref, cand):
            g = {"mean": side.mean, "cov": side.cov}
            if "prdc" in wanted:
                slot = "radii_%d" % k
                if slot in side.radii:
                    # the cache survives appends (reference quirk, data.py:60-66 vs 68-72): radii of FEWER rows than the set
                    # now holds cannot go through the raw-pointer call - the per-metric runner raises the shape error
                    if side.radii[slot].numel() != side.embeddings.shape[0]:
                        return None
                    g["radii"] = side.radii[slot]
                else:
                    g["radii_out"] = torch.empty(side.embeddings.shape[0], dtype=torch.float32, device=self.device)
            given.append(g)
        result = distributed.evaluate_single(ref.embeddings, cand.embeddings, wanted, k, hip_ops, given_ref=given[0],
                                             given_cand=given[1])
        for side, g in zip((ref, cand), given):
            if "radii_out" in g:
                side.radii["radii_%d" % k] = g["radii_out"]
        return result

    # -- single-process metric runners (result-key order of audio_metrics.py:254-274)
    def _run_fad(self, sets):
        return {"fad": frechet_distance(sets.stem_cand, sets.stem_ref)}

    def _run_kd(self, sets):
        return kernel_distance(sets.stem_cand, sets.stem_ref)       # candidate is features_1 (audio_metrics.py:260)

    def _run_prdc(self, sets):
        k = max(1, min(MAX_NEAREST_K, len(sets.stem_ref), len(sets.stem_cand)))
        return prdc(sets.stem_ref, sets.stem_cand, k)

    def _run_apa(self, sets):
        if self.apa_d_x_xp is None:
            self.apa_d_x_xp = apa_compute_d_x_xp(sets.mix_ref, sets.mix_anti)
        return {"apa": apa(sets.mix_cand, sets.mix_ref, sets.mix_anti, self.apa_d_x_xp)}

    # -- one process per GPU: this rank holds a shard of every set
    def _global_count(self, data):
        if data is None:
            return 0
        if self._group is None:
            return len(data)
        from . import distributed
        return distributed.global_count(len(data), self.device, self._group)

    def _evaluate_sharded(self, sets):
        from . import distributed
        group = self._group
        result = {}
        if "fad" in self.metrics:
            cand, ref = (distributed.merged_stats(d, group) for d in (sets.stem_cand, sets.stem_ref))
            result["fad"] = frechet_distance(cand, ref)
        rows = [m for m in ("kd", "prdc") if m in self.metrics]
        if rows:
            n_ref, n_cand = self._global_count(sets.stem_ref), self._global_count(sets.stem_cand)
            k = max(1, min(MAX_NEAREST_K, n_ref, n_cand))
            result.update(distributed.evaluate_sharded(
                distributed.local_rows(sets.stem_ref, group), distributed.local_rows(sets.stem_cand, group),
                metrics=rows, nearest_k=k, group=group))
        if self.need_apa:
            cand, ref, anti = (distributed.merged_stats(d, group) for d in (sets.mix_cand, sets.mix_ref, sets.mix_anti))
            if self.apa_d_x_xp is None:
                self.apa_d_x_xp = apa_compute_d_x_xp(ref, anti)
            result["apa"] = apa(cand, ref, anti, self.apa_d_x_xp)
        return result

    # ------------------------------------------------------------ state files (audio_metrics.py:78-104)
    def save_state(self, fp: str | Path):
        """The reference's layout: its instance dictionary without embedder / mix function, reference sets as
        ``AudioMetricsData.serialize()`` dicts (host tensors), projections as their ``__getstate__`` dicts."""
        state = {key: getattr(self, key) for key in PLAIN_STATE}
        for name in PROJECTIONS:
            projection = getattr(self, name)
            state[name] = projection.__getstate__().copy() if projection is not None else None
        for name, spec in REFERENCE_SETS.items():
            for attr in (name, spec.shadow):
                data = getattr(self, attr)
                state[attr] = data.serialize() if data is not None else None
        torch.save(state, fp)

    def load_state(self, fp: str | Path):
        state = dict(torch.load(fp, weights_only=True))
        for name, spec in REFERENCE_SETS.items():
            for attr in (name, spec.shadow):
                if attr in state:
                    item = state.pop(attr)
                    setattr(self, attr, AudioMetricsData.deserialize(item, device=self.device) if item else item)
        for name in PROJECTIONS:
            item = state.pop(name, None)
            if item:
                if getattr(self, name) is None:
                    setattr(self, name, IncrementalPCA(n_components=item.get("n_components"), device=self.device))
                getattr(self, name).__setstate__(item)
        for key, value in state.items():                   # plain fields (and anything else a reference file carries)
            if key not in ("embedder", "mix_function", "device"):
                setattr(self, key, value)


METRIC_TABLE = (("fad", AudioMetrics._run_fad), ("kd", AudioMetrics._run_kd), ("prdc", AudioMetrics._run_prdc),
                ("apa", AudioMetrics._run_apa))
