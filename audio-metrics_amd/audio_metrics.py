"""``AudioMetrics(add_reference / evaluate)`` - drop-in front end.

Mirrors src/audio_metrics/audio_metrics.py:15-313: same constructor kwargs, methods,
result keys (``fad``, ``kernel_distance_mean``, ``kernel_distance_std``, ``precision``,
``recall``, ``density``, ``coverage``, ``apa``), state-dict layout and error behaviour.
Everything below the embedder runs on the MI355X through the HIP library."""
from pathlib import Path

import torch

from .data import AudioMetricsData
from .embed import ItemCategory, embedding_pipeline
from .embedders import DEFAULT_EMBEDDER, EMBEDDERS
from .metrics.apa import apa, apa_compute_d_x_xp
from .metrics.fad import frechet_distance
from .metrics.kd import kernel_distance
from .metrics.prdc import prdc
from .mix_functions import DEFAULT_MIX_FUNCTION, MIX_FUNCTIONS
from .projection import IncrementalPCA


class AudioMetrics:
    _need_embeddings = set(("kd", "precision", "prdc"))       # audio_metrics.py:17
    _amd = ("stem_reference", "mix_reference", "mix_anti_reference",
            "stem_reference_pca", "mix_reference_pca", "mix_anti_reference_pca")

    def __init__(self, metrics=["apa", "fad"], n_pca=None, device_indices=None, embedder=None, mix_function=None,
                 win_dur=5.0, input_sr=None):
        self.device = self._pick_device(device_indices)
        self.metrics = metrics
        self.need_apa = "apa" in self.metrics
        self.win_dur = win_dur
        self.input_sr = input_sr
        if n_pca is None:
            self.stem_projection = None
            self.mix_projection = None
        else:
            self.stem_projection = IncrementalPCA(n_components=n_pca, device=self.device)
            self.mix_projection = IncrementalPCA(n_components=n_pca, device=self.device)
        self.embedder = self.get_embedder(embedder) if embedder is None or isinstance(embedder, str) else embedder
        self.mix_function = (self.get_mix_function(mix_function)
                             if mix_function is None or isinstance(mix_function, str) else mix_function)
        self.apa_d_x_xp = None
        self.reset_reference(_init=True)
        self.mix_reference_pca = None
        self.mix_anti_reference_pca = None
        self.stem_reference_pca = None

    # ------------------------------------------------------------ configuration
    @staticmethod
    def _pick_device(device_indices):
        """The reference spreads embedder replicas over `device_indices`; here the embedder and
        the metric kernels share ONE GPU: the first listed index (default: the current device)."""
        if not torch.cuda.is_available():
            raise RuntimeError("No GPUs found, cannot compute audio metrics")       # gpu_parallel.py:27-28
        if device_indices:
            return torch.device("cuda", int(device_indices[0]))
        return torch.device("cuda", torch.cuda.current_device())

    @property
    def stems_mode(self):
        return any(metric for metric in self.metrics if metric != "apa")

    @property
    def store_mix_embeddings(self):
        return self.need_apa and self.mix_projection is not None

    @property
    def store_stem_embeddings(self):
        return self.stem_projection is not None or any(metric in self._need_embeddings for metric in self.metrics)

    def get_mix_function(self, mix_function):
        if mix_function is None:
            mix_function = DEFAULT_MIX_FUNCTION
        func = MIX_FUNCTIONS.get(mix_function)
        if func is None:
            raise ValueError(f"Unknown mix_function {mix_function}, must be one of {MIX_FUNCTIONS.keys()}")
        return func

    def get_embedder(self, embedder):
        if embedder is None:
            embedder = DEFAULT_EMBEDDER
        info = EMBEDDERS.get(embedder)
        if info is None:
            raise ValueError(f"Unknown embedder {embedder}, must be one of {EMBEDDERS.keys()}")
        cls, kwargs = info
        return cls(**kwargs, device=self.device)

    # ------------------------------------------------------------ state
    def save_state(self, fp: str | Path):
        state = dict(self.__dict__)
        for key in ("mix_function", "embedder", "device"):
            state.pop(key, None)
        for attr in self._amd:
            item = state.get(attr)
            if item:
                state[attr] = item.serialize()
        for attr in ("stem_projection", "mix_projection"):
            item = state.get(attr)
            if item:
                state[attr] = item.__getstate__().copy()
        torch.save(state, fp)

    def load_state(self, fp: str | Path):
        state = torch.load(fp, weights_only=True)
        for attr in self._amd:
            item = state.get(attr)
            if item:
                state[attr] = AudioMetricsData.deserialize(item, device=self.device)
        for attr in ("stem_projection", "mix_projection"):
            item = state.get(attr)
            if item:
                getattr(self, attr).__setstate__(item)
                del state[attr]
        self.__dict__.update(state)

    def reset_reference(self, _init=False):
        if self.need_apa:
            self.apa_d_x_xp = None
            self.mix_reference = AudioMetricsData(self.store_mix_embeddings, device=self.device)
            self.mix_anti_reference = AudioMetricsData(self.store_mix_embeddings, device=self.device)
            self.mix_reference_pca = None
            self.mix_anti_reference_pca = None
        elif _init:
            self.mix_reference = None
            self.mix_anti_reference = None
        if self.stems_mode:
            self.stem_reference = AudioMetricsData(self.store_stem_embeddings, device=self.device)
            self.stem_reference_pca = None
        elif _init:
            self.stem_reference = None

    def assert_reference(self):
        msg = ("The reference dataset is empty. This can have various causes:"
               "  - You have not called AudioMetrics.add_reference()"
               "  - You have called AudioMetrics.add_reference() with an empty dataset"
               f"  - The duration of your audio is shorter than `win_dur` ({self.win_dur}s)."
               "    (You can specify your own `win_dur` when instantiating AudioMetrics)")
        if self.stems_mode and self.stem_reference.n is None:
            raise ValueError(msg)
        if self.need_apa and self.mix_reference.n is None:
            raise ValueError(msg)

    # ------------------------------------------------------------ PCA projection (audio_metrics.py:163-209)
    def ensure_stem_projection(self, ref, cand):
        if self.stem_projection is None:
            return ref, cand
        store_embs = any(metric in self._need_embeddings for metric in self.metrics)
        if self.stem_reference_pca is None:
            self.stem_projection.partial_fit(ref.embeddings)
            ref_emb = self.stem_projection.transform(ref.embeddings)
            ref = AudioMetricsData(store_embs, device=self.device)
            ref.add(ref_emb)
            self.stem_reference_pca = ref
        ref = self.stem_reference_pca
        cand_emb = self.stem_projection.transform(cand.embeddings)
        cand = AudioMetricsData(store_embs, device=self.device)
        cand.add(cand_emb)
        return ref, cand

    def ensure_mix_projection(self, ref, anti_ref, cand):
        if self.mix_projection is None:
            return ref, anti_ref, cand
        if self.mix_reference_pca is None:
            self.mix_projection.partial_fit(ref.embeddings)
            ref_emb = self.mix_projection.transform(ref.embeddings)
            anti_ref_emb = self.mix_projection.transform(anti_ref.embeddings)
            ref = AudioMetricsData(store_embeddings=False, device=self.device)
            anti_ref = AudioMetricsData(store_embeddings=False, device=self.device)
            ref.add(ref_emb)
            anti_ref.add(anti_ref_emb)
            self.mix_reference_pca = ref
            self.mix_anti_reference_pca = anti_ref
        ref, anti_ref = self.mix_reference_pca, self.mix_anti_reference_pca
        cand_emb = self.mix_projection.transform(cand.embeddings)
        cand = AudioMetricsData(store_embeddings=False, device=self.device)
        cand.add(cand_emb)
        return ref, anti_ref, cand

    # ------------------------------------------------------------ the two entry points
    def _pipeline(self, waveforms, apa_mode):
        return embedding_pipeline(
            waveforms, embedder=self.embedder, mix_function=self.mix_function,
            apa_mode=apa_mode if self.need_apa else None, stems_mode=self.stems_mode,
            store_mix_embeddings=self.store_mix_embeddings, store_stem_embeddings=self.store_stem_embeddings,
            win_dur=self.win_dur, input_sr=self.input_sr, device=self.device)

    def add_reference(self, reference):
        metrics = self._pipeline(reference, "reference")
        stem_reference = metrics.get(ItemCategory.stem)
        if stem_reference is not None:
            self.stem_reference_pca = None
            self.stem_reference += stem_reference
            self.stem_reference.recompute_stats()            # audio_metrics.py:139
        mix_reference = metrics.get(ItemCategory.aligned)
        if mix_reference is not None:
            self.mix_reference_pca = None
            self.mix_anti_reference_pca = None
            self.mix_reference += mix_reference
        mix_anti_reference = metrics.get(ItemCategory.misaligned)
        if mix_anti_reference is not None:
            self.mix_anti_reference += mix_anti_reference

    def __call__(self, candidate):
        return self.evaluate(candidate)

    def evaluate(self, candidate):
        self.assert_reference()
        metrics = self._pipeline(candidate, "candidate")
        stem_cand = metrics.get(ItemCategory.stem)
        apa_cand = metrics.get(ItemCategory.aligned)
        stem_ref, apa_ref, apa_anti_ref = self.stem_reference, self.mix_reference, self.mix_anti_reference
        if self.stems_mode and (stem_cand is None or stem_cand.n is None):
            raise ValueError("No stem candidate embeddings were computed")
        if self.need_apa and (apa_cand is None or apa_cand.n is None):
            raise ValueError("No apa candidate embeddings were computed")
        if self.stems_mode:
            stem_ref, stem_cand = self.ensure_stem_projection(stem_ref, stem_cand)
        if self.need_apa:
            apa_ref, apa_anti_ref, apa_cand = self.ensure_mix_projection(apa_ref, apa_anti_ref, apa_cand)
            if self.apa_d_x_xp is None:
                self.apa_d_x_xp = apa_compute_d_x_xp(apa_ref, apa_anti_ref)

        result = {}
        if "fad" in self.metrics:
            result["fad"] = frechet_distance(stem_cand, stem_ref)
        if "kd" in self.metrics:
            result.update(kernel_distance(stem_cand, stem_ref))          # candidate is features_1 (audio_metrics.py:260)
        if "prdc" in self.metrics:
            k = max(1, min(10, len(stem_ref), len(stem_cand)))
            result.update(prdc(stem_ref, stem_cand, k))
        if self.need_apa:
            result["apa"] = apa(apa_cand, apa_ref, apa_anti_ref, self.apa_d_x_xp)
        return result
