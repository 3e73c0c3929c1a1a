"""Deterministic synthetic inputs shared by the golden generator and the tests.

All inputs come from numpy's PCG64 ``default_rng(seed)`` (stable across numpy
versions and machines), cast to float32 - never from torch's generator.
"""
import numpy as np


def randn(seed, n, d, scale=1.0, shift=0.0):
    x = np.random.default_rng(seed).standard_normal((n, d))
    return (x * scale + shift).astype(np.float32)


def unit_norm(seed, n, d, shift=0.5):
    """CLAP-like rows: offset Gaussian, L2-normalised."""
    x = np.random.default_rng(seed).standard_normal((n, d)) + shift
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def decaying(seed, n, d, decades=4.0, scale=1.0, shift=0.0):
    """Gaussian with a geometrically decaying column scale (ill-conditioned cov)."""
    x = np.random.default_rng(seed).standard_normal((n, d)) * scale + shift
    return (x * np.logspace(0.0, -decades, d)).astype(np.float32)


def pair(kind, seed, n_ref, n_cand, d):
    """(reference, candidate) embedding sets of a named family."""
    if kind == "randn":
        return randn(seed, n_ref, d), randn(seed + 1, n_cand, d, 1.05, 0.05)
    if kind == "unit":
        return unit_norm(seed, n_ref, d, 0.5), unit_norm(seed + 1, n_cand, d, 0.55)
    if kind == "decay":
        return decaying(seed, n_ref, d), decaying(seed + 1, n_cand, d, 4.0, 1.05, 0.05)
    if kind == "shifted":
        return randn(seed, n_ref, d), randn(seed + 1, n_cand, d, 1.0, 0.1)
    raise ValueError(kind)


# ---- case tables (name -> parameters); the generator stores outputs under the same names
STATS_CASES = {
    # name: (seed, d, row-splits fed to add() one after the other)
    "tiny_1": (11, 8, [1]),
    "tiny_33": (12, 8, [33]),
    "chan_1_100_1000": (13, 8, [1, 100, 1000]),          # shape of the reference's tests/test_data.py:10-12
    "vggish_1000_b32": (14, 128, [32] * 31 + [8]),
    "clap_4096_b32": (15, 512, [32] * 128),
    "clap_4096_oneshot": (15, 512, [4096]),
    "ragged": (16, 24, [5, 1, 17, 2, 64, 1, 1, 300]),
}

FAD_CASES = {
    # name: (kind, seed, n_ref, n_cand, d)
    "vggish_1k": ("shifted", 21, 1000, 1000, 128),        # BASELINE config 1 shape
    "clap_4k": ("randn", 22, 4096, 4096, 512),
    "clap_unit_4k": ("unit", 23, 4096, 4096, 512),
    "decay_2k": ("decay", 24, 2000, 2000, 128),
    "rankdef_300": ("randn", 25, 300, 300, 512),
    "rankdef_40_vs_full": ("randn", 26, 40, 2000, 128),
    "tiny_d8": ("shifted", 27, 50, 60, 8),
    "clap_unit_100_vs_100": ("unit", 29, 100, 100, 512),  # fewer clips than dimensions, CLAP-shaped: the common small-eval regime
    "clap_unit_100_vs_4k": ("unit", 30, 4096, 100, 512),
    "clap_100k": ("randn", 28, 100000, 100000, 512),      # BASELINE config 2/3 shape
}

KD_CASES = {
    # name: (kind, seed, n1, n2, d)   (set 1 = features_1 = candidate side)
    "shrink_300_500": ("shifted", 31, 300, 500, 32),      # exercises kd.py:160-168
    "mid_3000": ("randn", 32, 3000, 3000, 64),
    "unit_2500": ("unit", 33, 2500, 2200, 128),
    "clap_100k": ("randn", 28, 100000, 100000, 512),
}

PRDC_CASES = {
    # name: (kind, seed, n_ref, n_cand, d, k)
    "randn_257_128_k1": ("randn", 41, 257, 257, 128, 1),
    "randn_257_128_k5": ("randn", 41, 257, 257, 128, 5),
    "shifted_1000_128_k5": ("shifted", 42, 1000, 1000, 128, 5),
    "randn_2000_512_k5": ("randn", 43, 2000, 2000, 512, 5),
    "randn_2000_512_k10": ("randn", 43, 2000, 2000, 512, 10),
    "unit_2000_512_k5": ("unit", 44, 2000, 2000, 512, 5),
    "unit_2000_512_k10": ("unit", 44, 2000, 2000, 512, 10),
    "ragged_300_500_24_k3": ("shifted", 45, 300, 500, 24, 3),
    "tiny_12_9_8_k2": ("shifted", 46, 12, 9, 8, 2),
    # N >= 8192 and D >= 128: the sets the HIP path sends through the symmetric k-NN kernel
    "randn_8300_128_k5": ("randn", 47, 8300, 8300, 128, 5),
    "unit_8300_8200_128_k10": ("unit", 48, 8300, 8200, 128, 10),
}

# Sets large enough for the production f16 filter kernels (am_knn_path == am_prdc_path == 3: >= 32768 rows) and
# still small enough for the reference's N x N formulation to fit this container's RAM (26 GB at 40k rows).
PRDC_LARGE_CASES = {
    "randn_33000_128_k5": ("randn", 141, 33000, 33000, 128, 5),
    "randn_33000_128_k10": ("randn", 141, 33000, 33000, 128, 10),
    "unit_33000_35000_128_k5": ("unit", 142, 33000, 35000, 128, 5),
    "unit_33000_35000_128_k10": ("unit", 142, 33000, 35000, 128, 10),
    "randn_40000_512_k5": ("randn", 143, 40000, 40000, 512, 5),
    "unit_40000_512_k10": ("unit", 144, 40000, 40000, 512, 10),
}

# float64 rows (round 5): what the reference's PCA projection hands to KD / PRDC (projection.py:20-21) and what a float64
# embedder yields.  The inputs are NOT cast to float32.
def randn64(seed, n, d, scale=1.0, shift=0.0):
    return np.random.default_rng(seed).standard_normal((n, d)) * scale + shift


def decaying64(seed, n, d, decades=1.0, scale=1.0, shift=0.0):
    x = np.random.default_rng(seed).standard_normal((n, d)) * scale + shift
    return x * np.logspace(0.0, -decades, d)


def pair64(kind, seed, n_ref, n_cand, d):
    if kind == "randn":
        return randn64(seed, n_ref, d), randn64(seed + 1, n_cand, d, 1.05, 0.05)
    if kind == "decay":
        return decaying64(seed, n_ref, d, 1.0, 1.0, 0.3), decaying64(seed + 1, n_cand, d, 1.0, 1.1, 0.35)
    raise ValueError(kind)


F64_CASES = {
    # name: (kind, seed, n_ref, n_cand, d of the rows fed in, n_pca or 0, k)
    "direct_2000_24_k5": ("randn", 201, 2000, 2000, 24, 0, 5),
    "direct_300_500_9_k3": ("randn", 202, 300, 500, 9, 0, 3),          # odd width, unequal sizes
    "direct_1500_40_k40": ("randn", 203, 1500, 1400, 40, 0, 40),      # k + 1 > 32: the select path
    "direct_2100_130_k10": ("decay", 204, 2100, 2050, 130, 0, 10),    # more than eight 16-element slabs, a ragged last one
    "pca8_20000_k5": ("decay", 205, 20000, 20000, 96, 8, 5),          # the reference's own projection (n_pca = 8), 20 000 rows
    "pca64_20000_k10": ("decay", 206, 20000, 20000, 96, 64, 10),
}


# The headline benchmark's inputs (bench.py): numpy-seeded so that the CPU oracle can reproduce them.
BENCH_SEED = 2026


def bench_pair(kind, n, d, seed=BENCH_SEED):
    """(reference, candidate) of bench.py: 'randn'  or 'clap' (unit-norm rows, offsets 0.5 / 0.55)."""
    return pair("unit" if kind == "clap" else "randn", seed, n, n, d)


# ---- end-to-end (A12/A13) case: synthetic (context, stem) pairs + a host-side numpy embedder
E2E = dict(sr=16000, win_dur=1.0, n_ref=60, n_cand=50, seconds=3, dim=24, seed=77, random_seed=1234)


def e2e_pairs(seed, n_items, seconds, sr, stem_gain=1.0):
    """n_items arrays of shape (seconds*sr, 2): column 0 = context, column 1 = stem."""
    rng = np.random.default_rng(seed)
    t = np.arange(seconds * sr) / sr
    out = []
    for _ in range(n_items):
        f0, f1 = rng.uniform(80, 400), rng.uniform(200, 1200)
        ctx = np.sin(2 * np.pi * f0 * t) * rng.uniform(0.2, 0.9) + 0.05 * rng.standard_normal(len(t))
        stem = np.sign(np.sin(2 * np.pi * f1 * t)) * rng.uniform(0.1, 0.6) * stem_gain + 0.05 * rng.standard_normal(len(t))
        out.append(np.stack([ctx, stem], axis=1).astype(np.float32))
    return out


class NumpyEmbedder:
    """Embedder protocol (sr / get_device / forward) computed entirely with numpy on the host,
    so the reference run and this build see bit-identical embeddings."""

    def __init__(self, dim=24, sr=16000, frame=400, seed=5):
        rng = np.random.default_rng(seed)
        self._sr, self.frame = sr, frame
        self.w = (rng.standard_normal((frame, dim)) / np.sqrt(frame)).astype(np.float64)

    @property
    def sr(self):
        return self._sr

    def get_device(self):
        import torch
        return torch.device("cpu")

    def forward(self, data, sr=None):
        import torch
        audio = np.asarray(data["audio"], dtype=np.float64)
        if audio.ndim == 1:
            audio = audio[None]
        n = audio.shape[1] // self.frame * self.frame
        frames = audio[:, :n].reshape(len(audio), -1, self.frame)
        h = np.tanh(3.0 * frames @ self.w)
        emb = np.concatenate([h.mean(1)[:, : self.w.shape[1] // 2], h.std(1)[:, self.w.shape[1] // 2:]], axis=1)
        return {"embedding": torch.as_tensor(emb.astype(np.float32))}


def e2e_mix(audio, sr):
    """custom mix function (protocol f(audio[n,2], sr) -> audio[n])"""
    return 0.6 * audio[:, 0] + 0.4 * audio[:, 1]
