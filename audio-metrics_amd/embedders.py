"""Embedder plug-ins.  The embedder forward is ordinary PyTorch-ROCm host code on one
GPU (north star); only the protocol of the reference is part of this build:

    embedder.sr : int
    embedder.get_device() -> torch.device
    embedder.forward({"audio": ndarray[B, n], "category": ndarray[B]}) -> {"embedding": Tensor[B, D]}

Registry names follow the reference's embedders/__init__.py:9-56.  LAION-CLAP and VGGish
need third-party packages and checkpoints that cannot be fetched here; they are
constructed lazily and fail with a clear message.  ``SyntheticEmbedder`` is a small
deterministic torch module with the same protocol (used for benchmarks / tests when no
checkpoint is available - say so in any result that uses it)."""
import numpy as np
import torch

LAION_CLAP_MUSIC_SPEECH_CHECKPOINT_URL = "https://huggingface.co/lukewys/laion_clap/resolve/main/music_speech_audioset_epoch_15_esc_89.98.pt"
LAION_CLAP_MUSIC_CHECKPOINT_URL = "https://huggingface.co/lukewys/laion_clap/resolve/main/music_audioset_epoch_15_esc_90.14.pt"
LAION_CLAP_LAYERS = ["audio_projection.0", "audio_projection.2"]


def local_checkpoint(ckpt):
    """Path of a checkpoint given as a path or URL: URLs are fetched once into ~/.cache/audio_metrics (the reference
    downloads into its appdirs cache, util/get_url.py); an unreachable URL is a clear error here, not a confusing one
    inside the model loader."""
    import os
    import urllib.request
    if "://" not in str(ckpt):
        return ckpt
    cache = os.path.join(os.path.expanduser("~"), ".cache", "audio_metrics")
    path = os.path.join(cache, os.path.basename(str(ckpt)))
    if not os.path.exists(path):
        os.makedirs(cache, exist_ok=True)
        try:
            urllib.request.urlretrieve(ckpt, path + ".part")
        except OSError as e:
            raise RuntimeError(f"cannot fetch the checkpoint {ckpt} ({e}); download it yourself and pass its path "
                               f"as `ckpt`, or place it at {path}") from e
        os.replace(path + ".part", path)
    return path


class LaionCLAP:
    """LAION-CLAP HTSAT-base audio tower, 512-d, 48 kHz (reference embedders/clap.py:10-60)."""

    def __init__(self, ckpt=None, layer=None, device=None):
        try:
            import laion_clap
        except ImportError as e:
            raise ImportError("embedder 'laion_clap_*' needs the `laion_clap` package and its checkpoint, which are not "
                              "available in this environment; pass an embedder object or embedder='synthetic_clap512'") from e
        if ckpt is None:
            ckpt = LAION_CLAP_MUSIC_CHECKPOINT_URL
        ckpt = local_checkpoint(ckpt)                       # load_ckpt takes a file, not a URL (clap.py:16)
        self.clap = laion_clap.CLAP_Module(enable_fusion=False, amodel="HTSAT-base")
        self.clap.load_ckpt(ckpt, verbose=False)
        if device is not None:
            self.clap = self.clap.to(device)
        self.layer = layer

    @property
    def sr(self):
        return self.clap.model.audio_cfg.sample_rate

    def get_device(self):
        return next(self.clap.parameters()).device

    @torch.no_grad()
    def forward(self, data, sr=None):
        audio = torch.from_numpy(data["audio"]).float().to(self.get_device(), non_blocking=True)
        if audio.ndim == 1:
            audio = audio.unsqueeze(0)
        captured = {}
        hook = None
        if self.layer:
            hook = self.clap.get_submodule(f"model.{self.layer}").register_forward_hook(
                lambda mod, inp, out: captured.__setitem__("out", out.detach()))
        embedding = self.clap.get_audio_embedding_from_data(audio, use_tensor=True)
        if hook is not None:
            hook.remove()
            embedding = captured["out"]
        return {"embedding": embedding}


class VGGish:
    """VGGish 128-d pre-activation embedding, 16 kHz, time-averaged (reference embedders/vggish.py:5-33)."""

    def __init__(self, device=None):
        try:
            self.model = torch.hub.load("harritaylor/torchvggish", "vggish")
        except Exception as e:  # no network / no cache
            raise ImportError("embedder 'vggish' needs the torch.hub model 'harritaylor/torchvggish', which cannot be "
                              "fetched in this environment; pass an embedder object instead") from e
        self.model.eval()
        self.model.postprocess = False
        self.model.preprocess = False
        self.model.embeddings[5] = torch.nn.Identity()
        if device is not None:
            self.model = self.model.to(device)

    @property
    def sr(self):
        return 16000

    def get_device(self):
        return next(self.model.parameters()).device

    @torch.no_grad()
    def forward(self, data, sr=None):
        device = self.get_device()
        self.model.device = device
        audio = data["audio"]
        spec = torch.cat([self.model._preprocess(item, self.sr) for item in audio]).to(device)
        per_item = spec.shape[0] // len(audio)
        emb = self.model(spec).reshape(len(audio), per_item, -1).mean(1)
        return {"embedding": emb}


class SyntheticEmbedder(torch.nn.Module):
    """Deterministic stand-in with the embedder protocol: framed log-power features ->
    two random-init linear layers -> D-dim embedding, L2-normalised like CLAP's."""

    def __init__(self, dim=512, sr=48000, frame=1024, seed=0, device=None, normalize=True):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self._sr, self.frame, self.normalize = sr, frame, normalize
        nfreq = frame // 2 + 1
        self.w1 = torch.nn.Parameter(torch.randn(nfreq, 256, generator=g) / nfreq ** 0.5, requires_grad=False)
        self.w2 = torch.nn.Parameter(torch.randn(512, dim, generator=g) / 512 ** 0.5, requires_grad=False)
        if device is not None:
            self.to(device)

    @property
    def sr(self):
        return self._sr

    def get_device(self):
        return self.w1.device

    @torch.no_grad()
    def forward(self, data, sr=None):
        audio = torch.as_tensor(np.asarray(data["audio"])).float().to(self.get_device(), non_blocking=True)
        if audio.ndim == 1:
            audio = audio.unsqueeze(0)
        n = audio.shape[1] // self.frame * self.frame
        frames = audio[:, :n].reshape(audio.shape[0], -1, self.frame)
        spec = torch.log1p(torch.fft.rfft(frames * torch.hann_window(self.frame, device=audio.device)).abs() ** 2)
        h = torch.tanh(spec @ self.w1)                                    # [B, T, 256]
        pooled = torch.cat([h.mean(1), h.std(1, unbiased=False)], dim=1)  # [B, 512]
        emb = pooled @ self.w2
        if self.normalize:
            emb = torch.nn.functional.normalize(emb, dim=-1)
        return {"embedding": emb}


EMBEDDERS = {
    "laion_clap_music": (LaionCLAP, {"ckpt": LAION_CLAP_MUSIC_CHECKPOINT_URL}),
    "laion_clap_music_l-2": (LaionCLAP, {"ckpt": LAION_CLAP_MUSIC_CHECKPOINT_URL, "layer": LAION_CLAP_LAYERS[0]}),
    "laion_clap_music_l-1": (LaionCLAP, {"ckpt": LAION_CLAP_MUSIC_CHECKPOINT_URL, "layer": LAION_CLAP_LAYERS[1]}),
    "laion_clap_music_speech": (LaionCLAP, {"ckpt": LAION_CLAP_MUSIC_SPEECH_CHECKPOINT_URL}),
    "laion_clap_music_speech_l-2": (LaionCLAP, {"ckpt": LAION_CLAP_MUSIC_SPEECH_CHECKPOINT_URL, "layer": LAION_CLAP_LAYERS[0]}),
    "laion_clap_music_speech_l-1": (LaionCLAP, {"ckpt": LAION_CLAP_MUSIC_SPEECH_CHECKPOINT_URL, "layer": LAION_CLAP_LAYERS[1]}),
    "vggish": (VGGish, {}),
    "synthetic_clap512": (SyntheticEmbedder, {"dim": 512, "sr": 48000}),
    "synthetic_vggish128": (SyntheticEmbedder, {"dim": 128, "sr": 16000, "normalize": False}),
}
DEFAULT_EMBEDDER = "laion_clap_music"
