"""Device-resident sufficient statistics container.

Host-side mirror of the reference's ``AudioMetricsData``
(src/audio_metrics/data.py:18-112): same attributes, same method names and
argument meaning, same quirks - but every tensor lives in MI355X HBM and every
computation is a call into the HIP library:

  add / recompute_stats  -> am_stats_f32 / am_stats_f64        (data.py:37-58; in the dtype of the rows, as the reference)
  _update_stats          -> am_stats_merge_f64                 (data.py:77-94)
  get_radii              -> am_knn_radii_f32 / am_knn_radii_f64 (data.py:60-66, prdc.py:4-14)
  _update_embeddings     -> amortised-doubling HBM buffer instead of the
                            reference's per-batch torch.cat (data.py:68-72)
"""
import numpy as np
import torch

from . import hip_ops as ops


def default_device():
    if not torch.cuda.is_available():
        from ._lib import HipLibraryError
        raise HipLibraryError("no MI355X visible (torch.cuda.is_available() is False); "
                              "audio_metrics_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def ensure_tensor(x, device=None):
    """Anything array-like as a torch tensor, optionally moved to `device` (role of data.py:6-9)."""
    t = x if torch.is_tensor(x) else torch.as_tensor(x)
    if device is None or device == "":
        return t
    return t.to(device, non_blocking=True)


def ensure_ndarray(x):
    """Tensors come back to the host as numpy arrays; everything else passes through (role of data.py:12-15)."""
    return x.detach().cpu().numpy() if torch.is_tensor(x) else x


def _ld_for(d):
    return (d + 3) // 4 * 4


class AudioMetricsData:
    def __init__(self, store_embeddings=True, device=None):
        self.n = self.mean = self.cov = None
        self.store_embeddings = bool(store_embeddings)
        self._embeddings = None           # [n, D] view of self._buf (property `embeddings`): f32, or f64 once float64 rows came in
        self.radii = {}                   # "radii_{k}" -> [n] in the dtype of the rows
        self.dtype = torch.float64        # dtype of the statistics
        self.stats_rows_dtype = None      # dtype of the rows the statistics were accumulated from (float32 once any float32 row came in;
                                          # None: unknown, e.g. a loaded state) - metrics/fad.py warns about n <= D unless float64
        self._device = torch.device(device) if device is not None else None
        self._buf = None                  # [capacity, ld] f32 / f64, rows 16-B aligned
        self._mean_spare = None           # second mean buffer of the one-launch add (am_stats_push_f32 reads one, writes the other)
        self._content_version = 0         # bumped whenever the stored rows change: validity of the cached PreparedSet
        self._prepared = None
        self._prepared_version = -1

    @property
    def embeddings(self):
        return self._embeddings

    @embeddings.setter
    def embeddings(self, rows):
        # any assignment - the class's own appends or a caller's - makes the cached PreparedSet (norms, f16 copy) stale
        self._embeddings = rows
        self._content_version += 1

    def _note_rows_dtype(self, dtype):
        """called BEFORE the rows are merged in: float64 only while every row so far was float64"""
        first = not self.n
        if dtype == torch.float64 and (first or self.stats_rows_dtype == torch.float64):
            self.stats_rows_dtype = torch.float64
        else:
            self.stats_rows_dtype = torch.float32

    # ------------------------------------------------------------ plumbing
    @property
    def device(self):
        if self._device is None:
            self._device = default_device()
        return self._device

    def _to_device_matrix(self, embeddings):
        e = ensure_tensor(embeddings)
        if e.dim() != 2:
            raise ValueError(f"embeddings must have shape [n, d], got {tuple(e.shape)}")
        if not e.is_cuda:
            e = e.to(self.device, non_blocking=True)
        elif self._device is None:
            self._device = e.device
        return ops.as_rows(e)             # float64 rows stay float64 (the reference keeps the dtype it is given), the rest is f32

    # ------------------------------------------------------------ reference API
    def serialize(self):
        """Plain dict in the reference's layout (data.py:28-29) with HOST tensors,
        so that ``torch.save`` / ``torch.load(weights_only=True)`` round-trip and
        reference-written state files stay loadable."""
        cpu = lambda t: t.detach().cpu() if isinstance(t, torch.Tensor) else t   # noqa: E731
        return dict(
            mean=cpu(self.mean), n=self.n, cov=cpu(self.cov), store_embeddings=self.store_embeddings,
            embeddings=None if self.embeddings is None else self.embeddings.detach().cpu().contiguous(),
            radii={k: cpu(v) for k, v in self.radii.items()}, dtype=self.dtype)

    @classmethod
    def deserialize(cls, state, device=None):
        self = cls(store_embeddings=state.get("store_embeddings", True), device=device)
        self.n = state.get("n")
        dev = self.device
        if state.get("mean") is not None:
            self.mean = ensure_tensor(state["mean"]).to(dev, torch.float64)
            self.cov = ensure_tensor(state["cov"]).to(dev, torch.float64)
        if state.get("embeddings") is not None:
            self._append(self._to_device_matrix(state["embeddings"]))
        # radii in the dtype they were written in (f64 for float64 rows - the reference's PCA output -, else f32)
        self.radii = {k: ensure_tensor(v).to(dev) for k, v in (state.get("radii") or {}).items()}
        self.dtype = state.get("dtype", torch.float64)
        return self

    def to(self, device):
        """This set on another GPU (statistics, stored rows and cached radii copied device to device)."""
        device = torch.device(device)
        if self.n is not None and device == self.device:
            return self
        other = AudioMetricsData(self.store_embeddings, device=device)
        if self.n is None:
            return other
        other.n = self.n
        other.mean, other.cov = self.mean.to(device), self.cov.to(device)
        if self.embeddings is not None:
            other._append(ops.as_rows(self.embeddings.to(device)))
        other.radii = {key: r.to(device) for key, r in self.radii.items()}
        return other

    def add(self, embeddings):
        """Batch statistics, Chan merge, row append (data.py:37-47).  Batches of up to am_stats_push_max_rows() rows - what
        the embedding pipeline feeds (embed.py:231-236) - take ONE kernel launch (am_stats_push_f32) instead of the
        statistics / merge / copy chain."""
        e = embeddings
        if torch.is_tensor(e) and e.dtype == torch.float64 or (isinstance(e, np.ndarray) and e.dtype == np.float64):
            return self._add_f64(e)
        if not (torch.is_tensor(e) and e.is_cuda and e.dtype == torch.float32 and e.dim() == 2 and e.stride(1) == 1):
            e = self._to_device_matrix(embeddings)
        elif self._device is None:
            self._device = e.device
        n = e.shape[0]
        if n == 0:
            raise ValueError("cannot add an empty batch of embeddings")
        self._note_rows_dtype(torch.float32)
        f32_store = self._buf is None or self._buf.dtype == torch.float32      # (a float64 store takes the general path)
        if n <= ops.stats_push_max_rows() and f32_store and self._push(e):
            return
        e = ops.as_matrix(e)
        mean, cov = ops.stats(e)          # n == 1 -> zero covariance (data.py:40-42)
        self._update_stats(mean, cov, n)
        if self.store_embeddings:
            self._update_embeddings(e)

    def _add_f64(self, embeddings):
        """add() of float64 rows: statistics in f64 from the f64 values and rows stored as float64, as the reference does
        (data.py:39-44, 68-72): the k-NN radii, membership counts and kernel distance of such a set run in f64 too."""
        e = ensure_tensor(embeddings)
        if e.dim() != 2:
            raise ValueError(f"embeddings must have shape [n, d], got {tuple(e.shape)}")
        if not e.is_cuda:
            e = e.to(self.device, non_blocking=True)
        elif self._device is None:
            self._device = e.device
        n = e.shape[0]
        if n == 0:
            raise ValueError("cannot add an empty batch of embeddings")
        self._note_rows_dtype(torch.float64)
        mean, cov = ops.stats_f64(e)
        self._update_stats(mean, cov, n)
        if self.store_embeddings:
            self._update_embeddings(e)

    def _push(self, e):
        """The one-launch form of add().  False (nothing done) when the running state is not in the plain (D,), (D, D) f64
        form the kernel updates in place (recompute_stats' (1, 1) quirk, foreign dtypes, another device)."""
        b, d = e.shape
        n_old = 0 if self.n is None else int(self.n)
        dev = e.device
        if n_old:
            mean, cov = self.mean, self.cov
            if not (mean.is_cuda and mean.device == dev and mean.dtype == torch.float64 and mean.numel() == d and mean.is_contiguous()
                    and cov.dtype == torch.float64 and tuple(cov.shape) == (d, d) and cov.is_contiguous() and cov.device == dev):
                return False
            if self.store_embeddings and self.embeddings is not None and self.embeddings.shape[1] != d:
                return False
        else:
            self.mean = None
            self.cov = torch.empty((d, d), dtype=torch.float64, device=dev)
        spare = self._mean_spare
        if spare is None or spare.numel() != d or spare.device != dev:
            spare = torch.empty(d, dtype=torch.float64, device=dev)
        rows_out, ld_out = None, 0
        if self.store_embeddings:
            held = 0 if self.embeddings is None else self.embeddings.shape[0]
            self._reserve(held + b, d, dev)
            rows_out, ld_out = self._buf[held], self._buf.stride(0)
        ops.stats_push(e, n_old, self.mean, spare, self.cov, rows_out, ld_out)
        self.mean, self._mean_spare = spare, self.mean          # (two mean buffers alternate: the kernel reads one, writes the other)
        self.n = n_old + b
        if self.store_embeddings:
            self.embeddings = self._buf[:held + b, :d]
        return True

    def recompute_stats(self):
        """One-shot statistics of the stored rows, in their dtype (data.py:49-58); a no-op without stored rows."""
        rows = self.embeddings
        if rows is None:
            return
        self.n = int(rows.shape[0])
        self.stats_rows_dtype = rows.dtype
        self.mean, self.cov = ops.stats(rows)
        if self.n < 2:
            # reference quirk kept on purpose: a (1, 1) zero matrix, not (D, D) (data.py:56)
            self.cov = torch.zeros((1, 1), dtype=self.dtype, device=self.device)

    def prepared(self):
        """The PreparedSet of the stored rows (norms, maxima, scaled f16 copy), computed once per content: the k-NN sweep
        and the membership counts of an evaluate() share it.  None without stored rows."""
        rows = self.embeddings
        if rows is None or rows.dtype == torch.float64:          # (the f64 kernels take the rows as they are)
            return None
        rows = ops.as_matrix(rows)
        # valid for exactly the content it was computed from: every append / load / move bumps _content_version, and a
        # matrix assigned to .embeddings from outside no longer is the buffer view the cache was taken from
        cached = self._prepared
        if cached is None or self._prepared_version != self._content_version or not cached.matches(rows):
            cached = self._prepared = ops.prepare(rows)
            self._prepared_version = self._content_version
        return cached

    def invalidate_prepared(self):
        """Call after editing stored rows IN PLACE (the class never does): the derived norms / f16 copy are recomputed."""
        self._content_version += 1

    def get_radii(self, k_neighbor):
        """k-NN radii of the stored rows, computed once per k and kept (like the reference's cache, data.py:60-66, an
        append does NOT invalidate it); None when no rows are stored."""
        slot = "radii_%d" % int(k_neighbor)
        if slot not in self.radii:
            if self.embeddings is None:
                return None
            self.radii[slot] = ops.knn_radii(self.embeddings, int(k_neighbor), prepared=self.prepared())
        return self.radii[slot]

    def _update_embeddings(self, embeddings):
        self._append(self._to_device_matrix(embeddings))

    def _reserve(self, rows, d, device, dtype=torch.float32):
        """Capacity for `rows` stored rows of `dtype`: amortised doubling of the HBM buffer (the reference re-concatenates the
        whole matrix per batch, data.py:68-72).  A float32 store that receives float64 rows becomes float64, as torch.cat's
        type promotion makes the reference's."""
        held = 0 if self.embeddings is None else self.embeddings.shape[0]
        if self._buf is not None and self._buf.dtype != dtype and self._buf.dtype == torch.float64:
            dtype = torch.float64
        if self._buf is None or self._buf.shape[0] < rows or self._buf.device != device or self._buf.dtype != dtype:
            cap = max(rows, 2 * (0 if self._buf is None else self._buf.shape[0]), 1024)
            buf = torch.empty((cap, _ld_for(d)), dtype=dtype, device=device)
            if held:
                buf[:held, :d] = self.embeddings
            self._buf = buf

    def _append(self, e):
        n_new, d = e.shape
        n_old = 0 if self.embeddings is None else self.embeddings.shape[0]
        self._reserve(n_old + n_new, d, e.device, e.dtype if e.dtype == torch.float64 else torch.float32)
        self._buf[n_old:n_old + n_new, :d] = e
        self.embeddings = self._buf[:n_old + n_new, :d]

    def __len__(self):
        return 0 if self.n is None else int(self.n)

    def _update_stats(self, mean, cov, n):
        if self.n is None:
            self.mean, self.cov, self.n = mean, cov, n
            return
        d = self.mean.numel()
        if self.n == 1 and tuple(self.cov.shape) != (d, d):
            # recompute_stats' n == 1 quirk left a (1, 1) zero matrix; the reference's merge broadcasts it with
            # weight (n1 - 1) = 0 (data.py:85-92), which equals merging a D x D zero matrix
            self.cov = torch.zeros((d, d), dtype=self.dtype, device=self.mean.device)
        if n == 1 and tuple(cov.shape) != (d, d):
            cov = torch.zeros((d, d), dtype=self.dtype, device=self.mean.device)
        self.mean, self.cov = ops.stats_merge(self.n, self.mean, self.cov, n, mean, cov, inplace=True)
        self.n = self.n + n

    def __iadd__(self, other):
        """Absorb another set: Chan merge of the statistics and, when rows are kept, an append (data.py:96-106).  An empty
        receiver adopts the other side's storage mode; mixing modes afterwards is an error."""
        if not isinstance(other, AudioMetricsData):
            raise AssertionError("can only merge AudioMetricsData objects")
        if len(other) > 0:
            if len(self) == 0:
                self.store_embeddings = other.store_embeddings
            elif self.store_embeddings != other.store_embeddings:
                raise AssertionError("cannot merge a set that stores its embeddings with one that does not")
            self._note_rows_dtype(other.stats_rows_dtype)          # (unknown counts as float32: the warning stays on)
            self._update_stats(other.mean.clone(), other.cov.clone(), other.n)
            if self.store_embeddings:
                self._update_embeddings(other.embeddings)
        return self

    def __add__(self, other):
        total = AudioMetricsData(device=self._device)
        for part in (self, other):
            total += part
        return total
