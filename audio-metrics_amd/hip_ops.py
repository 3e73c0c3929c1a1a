"""Thin torch<->C-ABI glue: device tensors in, device tensors out.

PyTorch is used only as the device-memory allocator and stream owner; every
computation below is a call into libaudio_metrics_hip.so on torch's current
HIP stream.  There is no CPU path: CPU tensors are rejected.
"""
import ctypes
import os

import torch

from . import _lib


def _stream(device=None):
    """torch's current stream ON `device` (the tensors' device, not the thread's current device)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


_SIDE_STREAMS = {}


def upload_host_array(array, device):
    """Host array -> device tensor usable on the current stream, without blocking the host behind the work already
    queued there: the (pageable, host-synchronous) copy is issued on an otherwise idle side stream and the current
    stream waits for it on the device.  Pinned host memory is deliberately NOT used: on this platform both
    `Tensor.pin_memory()` per call and kernels reading a pooled pinned buffer in place stall the host for 75-100 ms
    every few calls (measured on the 2 x 800 KB kernel-distance index tables: a 10 ms evaluate became a 90 ms one every
    third call), while pageable copies of the same data never did."""
    import numpy as np
    t = torch.as_tensor(np.ascontiguousarray(array))
    if getattr(device, "type", "cpu") != "cuda":
        return t
    main = torch.cuda.current_stream(device)
    side = _SIDE_STREAMS.get(device.index)
    if side is None:
        side = _SIDE_STREAMS[device.index] = torch.cuda.Stream(device)
    with torch.cuda.stream(side):
        d = t.to(device)
    done = torch.cuda.Event()
    done.record(side)
    main.wait_event(done)
    d.record_stream(main)
    return d


def _index_table(t, name):
    """int64 [S, m] index table on the device."""
    _require_cuda(t, name)
    return t.to(torch.int64).contiguous()


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the stream the
    kernels are launched on (torch's current stream).  Used by bench.py:
        with KernelTimer() as t: ...;  t.summary() -> {name: (calls, total_ms)}"""
    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.records:
            c, t = out.get(name, (0, 0.0))
            out[name] = (c + 1, t + a.elapsed_time(b))
        return out


KERNEL_KNN, KERNEL_PRDC_CROSS, KERNEL_KNN_VERIFY, KERNEL_PRDC_VERIFY = 0, 1, 2, 3        # enum am_clocked_kernel


def kernel_clock_enable(on=True):
    """Bracket every launch of the two tile kernels with hipEvents inside the library (bench.py)."""
    _lib.check(_lib.load().am_kernel_clock_enable(1 if on else 0), "am_kernel_clock_enable")


def kernel_clock_read(kernel):
    """(launches, total_ms) of the clocked kernel since the last read; waits for the recorded launches."""
    import ctypes
    n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
    _lib.check(_lib.load().am_kernel_clock_read(int(kernel), ctypes.byref(n), ctypes.byref(ms)), "am_kernel_clock_read")
    return n.value, ms.value


FILTER_STATS_NAMES = ("knn_calls", "knn_queued", "knn_spilled", "knn_verified_pairs", "knn_fallback_rows",
                      "prdc_calls", "prdc_queued", "prdc_overflow_queue", "prdc_fallback_calls", "bound_ratio_max", "bound_pairs")
_FILTER_STATS = {}


def filter_stats_enable(device, on=True):
    """Have the filter-form PRDC entry points add their queue / verification / fallback counts to a device buffer
    (am_filter_stats_enable); filter_stats_read() returns and clears them."""
    device = torch.device(device)
    with torch.cuda.device(device):
        if on:
            buf = torch.zeros(16, dtype=torch.int64, device=device)
            _FILTER_STATS[device.index] = buf
            _lib.check(_lib.load().am_filter_stats_enable(_ptr(buf)), "am_filter_stats_enable")
        else:
            _FILTER_STATS.pop(device.index, None)
            _lib.check(_lib.load().am_filter_stats_enable(ctypes.c_void_p(None)), "am_filter_stats_enable")


def filter_stats_read(device):
    buf = _FILTER_STATS[torch.device(device).index]
    values = buf.cpu().tolist()
    buf.zero_()
    out = dict(zip(FILTER_STATS_NAMES, values))
    # slot 9 holds the bit pattern of a float: the largest measured |f16 value - exact value| / (fast_c (|x|^2 + G))
    import struct
    out["bound_ratio_max"] = struct.unpack("<f", struct.pack("<I", int(out["bound_ratio_max"]) & 0xffffffff))[0]
    return out


def _call(lib, name, device, *args):
    """Call entry point `name` with `device` current and torch's current stream of THAT device appended as the
    trailing am_stream_t argument: the library launches on the device that is current in the calling thread, and the
    tensors may live on another one than the thread's (AudioMetrics(device_indices=[1]) while cuda:0 is current)."""
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device)
        args = (*args, ctypes.c_void_p(stream.cuda_stream))
        timer = KernelTimer.active
        if timer is not None:
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record(stream)
            status = getattr(lib, name)(*args)
            b.record(stream)
            timer.records.append((name, a, b))
        else:
            status = getattr(lib, name)(*args)
    _lib.check(status, name)


def _require_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HipLibraryError(
            f"{name} must be a tensor resident on the MI355X (got {type(t).__name__} on "
            f"{getattr(t, 'device', 'host')}); this package has no CPU fallback")


def _same_device(*tensors):
    dev = tensors[0].device
    for t in tensors[1:]:
        if t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} and {t.device}")
    return dev


def is_f64(t):
    return torch.is_tensor(t) and t.dtype == torch.float64


def as_matrix(e, name="embeddings"):
    """(N, D) f32 device matrix with unit column stride, 16-B aligned rows
    (row stride % 4 == 0).  Copies (zero-padding the row stride) only if needed."""
    _require_cuda(e, name)
    if e.dim() != 2:
        raise ValueError(f"{name} must be 2-D, got shape {tuple(e.shape)}")
    if e.dtype != torch.float32:
        e = e.to(torch.float32)
    n, d = e.shape
    ok = e.stride(1) == 1 and e.stride(0) % 4 == 0 and e.stride(0) >= d and e.data_ptr() % 16 == 0
    if n > 0 and not ok:
        ld = (d + 3) // 4 * 4
        buf = torch.zeros((n, ld), dtype=torch.float32, device=e.device)
        buf[:, :d] = e
        e = buf[:, :d]
    return e


def as_matrix64(e, name="embeddings"):
    """(N, D) f64 device matrix with unit column stride (the f64 kernels need no alignment: any row stride >= D)."""
    _require_cuda(e, name)
    if e.dim() != 2:
        raise ValueError(f"{name} must be 2-D, got shape {tuple(e.shape)}")
    if e.dtype != torch.float64:
        e = e.to(torch.float64)
    n, d = e.shape
    if n > 0 and (e.stride(1) != 1 or (n > 1 and e.stride(0) < d)):
        e = e.contiguous()
    return e


def as_rows(e, name="embeddings"):
    """The matrix form the kernels take, in the dtype the reference computes in (data.py:39-44, prdc.py:12, kd.py:115):
    float64 rows stay float64 (the *_f64 entry points), everything else is float32."""
    return as_matrix64(e, name) if is_f64(e) else as_matrix(e, name)


def _ld64(e):
    return e.stride(0) if e.shape[0] > 1 else e.shape[1]


def _ld(e):
    """Leading dimension handed to the C ABI (a 1-row view may carry any stride)."""
    if e.shape[0] > 1:
        return e.stride(0)
    return (e.shape[1] + 3) // 4 * 4


def _f64(t, name):
    _require_cuda(t, name)
    if t.dtype != torch.float64 or not t.is_contiguous():
        t = t.to(torch.float64).contiguous()
    return t


# ------------------------------------------------------------------ statistics
def stats(e):
    """mean f64[D], unbiased covariance f64[D, D] of the rows of e (data.py:37-58), computed in the dtype of e."""
    if is_f64(e):
        return stats_f64(e)
    lib = _lib.load()
    e = as_matrix(e)
    n, d = e.shape
    mean = torch.empty(d, dtype=torch.float64, device=e.device)
    cov = torch.empty((d, d), dtype=torch.float64, device=e.device)
    nb = lib.am_stats_workspace_bytes(n, d)
    ws = _workspace(nb, e.device)
    _call(lib, "am_stats_f32", e.device, _ptr(e), n, d, _ld(e), _ptr(mean), _ptr(cov), _ptr(ws), nb)
    return mean, cov


def stats_f64(e):
    """mean f64[D], unbiased covariance f64[D, D] of the rows of a FLOAT64 device matrix, every step in f64 (am_stats_f64):
    the reference computes in the dtype it is given (data.py:39-44)."""
    lib = _lib.load()
    _require_cuda(e, "embeddings")
    if e.dim() != 2 or e.dtype != torch.float64:
        raise ValueError(f"stats_f64 takes a 2-D float64 matrix, got {tuple(e.shape)} {e.dtype}")
    if e.stride(1) != 1 or (e.shape[0] > 1 and e.stride(0) < e.shape[1]):
        e = e.contiguous()
    n, d = e.shape
    mean = torch.empty(d, dtype=torch.float64, device=e.device)
    cov = torch.empty((d, d), dtype=torch.float64, device=e.device)
    nb = lib.am_stats_f64_workspace_bytes(n, d)
    ws = _workspace(nb, e.device)
    _call(lib, "am_stats_f64", e.device, _ptr(e), n, d, e.stride(0) if n > 1 else d, _ptr(mean), _ptr(cov), _ptr(ws), nb)
    return mean, cov


def colsum(e):
    lib = _lib.load()
    if is_f64(e):
        e = as_matrix64(e)
        n, d = e.shape
        out = torch.empty(d, dtype=torch.float64, device=e.device)
        nb = lib.am_stats_f64_workspace_bytes(n, d)
        ws = _workspace(nb, e.device)
        _call(lib, "am_colsum_f64", e.device, _ptr(e), n, d, _ld64(e), _ptr(out), _ptr(ws), nb)
        return out
    e = as_matrix(e)
    n, d = e.shape
    out = torch.empty(d, dtype=torch.float64, device=e.device)
    nb = lib.am_stats_workspace_bytes(n, d)
    ws = _workspace(nb, e.device)
    _call(lib, "am_colsum_f32", e.device, _ptr(e), n, d, _ld(e), _ptr(out), _ptr(ws), nb)
    return out


def scatter(e, mean):
    """sum_n (x_n - mean)(x_n - mean)^T, not divided (multi-GPU building block)."""
    lib = _lib.load()
    mean = _f64(mean, "mean")
    if is_f64(e):
        e = as_matrix64(e)
        n, d = e.shape
        out = torch.empty((d, d), dtype=torch.float64, device=e.device)
        nb = lib.am_stats_f64_workspace_bytes(n, d)
        ws = _workspace(nb, e.device)
        _call(lib, "am_scatter_f64", e.device, _ptr(e), n, d, _ld64(e), _ptr(mean), _ptr(out), _ptr(ws), nb)
        return out
    e = as_matrix(e)
    n, d = e.shape
    out = torch.empty((d, d), dtype=torch.float64, device=e.device)
    nb = lib.am_stats_workspace_bytes(n, d)
    ws = _workspace(nb, e.device)
    _call(lib, "am_scatter_f32", e.device, _ptr(e), n, d, _ld(e), _ptr(mean), _ptr(out), _ptr(ws), nb)
    return out


def stats_merge(n1, mean1, cov1, n2, mean2, cov2, inplace=False):
    """Chan merge (data.py:77-94).  inplace=True overwrites (mean1, cov1)."""
    lib = _lib.load()
    mean1, cov1, mean2, cov2 = (_f64(t, "stats") for t in (mean1, cov1, mean2, cov2))
    d = mean1.numel()
    # the C ABI sees raw pointers: a (1, 1) covariance (recompute_stats' n == 1 quirk) would be read and written as D x D
    if mean2.numel() != d or tuple(cov1.shape) != (d, d) or tuple(cov2.shape) != (d, d):
        raise ValueError(f"stats_merge: inconsistent shapes mean {tuple(mean1.shape)}/{tuple(mean2.shape)}, "
                         f"cov {tuple(cov1.shape)}/{tuple(cov2.shape)}")
    _same_device(mean1, cov1, mean2, cov2)
    om = mean1 if inplace else torch.empty_like(mean1)
    oc = cov1 if inplace else torch.empty_like(cov1)
    _call(lib, "am_stats_merge_f64", mean1.device, int(n1), _ptr(mean1), _ptr(cov1), int(n2), _ptr(mean2), _ptr(cov2), d,
                                      _ptr(om), _ptr(oc))
    return om, oc


_PUSH_MAX = None


def stats_push_max_rows():
    global _PUSH_MAX
    if _PUSH_MAX is None:
        _PUSH_MAX = int(_lib.load().am_stats_push_max_rows())
    return _PUSH_MAX


def stats_push(e, n_old, mean_in, mean_out, cov, rows_out=None, ld_out=0):
    """One launch for AudioMetricsData.add(batch) of a small batch (am_stats_push_f32; data.py:37-47, 68-72, 77-94): batch
    statistics, Chan merge into (n_old, mean_in, cov) -> (mean_out, cov in place), rows copied to `rows_out` (the row
    n_old of the store, row stride ld_out) when given.  The caller owns every buffer; nothing is allocated here - this is
    the per-batch hot path of the embedding pipeline, so the glue is kept to one ctypes call."""
    lib = _lib.load()
    dev = e.device
    with torch.cuda.device(dev):
        status = lib.am_stats_push_f32(e.data_ptr(), e.shape[0], e.shape[1], e.stride(0) if e.shape[0] > 1 else e.shape[1],
                                       int(n_old), mean_in.data_ptr() if mean_in is not None else None, mean_out.data_ptr(),
                                       cov.data_ptr(), rows_out.data_ptr() if rows_out is not None else None, int(ld_out),
                                       torch.cuda.current_stream(dev).cuda_stream)
    if status != 0:
        _lib.check(status, "am_stats_push_f32")


# ------------------------------------------------------------------ PCA projection support
def eigh_descending(a, max_sweeps=40):
    """(eigenvalues f64[D] descending, eigenvectors f64[D, D] with row i = vector i) of a symmetric PSD matrix."""
    lib = _lib.load()
    a = _f64(a, "matrix")
    d = a.shape[0]
    if a.dim() != 2 or a.shape[1] != d:
        raise ValueError(f"expected a square matrix, got {tuple(a.shape)}")
    evals = torch.empty(d, dtype=torch.float64, device=a.device)
    evecs = torch.empty((d, d), dtype=torch.float64, device=a.device)
    nb = lib.am_eigh_workspace_bytes(d)
    ws = _workspace(nb, a.device)
    _call(lib, "am_eigh_sym_f64", a.device, _ptr(a), d, _ptr(evals), _ptr(evecs), int(max_sweeps), _ptr(ws), nb)
    return evals, evecs


def project(x, mean, components):
    """(x - mean) @ components.T as f64 [N, p] (IncrementalPCA.transform)."""
    lib = _lib.load()
    x = as_rows(x)
    mean, components = _f64(mean, "mean"), _f64(components, "components")
    n, d = x.shape
    p = components.shape[0]
    if mean.numel() != d or components.dim() != 2 or components.shape[1] != d:
        raise ValueError(f"projection shapes do not match: x {tuple(x.shape)}, mean {tuple(mean.shape)}, components {tuple(components.shape)}")
    _same_device(x, mean, components)
    out = torch.empty((n, p), dtype=torch.float64, device=x.device)
    if n > 0 and is_f64(x):
        _call(lib, "am_project_rows_f64", x.device, _ptr(x), n, _ld64(x), d, _ptr(mean), _ptr(components), p, _ptr(out))
    elif n > 0:
        _call(lib, "am_project_f64", x.device, _ptr(x), n, _ld(x), d, _ptr(mean), _ptr(components), p, _ptr(out))
    return out


# ------------------------------------------------------------------ Frechet
def frechet(mu_x, cov_x, mu_y, cov_y, max_iter=64, tol=1e-13):
    lib = _lib.load()
    mu_x, cov_x, mu_y, cov_y = (_f64(t, "stats") for t in (mu_x, cov_x, mu_y, cov_y))
    d = mu_x.numel()
    if cov_x.shape != (d, d) or cov_y.shape != (d, d) or mu_y.numel() != d:
        raise ValueError(f"inconsistent shapes: mu {tuple(mu_x.shape)}/{tuple(mu_y.shape)}, "
                         f"cov {tuple(cov_x.shape)}/{tuple(cov_y.shape)}")
    _same_device(mu_x, cov_x, mu_y, cov_y)
    out = (ctypes.c_double * 4)()
    nb = lib.am_frechet_workspace_bytes(d)
    ws = _workspace(nb, mu_x.device)
    _call(lib, "am_frechet_f64", mu_x.device, _ptr(mu_x), _ptr(cov_x), _ptr(mu_y), _ptr(cov_y), d, int(max_iter), float(tol),
                                  ctypes.cast(out, ctypes.c_void_p), _ptr(ws), nb)
    return dict(fd=out[0], tr_sqrt=out[1], iters=int(out[2]), resid=out[3])


class FrechetJob:
    """A Frechet solve in flight on a side stream (frechet_async).  Nothing here blocks the host until .result()."""

    def __init__(self, stats, max_iter, tol):
        lib = _lib.load()
        self._stats = stats                                  # keeps the operands alive
        self._max_iter, self._tol = int(max_iter), float(tol)
        dev = stats[0].device
        self._d = stats[0].numel()
        self._nb = lib.am_frechet_workspace_bytes(self._d)
        main = torch.cuda.current_stream(dev)
        self._side = _SIDE_STREAMS.get(("fad", dev.index))
        if self._side is None:
            self._side = _SIDE_STREAMS[("fad", dev.index)] = torch.cuda.Stream(dev)
        self._side.wait_stream(main)                         # the statistics were produced on the caller's stream
        with torch.cuda.stream(self._side):
            self._ws = _workspace(self._nb, dev)
            self._out = torch.empty(8, dtype=torch.float64, device=dev)
            self._next = 0
            self._enqueue(lib.am_frechet_first_block())
        for t in stats:
            t.record_stream(self._side)

    def _enqueue(self, n_iter):
        lib = _lib.load()
        mu_x, cov_x, mu_y, cov_y = self._stats
        n_iter = min(int(n_iter), self._max_iter - self._next)
        with torch.cuda.stream(self._side):
            _call(lib, "am_frechet_enqueue_f64", mu_x.device, _ptr(mu_x), _ptr(cov_x), _ptr(mu_y), _ptr(cov_y), self._d,
                  self._next, n_iter, self._max_iter, self._tol, _ptr(self._out), _ptr(self._ws), self._nb)
        self._next += n_iter

    def result(self):
        lib = _lib.load()
        while True:
            with torch.cuda.stream(self._side):
                out = self._out.cpu().tolist()               # the one host read (synchronises the side stream only)
            if int(out[4]) != 0 or self._next >= self._max_iter:
                break
            self._enqueue(lib.am_frechet_first_block())      # ill-conditioned product: another block of iterations
        torch.cuda.current_stream(self._stats[0].device).wait_stream(self._side)
        if int(out[4]) == 4:
            raise _lib.HipLibraryError("am_frechet_enqueue_f64: non-finite covariance product or trace in Newton-Schulz")
        return dict(fd=out[0], tr_sqrt=out[1], iters=int(out[2]), resid=out[3])


def frechet_async(mu_x, cov_x, mu_y, cov_y, max_iter=64, tol=1e-13):
    """frechet() enqueued on a side stream behind the work already queued on the current one; the kernels that the
    caller launches next on its own stream overlap it.  The solver decides convergence on the device, so no helper thread
    and no host polling are involved: .result() reads five doubles."""
    stats = tuple(_f64(t, "stats") for t in (mu_x, cov_x, mu_y, cov_y))
    d = stats[0].numel()
    if stats[1].shape != (d, d) or stats[3].shape != (d, d) or stats[2].numel() != d:
        raise ValueError("inconsistent shapes of the statistics")
    _same_device(*stats)
    return FrechetJob(stats, max_iter, tol)


def apa_scalar(d_y_x, d_y_xp, d_x_xp):
    return _lib.load().am_apa_f64(float(d_y_x), float(d_y_xp), float(d_x_xp))


# ------------------------------------------------------------------ kernel distance
def kd_poly(x, y, idx1, idx2, gamma, coef0, degree):
    """Per-subset unbiased MMD^2 (f64[S] device tensor).  idx1/idx2: int64 [S, m].  float64 features (either side: numpy's
    matmul promotes, kd.py:115) run every product and sum in f64 (am_kd_poly_f64)."""
    lib = _lib.load()
    idx1, idx2 = _index_table(idx1, "idx1"), _index_table(idx2, "idx2")
    s, m = idx1.shape
    if is_f64(x) or is_f64(y):
        x, y = as_matrix64(x, "features_1"), as_matrix64(y, "features_2")
        out = torch.empty(s, dtype=torch.float64, device=x.device)
        nb = lib.am_kd_f64_workspace_bytes(s, m)
        ws = _workspace(nb, x.device)
        _call(lib, "am_kd_poly_f64", x.device, _ptr(x), x.shape[0], _ld64(x), _ptr(y), y.shape[0], _ld64(y), x.shape[1],
              _ptr(idx1), _ptr(idx2), s, m, float(gamma), float(coef0), int(degree), _ptr(out), _ptr(ws), nb)
        return out
    x, y = as_matrix(x, "features_1"), as_matrix(y, "features_2")
    out = torch.empty(s, dtype=torch.float64, device=x.device)
    nb = lib.am_kd_poly_workspace_bytes(s, m, x.shape[1])
    ws = _workspace(nb, x.device)
    _call(lib, "am_kd_poly_f32", x.device, _ptr(x), x.shape[0], _ld(x), _ptr(y), y.shape[0], _ld(y), x.shape[1],
                                  _ptr(idx1), _ptr(idx2), s, m, float(gamma), float(coef0), int(degree),
                                  _ptr(out), _ptr(ws), nb)
    return out


def kd_rbf(x, y, idx1, idx2, sigma):
    """Per-subset unbiased MMD^2 with the RBF kernel exp(-|x-y|^2 / (2 sigma^2)) (f64[S] device tensor)."""
    lib = _lib.load()
    idx1, idx2 = _index_table(idx1, "idx1"), _index_table(idx2, "idx2")
    s, m = idx1.shape
    if is_f64(x) or is_f64(y):
        x, y = as_matrix64(x, "features_1"), as_matrix64(y, "features_2")
        out = torch.empty(s, dtype=torch.float64, device=x.device)
        nb = lib.am_kd_f64_workspace_bytes(s, m)
        ws = _workspace(nb, x.device)
        _call(lib, "am_kd_rbf_f64", x.device, _ptr(x), x.shape[0], _ld64(x), _ptr(y), y.shape[0], _ld64(y), x.shape[1], _ptr(idx1),
              _ptr(idx2), s, m, float(sigma), _ptr(out), _ptr(ws), nb)
        return out
    x, y = as_matrix(x, "features_1"), as_matrix(y, "features_2")
    out = torch.empty(s, dtype=torch.float64, device=x.device)
    nb = lib.am_kd_rbf_workspace_bytes(s, m)
    ws = _workspace(nb, x.device)
    _call(lib, "am_kd_rbf_f32", x.device, _ptr(x), x.shape[0], _ld(x), _ptr(y), y.shape[0], _ld(y), x.shape[1], _ptr(idx1), _ptr(idx2),
          s, m, float(sigma), _ptr(out), _ptr(ws), nb)
    return out


# ------------------------------------------------------------------ PRDC
class PreparedSet:
    """What every PRDC entry point derives from a set before its tile kernels run - squared row norms, their maximum, the
    largest |element| and the scaled f16 copy (am_prepare_set_f32) - computed once and handed to the *_prepared_* entry
    points.  `rows(lo, hi)` is the prepared form of a row shard (same statistics, offset pointers)."""

    def __init__(self, norms, stats, half, ld_half, source):
        self.norms, self.stats, self.half, self.ld_half = norms, stats, half, ld_half
        self.source = source                      # (data_ptr, n, d) of the matrix it was prepared from

    def matches(self, x):
        return self.source == (x.data_ptr(), x.shape[0], x.shape[1])

    def rows(self, lo, hi):
        return PreparedSet(self.norms[lo:hi], self.stats, self.half[lo * self.ld_half:hi * self.ld_half], self.ld_half,
                           (self.source[0] + 0, hi - lo, self.source[2]))

    def struct(self):
        return _lib.PreparedSetStruct(self.norms.data_ptr(), self.stats.data_ptr(), self.half.data_ptr())


def prepare(x):
    """PreparedSet of a float32 set; None for float64 rows (the f64 kernels derive nothing ahead of their tile loop)."""
    if is_f64(x):
        return None
    lib = _lib.load()
    x = as_matrix(x)
    n, d = x.shape
    ldh = int(lib.am_prepared_half_ld(d))
    norms = torch.empty(n, dtype=torch.float32, device=x.device)
    stats = torch.empty(4, dtype=torch.int32, device=x.device)
    half = torch.empty(n * ldh, dtype=torch.int16, device=x.device)
    _call(lib, "am_prepare_set_f32", x.device, _ptr(x), n, _ld(x), d, _ptr(norms), _ptr(stats), _ptr(half))
    return PreparedSet(norms, stats, half, ldh, (x.data_ptr(), n, d))


def knn_radii(x, k, columns=None, prepared=None):
    """(k+1)-th smallest distance from each row of x to the rows of `columns`
    (default: x itself) - prdc.py:4-14, in the dtype of the rows (float64 rows: float64 radii, am_knn_radii_f64).
    `prepared`: the PreparedSet of x (self-distance form of float32 sets only)."""
    lib = _lib.load()
    if is_f64(x) or is_f64(columns):
        x = as_matrix64(x)
        y = x if columns is None else as_matrix64(columns, "columns")
        n, d = x.shape
        if y.shape[1] != d:
            raise ValueError("feature dimensions differ")
        out = torch.empty(n, dtype=torch.float64, device=x.device)
        nb = lib.am_knn_f64_workspace_bytes(n, y.shape[0], d, int(k))
        ws = _workspace(nb, x.device)
        _call(lib, "am_knn_radii_f64", x.device, _ptr(x), n, _ld64(x), _ptr(y), y.shape[0], _ld64(y), d, int(k), _ptr(out), _ptr(ws), nb)
        return out
    x = as_matrix(x)
    y = x if columns is None else as_matrix(columns, "columns")
    n, d = x.shape
    if y.shape[1] != d:
        raise ValueError("feature dimensions differ")
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    nb = lib.am_knn_workspace_bytes(n, y.shape[0], d, int(k))
    ws = _workspace(nb, x.device)
    if prepared is not None and columns is None:
        ps = prepared.struct()
        _call(lib, "am_knn_radii_prepared_f32", x.device, _ptr(x), n, _ld(x), d, ctypes.byref(ps), int(k), _ptr(out), _ptr(ws), nb)
        return out
    _call(lib, "am_knn_radii_f32", x.device, _ptr(x), n, _ld(x), _ptr(y), y.shape[0], _ld(y), d, int(k), _ptr(out),
                                    _ptr(ws), nb)
    return out


# ---- partitioned symmetric k-NN (multi-GPU, every rank holds the full set) ----
def knn_path(n, m, d, k, self_distance=True):
    """0 exact general kernel, 1 exact symmetric kernel, 2 / 3 f16 filter + exact verification (128 / 256-row engine)."""
    return int(_lib.load().am_knn_path(int(n), int(m), int(d), int(k), 1 if self_distance else 0))


def prdc_path(n_ref, n_cand, d):
    """0 exact kernel, 2 / 3 f16 filter + exact verification (128 / 256-row engine)."""
    return int(_lib.load().am_prdc_path(int(n_ref), int(n_cand), int(d)))


def filter_engine(d):
    """Tile engine of the path-3 filter kernels for rows of d elements: 1 operand-stationary (pstat, D <= 512), 2 the same as two
    256-thread workgroups per CU (pstat64, D <= 128), 0 streamed (wide)."""
    return int(_lib.load().am_filter_engine(int(d)))


def knn_sym_eligible(n, d, k, dtype=None):
    """The partitioned symmetric sweep is a float32 form (float64 sets: row shards through knn_radii)."""
    if dtype == torch.float64:
        return False
    return bool(_lib.load().am_knn_sym_eligible(int(n), int(d), int(k)))


def knn_bounds(x_full, k, row0, nrows, prepared=None):
    """Squared upper bounds for rows [row0, row0+nrows) of the full set (column-sample pre-pass)."""
    lib = _lib.load()
    x = as_matrix(x_full)
    n, d = x.shape
    out = torch.empty(int(nrows), dtype=torch.float32, device=x.device)
    nb = lib.am_knn_part_workspace_bytes(n, d, int(k))
    ws = _workspace(nb, x.device)
    if prepared is not None:
        ps = prepared.struct()
        _call(lib, "am_knn_bounds_prepared_f32", x.device, _ptr(x), n, _ld(x), d, ctypes.byref(ps), int(k), int(row0), int(nrows),
              _ptr(out), _ptr(ws), nb)
        return out
    _call(lib, "am_knn_bounds_f32", x.device, _ptr(x), n, _ld(x), d, int(k), int(row0), int(nrows), _ptr(out), _ptr(ws), nb)
    return out


def knn_sym_part(x_full, k, part, nparts, bounds_sq, prepared=None):
    """This rank's share of the symmetric k-NN: [N, width] smallest entries per row (+inf padded)."""
    lib = _lib.load()
    x = as_matrix(x_full)
    n, d = x.shape
    width = lib.am_knn_list_width(int(k))
    bounds_sq = bounds_sq.to(torch.float32).contiguous().clone()
    out = torch.empty((n, width), dtype=torch.float32, device=x.device)
    nb = lib.am_knn_part_workspace_bytes(n, d, int(k))
    ws = _workspace(nb, x.device)
    if prepared is not None:
        ps = prepared.struct()
        _call(lib, "am_knn_sym_part_prepared_f32", x.device, _ptr(x), n, _ld(x), d, ctypes.byref(ps), int(k), int(part), int(nparts),
              _ptr(bounds_sq), _ptr(out), _ptr(ws), nb)
        return out
    _call(lib, "am_knn_sym_part_f32", x.device, _ptr(x), n, _ld(x), d, int(k), int(part), int(nparts), _ptr(bounds_sq), _ptr(out),
          _ptr(ws), nb)
    return out


def knn_lists_finish(lists, x_full, k):
    """[nparts, N, width] per-rank lists -> radii f32[N] (bit-identical to knn_radii(x_full, k))."""
    lib = _lib.load()
    x = as_matrix(x_full)
    n, d = x.shape
    lists = lists.to(torch.float32).contiguous()
    nparts = lists.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    nb = lib.am_knn_lists_finish_workspace_bytes(n, d, int(k))
    ws = _workspace(nb, x.device)
    _call(lib, "am_knn_lists_finish_f32", x.device, _ptr(lists), nparts, _ptr(x), n, _ld(x), d, int(k), _ptr(out), _ptr(ws), nb)
    return out


def prdc_counts(ref, cand, r_ref, r_cand, want_min=False, prepared_ref=None, prepared_cand=None):
    """col_count i32[Nc], row_any u8[Nr], row_cover u8[Nr] (prdc.py:34-48); with want_min=True also the row
    minimum f32[Nr] (not needed by any metric; costs extra).  float64 rows (either set): distances, radii and the comparisons
    in f64 (am_prdc_counts_f64; the row minimum is then f64 too)."""
    lib = _lib.load()
    _require_cuda(r_ref, "r_ref")
    _require_cuda(r_cand, "r_cand")
    if is_f64(ref) or is_f64(cand):
        ref, cand = as_matrix64(ref, "reference"), as_matrix64(cand, "candidate")
        r_ref, r_cand = r_ref.to(torch.float64).contiguous(), r_cand.to(torch.float64).contiguous()
        nr, d = ref.shape
        nc = cand.shape[0]
        if r_ref.numel() != nr or r_cand.numel() != nc or cand.shape[1] != d:
            raise ValueError("radius / embedding shapes do not match")
        col = torch.empty(nc, dtype=torch.int32, device=ref.device)
        rany = torch.empty(nr, dtype=torch.uint8, device=ref.device)
        rcov = torch.empty(nr, dtype=torch.uint8, device=ref.device)
        rmin = torch.empty(nr, dtype=torch.float64, device=ref.device) if want_min else None
        nb = lib.am_prdc_f64_workspace_bytes(nr, nc, d)
        ws = _workspace(nb, ref.device)
        _call(lib, "am_prdc_counts_f64", ref.device, _ptr(ref), nr, _ld64(ref), _ptr(cand), nc, _ld64(cand), d, _ptr(r_ref), _ptr(r_cand),
              _ptr(col), _ptr(rany), _ptr(rcov), _ptr(rmin) if want_min else ctypes.c_void_p(None), _ptr(ws), nb)
        return (col, rany, rcov, rmin) if want_min else (col, rany, rcov)
    ref, cand = as_matrix(ref, "reference"), as_matrix(cand, "candidate")
    r_ref = r_ref.to(torch.float32).contiguous()
    r_cand = r_cand.to(torch.float32).contiguous()
    nr, d = ref.shape
    nc = cand.shape[0]
    if r_ref.numel() != nr or r_cand.numel() != nc or cand.shape[1] != d:
        raise ValueError("radius / embedding shapes do not match")
    col = torch.empty(nc, dtype=torch.int32, device=ref.device)
    rany = torch.empty(nr, dtype=torch.uint8, device=ref.device)
    rcov = torch.empty(nr, dtype=torch.uint8, device=ref.device)
    rmin = torch.empty(nr, dtype=torch.float32, device=ref.device) if want_min else None
    nb = lib.am_prdc_workspace_bytes(nr, nc, d)
    ws = _workspace(nb, ref.device)
    if prepared_ref is not None and prepared_cand is not None:
        pr, pc = prepared_ref.struct(), prepared_cand.struct()
        _call(lib, "am_prdc_counts_prepared_f32", ref.device, _ptr(ref), nr, _ld(ref), ctypes.byref(pr), _ptr(cand), nc, _ld(cand),
              ctypes.byref(pc), d, _ptr(r_ref), _ptr(r_cand), _ptr(col), _ptr(rany), _ptr(rcov),
              _ptr(rmin) if want_min else ctypes.c_void_p(None), _ptr(ws), nb)
        return (col, rany, rcov, rmin) if want_min else (col, rany, rcov)
    _call(lib, "am_prdc_counts_f32", ref.device, _ptr(ref), nr, _ld(ref), _ptr(cand), nc, _ld(cand), d, _ptr(r_ref),
                                      _ptr(r_cand), _ptr(col), _ptr(rany), _ptr(rcov),
                                      _ptr(rmin) if want_min else ctypes.c_void_p(None), _ptr(ws), nb)
    return (col, rany, rcov, rmin) if want_min else (col, rany, rcov)


def prdc_reduce(col, rany, rcov):
    """Four integer totals as a device int64[4]:
    (#cols with count>0, #rows with any, sum of counts, #rows covered)."""
    lib = _lib.load()
    for t, name in ((col, "col_count"), (rany, "row_any"), (rcov, "row_cover")):
        _require_cuda(t, name)                 # (empty row arrays are allowed: column totals only)
    # the C ABI takes raw pointers: hand it exactly the element types it reads
    col = col.to(torch.int32).contiguous()
    rany = rany.to(torch.uint8).contiguous()
    rcov = rcov.to(torch.uint8).contiguous()
    if rcov.numel() != rany.numel():
        raise ValueError("row_any / row_cover lengths differ")
    out = torch.empty(4, dtype=torch.int64, device=col.device)
    _call(lib, "am_prdc_reduce", col.device, _ptr(col), col.numel(), _ptr(rany), _ptr(rcov), rany.numel(), _ptr(out))
    return out


# ------------------------------------------------------------------ one call = one evaluate()
EVAL_FAD, EVAL_KD, EVAL_PRDC, EVAL_HEAD = 1, 2, 4, 16
_EVAL_WS = {}


def _eval_workspace(nbytes, device):
    """The evaluate workspace is kept per device and grown on demand: its size is a function of the shapes, an evaluate
    runs start to finish on one stream, and re-allocating ~1 GB per call costs the allocator round trips this entry point
    exists to avoid."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _EVAL_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _EVAL_WS.pop(key, None)                                  # (the old block goes back to the allocator first)
        ws = None
        ws = _EVAL_WS[key] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    return ws


def release_workspaces(device=None):
    """Hand the cached evaluate workspaces (1.2 GB per (device, stream) at 2 x 100k x 512, 19.6 GB at 2 x 1M) back to
    torch's caching allocator - for callers that run one large evaluate and then need the memory for something else.
    The next evaluate allocates again.  `device`: only that GPU's."""
    index = None if device is None else torch.device(device).index
    for key in [k for k in _EVAL_WS if index is None or k[0] == index]:
        del _EVAL_WS[key]


def evaluate(ref, cand, what, nearest_k=5, idx_cand=None, idx_ref=None, gamma=None, coef0=1.0, degree=3,
             given_ref=None, given_cand=None):
    """am_evaluate_f32: FAD / KD / PRDC of (candidate vs reference) as ONE stream-ordered chain with ONE read-back.
    `what`: iterable of "fad", "kd", "prdc".  given_*: dict with any of mean, cov, radii (already computed: used as they
    are) and mean_out, cov_out, radii_out (device tensors the chain writes its own results to, so that the caller keeps
    them).  Returns the raw host record: (head f64[16], mmds f64[S] or None)."""
    lib = _lib.load()
    if is_f64(ref) or is_f64(cand):
        raise ValueError("am_evaluate_f32 is the float32 chain; float64 sets run entry point by entry point (evaluate_sharded)")
    ref, cand = as_matrix(ref, "reference"), as_matrix(cand, "candidate")
    dev = _same_device(ref, cand)
    n_ref, d = ref.shape
    n_cand = cand.shape[0]
    if cand.shape[1] != d:
        raise ValueError("feature dimensions differ")
    flags = sum(bit for name, bit in (("fad", EVAL_FAD), ("kd", EVAL_KD), ("prdc", EVAL_PRDC)) if name in what)
    s = m = 0
    if flags & EVAL_KD:
        idx_cand, idx_ref = _index_table(idx_cand, "idx_cand"), _index_table(idx_ref, "idx_ref")
        s, m = idx_cand.shape
    keep = []

    def side(given, rows):
        # the C ABI reads and writes raw pointers: every tensor is checked against the shape the chain assumes for it
        if not given:
            return None
        st = _lib.EvaluateSideStruct()
        for key in ("mean", "cov", "mean_out", "cov_out"):
            t = given.get(key)
            if t is not None:
                want = (d,) if key.startswith("mean") else (d, d)
                if tuple(t.shape) != want:
                    raise ValueError(f"{key} has shape {tuple(t.shape)}, the embeddings need {want}")
                if key.endswith("_out") and (t.dtype != torch.float64 or not t.is_contiguous()):
                    raise ValueError(f"{key} must be a contiguous f64 tensor (the chain writes into it)")
                t = _f64(t, key)
                if t.device != dev:
                    raise ValueError(f"{key} lives on {t.device}, the embeddings on {dev}")
                keep.append(t)
                setattr(st, key, t.data_ptr())
        for key in ("radii", "radii_out"):
            t = given.get(key)
            if t is not None:
                _require_cuda(t, key)
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                    raise ValueError(f"{key} must be a contiguous f32 tensor on {dev}")
                if t.numel() != rows:
                    raise ValueError(f"radius / embedding shapes do not match: {key} holds {t.numel()} values, the set {rows} rows")
                keep.append(t)
                setattr(st, key, t.data_ptr())
        return st

    s_ref, s_cand = side(given_ref, n_ref), side(given_cand, n_cand)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side_stream = _SIDE_STREAMS.get(("fad", dev.index))
        if side_stream is None:
            side_stream = _SIDE_STREAMS[("fad", dev.index)] = torch.cuda.Stream(dev)
        nb = lib.am_evaluate_workspace_bytes(n_ref, n_cand, d, int(nearest_k), s, m, flags)
        ws = _eval_workspace(nb, dev)
        out = torch.empty(EVAL_HEAD + s, dtype=torch.float64, device=dev)
        timer = KernelTimer.active
        if timer is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main)
        status = lib.am_evaluate_f32(_ptr(ref), n_ref, _ld(ref), _ptr(cand), n_cand, _ld(cand), d, flags, int(nearest_k),
                                     _ptr(idx_cand) if s else None, _ptr(idx_ref) if s else None, s, m,
                                     float(1.0 / d if gamma is None else gamma), float(coef0), int(degree),
                                     ctypes.byref(s_ref) if s_ref is not None else None,
                                     ctypes.byref(s_cand) if s_cand is not None else None, _ptr(out), _ptr(ws), nb,
                                     ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(side_stream.cuda_stream))
        if timer is not None:
            b.record(main)
            timer.records.append(("am_evaluate_f32", a, b))
        _lib.check(status, "am_evaluate_f32")
        for t in (ref, cand, *keep):
            t.record_stream(side_stream)
        host = out.cpu()                                             # the one read-back
    head = host[:EVAL_HEAD].tolist()
    return head, (host[EVAL_HEAD:].numpy() if s else None)


def evaluate_sharded_c(ref_local, cand_local, ref_counts, cand_counts, what, coll, nearest_k=5, idx_cand=None, idx_ref=None,
                       gamma=None, coef0=1.0, degree=3, overlap=True):
    """am_evaluate_sharded_f32: this rank's share of a row-sharded evaluate as ONE library call - the exchange schedule of
    distributed.evaluate_sharded behind the C ABI, with the collectives supplied as hooks (`coll`: collectives.TorchCollectives
    here, am_rccl_collectives for a host that is not Python).  ref_counts / cand_counts: rows of every rank.  overlap=False
    issues the collectives on the compute stream.  Returns the host record like evaluate(): (head f64[16], mmds or None)."""
    lib = _lib.load()
    if is_f64(ref_local) or is_f64(cand_local):
        raise ValueError("am_evaluate_sharded_f32 is the float32 schedule")
    rank, world = coll.rank, coll.world
    ref_counts, cand_counts = [int(c) for c in ref_counts], [int(c) for c in cand_counts]
    if len(ref_counts) != world or len(cand_counts) != world:
        raise ValueError("one shard size per rank")
    ref_local, cand_local = as_matrix(ref_local, "reference"), as_matrix(cand_local, "candidate")
    dev = _same_device(ref_local, cand_local)
    d = ref_local.shape[1]
    if cand_local.shape[1] != d or ref_local.shape[0] != ref_counts[rank] or cand_local.shape[0] != cand_counts[rank]:
        raise ValueError("the shards do not match the shard sizes / each other")
    flags = sum(bit for name, bit in (("fad", EVAL_FAD), ("kd", EVAL_KD), ("prdc", EVAL_PRDC)) if name in what)
    s = m = 0
    if flags & EVAL_KD:
        idx_cand, idx_ref = _index_table(idx_cand, "idx_cand"), _index_table(idx_ref, "idx_ref")
        s, m = idx_cand.shape
    rc_arr = (ctypes.c_int64 * world)(*ref_counts)
    cc_arr = (ctypes.c_int64 * world)(*cand_counts)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side_stream = _SIDE_STREAMS.get(("fad", dev.index))
        if side_stream is None:
            side_stream = _SIDE_STREAMS[("fad", dev.index)] = torch.cuda.Stream(dev)
        comm_stream = _SIDE_STREAMS.get(("comm", dev.index))
        if comm_stream is None:
            comm_stream = _SIDE_STREAMS[("comm", dev.index)] = torch.cuda.Stream(dev)
        nb = lib.am_evaluate_sharded_workspace_bytes(rc_arr, cc_arr, rank, world, d, int(nearest_k), s, m, flags)
        if nb == 0:
            raise ValueError(f"empty embedding set: {sum(ref_counts)} reference and {sum(cand_counts)} candidate rows over {world} ranks")
        ws = coll.expose(_eval_workspace(nb, dev))            # (kept per device and stream, like the fused call's)
        out = torch.empty(EVAL_HEAD + s, dtype=torch.float64, device=dev)
        status = lib.am_evaluate_sharded_f32(
            _ptr(ref_local) if ref_counts[rank] else None, _ld(ref_local), _ptr(cand_local) if cand_counts[rank] else None,
            _ld(cand_local), d, rc_arr, cc_arr, coll.byref(), flags, int(nearest_k), _ptr(idx_cand) if s else None,
            _ptr(idx_ref) if s else None, s, m, float(1.0 / d if gamma is None else gamma), float(coef0), int(degree), _ptr(out),
            _ptr(ws), nb, ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(side_stream.cuda_stream),
            ctypes.c_void_p(comm_stream.cuda_stream if overlap else main.cuda_stream))
        if getattr(coll, "error", None) is not None:
            raise coll.error
        _lib.check(status, "am_evaluate_sharded_f32")
        for t in (ref_local, cand_local):
            t.record_stream(side_stream)
            t.record_stream(comm_stream)
        host = out.cpu()                                     # the one read-back: the pack kernel behind it waits for all three streams' work
    head = host[:EVAL_HEAD].tolist()
    return head, (host[EVAL_HEAD:].numpy() if s else None)
