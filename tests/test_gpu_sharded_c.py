"""am_evaluate_sharded_f32: one rank's share of a row-sharded evaluate as ONE C call (VERDICT r4 "missing" item 4) - the exchange
schedule of audio-metrics_amd/distributed.py behind the C ABI, the collectives handed in as hooks.

  * one rank, hooks over torch (no-ops) and hooks over a world-1 RCCL communicator (libaudio_metrics_rccl.so: the adapter a
    host that is not Python links) against the fused one-GPU call am_evaluate_f32;
  * two and three ranks sharing cuda:0 over gloo (the 1-GPU box cannot host several RCCL ranks), hooks over
    torch.distributed, against the Python schedule and the one-rank result: shapes that take the general k-NN kernel on row
    shards, shapes that take the partitioned symmetric sweep, unequal shards, a rank without rows, kd-only / prdc-only."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import inputs as gi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shards(n, world, uneven):
    if uneven == "empty_rank":                              # the last rank holds nothing
        cut = [n * r // (world - 1) for r in range(world)] + [n]
        return [(cut[r], cut[r + 1]) for r in range(world - 1)] + [(n, n)]
    if uneven:
        edges = [0] + [min(n, n * (r + 1) // world + (37 if r % 2 == 0 else -21)) for r in range(world - 1)] + [n]
        return [(edges[r], edges[r + 1]) for r in range(world)]
    return [(n * r // world, n * (r + 1) // world) for r in range(world)]


def _worker(rank, world, port, n_ref, n_cand, d, k, metrics, uneven, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audio_metrics_amd.distributed import evaluate_sharded
    ref, cand = gi.pair("randn", 95, n_ref, n_cand, d)
    (rl, rh), (cl, ch) = _shards(n_ref, world, uneven)[rank], _shards(n_cand, world, uneven)[rank]
    dev = torch.device("cuda:0")
    rloc, cloc = torch.as_tensor(ref[rl:rh]).to(dev), torch.as_tensor(cand[cl:ch]).to(dev)
    kw = dict(metrics=metrics, nearest_k=k, kid_subsets=8, kid_subset_size=300)
    c_res = evaluate_sharded(rloc, cloc, c_entry=True, **kw)
    py_res = evaluate_sharded(rloc, cloc, **kw)
    out_q.put((rank, c_res, py_res))
    dist.barrier()
    dist.destroy_process_group()


def _close(a, b, key):
    if key in ("precision", "recall", "density", "coverage"):
        return a == b                                        # integer counts: identical whatever the sharding and the form
    if key == "fad":
        return abs(a - b) <= 1e-5 * abs(b)                   # (covariance chains break at shard boundaries: test_gpu_distributed.py)
    return abs(a - b) <= max(1e-9 * abs(b), 1e-12)


@pytest.mark.parametrize("world,n_ref,n_cand,d,k,metrics,uneven", [
    (2, 2600, 2400, 96, 4, ("fad", "kd", "prdc"), False),             # general k-NN kernel on row shards
    (2, 2501, 2333, 96, 4, ("fad", "kd", "prdc"), True),
    (3, 2501, 2333, 67, 5, ("fad", "kd", "prdc"), "empty_rank"),      # a rank without rows takes part in every collective
    (2, 8300, 8200, 128, 4, ("fad", "kd", "prdc"), True),             # the partitioned symmetric sweep
    (3, 9100, 8300, 128, 5, ("prdc",), False),
    (2, 2600, 2400, 96, 4, ("kd",), False),
    (2, 2600, 2400, 96, 4, ("fad",), True)])
def test_c_entry_over_gloo_equals_the_python_schedule(world, n_ref, n_cand, d, k, metrics, uneven):
    from audio_metrics_amd.distributed import evaluate_sharded
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_ref, n_cand, d, k, metrics, uneven, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c_res = {rank: c for rank, c, _ in got}
    py_res = {rank: p for rank, _, p in got}
    for r in range(1, world):
        assert c_res[r] == c_res[0], "every rank holds the same record"
    ref, cand = gi.pair("randn", 95, n_ref, n_cand, d)
    dev = torch.device("cuda:0")
    single = evaluate_sharded(torch.as_tensor(ref).to(dev), torch.as_tensor(cand).to(dev), metrics=metrics, nearest_k=k,
                              kid_subsets=8, kid_subset_size=300)
    assert set(c_res[0]) == set(single) == set(py_res[0])
    for key in single:
        assert _close(c_res[0][key], py_res[0][key], key), (key, c_res[0][key], py_res[0][key])
        assert _close(c_res[0][key], single[key], key), (key, c_res[0][key], single[key])


def _one_rank_inputs(n_ref, n_cand, d):
    ref, cand = gi.pair("randn", 96, n_ref, n_cand, d)
    dev = torch.device("cuda:0")
    return torch.as_tensor(ref).to(dev), torch.as_tensor(cand).to(dev)


@pytest.mark.parametrize("n_ref,n_cand,d,k,overlap", [(3000, 2800, 64, 5, True), (9000, 8800, 128, 5, True), (3000, 2800, 67, 3, False)])
def test_c_entry_one_rank_equals_the_fused_call(n_ref, n_cand, d, k, overlap):
    from audio_metrics_amd import hip_ops as ops
    from audio_metrics_amd.collectives import TorchCollectives
    from audio_metrics_amd.metrics.kd import device_subset_indices
    ref, cand = _one_rank_inputs(n_ref, n_cand, d)
    i1, i2 = device_subset_indices(n_cand, n_ref, 8, 300, 1234, ref.device)
    want_head, want_mmds = ops.evaluate(ref, cand, ("fad", "kd", "prdc"), k, i1, i2)
    coll = TorchCollectives(None)
    head, mmds = ops.evaluate_sharded_c(ref, cand, [n_ref], [n_cand], ("fad", "kd", "prdc"), coll, k, i1, i2, overlap=overlap)
    assert head[5:9] == want_head[5:9]                                          # the four PRDC totals
    assert np.array_equal(mmds, want_mmds)
    assert abs(head[0] - want_head[0]) <= 1e-9 * abs(want_head[0]) and head[4] == want_head[4]
    names = [name for name, _ in coll.calls]
    assert names.count("all_reduce_sum") == 4 and names.count("all_gather_v") == 2, coll.calls     # sums, scatters, counts, kd; two row gathers
    assert names[0] == "all_reduce_sum" and coll.calls[0][1] == 2 * d * 8                          # ... column sums first


def test_c_entry_over_a_world_one_rccl_communicator():
    """The adapter a host that is not Python links (csrc/rccl/am_rccl.cpp): an ncclComm_t of one rank, its two hooks handed to
    am_evaluate_sharded_f32 through am_rccl_collectives - RCCL really runs the all-reduces and all-gathers."""
    from audio_metrics_amd import _build, _lib, hip_ops as ops
    from audio_metrics_amd.metrics.kd import device_subset_indices
    path = _build.RCCL_ADAPTER_PATH
    if not os.path.exists(path):
        pytest.skip("libaudio_metrics_rccl.so was not built (no RCCL headers in this image)")
    adapter = ctypes.CDLL(path)
    adapter.am_rccl_comm_init_single.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    adapter.am_rccl_collectives.argtypes = [ctypes.c_void_p, ctypes.POINTER(_lib.CollectivesStruct)]
    adapter.am_rccl_comm_destroy.argtypes = [ctypes.c_void_p]
    n_ref, n_cand, d, k = 9000, 8800, 128, 5
    ref, cand = _one_rank_inputs(n_ref, n_cand, d)
    i1, i2 = device_subset_indices(n_cand, n_ref, 8, 300, 1234, ref.device)
    want_head, want_mmds = ops.evaluate(ref, cand, ("fad", "kd", "prdc"), k, i1, i2)
    comm = ctypes.c_void_p()
    assert adapter.am_rccl_comm_init_single(ctypes.byref(comm)) == 0
    try:
        class RcclHooks:                                    # what hip_ops.evaluate_sharded_c needs of a collectives object
            rank, world, error = 0, 1, None
            struct = _lib.CollectivesStruct()

            def expose(self, t):
                return t

            def byref(self):
                return ctypes.byref(self.struct)
        hooks = RcclHooks()
        assert adapter.am_rccl_collectives(comm, ctypes.byref(hooks.struct)) == 0
        assert hooks.struct.rank == 0 and hooks.struct.world == 1
        head, mmds = ops.evaluate_sharded_c(ref, cand, [n_ref], [n_cand], ("fad", "kd", "prdc"), hooks, k, i1, i2)
        torch.cuda.synchronize()
    finally:
        adapter.am_rccl_comm_destroy(comm)
    assert head[5:9] == want_head[5:9]
    assert np.array_equal(mmds, want_mmds)
    assert abs(head[0] - want_head[0]) <= 1e-9 * abs(want_head[0])


def test_c_entry_errors():
    from audio_metrics_amd import _lib, hip_ops as ops
    from audio_metrics_amd.collectives import TorchCollectives
    ref, cand = _one_rank_inputs(600, 500, 32)
    with pytest.raises(ValueError):
        ops.evaluate_sharded_c(ref, cand, [600, 1], [500], ("prdc",), TorchCollectives(None), 3)        # one size per rank
    with pytest.raises(ValueError):
        ops.evaluate_sharded_c(ref.double(), cand.double(), [600], [500], ("prdc",), TorchCollectives(None), 3)
    with pytest.raises(_lib.HipLibraryError):
        ops.evaluate_sharded_c(ref, cand, [600], [500], ("prdc",), TorchCollectives(None), 700)         # k + 1 > rows, as torch.kthvalue
