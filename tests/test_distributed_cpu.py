"""world_size-2 gloo test of the row-sharded evaluate (sharding, all-gathers of
unequal shards, all-reduces), with the oracle standing in for the HIP kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import inputs as gi
import oracle


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_ref, n_cand, d, k, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_ops
    from audio_metrics_amd.distributed import evaluate_sharded, shard_bounds
    ref, cand = gi.pair("shifted", 91, n_ref, n_cand, d)
    rl, rh = shard_bounds(n_ref, world, rank)
    cl, ch = shard_bounds(n_cand, world, rank)
    res = evaluate_sharded(torch.as_tensor(ref[rl:rh]), torch.as_tensor(cand[cl:ch]), nearest_k=k, ops=cpu_ops,
                           kid_subsets=6, kid_subset_size=200)
    # the same with the shard sizes handed in (no count exchange): identical result
    counts = ([shard_bounds(n_ref, world, r)[1] - shard_bounds(n_ref, world, r)[0] for r in range(world)],
              [shard_bounds(n_cand, world, r)[1] - shard_bounds(n_cand, world, r)[0] for r in range(world)])
    again = evaluate_sharded(torch.as_tensor(ref[rl:rh]), torch.as_tensor(cand[cl:ch]), nearest_k=k, ops=cpu_ops,
                             kid_subsets=6, kid_subset_size=200, shard_counts=counts)
    assert again == res, (again, res)
    out_q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_ref,n_cand", [(700, 700), (701, 655)])
def test_sharded_evaluate_matches_single_process(n_ref, n_cand):
    d, k, world = 24, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_ref, n_cand, d, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, cand = gi.pair("shifted", 91, n_ref, n_cand, d)
    a = oracle.OracleData(True).add(torch.as_tensor(cand))
    b = oracle.OracleData(True).add(torch.as_tensor(ref))
    want = {"fad": oracle.frechet_distance(a, b)}
    want.update(oracle.kid_from_features(cand, ref, subsets=6, subset_size=200))
    want.update(oracle.prdc(b, a, k))
    assert results[0] == results[1]                     # every rank reports the same values
    for key, w in want.items():
        assert abs(results[0][key] - w) <= max(2e-5 * abs(w), 5e-7), (key, results[0][key], w)


def test_shard_bounds_cover_rows():
    from audio_metrics_amd.distributed import shard_bounds
    for n in (1, 7, 100000, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def _empty_rank_worker(rank, world, port, n_ref, n_cand, d, k, out_q):
    """Rank 1 was fed no audio: its sets never received a row (embeddings is None).  The front end's local_rows() must
    hand evaluate_sharded an empty [0, D] matrix there instead of raising on that rank alone (which left the others
    waiting in the first all-gather)."""
    from types import SimpleNamespace
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_ops
    from audio_metrics_amd.distributed import evaluate_sharded, local_rows
    ref, cand = gi.pair("shifted", 92, n_ref, n_cand, d)
    cpu = torch.device("cpu")
    if rank == 0:
        sets = [SimpleNamespace(embeddings=torch.as_tensor(x), store_embeddings=True, device=cpu) for x in (ref, cand)]
    else:
        sets = [SimpleNamespace(embeddings=None, store_embeddings=True, device=cpu) for _ in range(2)]
    rows = [local_rows(s, None if world == 1 else dist.group.WORLD) for s in sets]
    assert rows[0].shape == ((n_ref, d) if rank == 0 else (0, d)) and rows[0].dtype == torch.float32
    res = evaluate_sharded(rows[0], rows[1], nearest_k=k, ops=cpu_ops, kid_subsets=6, kid_subset_size=200)
    # a set that keeps no embeddings is an error on EVERY rank (same exception, no hang)
    keeps_none = SimpleNamespace(embeddings=None, store_embeddings=rank == 0, device=cpu)
    try:
        local_rows(keeps_none, dist.group.WORLD)
        raised = False
    except ValueError:
        raised = True
    out_q.put((rank, res, raised))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_without_rows_takes_part_in_the_evaluation():
    n_ref, n_cand, d, k, world = 500, 450, 16, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, world, port, n_ref, n_cand, d, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results = {rank: res for rank, res, _ in got}
    assert all(raised for _, _, raised in got)
    ref, cand = gi.pair("shifted", 92, n_ref, n_cand, d)
    a = oracle.OracleData(True).add(torch.as_tensor(cand))
    b = oracle.OracleData(True).add(torch.as_tensor(ref))
    want = {"fad": oracle.frechet_distance(a, b)}
    want.update(oracle.kid_from_features(cand, ref, subsets=6, subset_size=200))
    want.update(oracle.prdc(b, a, k))
    assert results[0] == results[1]
    for key, w in want.items():
        assert abs(results[0][key] - w) <= max(2e-5 * abs(w), 5e-7), (key, results[0][key], w)


def _global_stats_worker(rank, world, port, n, d, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_ops
    from audio_metrics_amd.distributed import global_stats
    x = torch.as_tensor(gi.pair("shifted", 93, n, n, d)[0])
    # rank 0 holds everything but three rows, rank 1 three rows; then rank 1 holds nothing at all
    for cut in (n - 3, n):
        local = x[:cut] if rank == 0 else x[cut:]
        count, mean, cov = global_stats(local, dist.group.WORLD, ops=cpu_ops)
        out_q.put((rank, cut, count, mean.numpy(), cov.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_global_stats_of_one_sharded_set():
    """The statistics the row-sharded PCA fit consumes (n_pca under a process group): count + column sums in one all-reduce,
    centred scatters in a second; a rank without rows contributes zeros."""
    n, d, world = 300, 12, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_global_stats_worker, args=(r, world, port, n, d, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=90) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = torch.as_tensor(gi.pair("shifted", 93, n, n, d)[0]).double()
    for rank, cut, count, mean, cov in got:
        assert count == n
        np.testing.assert_allclose(mean, x.mean(0).numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(cov, torch.cov(x.T).numpy(), rtol=1e-10, atol=1e-13)


def _hooks_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import ctypes
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audio_metrics_amd.collectives import COLL_F64, COLL_I32, TorchCollectives
    coll = TorchCollectives(dist.group.WORLD)
    ws = coll.expose(torch.zeros(4096, dtype=torch.uint8))
    base = ws.data_ptr()
    # all_reduce_sum on an f64 and an i32 window of the exposed tensor, through the C function pointers of the struct
    ws[256:256 + 40].view(torch.float64)[:] = torch.arange(5, dtype=torch.float64) + rank
    ws[512:512 + 12].view(torch.int32)[:] = torch.tensor([1, 2, 3], dtype=torch.int32) * (rank + 1)
    assert coll.struct.all_reduce_sum(None, base + 256, 5, COLL_F64, None) == 0
    assert coll.struct.all_reduce_sum(None, base + 512, 3, COLL_I32, None) == 0
    # all_gather_v, in place, unequal shares (rank r contributes 7 + 5 r bytes; the last rank nothing)
    sizes = [7 + 5 * r for r in range(world - 1)] + [0]
    offs = [sum(sizes[:r]) for r in range(world)]
    ws[1024 + offs[rank]:1024 + offs[rank] + sizes[rank]] = rank + 10
    arr = (ctypes.c_int64 * world)(*sizes)
    assert coll.struct.all_gather_v(None, base + 1024 + offs[rank], base + 1024, arr, None) == 0
    forms = [coll.gather_form]
    # ... and equal shares: ONE all_gather_into_tensor into the caller's buffer (in place on RCCL; gloo takes the own share from a copy)
    ws[2048 + 16 * rank:2048 + 16 * (rank + 1)] = rank + 50
    arr_eq = (ctypes.c_int64 * world)(*([16] * world))
    assert coll.struct.all_gather_v(None, base + 2048 + 16 * rank, base + 2048, arr_eq, None) == 0
    forms.append(coll.gather_form)
    # a pointer outside every exposed tensor is an error code, not an exception through the C frames
    assert coll.struct.all_reduce_sum(None, base + 4096, 4, COLL_F64, None) == 1 and coll.error is not None
    out_q.put((rank, ws[256:296].view(torch.float64).tolist(), ws[512:524].view(torch.int32).tolist(),
               ws[1024:1024 + sum(sizes)].tolist(), [name for name, _ in coll.calls], ws[2048:2048 + 16 * world].tolist(), forms))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_hooks_over_gloo():
    """collectives.TorchCollectives - the am_collectives hooks am_evaluate_sharded_f32 is handed in this package - called
    through their C function pointers on host tensors, three ranks over gloo (the GPU tests run the whole C entry point)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hooks_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [7 + 5 * r for r in range(world - 1)] + [0]
    for rank, f64, i32, gathered, names, gathered_eq, forms in got:
        assert f64 == [float(3 * i + 0 + 1 + 2) for i in range(5)]
        assert i32 == [6, 12, 18]
        assert gathered == [r + 10 for r in range(world) for _ in range(sizes[r])]
        assert names == ["all_reduce_sum", "all_reduce_sum", "all_gather_v", "all_gather_v", "all_reduce_sum"]
        assert gathered_eq == [r + 50 for r in range(world) for _ in range(16)]
        assert forms == ["broadcast per rank, in place", "all_gather_into_tensor, own share copied"]
