"""N2: the embedder pool / device-side aggregation and the one-process-per-GPU form of ``AudioMetrics``.

* several embedder replicas (one worker thread each, batches dealt round-robin, per-replica device-side aggregation,
  Chan merge of the partial statistics) must give the single-replica results - exercised on one GPU by listing the same
  device twice (the reference spreads replicas over GPUs the same way: util/gpu_parallel.py:20-118);
* ``AudioMetrics(process_group=...)``: every rank feeds its shard of the audio, the metrics are reduced across ranks by
  distributed.py - two ranks sharing cuda:0 over gloo (RCCL needs one GPU per rank);
* an RCCL run of the same thing, skipped when the box has fewer than two GPUs (so that the first 8-GPU run is not also
  the first RCCL run)."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import inputs as gi

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(am, metrics, **kw):
    c = gi.E2E
    # one embedder replica unless a test asks otherwise: the goldens were produced by ONE process, and the kernel-distance
    # subsets are drawn by row index - on a box with several GPUs the default (None = every visible GPU) stores another order
    kw.setdefault("device_indices", [0])
    return am.AudioMetrics(metrics=metrics, embedder=gi.NumpyEmbedder(c["dim"], c["sr"]), mix_function=gi.e2e_mix,
                           win_dur=c["win_dur"], **kw)


def _audio():
    c = gi.E2E
    return (gi.e2e_pairs(c["seed"], c["n_ref"], c["seconds"], c["sr"]),
            gi.e2e_pairs(c["seed"] + 1, c["n_cand"], c["seconds"], c["sr"], stem_gain=1.3))


def test_free_gpu_dealing_like_the_reference():
    """replica_dealing="free": every replica pulls from ONE shared queue - a batch goes to whichever GPU is free, the
    reference's scheme (util/gpu_parallel.py:20-76, submit :59-76) - so a slow replica does not stall the others: with one of
    two replicas slowed down it ends up with fewer batches, every row still arrives exactly once, and the row-order-free
    metrics equal the single-replica run's."""
    import time
    import audio_metrics_amd as am
    ref, cand = _audio()
    random.seed(5)
    one = _make(am, ["fad", "prdc"], device_indices=[0])
    one.add_reference(ref)
    want = one.evaluate(cand)
    random.seed(5)
    two = _make(am, ["fad", "prdc"], device_indices=[0, 0], replica_dealing="free")
    slow = two._pool.replicas[1]
    fast_forward = type(slow).forward

    class Slowed(type(slow)):                                  # replica 1 takes 100 ms longer per batch (six batches in all: [5, 1] or [4, 2])
        def forward(self, batch):
            time.sleep(0.1)
            return fast_forward(self, batch)

    import copy
    two._pool.replicas[1] = copy.copy(slow)
    two._pool.replicas[1].__class__ = Slowed
    two.add_reference(ref)
    taken = list(two._pool.batches_per_replica)
    got = two.evaluate(cand)
    assert sum(taken) > 2 and taken[1] < taken[0], taken         # the free replica took more of the work
    assert two.stem_reference.n == one.stem_reference.n
    assert torch.equal(torch.sort(two.stem_reference.embeddings.sum(1))[0], torch.sort(one.stem_reference.embeddings.sum(1))[0])
    for key in ("precision", "recall", "density", "coverage"):
        assert got[key] == want[key], key
    assert abs(got["fad"] - want["fad"]) <= 1e-6 * abs(want["fad"])          # (other merge order of the partial statistics)
    with pytest.raises(ValueError):
        _make(am, ["fad"], replica_dealing="whoever")


def test_two_replicas_match_one():
    import audio_metrics_amd as am
    ref, cand = _audio()
    results = []
    for devices in ([0], [0, 0]):
        random.seed(5)
        m = _make(am, ["fad", "kd", "prdc", "apa"], device_indices=devices)
        m.add_reference(ref)
        results.append((m.evaluate(cand), m.stem_reference.n, m.stem_reference.embeddings.clone()))
    (one, n1, rows1), (two, n2, rows2) = results
    assert n1 == n2 and one.keys() == two.keys()
    # the same rows reach the statistics, in a different (per-replica) order: row-order-free metrics agree closely, KD
    # draws subsets by row index and therefore differs like any two index draws do
    assert torch.equal(torch.sort(rows1.sum(1))[0], torch.sort(rows2.sum(1))[0])
    for key in ("fad", "apa"):
        assert abs(one[key] - two[key]) <= 1e-5 * max(1.0, abs(one[key])), (key, one[key], two[key])
    for key in ("precision", "recall", "density", "coverage"):
        assert one[key] == two[key], key


def test_embedder_exception_reaches_the_caller():
    import audio_metrics_amd as am

    class Broken(gi.NumpyEmbedder):
        def forward(self, data, sr=None):
            raise RuntimeError("embedder failed")

    c = gi.E2E
    m = am.AudioMetrics(metrics=["fad"], embedder=Broken(c["dim"], c["sr"]), mix_function=gi.e2e_mix, win_dur=c["win_dur"],
                        device_indices=[0, 0])
    with pytest.raises(RuntimeError, match="embedder failed"):
        m.add_reference([x[:, 1] for x in _audio()[0]])


def _rank_worker(rank, world, port, backend, out_q, metrics=("fad", "kd", "prdc"), n_pca=None, contiguous=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    device = rank if backend == "nccl" else 0
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import audio_metrics_amd as am
    ref, cand = _audio()
    if "apa" not in metrics:
        ref, cand = [x[:, 1] for x in ref], [x[:, 1] for x in cand]      # stems only: no shuffled pairing across ranks
    random.seed(11 + rank)
    # device_indices=None under a process group = this rank's device only
    m = _make(am, list(metrics), device_indices=None if backend == "gloo" else [device], process_group=dist.group.WORLD,
              n_pca=n_pca)
    assert m._devices == [torch.device("cuda", device)]

    def shard(items):
        if contiguous:                                   # rank order = row order of the one-process run (KD draws by row index)
            return items[len(items) * rank // world:len(items) * (rank + 1) // world]
        return items[rank::world]

    m.add_reference(shard(ref))
    res = m.evaluate(shard(cand))
    again = m.evaluate(shard(cand))                      # cached projected reference sets, radii, d_x_xp
    out_q.put((rank, res, again))
    dist.barrier()
    dist.destroy_process_group()


def _launch(backend, **kw):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, backend, q), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    results, seconds = {}, {}
    for _ in range(world):
        rank, res, again = q.get(timeout=300)
        results[rank], seconds[rank] = res, again
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0] == results[1]                       # every rank reports the same values
    assert seconds[0] == seconds[1]
    return results[0], seconds[0]


def _run_ranks(backend):
    import audio_metrics_amd as am
    got, _ = _launch(backend)
    ref, cand = _audio()
    ref, cand = [x[:, 1] for x in ref], [x[:, 1] for x in cand]
    m = _make(am, ["fad", "kd", "prdc"])
    m.add_reference(ref)
    single = m.evaluate(cand)
    assert got.keys() == single.keys()
    for key in ("precision", "recall", "density", "coverage"):
        assert got[key] == single[key], key         # integer counts do not depend on who holds which rows
    assert abs(got["fad"] - single["fad"]) <= 1e-5 * abs(single["fad"])


def test_process_group_audio_metrics_two_ranks_gloo():
    _run_ranks("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_process_group_audio_metrics_two_ranks_rccl():
    _run_ranks("nccl")


def _check_pca_against_golden(golden, got, keys):
    g = golden("e2e")
    for key in keys:
        want = float(g[f"pca/{key}"])
        slack = max(1e-4 * abs(want), 1.0 / 150) if key != "fad" else 1e-4 * abs(want)
        assert abs(got[key] - want) <= slack, (key, got[key], want)


def _pca_two_ranks(backend, golden):
    """n_pca under a process group (reference audio_metrics.py:163-209, projection.py:6-46): the fit sees the union of the
    ranks' rows through two all-reduces and runs replicated.  Order-free values (FAD, PRDC) must equal the reference's own
    `pca` golden - produced by ONE process - and everything must equal this build's one-process run on the same rows."""
    import audio_metrics_amd as am
    got, again = _launch(backend, metrics=("fad", "kd", "prdc"), n_pca=8, contiguous=True)
    _check_pca_against_golden(golden, got, ("fad", "precision", "recall", "density", "coverage"))
    ref, cand = _audio()
    m = _make(am, ["fad", "kd", "prdc"], n_pca=8)
    m.add_reference([x[:, 1] for x in ref])
    single = m.evaluate([x[:, 1] for x in cand])
    assert list(got) == list(single)
    for key, want in single.items():
        slack = 1.0 / 150 if key in ("precision", "recall", "density", "coverage") else max(1e-5 * abs(want), 5e-7)
        assert abs(got[key] - want) <= slack, (key, got[key], want)
        assert abs(again[key] - got[key]) <= 1e-9 * max(1.0, abs(got[key])), key
    # with APA: both projections fitted across ranks; the misaligned pairs are drawn per rank, so APA itself is only sane
    got, again = _launch(backend, metrics=("fad", "apa"), n_pca=8, contiguous=True)
    _check_pca_against_golden(golden, got, ("fad",))
    assert 0.0 <= got["apa"] <= 1.0 and abs(again["apa"] - got["apa"]) <= 1e-9


def test_process_group_with_pca_two_ranks_gloo(golden):
    _pca_two_ranks("gloo", golden)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_process_group_with_pca_two_ranks_rccl(golden):
    _pca_two_ranks("nccl", golden)


def test_device_indices_none_means_every_visible_gpu():
    """The reference's default (util/gpu_parallel.py:24-25, audio_metrics.py:276-279): None = all GPUs, an empty list = no
    replica handler (the embedder runs where it lives)."""
    import audio_metrics_amd as am
    from audio_metrics_amd.audio_metrics import _visible_devices
    count = torch.cuda.device_count()
    current = torch.cuda.current_device()
    devices = _visible_devices(None)
    assert len(devices) == count and devices[0] == torch.device("cuda", current)
    assert sorted(d.index for d in devices) == list(range(count))
    assert _visible_devices([]) == [torch.device("cuda", current)]
    assert _visible_devices(None, one_process_per_gpu=True) == [torch.device("cuda", current)]
    ref, cand = _audio()
    results = []
    for indices in (None, [current]) if count == 1 else (None, list(range(count))):
        random.seed(5)
        m = _make(am, ["fad", "kd", "prdc", "apa"], device_indices=indices)
        assert len(m._pool.devices) == count
        m.add_reference(ref)
        results.append(m.evaluate(cand))
    assert results[0] == results[1]
