"""N>1 host logic on CPU: world_size-2 gloo processes shard a batch, all-gather embeddings, merge ECE bins."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clip_calibration_amd import metrics, parallel
from oracle import clip_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_global, ragged, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        emb = torch.from_numpy(rng.normal(size=(n_global, 32)).astype(np.float32))
        emb = orc.l2_normalize(emb)
        txt = orc.l2_normalize(torch.from_numpy(rng.normal(size=(11, 32)).astype(np.float32)))
        labels = rng.integers(0, 11, n_global)
        lo, hi = parallel.shard_bounds(n_global, rank, world)
        local = emb[lo:hi].clone()
        full = parallel.all_gather_ragged(local, n_global) if ragged else parallel.all_gather_embeddings(local)
        assert torch.equal(full, emb), "gather must reproduce the unsharded batch bitwise"
        # every rank computes the shared logits on the gathered embeddings -> identical on all ranks
        logits = (100.0 * full) @ txt.t()
        probs = orc.softmax_probs(logits.numpy().astype(np.float64))
        conf, pred = orc.conf_pred(probs)
        # each rank accumulates ECE statistics for ITS rows only, then one all-reduce merges them
        bins = torch.from_numpy(metrics.bin_statistics(conf[lo:hi], pred[lo:hi], labels[lo:hi], 10)).reshape(-1)
        parallel.merge_ece_bins(bins)
        ece = metrics.ece_from_bins(bins.numpy(), 10)
        want = orc.ece(conf, pred, labels, 10)
        assert abs(ece - want) < 1e-12
        np.save(os.path.join(out_dir, f"logits{rank}.npy"), logits.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global,ragged", [(16, False), (17, True), (3, True)])
def test_shard_gather_merge_world2(tmp_path, n_global, ragged):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_global, ragged, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "logits0.npy")
    b = np.load(tmp_path / "logits1.npy")
    assert np.array_equal(a, b)


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 256, 1024, 1001):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def test_single_process_is_identity():
    t = torch.arange(6.0).reshape(3, 2)
    assert parallel.all_gather_embeddings(t) is t
    assert parallel.all_gather_ragged(t, 3) is t


class _HostEvaluator:
    """CPU stand-in with the fields gather_samples touches (the real one accumulates through the HIP kernel)."""
    keep_samples = True

    def __init__(self, conf, pred, gt, n_bins=10):
        self.bins = torch.from_numpy(metrics.bin_statistics(conf, pred, gt, n_bins)).reshape(-1)
        half = len(conf) // 2                           # two "batches", as process() would have appended them
        self._conf = [torch.from_numpy(conf[:half]), torch.from_numpy(conf[half:])]
        self._pred = [torch.from_numpy(pred[:half]), torch.from_numpy(pred[half:])]
        self._gt = [torch.from_numpy(gt[:half]), torch.from_numpy(gt[half:])]


def _sample_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        n = 301                                          # uneven shards: 151 / 150
        conf = rng.uniform(0.1, 1.0, n).astype(np.float32)
        pred = rng.integers(0, 9, n)
        gt = np.where(rng.uniform(size=n) < conf, pred, rng.integers(0, 9, n))
        prox = rng.uniform(0.2, 0.9, n).astype(np.float32)
        lo, hi = parallel.shard_bounds(n, rank, world)
        ev = _HostEvaluator(conf[lo:hi], pred[lo:hi], gt[lo:hi])
        got_prox = parallel.gather_samples(ev, torch.from_numpy(prox[lo:hi]))
        assert np.array_equal(got_prox.numpy(), prox)
        assert np.array_equal(torch.cat(ev._conf).numpy(), conf) and np.array_equal(torch.cat(ev._gt).numpy(), gt)
        assert abs(metrics.ece_from_bins(ev.bins.numpy(), 10) - orc.ece(conf.astype(np.float64), pred, gt, 10)) < 1e-12
        c, p, g = torch.cat(ev._conf).numpy(), torch.cat(ev._pred).numpy(), torch.cat(ev._gt).numpy()
        res = np.array([metrics.AdaptiveECE(c, p, g, 10), metrics.PIECE(c, got_prox.numpy(), p, g, 10, 10), metrics.macro_f1(p, g)])
        want = np.array([orc.ace(conf, pred, gt, 10), orc.piece(conf, prox, pred, gt, 10, 10), orc.macro_f1(pred, gt)])
        assert np.abs(res - want).max() < 1e-6
        np.save(os.path.join(out_dir, f"res{rank}.npy"), res)
    finally:
        dist.destroy_process_group()


def _empty_shard_worker(rank, world, port, out_dir):
    """world 2, every sample on rank 0 (fewer batches than ranks): rank 1 must still enter the collectives."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        rng = np.random.default_rng(9)
        n = 40
        conf = rng.uniform(0.1, 1.0, n).astype(np.float32)
        pred = rng.integers(0, 5, n)
        gt = np.where(rng.uniform(size=n) < conf, pred, rng.integers(0, 5, n))
        prox = rng.uniform(0.2, 0.9, n).astype(np.float32)
        lo, hi = (0, n) if rank == 0 else (n, n)
        ev = _HostEvaluator(conf[lo:hi], pred[lo:hi], gt[lo:hi])
        if rank == 1:
            ev._conf, ev._pred, ev._gt = [], [], []      # what process() leaves behind when it was never called
        got = parallel.gather_samples(ev, torch.from_numpy(prox[lo:hi]) if rank == 0 else None, has_proximity=True)
        assert np.array_equal(got.numpy(), prox)
        assert np.array_equal(torch.cat(ev._conf).numpy(), conf) and np.array_equal(torch.cat(ev._pred).numpy(), pred)
        assert abs(metrics.ece_from_bins(ev.bins.numpy(), 10) - orc.ece(conf.astype(np.float64), pred, gt, 10)) < 1e-12
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


def test_gather_samples_with_an_empty_shard_world2(tmp_path):
    mp.spawn(_empty_shard_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0.npy").exists() and (tmp_path / "ok1.npy").exists()


def _exchange_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = parallel.EmbeddingExchange(torch.device("cpu"), backend="auto")     # no GPU: the torch.distributed backend
        assert ex.backend == "torch" and ex.rccl_ranks is None and ex.world == world and ex.rank == rank
        x = (torch.arange(12, dtype=torch.float32).reshape(3, 4) + 100 * rank).half()   # fp16: the exchange format
        y = ex.all_gather(x)
        assert y.dtype == torch.float16 and y.shape == (3 * world, 4)
        for r in range(world):
            assert torch.equal(y[3 * r: 3 * r + 3], (torch.arange(12, dtype=torch.float32).reshape(3, 4) + 100 * r).half())
        ex.close()
        np.save(os.path.join(out_dir, f"ex{rank}.npy"), y.float().numpy())
    finally:
        dist.destroy_process_group()


def test_embedding_exchange_world2(tmp_path):
    """The exchange object of the N > 1 path (one all-gather of fp16 embeddings per step) on its torch.distributed backend;
    the RCCL backend needs GPUs (tests/test_gpu_multirank.py)."""
    mp.spawn(_exchange_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "ex0.npy"), np.load(tmp_path / "ex1.npy"))


def test_sample_level_metrics_world2(tmp_path):
    mp.spawn(_sample_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "res0.npy"), np.load(tmp_path / "res1.npy"))
