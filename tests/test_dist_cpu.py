"""CPU tests of the data-parallel path (gloo, world_size 2): bucketed SUM all-reduce over the flat gradient arena +
1/world scaling reproduces the single-process gradients of the global batch (U-Net has no BatchNorm)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import unet as ounet
from segmentation_amd.dist import DataParallel, shard_batch


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _flat(g, order):
    return np.concatenate([np.concatenate([np.asarray(g[n]['weights']).ravel(), np.asarray(g[n]['biases']).ravel()]) for n in order])


def _worker(rank, world, port, x, y, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        p = ounet.init_params(2, 2, seed=11)
        lo, hi = shard_batch(x.shape[0], world, rank)
        loss, g, _ = ounet.loss_and_grads(p, x[lo:hi], y[lo:hi])
        order = list(reversed(ounet.CONV_ORDER))               # backward-production order, as the arena
        flat = torch.from_numpy(_flat(g, order).copy())
        dp = DataParallel(None, overlap=True)
        assert dp.world == world and dp.rank == rank and dp.enabled
        cut = flat.numel() // 3
        dp.all_reduce_bucket(flat, 0, cut)                      # bucket 0 while "backward continues"
        dp.all_reduce_bucket(flat, cut, flat.numel())
        dp.wait_all()
        flat /= world
        losses = torch.tensor([loss]); dist.all_reduce(losses); 
        if rank == 0:
            out['flat'] = flat.numpy().copy(); out['loss'] = float(losses.item() / world)
        dp.broadcast_(flat, src=0)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_global_batch():
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (2, 188, 188, 3)).astype(np.float32)
    y = rng.integers(0, 2, (2, 188, 188, 1)).astype(np.uint8)
    port = _free_port()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(2, port, x, y, out), nprocs=2, join=True)
    p = ounet.init_params(2, 2, seed=11)
    loss, g, _ = ounet.loss_and_grads(p, x, y)
    ref = _flat(g, list(reversed(ounet.CONV_ORDER)))
    assert abs(out['loss'] - loss) < 1e-12
    assert np.allclose(out['flat'], ref, rtol=1e-10, atol=1e-14)


def test_shard_batch():
    assert shard_batch(128, 8, 3) == (48, 64)
    with pytest.raises(ValueError):
        shard_batch(10, 4, 0)


def test_single_process_is_disabled():
    dp = DataParallel(None)
    assert dp.world == 1 and dp.rank == 0 and not dp.enabled
    t = torch.ones(4)
    dp.all_reduce_bucket(t, 0, 4); dp.wait_all()
    assert (t == 1).all()
