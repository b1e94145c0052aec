"""-m gpu: the full data-parallel train step with world_size 2 (two processes sharing the one GPU of the test box, gloo
backend on device tensors -- RCCL itself needs one GPU per rank): segmented backward, bucketed all-reduce, 1/world Adam
scaling.  Two ranks with one image each must reproduce the single-process step on the 2-image batch."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, x, y, out, backend='gloo', cuts=None, use_graph=True, algo=None):
    import torch.distributed as dist
    from segmentation_amd.datasets import ArrayDataSet
    from segmentation_amd.dist import shard_batch
    from segmentation_amd.unet import UNetModel
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if algo:
        os.environ['SEG_DP_ALGO'] = algo
    dev = rank if backend == 'nccl' else 0                       # RCCL needs one GPU per rank; gloo shares the single test GPU
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_batch(x.shape[1], world, rank)
        m = UNetModel(sess=None, dataset=ArrayDataSet(x[:, lo:hi], y[:, lo:hi]), n_classes=2, input_dims=188, learning_rate=1e-3,
                      log_dir=None, save_dir=None, load_snapshot=False, dtype='f32', use_graph=use_graph, dp_cuts=cuts)
        assert m.pg.world == world and m.pg.enabled and m.pg.algo == (algo or 'allreduce')
        if cuts:
            assert len(m.bwd_segments) == len(cuts.split(',')) + 1
        m.pg.broadcast_(m.store.p); m._repack()
        m.train_step()
        torch.cuda.synchronize()
        if rank == 0:
            out['g1'] = m.store.g.cpu().numpy() / world          # the arena holds the SUM over ranks; Adam scales by 1/world
        for _ in range(2):
            m.train_step()
        torch.cuda.synchronize()
        if rank == 0:
            out['p'] = m.store.p.cpu().numpy()
        rep = m.dp_exposure_report(steps=2)                      # (collective: every rank takes the instrumented steps)
        if rank == 0:
            out['rep'] = dict(rep)
    finally:
        dist.destroy_process_group()


def _run_two_ranks(backend, cuts=None):
    return _run_ranks(backend, cuts, 2)


def _run_ranks(backend, cuts=None, world=2, use_graph=True, algo=None):
    import torch.multiprocessing as mp
    from segmentation_amd.datasets import ArrayDataSet
    from segmentation_amd.unet import UNetModel
    rng = np.random.default_rng(7)
    x = rng.uniform(0, 1, (2, world, 188, 188, 3)).astype(np.float32)
    y = rng.integers(0, 2, (2, world, 188, 188, 1)).astype(np.uint8)
    ref = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=None,
                    save_dir=None, load_snapshot=False, dtype='f32', use_graph=False)
    ref.train_step()
    torch.cuda.synchronize()
    gref = ref.store.g.cpu().numpy()
    for _ in range(2):
        ref.train_step()
    torch.cuda.synchronize()
    pref = ref.store.p.cpu().numpy()
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), x, y, out, backend, cuts, use_graph, algo), nprocs=world, join=True)
    # same math up to summation order (mean over 2 images vs mean of two per-image means)
    for name, l in ref.store.layers.items():
        for lo, n in ((l.w_off, l.wsize), (l.b_off, l.cout)):
            a, b = out['g1'][lo:lo + n], gref[lo:lo + n]
            assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-12, name
    # after three Adam steps: identical except where a cancellation-dominated gradient flips sign under round-off
    # (Adam then moves that weight by a full +-lr); such elements must stay a vanishing fraction
    d = np.abs(out['p'] - pref)
    assert np.median(d) < 1e-6 and (d > 1e-4).mean() < 1e-3 and d.max() < 3.5e-3
    rep = out['rep']
    assert rep['world'] == world and len(rep['buckets_mb']) == len(rep['exposed_us']) and abs(sum(rep['buckets_mb']) - 31.04) < 0.05
    return rep


def test_two_ranks_equal_single_process_global_batch():
    """default bucket plan (4 buckets) over gloo on the one test GPU"""
    rep = _run_two_ranks('gloo')
    assert len(rep['buckets_mb']) == 4


def test_two_ranks_six_bucket_plan():
    rep = _run_two_ranks('gloo', 'conv6_2,upconv1,conv5_2,conv5_1,conv3_1')
    assert len(rep['buckets_mb']) == 6 and max(rep['buckets_mb']) < 9.5


def test_four_ranks_eager_rs_ag_equal_the_global_batch():
    """Config C4's control flow as far as a one-GPU box allows it (at most 6 processes may use the card: this one + 4 ranks): four
    ranks with ONE image each, the eager one-plan data-parallel step bench.py times (bucket markers, collectives issued from a side
    stream), SEG_DP_ALGO=rs_ag (reduce-scatter emulated over gloo, tail all-reduce) == the single-process step on the batch of 4.
    The 8-rank form of the same exchange runs on the CPU (tests/test_dist_cpu.py)."""
    rep = _run_ranks('gloo', None, world=4, use_graph=False, algo='rs_ag')
    assert len(rep['buckets_mb']) == 4


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank: runs only where >= 2 GPUs are visible')
def test_two_ranks_nccl_over_xgmi():
    """the same equality through the real backend: 2 ranks, 2 GPUs, RCCL all-reduce of the gradient buckets"""
    _run_two_ranks('nccl')
