"""-m gpu tests: 1-GPU RCCL data-parallel path (segmented backward + bucket all-reduce), MC-dropout inference,
dropout kernel statistics, snapshot/restore."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from segmentation_amd import _lib as L
from segmentation_amd import engine as E
from segmentation_amd.datasets import ArrayDataSet, SyntheticDataSet
from segmentation_amd.unet import UNetModel

pytestmark = pytest.mark.gpu


def _data(B, S, nc, seed=5555, n=1):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0, 1, (n, B, S, S, 3)).astype(np.float32), rng.integers(0, nc, (n, B, S, S, 1)).astype(np.uint8))


def test_dp_world1_rccl_path_equals_plain_step():
    x, y = _data(2, 188, 2, n=2)
    kw = dict(sess=None, n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None, load_snapshot=False, dtype='f32')
    plain = UNetModel(dataset=ArrayDataSet(x, y), use_graph=True, **kw)
    for _ in range(4):
        plain.train_step()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        dp = UNetModel(dataset=ArrayDataSet(x, y), use_graph=True, **kw)
        assert dp.pg.enabled and dp.pg.world == 1 and len(dp.bwd_segments) == 2
        bounds = [b for _, b in dp.bwd_segments]
        assert bounds[0][0] == 0 and bounds[0][1] == bounds[1][0] and bounds[1][1] == dp.store.n
        assert (bounds[1][1] - bounds[1][0]) * 4 < 300e3            # last (exposed) bucket: conv2_x + conv1_x only
        for _ in range(4):
            dp.train_step()
        torch.cuda.synchronize()
        # filter gradients are reduced in a fixed order -> the two paths agree bit for bit
        assert torch.equal(plain.store.p, dp.store.p)
        assert plain.last_loss() == dp.last_loss()
    finally:
        torch.distributed.destroy_process_group()


def test_dropout_kernel_statistics_and_determinism():
    lib = L.load()
    net = E.Net(None, 2, L.SEG_F32, torch.device('cuda', 0))
    a = net.act(64, 64, 32); a.t.fill_(1.0)
    b = net.act(64, 64, 32); c = net.act(64, 64, 32)
    av, bv, cv = a.view(), b.view(), c.view()
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.seg_dropout(C.byref(av), C.byref(bv), 2, 64, 64, 32, 0.5, 7, 0, L.SEG_F32, s))
    L.check(lib.seg_dropout(C.byref(av), C.byref(cv), 2, 64, 64, 32, 0.5, 7, 0, L.SEG_F32, s))
    torch.cuda.synchronize()
    assert torch.equal(b.t, c.t)                                  # same (seed, offset) -> same mask
    vals = torch.unique(b.t).cpu().numpy().tolist()
    assert vals == [0.0, 2.0]
    assert abs(float((b.t > 0).float().mean()) - 0.5) < 0.01
    L.check(lib.seg_dropout(C.byref(av), C.byref(cv), 2, 64, 64, 32, 0.5, 7, 1 << 40, L.SEG_F32, s))
    torch.cuda.synchronize()
    assert not torch.equal(b.t, c.t)


def test_mc_dropout_inference():
    x, y = _data(2, 188, 4, seed=3)
    m = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=4, input_dims=188, log_dir=None, save_dir=None,
                  load_snapshot=False, dtype='bf16', bayesian=True)
    mean, var, amax = m.infer_mc(x[0], passes=8)
    assert mean.shape == (2, 4, 4, 4) and var.shape == mean.shape and amax.shape == (2, 4, 4, 1)
    assert np.isfinite(mean).all() and (var >= 0).all() and var.max() > 0       # stochastic passes differ
    sig, out = m.infer(x[0])                                                     # `bayesian` is ignored by infer(), as in the reference
    sig2, out2 = m.infer(x[0])
    assert np.array_equal(sig, sig2)


def test_snapshot_restore_roundtrip(tmp_path):
    x, y = _data(1, 188, 2, seed=4)
    kw = dict(sess=None, n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=str(tmp_path / 'logs'), dtype='f32', use_graph=False)
    m = UNetModel(dataset=ArrayDataSet(x, y), save_dir=str(tmp_path / 'snap'), load_snapshot=False, **kw)
    m.train_step(); m.train_step(); m.snapshot(); m.test()
    assert m.last_test_loss is not None
    m2 = UNetModel(dataset=ArrayDataSet(x, y), save_dir=str(tmp_path / 'snap'), load_snapshot=True, **kw)
    assert m2.global_step == 2 and int(m2.store.step[0].item()) == 2
    assert torch.equal(m.store.p, m2.store.p) and torch.equal(m.store.m, m2.store.m)
    m.train_step(); m2.train_step()
    torch.cuda.synchronize()
    assert torch.equal(m.store.p, m2.store.p)                    # resume continues the same trajectory
    names = set(np.load(m2._latest_checkpoint()).files)
    assert {'conv1_1/weights', 'conv1_1/biases', 'output/weights', 'global_step'} <= names


def test_device_prefetcher_feeds_identical_batches():
    """host dataset -> pinned ring -> async H2D on a copy stream -> train_step: same trajectory as the direct feed"""
    from segmentation_amd.datasets import DevicePrefetcher
    x, y = _data(2, 188, 2, seed=8, n=3)
    kw = dict(sess=None, n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None, load_snapshot=False, dtype='f32')
    a = UNetModel(dataset=ArrayDataSet(x, y), **kw)
    pf = DevicePrefetcher(ArrayDataSet(x, y), depth=2)
    b = UNetModel(dataset=pf, **kw)
    try:
        for _ in range(7):
            a.train_step(); b.train_step()
        torch.cuda.synchronize()
        assert torch.equal(a.store.p, b.store.p)
    finally:
        pf.stop()


def test_zero_copy_device_batches_equal_copied_ones(monkeypatch):
    """Device-resident batches consumed in place (launch arguments re-pointed, one graph per buffer) vs copied into the
    model's own input buffers: identical parameters after a few steps, in graph and in eager mode."""
    import numpy as np
    from segmentation_amd.datasets import SyntheticDataSet
    from segmentation_amd.unet import UNetModel
    out = {}
    for zc in ('1', '0'):
        for graph in (True, False):
            monkeypatch.setenv('SEG_ZERO_COPY', zc)
            ds = SyntheticDataSet(2, 188, 3, seed=11, n_batches=3)
            m = UNetModel(sess=None, dataset=ds, n_classes=3, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None,
                          load_snapshot=False, dtype='bf16', use_graph=graph, seed=3)
            for _ in range(7):
                m.train_step()
            torch.cuda.synchronize()
            out[(zc, graph)] = (m.store.p.clone(), m.last_loss())
            if zc == '1':
                assert len(m._slots) == 3          # three dataset buffers were bound in place
    ref = out[('0', False)]
    for k, v in out.items():
        assert torch.equal(v[0], ref[0]) and v[1] == ref[1], k
