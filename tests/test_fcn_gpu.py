"""-m gpu parity tests: FCNModel (32s/16s/8s) through the C-ABI vs the oracle on identical weights and inputs."""
import numpy as np
import pytest
import torch

from oracle import np_ops as ops
from oracle import fcn as ofcn
from segmentation_amd.datasets import ArrayDataSet
from segmentation_amd.fcn import FCNModel

pytestmark = pytest.mark.gpu


def _data(B, S, nc, seed=5555):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0, 1, (1, B, S, S, 3)).astype(np.float32), rng.integers(0, nc, (1, B, S, S, 1)).astype(np.uint8))


def _model(x, y, nc, S, fcn_type, dtype, nk=16, **kw):
    kw.setdefault('keep_logits', True)          # (the fused training head does not store the logits otherwise)
    return FCNModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=S, fcn_type=fcn_type, n_kernels=nk,
                    learning_rate=1e-3, log_dir=None, save_dir=None, load_snapshot=False, dtype=dtype, autoencoder=False, **kw)


@pytest.mark.parametrize('fcn_type', ['32s', '16s', '8s'])
def test_fcn_f32_parity(fcn_type):
    B, S, nc = 2, 64, 5
    x, y = _data(B, S, nc)
    m = _model(x, y, nc, S, fcn_type, 'f32', use_graph=False)
    p = m.store.get_params()
    for n in p:      # non-zero biases so that the ReLU'd score layers are not all dead
        p[n]['biases'] = (np.random.default_rng(3).standard_normal(p[n]['biases'].shape) * 0.05 + 0.05).astype(np.float32)
    m.set_weights(p)
    m._load_batch(m.dataset, m.input_x, m.input_y)
    m.store.g.fill_(float('nan'))
    m._run_fwd_bwd()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.store.g).all())
    loss_ref, g_ref, c = ofcn.loss_and_grads(p, x[0], y[0], fcn_type)
    logits = m.acts['logits'].t[..., :nc].cpu().numpy()
    assert logits.shape == (B, S, S, nc)
    assert np.abs(logits - c['logits']).max() < 1e-4
    assert abs(m.last_loss() - loss_ref) < 1e-5
    g = m.store.get_grads()
    for n in g_ref:
        for k in ('weights', 'biases'):
            ref = np.asarray(g_ref[n][k])
            err = np.abs(g[n][k] - ref).max() / (np.abs(ref).max() + 1e-20)
            assert err < 3e-4, (n, k, err)
    sig, out = m.infer(x[0])
    sref, oref = ofcn.infer(p, x[0], fcn_type)
    assert sig.shape == (B, S, S, nc) and out.shape == (B, S, S, 1)
    assert np.abs(sig - sref).max() < 1e-5
    srt = np.sort(sref.astype(np.float64), -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(out[..., 0][decided], oref[..., 0][decided])


def test_fcn8s_bf16_trains_and_matches_loosely():
    B, S, nc = 2, 96, 21
    x, y = _data(B, S, nc, seed=9)
    m = _model(x, y, nc, S, '8s', 'bf16', nk=32, use_graph=True)
    p = m.store.get_params()
    loss_ref, g_ref, c = ofcn.loss_and_grads(p, x[0], y[0], '8s')
    l0 = None
    for i in range(6):
        m.train_step()
        if i == 0:
            l0 = m.last_loss()
    assert abs(l0 - loss_ref) < 3e-2 * abs(loss_ref)
    assert m.last_loss() < l0
    assert m.global_step == 6


def test_fcn_odd_input_size_crop_or_pad():
    # 70 -> pools 35,17,8,4,2: the x32 up-sampled map (64) is smaller than the input -> zero-padded, floor offsets
    B, S, nc = 1, 70, 3
    x, y = _data(B, S, nc, seed=2)
    m = _model(x, y, nc, S, '32s', 'f32', use_graph=False)
    p = m.store.get_params()
    for n in p:
        p[n]['biases'] = np.full(p[n]['biases'].shape, 0.05, np.float32)
    m.set_weights(p)
    m._load_batch(m.dataset, m.input_x, m.input_y)
    m._run_fwd_bwd()
    torch.cuda.synchronize()
    loss_ref, g_ref, c = ofcn.loss_and_grads(p, x[0], y[0], '32s')
    logits = m.acts['logits'].t[..., :nc].cpu().numpy()
    assert np.abs(logits - c['logits']).max() < 1e-4
    assert (logits[:, :3] == 0).all() and (logits[:, -3:] == 0).all()
    g = m.store.get_grads()
    ref = g_ref['conv_fr']['weights']
    assert np.abs(g['conv_fr']['weights'] - ref).max() / (np.abs(ref).max() + 1e-20) < 3e-4


def test_fcn_fused_head_equals_separate_launches(monkeypatch):
    """the fused up-sampling + x-entropy head (default) against bilinear_fwd + softmax_xent (SEG_FUSE_HEAD=0): same loss and
    gradients; without keep_logits the float logits are never stored (y_hat is None)"""
    B, S, nc = 2, 96, 5
    x, y = _data(B, S, nc)
    res = []
    for fuse in ('1', '0'):
        monkeypatch.setenv('SEG_FUSE_HEAD', fuse)
        m = _model(x, y, nc, S, '8s', 'f32', use_graph=False, keep_logits=False)
        assert (m.y_hat is None) == (fuse == '1')
        names = [o[0] for o in m.fwd_plan.ops]
        assert ('up8+xent' in names) == (fuse == '1') and ('xent' in names) == (fuse == '0')
        m._load_batch(m.dataset, m.input_x, m.input_y)
        m._run_fwd_bwd(); torch.cuda.synchronize()
        res.append((m.last_loss(), m.store.g.clone()))
    assert abs(res[0][0] - res[1][0]) < 1e-6
    assert float((res[0][1] - res[1][1]).abs().max()) < 2e-6 * float(res[1][1].abs().max())
