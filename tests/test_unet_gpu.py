"""-m gpu parity tests, model level: UNetModel.train_step()/infer() through the C-ABI vs the oracle on
identical weights and inputs.  f32 mode: logits within 1e-4 (north_star), argmax bit-exact on the same logits,
gradients/updated weights to fp32 round-off.  bf16 mode: stated looser bounds."""
import numpy as np
import pytest
import torch

from oracle import np_ops as ops
from oracle import unet as ounet
from segmentation_amd import _lib as L
from segmentation_amd.datasets import ArrayDataSet, SyntheticDataSet
from segmentation_amd.unet import UNetModel

pytestmark = pytest.mark.gpu


def _data(B, S, nc, seed=5555, n=1):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (n, B, S, S, 3)).astype(np.float32)
    y = rng.integers(0, nc, (n, B, S, S, 1)).astype(np.uint8)
    return x, y


def _model(B, S, nc, dtype, x, y, lr=1e-4, **kw):
    ds = ArrayDataSet(x, y)
    return UNetModel(sess=None, dataset=ds, n_classes=nc, input_dims=S, learning_rate=lr, log_dir=None, save_dir=None,
                     load_snapshot=False, dtype=dtype, **kw)


def _grad_check(m, g_ref, rtol):
    g = m.store.get_grads()
    worst = 0.0
    for n in g_ref:
        for k in ('weights', 'biases'):
            ref = np.asarray(g_ref[n][k]); got = g[n][k]
            err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-20)
            worst = max(worst, err)
            assert err < rtol, (n, k, err)
    return worst


@pytest.mark.parametrize('crop_aware', [True, False])
def test_unet_f32_forward_backward_parity(crop_aware):
    B, S, nc = 2, 188, 2
    x, y = _data(B, S, nc)
    m = _model(B, S, nc, 'f32', x, y, use_graph=False, crop_aware=crop_aware)
    p = m.store.get_params()
    assert sum(v['weights'].size + v['biases'].size for v in p.values()) == 7760130
    m._load_batch(m.dataset, m.input_x, m.input_y)
    m.store.g.fill_(float('nan'))          # every gradient entry must be (over)written by the backward plan
    m._run_fwd_bwd()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.store.g).all())
    loss_ref, g_ref, c = ounet.loss_and_grads(p, x[0], y[0])
    logits = m.acts['logits'].t[..., :nc].cpu().numpy()
    assert logits.shape == (B, 4, 4, nc)
    assert np.abs(logits - c['logits']).max() < 1e-4                      # north_star: fp32 logits within 1e-4
    assert abs(m.last_loss() - loss_ref) < 1e-5
    for name in ('conv1_1', 'conv2_2', 'conv5_2', 'upconv1', 'conv6_1', 'conv9_2'):
        a = m.acts[name]
        assert np.abs(a.t[..., :a.C].cpu().numpy() - c[name]).max() < 1e-4, name
    _grad_check(m, g_ref, 2e-4)


def test_unet_f32_train_steps_match_oracle_adam():
    B, S, nc = 1, 188, 3
    x, y = _data(B, S, nc, seed=7, n=2)
    m = _model(B, S, nc, 'f32', x, y, lr=1e-3, use_graph=False)
    p = m.store.get_params()
    mm, vv = ounet.init_opt_state(p)
    losses = []
    for t in (1, 2):
        m.train_step()
        losses.append(m.last_loss())
    ref_losses = []
    pr = p
    for t in (1, 2):
        l, pr, mm, vv = ounet.train_step(pr, mm, vv, t, x[t - 1], y[t - 1], lr=1e-3)
        ref_losses.append(l)
    assert m.global_step == 2 and int(m.store.step[0].item()) == 2
    assert np.allclose(losses, ref_losses, atol=2e-5)
    got = m.store.get_params()
    for n in pr:
        for k in ('weights', 'biases'):
            # Adam normalises the step to ~lr, so tiny gradients amplify round-off: compare at 5% of one step
            assert np.abs(got[n][k] - pr[n][k]).max() < 1e-4, (n, k, np.abs(got[n][k] - pr[n][k]).max())


def test_unet_graph_replay_equals_eager():
    B, S, nc = 2, 188, 4
    x, y = _data(B, S, nc, seed=3)
    m1 = _model(B, S, nc, 'f32', x, y, lr=1e-3, use_graph=False)
    m2 = _model(B, S, nc, 'f32', x, y, lr=1e-3, use_graph=True)
    for _ in range(4):          # eager warm-up, capture, then two replays
        m1.train_step(); m2.train_step()
    torch.cuda.synchronize()
    assert m2.global_step == 4 and int(m2.store.step[0].item()) == 4
    assert abs(m1.last_loss() - m2.last_loss()) < 1e-4
    # slab reductions have a fixed summation order (no atomics): eager (batched reductions) and graph replay
    # (per-layer reductions) produce the same bits
    assert torch.equal(m1.store.p, m2.store.p)


def test_unet_infer_matches_oracle_and_argmax_rule():
    B, S, nc = 2, 188, 4
    x, y = _data(B, S, nc, seed=11)
    m = _model(B, S, nc, 'f32', x, y, use_graph=False)
    p = m.store.get_params()
    sig, out = m.infer(x[0])
    assert sig.shape == (B, 4, 4, nc) and out.shape == (B, 4, 4, 1) and out.dtype == np.float32
    sref, oref = ounet.infer(p, x[0])
    assert np.abs(sig - sref).max() < 1e-5
    # the argmax op itself is bit-exact on identical logits ...
    plan, x_in, _, _ = m._infer_cache[tuple(x[0].shape)]
    logits = plan.acts['logits'].t[..., :nc].cpu().numpy()
    s2, o2 = ops.sigmoid_argmax(logits)
    assert np.array_equal(sig, s2) and np.array_equal(out, o2)
    # ... and end to end equal wherever the oracle's top-2 sigmoid margin exceeds the 1e-4 logit tolerance
    srt = np.sort(sref.astype(np.float64), -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert np.array_equal(out[..., 0][decided], oref[..., 0][decided])
    assert ops.miou(out, oref, nc) > 0.99
    # INFERENCE-mode model (F8: no dataset) shares nothing but the weights
    mi = UNetModel(sess=None, mode='INFERENCE', n_classes=nc, input_dims=S, save_dir=None, load_snapshot=False, dtype='f32')
    mi.set_weights(p)
    s3, o3 = mi.infer(x[0][:1])
    assert np.abs(s3 - sig[:1]).max() < 1e-6 and np.array_equal(o3, out[:1])


def test_unet_bf16_parity_bounds():
    B, S, nc = 2, 188, 4
    x, y = _data(B, S, nc, seed=5)
    m = _model(B, S, nc, 'bf16', x, y, use_graph=False)
    p = m.store.get_params()
    m._load_batch(m.dataset, m.input_x, m.input_y)
    m._run_fwd_bwd()
    torch.cuda.synchronize()
    loss_ref, g_ref, c = ounet.loss_and_grads(p, x[0], y[0])
    logits = m.acts['logits'].t[..., :nc].cpu().numpy()
    scale = np.abs(c['logits']).max()
    assert np.abs(logits - c['logits']).max() < 5e-2 * scale + 1e-3       # bf16 storage through 23 layers
    assert abs(m.last_loss() - loss_ref) < 2e-2 * abs(loss_ref)
    g = m.store.get_grads()
    for n in g_ref:
        a, b = g[n]['weights'].ravel(), np.asarray(g_ref[n]['weights']).ravel()
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        assert cos > 0.97, (n, cos)        # 23 bf16 layers at random init (first layer included: bf16 MFMA operands)


def test_unet_256_shapes_and_loss_decreases():
    B, S, nc = 2, 256, 4
    ds = SyntheticDataSet(B, S, nc)
    m = UNetModel(sess=None, dataset=ds, n_classes=nc, input_dims=S, learning_rate=1e-3, save_dir=None, load_snapshot=False,
                  dtype='bf16', use_graph=True)
    assert m.out_hw == (68, 68) and m.label_off == (94, 94)
    l0 = None
    for i in range(12):
        m.train_step()
        if i == 0:
            l0 = m.last_loss()
    l1 = m.last_loss()
    assert np.isfinite(l0) and np.isfinite(l1)
    assert abs(l0 - np.log(nc)) < 0.2            # random init -> ~uniform predictions
    assert l1 < l0                               # memorising one batch
    with pytest.raises(Exception):
        UNetModel(sess=None, dataset=SyntheticDataSet(1, 128, 2), n_classes=2, input_dims=128, save_dir=None)


@pytest.mark.parametrize('B,H,W,cin,nk,nc', [(1, 188, 188, 2, 32, 2), (3, 188, 188, 1, 16, 21), (2, 204, 204, 3, 8, 5)])
def test_unet_ragged_shapes_f32(B, H, W, cin, nk, nc):
    """1/2-channel images, channel counts that are not multiples of 32, odd batch sizes, 21 classes"""
    rng = np.random.default_rng(H + W + nk)
    x = rng.uniform(0, 1, (1, B, H, W, cin)).astype(np.float32)
    y = rng.integers(0, nc, (1, B, H, W, 1)).astype(np.uint8)
    m = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=[H, W], input_channel=cin, n_kernels=nk,
                  learning_rate=1e-4, log_dir=None, save_dir=None, load_snapshot=False, dtype='f32', use_graph=False)
    p = m.store.get_params()
    for n in p:
        p[n]['biases'] = (rng.standard_normal(p[n]['biases'].shape) * 0.05).astype(np.float32)
    m.set_weights(p)
    m._load_batch(m.dataset, m.input_x, m.input_y)
    m.store.g.fill_(float('nan'))
    m._run_fwd_bwd()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.store.g).all())
    import oracle.unet as ou
    logits_ref, c = ou.forward(p, x[0])
    oh, ow = logits_ref.shape[1:3]
    assert m.out_hw == (oh, ow)
    yc = y[0][:, (H - oh) // 2:(H - oh) // 2 + oh, (W - ow) // 2:(W - ow) // 2 + ow]
    loss_ref, _, _ = ops.softmax_xent(logits_ref, yc)
    logits = m.acts['logits'].t[..., :nc].cpu().numpy()
    assert np.abs(logits - logits_ref).max() < 1e-4
    assert abs(m.last_loss() - loss_ref) < 1e-5
    # gradients: the oracle's backward crops labels with a square target, so compare against torch autograd instead
    from oracle import torch_ref
    tp = torch_ref.to_torch_params(p, torch.float64)
    lg = torch_ref.unet_forward(tp, torch.as_tensor(x[0], dtype=torch.float64))
    loss_t = torch_ref.xent_mean(lg, torch.as_tensor(yc[..., 0].astype(np.int64)))
    loss_t.backward()
    g = m.store.get_grads()
    for n in tp:
        for k in ('weights', 'biases'):
            ref = tp[n][k].grad.numpy()
            err = np.abs(g[n][k] - ref).max() / (np.abs(ref).max() + 1e-20)
            assert err < 3e-4, (n, k, err)


def test_unet_rejects_non_square_like_the_reference():
    with pytest.raises(Exception):
        UNetModel(sess=None, dataset=SyntheticDataSet(1, 188, 2), n_classes=2, input_dims=[188, 220], save_dir=None, load_snapshot=False)


def test_limits_are_refused_at_the_constructor():
    """n_classes > 32 and input_channel > 3 are limits of the device path: the model constructors say so, no plan is built"""
    x = np.zeros((1, 2, 188, 188, 3), np.float32); y = np.zeros((1, 2, 188, 188, 1), np.uint8)
    from segmentation_amd.fcn import FCNModel
    for cls in (UNetModel, FCNModel):
        with pytest.raises(Exception, match='n_classes must be 1..32'):
            cls(sess=None, dataset=ArrayDataSet(x, y), n_classes=33, input_dims=188, log_dir=None, save_dir=None, load_snapshot=False)
        with pytest.raises(Exception, match='input_channel must be 1..3'):
            cls(sess=None, dataset=ArrayDataSet(x, y), n_classes=2, input_dims=188, input_channel=4, log_dir=None, save_dir=None, load_snapshot=False)


@pytest.mark.parametrize('nc,nk,dtype', [(4, 32, 'f32'), (6, 32, 'f32'), (3, 64, 'f32'), (4, 32, 'bf16')])
def test_unet_head_forms_agree(monkeypatch, nc, nk, dtype):
    """the three forms of the segmentation head -- one launch incl. the output layer's filter gradient (default), one launch
    with the filter gradient left to seg_conv2d_wgrad (SEG_FUSE_HEAD_DW=0), separate conv / x-entropy / dgrad launches
    (SEG_FUSE_HEAD=0) -- give the same loss and the same gradients of every layer"""
    B, S = 2, 188
    x, y = _data(B, S, nc)
    res = []
    for fuse, dw in (('1', '1'), ('1', '0'), ('0', '0')):
        monkeypatch.setenv('SEG_FUSE_HEAD', fuse)
        monkeypatch.setenv('SEG_FUSE_HEAD_DW', dw)
        m = _model(B, S, nc, dtype, x, y, use_graph=False, n_kernels=nk)
        names = [o[0] for o in m.fwd_plan.ops]
        assert ('output+xent+dx+dw' in names) == (fuse == '1' and dw == '1')
        assert ('output+xent+dx' in names) == (fuse == '1' and dw == '0')
        m._load_batch(m.dataset, m.input_x, m.input_y)
        m._run_fwd_bwd(); torch.cuda.synchronize()
        res.append((m.last_loss(), m.store.get_grads()))
    tol = 2e-5 if dtype == 'f32' else 2e-2
    for loss, g in res[1:]:
        assert abs(loss - res[0][0]) < (1e-6 if dtype == 'f32' else 1e-3)
        for n in g:
            for k in ('weights', 'biases'):
                ref = g[n][k]
                err = np.abs(res[0][1][n][k] - ref).max() / (np.abs(ref).max() + 1e-20)
                assert err < tol, (n, k, err)
