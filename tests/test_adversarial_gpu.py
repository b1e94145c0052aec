"""-m gpu parity tests for adversarial segmentation training (/root/reference/models/basemodel.py:215-355):
the new C-ABI ops against numpy, the adversary (forward of the real / fake batch, its gradients, the gradient it sends into the
segmentation logits, moving averages) against oracle/adversary.py, and UNetModel / FCNModel with adversarial_training=True
against the torch-autograd composition of segmentation loss + adversarial term."""
import ctypes as C
import numpy as np
import pytest
import torch

import gpu_util as U
from oracle import adversary as oadv
from oracle import np_ops as ops
from oracle import torch_ref as T
from segmentation_amd import _lib as L
from segmentation_amd import engine as E
from segmentation_amd.adversary import Adversary, ladder
from segmentation_amd.datasets import ArrayDataSet
from segmentation_amd.fcn import FCNModel
from segmentation_amd.unet import UNetModel

pytestmark = pytest.mark.gpu
DT = [L.SEG_F32, L.SEG_BF16]


def _net(B, dtype):
    store = E.ParamStore([E.Layer('d', 'bn', 1, [8], 8)], dtype, U.dev(), training=True)
    return E.Net(store, B, dtype, U.dev())


@pytest.mark.parametrize('dtype', DT)
def test_onehot_softmax_and_its_backward(dtype):
    B, H, W, nc = 2, 9, 11, 5
    rng = np.random.default_rng(3)
    net = _net(B, dtype); lib = net.lib
    LH, LW, off = H + 4, W + 3, (2, 1)
    y = rng.integers(0, nc, (B, LH, LW, 1)).astype(np.uint8)
    yd = torch.from_numpy(y).to(U.dev())
    oh = net.act(H, W, nc); ov = oh.view()
    oh.t.fill_(7.0)
    L.check(lib.seg_onehot(yd.data_ptr(), LH, LW, off[0], off[1], B, H, W, C.byref(ov), dtype, U.stream()), 'onehot')
    ref = oadv.one_hot(y[:, off[0]:off[0] + H, off[1]:off[1] + W], nc)
    assert np.array_equal(U.read_act(oh), ref) and U.pad_channels_zero(oh)

    z = (rng.standard_normal((B, H, W, nc)) * 3).astype(np.float32)
    lg = E.Act(B, H, W, nc, dtype, U.dev(), f32=True); U.fill_act(lg, z)
    pr = net.act(H, W, nc); lv, pv = lg.view(), pr.view()
    L.check(lib.seg_softmax_probs(C.byref(lv), B, H, W, nc, C.byref(pv), dtype, U.stream()), 'softmax')
    pref = oadv.softmax(z)
    assert np.abs(U.read_act(pr) - pref).max() < (1e-6 if dtype == L.SEG_F32 else 4e-3) and U.pad_channels_zero(pr)

    dp = U.round_dtype(rng.standard_normal((B, H, W, nc)), dtype)
    d0 = U.round_dtype(rng.standard_normal((B, H, W, nc)) * 0.1, dtype)
    dpa, dl = net.act(H, W, nc), net.act(H, W, nc); U.fill_act(dpa, dp); U.fill_act(dl, d0)
    gv, dv = dpa.view(), dl.view()
    L.check(lib.seg_softmax_bwd_add(C.byref(lv), C.byref(gv), B, H, W, nc, 2.0, C.byref(dv), dtype, U.stream()), 'softmax_bwd')
    want = d0 + 2.0 * oadv.softmax_bwd(pref, dp)
    assert np.abs(U.read_act(dl) - want).max() < (1e-5 if dtype == L.SEG_F32 else 2e-2) and U.pad_channels_zero(dl)


@pytest.mark.parametrize('dtype', DT)
def test_flatten_and_row_batch_norm(dtype):
    B, H, W, Cc = 6, 3, 2, 72
    rng = np.random.default_rng(5)
    net = _net(B, dtype); lib = net.lib
    F = H * W * Cc
    a = net.act(H, W, Cc); f = net.act(1, 1, F)
    av = U.round_dtype(rng.standard_normal((B, H, W, Cc)), dtype); U.fill_act(a, av)
    f.t.fill_(9.0)
    avw, fv = a.view(), f.view()
    L.check(lib.seg_flatten(C.byref(avw), B, H, W, Cc, C.byref(fv), 0, dtype, U.stream()), 'flatten')
    assert np.array_equal(U.read_act(f)[:, 0, 0, :], av.reshape(B, -1)) and U.pad_channels_zero(f)
    back = net.act(H, W, Cc); back.t.fill_(5.0); bv = back.view()
    L.check(lib.seg_flatten(C.byref(bv), B, H, W, Cc, C.byref(fv), 1, dtype, U.stream()), 'flatten bwd')
    assert np.array_equal(U.read_act(back), av) and U.pad_channels_zero(back)

    # batch norm over the rows: statistics per feature over the B rows; moving averages; gradient with / without the ReLU gate
    Fp = f.Cp
    beta = (rng.standard_normal(F) * 0.2).astype(np.float32)
    bd = torch.zeros(Fp, device=U.dev()); bd[:F] = torch.from_numpy(beta).to(U.dev())
    mov = torch.zeros(2 * Fp, device=U.dev()); mov[Fp:] = 1.0
    stats = torch.zeros(2 * Fp, device=U.dev())
    y = net.act(1, 1, F); yv = y.view()
    L.check(lib.seg_bn_rows_fwd(C.byref(fv), C.byref(yv), bd.data_ptr(), mov.data_ptr(), stats.data_ptr(), B, F, 0.999, 1e-3, dtype, U.stream()), 'bn_rows')
    x = av.reshape(B, 1, 1, F)
    yr, cache, nm, nv = ops.batch_norm(x, beta, np.zeros(F), np.ones(F), True)
    tol = 1e-5 if dtype == L.SEG_F32 else 3e-2
    assert np.abs(U.read_act(y) - yr).max() < tol and U.pad_channels_zero(y)
    m = mov.cpu().numpy()
    assert np.abs(m[:F] - nm).max() < 1e-6 and np.abs(m[Fp:Fp + F] - nv).max() < 1e-6
    dyv = U.round_dtype(rng.standard_normal((B, 1, 1, F)), dtype)
    dy, dz = net.act(1, 1, F), net.act(1, 1, F); U.fill_act(dy, dyv)
    gvw, zv = dy.view(), dz.view()
    dbeta = torch.full((Fp,), 1.0, device=U.dev())
    for mask in (0, 1):
        for add in (0, 1):
            dbeta.fill_(1.0)
            L.check(lib.seg_bn_rows_bwd(C.byref(fv), C.byref(gvw), C.byref(zv), stats.data_ptr(), dbeta.data_ptr(), add, B, F, mask, dtype, U.stream()), 'bn_rows_bwd')
            dxr, dbr = ops.batch_norm_bwd(dyv, cache)
            if mask:
                dxr = dxr * (x > 0)
            assert np.abs(U.read_act(dz) - dxr).max() < tol * 4, (mask, add)
            assert np.abs(dbeta.cpu().numpy()[:F] - (dbr + add)).max() < 1e-4 * max(1, np.abs(dbr).max())


@pytest.mark.parametrize('dtype', DT)
def test_bce2(dtype):
    B = 7
    rng = np.random.default_rng(9)
    net = _net(B, dtype); lib = net.lib
    zv = U.round_dtype(rng.standard_normal((B, 1, 1, 2)) * 2, dtype)
    lg, dl = net.act(1, 1, 2), net.act(1, 1, 2); U.fill_act(lg, zv)
    loss = torch.zeros(4, device=U.dev())
    lv, dv = lg.view(), dl.view()
    for label in (0, 1):
        L.check(lib.seg_bce2(C.byref(lv), B, label, 1.0, loss.data_ptr() + 4 * label, C.byref(dv), dtype, U.stream()), 'bce2')
        l, d = oadv.bce(zv[:, 0, 0, :], label)
        assert abs(float(loss[label]) - l.mean()) < 1e-5
        assert np.abs(U.read_act(dl)[:, 0, 0, :] - d).max() < (1e-6 if dtype == L.SEG_F32 else 2e-3) and U.pad_channels_zero(dl)


def _rand_adv_params(nc, h, w, rng):
    p = oadv.init_params(nc, h, w, seed=11)
    for n in p:
        for k in ('biases', 'beta'):
            if k in p[n]:
                p[n][k] = (rng.standard_normal(p[n][k].shape) * 0.1).astype(np.float32)
        if 'moving_mean' in p[n]:
            p[n]['moving_mean'] = (rng.standard_normal(p[n]['moving_mean'].shape) * 0.1).astype(np.float32)
            p[n]['moving_variance'] = rng.uniform(0.5, 1.5, p[n]['moving_variance'].shape).astype(np.float32)
    return p


def _load_adv(adv, p):
    adv.set_params(p)
    adv.set_moving({n: (p[n]['moving_mean'], p[n]['moving_variance']) for n in p if 'moving_mean' in p[n]})


def test_ladder_known_answers():
    assert ladder(324, 324)['pool2'] == (4, 4) and ladder(512, 512)['pool2'] == (7, 7) and ladder(84, 84)['pool2'] == (1, 1)
    assert ladder(100, 100) == oadv.sizes(100, 100)
    with pytest.raises(Exception):
        ladder(68, 68)                    # the U-Net at 256^2: output 68 -> 17 -> 8 -> 4 -> 1 -> pool2 collapses


@pytest.mark.parametrize('hw', [(100, 100), (96, 132)])
def test_adversary_f32_vs_oracle(hw):
    """everything the adversary contributes to a train step, on given logits / labels"""
    B, nc = 3, 3
    h, w = hw
    rng = np.random.default_rng(21)
    dt = L.SEG_F32
    step = torch.zeros(2, dtype=torch.int64, device=U.dev())
    adv = Adversary(B, h, w, nc, dt, U.dev(), step.data_ptr() + 8, lr=1e-3, lam=2.0)
    p = _rand_adv_params(nc, h, w, rng)
    assert sum(np.asarray(v).size for n, t in adv.get_params().items() for v in t.values()) == oadv.n_params(p)
    _load_adv(adv, p)
    LH, LW, off = h + 5, w + 2, (3, 1)
    z = (rng.standard_normal((B, h, w, nc)) * 2).astype(np.float32)
    y = rng.integers(0, nc, (B, LH, LW, 1)).astype(np.uint8)
    thin = adv.A['x'].thin            # (the models hand the adversary class maps of its own kind: thin up to 8 classes)
    lg = E.Act(B, h, w, nc, dt, U.dev(), f32=True, thin=thin); U.fill_act(lg, z)
    d0 = (rng.standard_normal((B, h, w, nc)) * 1e-3).astype(np.float32)
    dl = E.Act(B, h, w, nc, dt, U.dev(), thin=thin); U.fill_act(dl, d0)
    yd = torch.from_numpy(y).to(U.dev())
    plan = E.Plan('adv')
    adv.store.g.fill_(float('nan'))
    adv.emit(plan, lg, yd, LH, LW, off, dl)
    plan.run(U.stream()); U.sync()
    ref = oadv.adversarial_terms(p, z, y[:, off[0]:off[0] + h, off[1]:off[1] + w], nc, lam=2.0)
    got = adv.losses.cpu().numpy()
    for i, k in enumerate(('l_bce_real', 'l_bce_fake', 'l_bce_fake_one')):
        assert abs(got[i] - ref[k]) < 2e-5, (k, got[i], ref[k])
    assert np.abs(U.read_act(adv.A['lg'])[:B, 0, 0, :] - ref['logits_real']).max() < 1e-4
    assert np.abs(U.read_act(adv.A['lg'])[B:, 0, 0, :] - ref['logits_fake']).max() < 1e-4
    assert bool(torch.isfinite(adv.store.g).all())
    g = adv.get_grads()
    for n in ref['adv_grads']:
        for k, r in ref['adv_grads'][n].items():
            # (adv_bn2's beta has an analytically ZERO gradient when pool2 is 1x1: adv_bn3 removes any per-feature shift; the
            # device value is float32 cancellation noise, hence the absolute floor)
            scale = max(np.abs(r).max(), 1e-6)
            assert np.abs(g[n][k].reshape(r.shape) - r).max() < 5e-4 * scale + 2e-6, (n, k)
    want = d0 + ref['d_seg_logits']
    assert np.abs(U.read_act(dl) - want).max() < 5e-4 * np.abs(ref['d_seg_logits']).max() + 1e-8
    mv = adv.get_moving()
    for n, (m_, v_) in ref['moving'].items():
        assert np.abs(mv[n][0] - m_).max() < 1e-5 and np.abs(mv[n][1] - v_).max() < 1e-5, n
    # advAdam: one TF-Adam step of the adversary's arena at its own learning rate
    up = E.Plan('u'); adv.emit_update(up)
    p0 = adv.store.p.clone()
    up.run(U.stream()); U.sync()
    gflat = adv.store.g.cpu().numpy().astype(np.float64)
    want_p, _, _ = ops.adam_tf(p0.cpu().numpy(), gflat, np.zeros_like(gflat), np.zeros_like(gflat), 1, lr=1e-3)
    assert np.abs(adv.store.p.cpu().numpy() - want_p).max() < 1e-6


def _torch_seg_terms(kind, p_seg, p_adv, x, y_win_of, nc, lam, fcn_type='8s'):
    """torch-autograd composition: seg loss = mean xent + lam * mean_b bce(a(softmax(logits)), 1); adversary loss =
    mean_b bce(a(one_hot(y)), 1) + mean_b bce(a(softmax(logits)), 0).  Returns the scalars, d seg loss / d seg params and
    d adv loss / d adv params."""
    dtp = torch.float64
    tp = T.to_torch_params(p_seg, dtp)
    ta = {n: {k: torch.tensor(np.asarray(a), dtype=dtp, requires_grad=True) for k, a in t.items() if k in ('weights', 'biases', 'beta')} for n, t in p_adv.items()}
    xt = torch.as_tensor(x, dtype=dtp)
    logits = T.unet_forward(tp, xt) if kind == 'unet' else T.fcn_forward(tp, xt, fcn_type)
    yw = y_win_of(logits.shape[1], logits.shape[2])
    yl = torch.as_tensor(yw[..., 0].astype(np.int64))
    xent = T.xent_mean(logits, yl)
    real = torch.nn.functional.one_hot(yl, nc).to(dtp)
    fake = torch.softmax(logits, -1)
    lr_, lf_ = T.adversary_forward(ta, real), T.adversary_forward(ta, fake)
    Bn = logits.shape[0]
    ones, zeros = torch.ones(Bn, dtype=torch.int64), torch.zeros(Bn, dtype=torch.int64)
    ce = torch.nn.functional.cross_entropy
    l_real, l_fake, l_one = ce(lr_, ones), ce(lf_, zeros), ce(lf_, ones)
    seg_flat = [a for t in tp.values() for a in t.values()]
    adv_flat = [a for t in ta.values() for a in t.values()]
    gs = torch.autograd.grad(xent + lam * l_one, seg_flat, retain_graph=True)
    ga = torch.autograd.grad(l_real + l_fake, adv_flat)
    it = iter(gs); g_seg = {n: {k: next(it).numpy() for k in t} for n, t in tp.items()}
    it = iter(ga); g_adv = {n: {k: next(it).numpy() for k in t} for n, t in ta.items()}
    return ({'seg_xentropy': float(xent.detach()), 'l_bce_real': float(l_real.detach()), 'l_bce_fake': float(l_fake.detach()),
             'l_bce_fake_one': float(l_one.detach())}, g_seg, g_adv)


def _check_model_vs_torch(m, kind, x, y, nc, crop, fcn_type='8s'):
    rng = np.random.default_rng(31)
    oh, ow = m.out_hw if kind == 'unet' else m.input_dims
    p_adv = _rand_adv_params(nc, oh, ow, rng)
    m._load_batch(m.dataset, m.input_x, m.input_y)
    _load_adv(m.adversary, p_adv)
    m.store.g.fill_(float('nan')); m.adversary.store.g.fill_(float('nan'))
    m._run_fwd_bwd()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.store.g).all()) and bool(torch.isfinite(m.adversary.store.g).all())
    p_seg = m.store.get_params()
    sc, g_seg, g_adv = _torch_seg_terms(kind, p_seg, p_adv, x[0], lambda h, w: y[0][:, crop[0]:crop[0] + h, crop[1]:crop[1] + w], nc, 2.0, fcn_type)
    got = m.last_losses()
    for k, v in sc.items():
        assert abs(got[k] - v) < 3e-5, (k, got[k], v)
    assert abs(got['seg_loss'] - (sc['seg_xentropy'] + 2.0 * sc['l_bce_fake_one'])) < 1e-4
    assert abs(got['adv_loss'] - (sc['l_bce_real'] + sc['l_bce_fake'])) < 1e-4
    g = m.store.get_grads()
    for n in g_seg:
        for k, r in g_seg[n].items():
            r = r.reshape(g[n][k].shape)
            assert np.abs(g[n][k] - r).max() < 3e-3 * np.abs(r).max() + 1e-9, (n, k)      # float32 device vs float64 autograd
    ga = m.adversary.get_grads()
    for n in g_adv:
        for k, r in g_adv[n].items():
            if np.abs(r).max() < 1e-10:          # analytically zero (adv_bn2's beta under a 1x1 pool2): float32 cancellation noise
                assert np.abs(ga[n][k]).max() < 1e-4, (n, k)
                continue
            assert np.abs(ga[n][k].reshape(r.shape) - r).max() < 3e-3 * np.abs(r).max() + 2e-7, (n, k)


def test_unet_adversarial_f32_step_vs_torch_autograd():
    B, S, nc = 2, 268, 3                      # U-Net output 84 x 84: the smallest map the adversary accepts
    rng = np.random.default_rng(5555)
    x = rng.uniform(0, 1, (1, B, S, S, 3)).astype(np.float32); y = rng.integers(0, nc, (1, B, S, S, 1)).astype(np.uint8)
    m = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=S, learning_rate=1e-4, log_dir=None, save_dir=None,
                  load_snapshot=False, dtype='f32', use_graph=False, n_kernels=8, adversarial_training=True, adversarial_lr=1e-4)
    assert m.out_hw == (84, 84) and m.adversary is not None
    _check_model_vs_torch(m, 'unet', x, y, nc, m.label_off)


def test_fcn_adversarial_f32_step_vs_torch_autograd():
    B, S, nc = 2, 96, 3
    rng = np.random.default_rng(77)
    x = rng.uniform(0, 1, (1, B, S, S, 3)).astype(np.float32); y = rng.integers(0, nc, (1, B, S, S, 1)).astype(np.uint8)
    m = FCNModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=S, learning_rate=1e-4, log_dir=None, save_dir=None,
                 load_snapshot=False, dtype='f32', use_graph=False, n_kernels=8, fcn_type='8s', adversarial_training=True)
    _check_model_vs_torch(m, 'fcn', x, y, nc, (0, 0))


@pytest.mark.parametrize('use_graph', [False, True])
def test_adversarial_training_runs_and_snapshots(tmp_path, use_graph):
    """bf16, a few steps: every scalar finite, both networks move, the adversary learns to tell the halves apart on a fixed
    batch, graph replay = eager launches, snapshot / restore carries the adversary."""
    B, S, nc = 2, 96, 2
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 1, (1, B, S, S, 3)).astype(np.float32); y = rng.integers(0, nc, (1, B, S, S, 1)).astype(np.uint8)

    def make(**kw):
        return FCNModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=S, learning_rate=1e-4, log_dir=None,
                        load_snapshot=False, dtype='bf16', use_graph=use_graph, n_kernels=8, fcn_type='8s', adversarial_training=True,
                        adversarial_lr=1e-3, **kw)
    m = make(save_dir=str(tmp_path))
    p0, a0 = m.store.p.clone(), m.adversary.store.p.clone()
    hist = []
    for _ in range(12):
        m.train_step()
        hist.append(m.last_losses())
    assert all(np.isfinite(list(h.values())).all() for h in hist)
    assert float((m.store.p - p0).abs().max()) > 0 and float((m.adversary.store.p - a0).abs().max()) > 0
    assert hist[-1]['adv_loss'] < hist[0]['adv_loss']
    assert m.global_step == 12
    mv_before, l_before, g_before = m.adversary.get_moving(), m.adversary.losses.clone(), m.adversary.store.g.clone()
    m.test()                               # a forward on a held-out batch must not touch the adversary (moving averages, losses, gradients)
    mv_after = m.adversary.get_moving()
    assert all(np.array_equal(mv_before[n][0], mv_after[n][0]) and np.array_equal(mv_before[n][1], mv_after[n][1]) for n in mv_before)
    assert torch.equal(l_before, m.adversary.losses) and torch.equal(g_before, m.adversary.store.g) and m.global_step == 12
    m.snapshot()
    m2 = make(save_dir=None)
    m2.restore(m._latest_checkpoint())
    assert torch.equal(m2.store.p, m.store.p) and torch.equal(m2.adversary.store.p, m.adversary.store.p)
    assert torch.equal(m2.adversary.store.m, m.adversary.store.m) and torch.equal(m2.adversary.store.v, m.adversary.store.v)
    mv, mv2 = m.adversary.get_moving(), m2.adversary.get_moving()
    assert all(np.array_equal(mv[n][0], mv2[n][0]) and np.array_equal(mv[n][1], mv2[n][1]) for n in mv)
    if use_graph:
        e = FCNModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=S, learning_rate=1e-4, log_dir=None, save_dir=None,
                     load_snapshot=False, dtype='bf16', use_graph=False, n_kernels=8, fcn_type='8s', adversarial_training=True, adversarial_lr=1e-3)
        for _ in range(12):
            e.train_step()
        assert torch.equal(e.store.p, m.store.p) and torch.equal(e.adversary.store.p, m.adversary.store.p)


def test_adversarial_rejections():
    x = np.zeros((1, 2, 256, 256, 3), np.float32); y = np.zeros((1, 2, 256, 256, 1), np.uint8)
    with pytest.raises(Exception, match='too small for the adversary'):
        UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=2, input_dims=256, log_dir=None, save_dir=None, load_snapshot=False,
                  n_kernels=8, adversarial_training=True)
