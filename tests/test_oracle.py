"""CPU tests of the oracle: golden vectors from the reference's only executable file,
known-answer tests derivable from the reference text, and agreement of the
numpy restatement with the independent torch-CPU composition."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import np_ops as ops
from oracle import unet as ounet
from oracle import fcn as ofcn
from oracle import torch_ref

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'bilinear_golden.npz')


# ---------------- golden: utils/upsampling.py ----------------
def test_bilinear_golden_vectors():
    g = np.load(GOLD)
    for f in (1, 2, 3, 4, 8, 16, 32):
        assert ops.get_kernel_size(f) == int(g['ksize_%d' % f])
    for s in (1, 2, 3, 4, 5, 8, 16, 32, 64):
        ref = g['filt_%d' % s]
        mine = ops.upsample_filt(s)
        assert mine.dtype == np.float64 and np.array_equal(mine, ref)
    sha = dict(zip(g['sha_keys'].tolist(), g['sha_vals'].tolist()))
    for key, want in sha.items():
        f, c = map(int, key.split('_'))
        w = ops.bilinear_upsample_weights(f, c)
        assert w.dtype == np.float32
        assert hashlib.sha256(np.ascontiguousarray(w).tobytes()).hexdigest() == want
        if 'weights_' + key in g:
            assert np.array_equal(w, g['weights_' + key])
        assert np.isclose(w.sum(), c * f * f)


def test_product_upsampling_module_matches_golden_vectors():
    """segmentation_amd/upsampling.py (what FCNModel actually uploads) against the vectors generated from the reference's
    utils/upsampling.py by tests/golden/make_bilinear_golden.py -- bit for bit, like the oracle copy above."""
    from segmentation_amd import upsampling as up
    g = np.load(GOLD)
    for f in (1, 2, 3, 4, 8, 16, 32):
        assert up.get_kernel_size(f) == int(g['ksize_%d' % f])
    for s in (1, 2, 3, 4, 5, 8, 16, 32, 64):
        mine = up.upsample_filt(s)
        assert mine.dtype == np.float64 and np.array_equal(mine, g['filt_%d' % s])
    for key, want in zip(g['sha_keys'].tolist(), g['sha_vals'].tolist()):
        f, c = map(int, key.split('_'))
        w = up.bilinear_upsample_weights(f, c)
        assert w.dtype == np.float32 and w.shape == (up.get_kernel_size(f),) * 2 + (c, c)
        assert hashlib.sha256(np.ascontiguousarray(w).tobytes()).hexdigest() == want
        if 'weights_' + key in g:
            assert np.array_equal(w, g['weights_' + key])


def test_dropout_oracle_mask_properties():
    """Build-defined MC-dropout mask generator (a19): deterministic in (seed, offset), independent across the per-site
    seeds and the per-pass offsets the product uses, Bernoulli(keep) marginals, pad channels excluded from the index."""
    shape = (2, 9, 7, 40)
    m0 = ops.dropout_mask(shape, 0.5, 5557, 1 << 40)
    assert m0.dtype == bool and np.array_equal(m0, ops.dropout_mask(shape, 0.5, 5557, 1 << 40))
    assert abs(m0.mean() - 0.5) < 0.03
    # the aliasing the old generator had: site seed +3 at pass t vs site seed +0 at pass t+3 -- now uncorrelated
    for ds, dt in ((3, 3), (5, 1), (8, 2)):
        m1 = ops.dropout_mask(shape, 0.5, 5557 + ds, (1 + dt) << 40)
        assert abs((m0 == m1).mean() - 0.5) < 0.03
    assert ops.dropout_mask(shape, 1.0, 1, 0).all() and abs(ops.dropout_mask(shape, 0.25, 1, 0).mean() - 0.25) < 0.03
    # counter layout: element (b,y,x,c) -> ((b*H+y)*W+x)*c_pad + c with c_pad = 64 here
    a = ops.dropout_mask((1, 1, 2, 40), 0.5, 9, 0)
    b = ops.dropout_mask((1, 1, 1, 40), 0.5, 9, 64)
    assert np.array_equal(a[0, 0, 1], b[0, 0, 0])
    x = np.ones(shape)
    y = ops.dropout(x, 0.5, 5557, 1 << 40)
    assert set(np.unique(y)) == {0.0, 2.0}


# ---------------- known answers from the reference text ----------------
def test_unet_shape_ladder():
    assert ounet.output_size(256) == 68
    assert ounet.output_size(512) == 324
    assert ounet.output_size(186) == 4
    assert ounet.output_size(188) == 4
    assert ounet.output_size(572) == 388
    with pytest.raises(ValueError):
        ounet.output_size(128)


def test_unet_param_count():
    assert ounet.n_params(ounet.init_params(4, 32)) == 7760196
    assert ounet.n_params(ounet.init_params(2, 32)) == 7760130


def test_fcn_param_count():
    assert sum(v['weights'].size + v['biases'].size for v in ofcn.init_params(21, 32).values()) == 2320895
    assert sum(v['weights'].size + v['biases'].size for v in ofcn.init_params(21, 64).values()) == 9216959


def test_crop_offsets():
    # a5: crops 24->16 (off 4), 57->24 (16), 123->40 (41), 252->72 (90)
    for n, t, off in ((24, 16, 4), (57, 24, 16), (123, 40, 41), (252, 72, 90), (256, 68, 94)):
        x = np.arange(n * n, dtype=np.float64).reshape(1, n, n, 1)
        c = ops.crop_or_pad(x, t, t)
        assert c[0, 0, 0, 0] == x[0, off, off, 0]
    x = np.ones((1, 3, 3, 1))
    p = ops.crop_or_pad(x, 6, 6)      # pad before = floor(3/2) = 1
    assert p[0, 1, 1, 0] == 1 and p[0, 0, 0, 0] == 0 and p.sum() == 9 and p[0, 3, 3, 0] == 1 and p[0, 4, 4, 0] == 0


def test_pool_first_max_and_odd():
    x = np.zeros((1, 5, 5, 1))
    y, idx = ops.max_pool2x2(x)
    assert y.shape == (1, 2, 2, 1) and (idx == 0).all()      # ties -> first in window order
    x[0, 1, 0, 0] = 3; x[0, 1, 1, 0] = 3
    y, idx = ops.max_pool2x2(x)
    assert y[0, 0, 0, 0] == 3 and idx[0, 0, 0, 0] == 2
    dx = ops.max_pool2x2_bwd(np.ones_like(y), idx, (5, 5))
    assert dx.shape == (1, 5, 5, 1) and dx[0, 1, 0, 0] == 1 and dx[0, 4].sum() == 0 and dx.sum() == 4


def test_sigmoid_argmax_tie_rule():
    # F17: two saturated logits -> both sigmoid == 1.0f -> lower index wins
    z = np.array([[[[-1.0, 20.0, 30.0, 2.0]]]], np.float32)
    sig, out = ops.sigmoid_argmax(z)
    assert sig[0, 0, 0, 1] == np.float32(1.0) and sig[0, 0, 0, 2] == np.float32(1.0)
    assert out.shape == (1, 1, 1, 1) and out[0, 0, 0, 0] == 1.0 and out.dtype == np.float32


def test_adam_tf_eps_outside():
    p, m, v = ops.adam_tf(np.array([1.0]), np.array([0.5]), np.array([0.0]), np.array([0.0]), 1, lr=0.1)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    m1, v1 = 0.05, 0.001 * 0.25
    assert np.allclose(p, 1 - lr_t * m1 / (np.sqrt(v1) + 1e-8))


def test_unet_pool1_quirk_and_concat_order():
    # F12: perturbing conv1_2's weights must not change anything but via the last skip: compare
    # against torch composition which encodes the same quirk independently.
    p = ounet.init_params(2, 2, seed=1)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (1, 188, 188, 3)).astype(np.float32)
    logits, c = ounet.forward(p, x)
    assert c['pool1'].shape[1] == 93 and c['conv1_2'].shape[1] == 184
    assert logits.shape == (1, 4, 4, 2)
    assert c['cat4'].shape[-1] == 4 and np.array_equal(c['cat4'][..., :2], ops.crop_or_pad(c['conv1_2'], 8, 8))


# ---------------- numpy restatement vs independent torch composition ----------------
def test_ops_vs_torch_odd_sizes():
    rng = np.random.default_rng(3)
    import torch.nn.functional as F
    for (h, ci, co, k, pad) in ((13, 5, 7, 3, 'VALID'), (9, 4, 6, 3, 'SAME'), (7, 8, 3, 1, 'SAME'), (12, 3, 5, 2, 'VALID')):
        x = rng.standard_normal((2, h, h + 1, ci)); w = rng.standard_normal((k, k, ci, co)); b = rng.standard_normal(co)
        y = ops.conv2d(x, w, b, pad, 1, relu=False)
        xt = torch.tensor(x).permute(0, 3, 1, 2); wt = torch.tensor(w).permute(3, 2, 0, 1)
        if pad == 'SAME':
            xt = F.pad(xt, ((k - 1) // 2, k // 2, (k - 1) // 2, k // 2))
        yt = F.conv2d(xt, wt, torch.tensor(b)).permute(0, 2, 3, 1).numpy()
        assert np.allclose(y, yt, atol=1e-10)
    # transposed conv 2x2 s2 VALID and bilinear SAME
    x = rng.standard_normal((2, 5, 6, 4)); w = rng.standard_normal((2, 2, 3, 4))
    y = ops.conv2d_transpose(x, w, None, 2, 'VALID')
    yt = F.conv_transpose2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(3, 2, 0, 1), stride=2).permute(0, 2, 3, 1).numpy()
    assert y.shape == (2, 10, 12, 3) and np.allclose(y, yt, atol=1e-10)
    for f in (2, 8):
        x = rng.standard_normal((1, 3, 3, 2))
        y = ops.conv2d_transpose(x, ops.bilinear_upsample_weights(f, 2), None, f, 'SAME')
        yt = torch_ref._bilinear_up(torch.tensor(x).permute(0, 3, 1, 2), f).permute(0, 2, 3, 1).numpy()
        assert y.shape == (1, 3 * f, 3 * f, 2) and np.allclose(y, yt, atol=1e-10)


def test_unet_grads_vs_torch_autograd():
    p = ounet.init_params(3, 2, seed=7)
    for n in p:   # non-zero biases so bias paths are exercised
        p[n]['biases'] = (np.random.default_rng(1).standard_normal(p[n]['biases'].shape) * 0.1).astype(np.float32)
    rng = np.random.default_rng(5555)
    x = rng.uniform(0, 1, (2, 188, 188, 3)).astype(np.float32)
    y = rng.integers(0, 3, (2, 188, 188, 1)).astype(np.uint8)
    loss, g, c = ounet.loss_and_grads(p, x, y)
    tl, tg, tlog = torch_ref.unet_loss_and_grads(p, x, y)
    assert abs(loss - tl) < 1e-10
    assert np.allclose(c['logits'], tlog, atol=1e-10)
    for n in g:
        for k in ('weights', 'biases'):
            assert np.allclose(g[n][k], tg[n][k], atol=1e-9, rtol=1e-7), (n, k)


def test_unet_finite_difference():
    p = ounet.init_params(2, 2, seed=11)
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 1, (1, 188, 188, 3)).astype(np.float32)
    y = rng.integers(0, 2, (1, 188, 188, 1)).astype(np.uint8)
    p64 = {n: {k: v.astype(np.float64) for k, v in t.items()} for n, t in p.items()}
    loss, g, _ = ounet.loss_and_grads(p64, x, y)
    eps = 1e-6
    for (n, k, idx) in (('conv5_2', 'weights', (1, 1, 3, 5)), ('upconv2', 'weights', (0, 1, 2, 3)),
                        ('conv1_1', 'weights', (2, 0, 1, 1)), ('output', 'biases', (1,)), ('conv1_2', 'weights', (1, 1, 0, 1))):
        q = {a: {b: v.copy() for b, v in t.items()} for a, t in p64.items()}
        q[n][k][idx] += eps
        lp, _ = ounet.forward(q, x)
        lp = ops.softmax_xent(lp, ounet.crop_labels(y, lp.shape[1]))[0]
        q[n][k][idx] -= 2 * eps
        lm, _ = ounet.forward(q, x)
        lm = ops.softmax_xent(lm, ounet.crop_labels(y, lm.shape[1]))[0]
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[n][k][idx]) < 1e-6 + 1e-4 * abs(fd), (n, k, fd, g[n][k][idx])


@pytest.mark.parametrize('fcn_type', ['32s', '16s', '8s'])
def test_fcn_grads_vs_torch_autograd(fcn_type):
    p = ofcn.init_params(5, 2, fcn_type=fcn_type, seed=3)
    for n in p:
        p[n]['biases'] = (np.random.default_rng(1).standard_normal(p[n]['biases'].shape) * 0.1 + 0.05).astype(np.float32)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, 1, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 5, (2, 64, 64, 1)).astype(np.uint8)
    loss, g, c = ofcn.loss_and_grads(p, x, y, fcn_type)
    tl, tg, tlog = torch_ref.fcn_loss_and_grads(p, x, y, fcn_type)
    assert c['logits'].shape == (2, 64, 64, 5)
    assert np.allclose(c['logits'], tlog, atol=1e-9)
    assert abs(loss - tl) < 1e-10
    for n in g:
        for k in ('weights', 'biases'):
            assert np.allclose(g[n][k], tg[n][k], atol=1e-9, rtol=1e-7), (n, k)


def test_train_step_tf_adam_vs_torch():
    p = ounet.init_params(2, 2, seed=4)
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 1, (1, 188, 188, 3)).astype(np.float32)
    y = rng.integers(0, 2, (1, 188, 188, 1)).astype(np.uint8)
    m, v = ounet.init_opt_state(p)
    l1, p1, m1, v1 = ounet.train_step(p, m, v, 1, x, y, lr=1e-3)
    l2, p2, _, _ = ounet.train_step(p1, m1, v1, 2, x, y, lr=1e-3)
    st = torch_ref.TorchUNetStepper(p, lr=1e-3)
    t1 = st.train_step(x, y); t2 = st.train_step(x, y)
    assert abs(l1 - t1) < 1e-5 and abs(l2 - t2) < 1e-5
    names = [(n, k) for n in p for k in ('weights', 'biases')]
    for (n, k), t in zip(names, st.flat):
        assert np.allclose(p2[n][k], t.detach().numpy(), atol=2e-5), (n, k)


# ---------------- DeconvModel: numpy restatement vs the independent torch-autograd composition ----------------
@pytest.mark.parametrize('bayesian', [False, True])
def test_deconv_oracle_agrees_with_torch_autograd(bayesian):
    from oracle import deconv as odec
    p = odec.init_params(3, 8, 3, seed=3)
    assert odec.n_params(p) == 55682 and odec.n_params(odec.init_params(2, 32, 3)) == 877386
    rng = np.random.default_rng(1)
    for n in p:
        for k in ('biases', 'beta'):
            if k in p[n]:
                p[n][k] = (rng.standard_normal(p[n][k].shape) * 0.1).astype(np.float32)
    x = rng.uniform(0, 1, (2, 160, 160, 3)).astype(np.float32)
    y = rng.integers(0, 3, (2, 160, 160, 1)).astype(np.uint8)
    loss, g, c, newmov = odec.loss_and_grads(p, x, y, bayesian=bayesian, dropout={'keep': 0.5, 'seed': 7, 'offset': 1 << 40})
    # size ladder of models/deconvolution.py at 160: 80, 40, 38, 12, 10, 3, 1, then 5, 13, 29, resize 80, 160
    assert [c[k].shape[1] for k in ('conv1_0', 'pool1', 'conv2_0', 'pool2', 'conv3_0', 'pool3', 'conv4_0', 'd1', 'd2', 'd3', 'resize', 'd4')] == \
        [80, 40, 38, 12, 10, 3, 1, 5, 13, 29, 80, 160]
    masks = {bn: c[bn + '/mask'] for bn in odec.DROP_SITES} if bayesian else None
    l2, g2, lg2, st = torch_ref.deconv_loss_and_grads(p, x, y, masks)
    assert abs(loss - l2) < 1e-10 and np.abs(c['logits'] - lg2).max() < 1e-9
    for n in g:
        for k in g[n]:
            assert np.abs(g[n][k] - g2[n][k]).max() <= 1e-8 * (np.abs(g2[n][k]).max() + 1e-30), (n, k)
    for bn, (nm, nv) in newmov.items():        # UPDATE_OPS: decay 0.999 moving averages of the batch mean / population variance
        assert np.allclose(nm, 0.999 * p[bn]['moving_mean'] + 0.001 * st[bn][0], atol=1e-12)
        assert np.allclose(nv, 0.999 * p[bn]['moving_variance'] + 0.001 * st[bn][1], atol=1e-12)
    # test()-graph: moving averages, not batch statistics
    lt, _, nm2 = odec.forward(p, x, training=False)
    assert nm2 == {} and np.abs(lt - c['logits']).max() > 1e-3


def test_pool_k_and_resize_oracle_properties():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 11, 10, 3))
    y, idx = ops.max_pool_k(x, 3)
    assert y.shape == (2, 3, 3, 3) and y[0, 0, 0, 0] == x[0, :3, :3, 0].max()
    y2, idx2 = ops.max_pool_k(x, 2)
    y2b, idx2b = ops.max_pool2x2(x)
    assert np.array_equal(y2, y2b) and np.array_equal(idx2, idx2b)
    dy = rng.standard_normal(y.shape)
    dx = ops.max_pool_k_bwd(dy, idx, (11, 10), 3)
    assert dx.shape == x.shape and np.isclose(dx.sum(), dy.sum()) and (dx[:, 9:] == 0).all() and (dx[:, :, 9:] == 0).all()
    r = ops.resize_bilinear(x, (22, 25))
    g = rng.standard_normal(r.shape)
    assert abs((r * g).sum() - (x * ops.resize_bilinear_bwd(g, (11, 10))).sum()) < 1e-10       # adjoint
    assert np.array_equal(ops.resize_bilinear(x, (11, 10)), x)                                    # identity size
    assert np.allclose(ops.resize_bilinear(np.ones((1, 5, 5, 1)), (13, 9)), 1.0)


# ---------------- plain-C restatement (oracle/seg_cpu.c -> libseg_cpu.so) vs the numpy restatement ----------------
def test_c_restatement_ops_match_numpy():
    from oracle import c_ops as co
    assert co.load().segcpu_version() == 100 and co.load().segcpu_threads() >= 1
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 13, 11, 5)).astype(np.float32)
    w = rng.standard_normal((3, 3, 5, 7)).astype(np.float32); b = rng.standard_normal(7).astype(np.float32)
    for pad, s in (('VALID', 1), ('SAME', 1), ('SAME', 2), ('VALID', 2)):
        r = ops.conv2d(x, w, b, pad, s, True)
        assert np.abs(co.conv2d(x, w, b, pad, s, True) - r).max() < 2e-5
        dz = rng.standard_normal(r.shape).astype(np.float32)
        assert np.abs(co.conv2d_dgrad(dz, w, (13, 11), pad, s) - ops.conv2d_dgrad(dz, w, (13, 11), pad, s)).max() < 2e-5
        dw, db = co.conv2d_wgrad(x, dz, 3, pad, s)
        rdw, rdb = ops.conv2d_wgrad(x, dz, (3, 3), pad, s)
        assert np.abs(dw - rdw).max() < 1e-4 and np.abs(db - rdb).max() < 1e-5
    # crop by view: convolving a window of x == convolving the cropped copy
    assert np.array_equal(co.conv2d(x, w, b, window=(2, 1, 9, 8)), co.conv2d(np.ascontiguousarray(x[:, 2:11, 1:9]), w, b))
    wt = rng.standard_normal((2, 2, 6, 5)).astype(np.float32); bt = rng.standard_normal(6).astype(np.float32)
    r = ops.conv2d_transpose(x, wt, bt, 2, 'VALID', True)
    assert np.abs(co.convT2x2(x, wt, bt) - r).max() < 1e-5
    dz = rng.standard_normal(r.shape).astype(np.float32)
    dx, dw, db = co.convT2x2_bwd(x, wt, dz)
    rdw, rdb = ops.conv2d_transpose_wgrad(x, dz, (2, 2), 2, 'VALID')
    assert np.abs(dx - ops.conv2d_transpose_dgrad(dz, wt, (13, 11), 2, 'VALID')).max() < 2e-5
    assert np.abs(dw - rdw).max() < 1e-4 and np.abs(db - rdb).max() < 2e-5
    x[0, 0:2, 0:2, 0] = 0.5                                   # tie: first position wins
    yp, idx = co.maxpool2x2(x); rp, ridx = ops.max_pool2x2(x)
    assert np.array_equal(yp, rp.astype(np.float32)) and np.array_equal(idx, ridx)
    dyp = rng.standard_normal(rp.shape).astype(np.float32)
    assert np.array_equal(co.maxpool2x2_bwd(dyp, idx, (13, 11)), ops.max_pool2x2_bwd(dyp, ridx, (13, 11)))
    z = (rng.standard_normal((2, 5, 4, 4)) * 3).astype(np.float32); lab = rng.integers(0, 4, (2, 5, 4)).astype(np.uint8)
    l, dzc = co.softmax_xent(z, lab)
    lr_, _, dref = ops.softmax_xent(z, lab)
    assert abs(l - lr_) < 1e-6 and np.abs(dzc - dref).max() < 1e-7


def test_c_restatement_unet_train_step_matches_numpy():
    from oracle import c_ops as co
    p = ounet.init_params(2, 8, 3, seed=1)
    rng = np.random.default_rng(3)
    xb = rng.uniform(0, 1, (1, 188, 188, 3)).astype(np.float32); yb = rng.integers(0, 2, (1, 188, 188, 1)).astype(np.uint8)
    st = co.CUNetStepper(p, lr=1e-3, threads=2)
    l, g, c = st.loss_and_grads(xb, yb)
    lr_, gr, cr = ounet.loss_and_grads(p, xb, yb)
    assert abs(l - lr_) < 1e-6 and np.abs(c['logits'] - cr['logits']).max() < 1e-5
    for n in gr:
        for k in ('weights', 'biases'):
            assert np.abs(g[n][k] - gr[n][k]).max() <= 2e-4 * (np.abs(gr[n][k]).max() + 1e-30), (n, k)
    mm, vv = ounet.init_opt_state(p)
    st.train_step(xb, yb)
    _, p1, _, _ = ounet.train_step(p, mm, vv, 1, xb, yb, lr=1e-3)
    assert max(np.abs(st.p[n][k] - p1[n][k]).max() for n in p1 for k in ('weights', 'biases')) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# adversarial training terms: numpy restatement vs the torch-autograd composition, known-answer ladder
# ---------------------------------------------------------------------------------------------------------------------
def test_adversary_oracle_vs_torch_autograd():
    from oracle import adversary as A
    rng = np.random.default_rng(0)
    B, h, w, nc = 4, 96, 100, 3
    p = A.init_params(nc, h, w)
    for n in p:
        for k in ('biases', 'beta'):
            if k in p[n]:
                p[n][k] = (rng.standard_normal(p[n][k].shape) * 0.1).astype(np.float32)
    assert A.sizes(324, 324)['pool2'] == (4, 4) and A.sizes(512, 512)['pool2'] == (7, 7)
    with pytest.raises(ValueError):
        A.sizes(68, 68)                    # the U-Net's 256^2 output: the adversary's map collapses (like F11 at 128^2)
    assert A.n_params(A.init_params(4, 324, 324)) == 3 * 3 * 4 * 36 + 36 + 36 + 3 * 3 * 36 * 72 + 72 + 72 + 1152 + 1152 * 1024 + 1024 + 1024 + 1024 * 2 + 2
    z = rng.standard_normal((B, h, w, nc)) * 2
    y = rng.integers(0, nc, (B, h, w, 1)).astype(np.uint8)
    a, t = A.adversarial_terms(p, z, y, nc), torch_ref.adversarial_terms(p, z, y, nc)
    for k in ('l_bce_real', 'l_bce_fake', 'l_bce_fake_one'):
        assert abs(a[k] - t[k]) < 1e-12
    assert np.abs(a['d_seg_logits'] - t['d_seg_logits']).max() < 1e-12 * max(1.0, np.abs(t['d_seg_logits']).max())
    for n in a['adv_grads']:
        for k, v in a['adv_grads'][n].items():
            r = t['adv_grads'][n][k].reshape(v.shape)
            assert np.abs(v - r).max() < 1e-10 * max(np.abs(r).max(), 1e-3), (n, k)
    # the moving averages see the real pass, then the fake pass
    _, _, mv1 = A.forward(p, A.one_hot(y, nc))
    assert all(np.allclose(a['moving'][n][0], 0.999 * mv1[n][0] + 0.001 * (a['moving'][n][0] - 0.999 * mv1[n][0]) / 0.001) for n in mv1)
