"""Oracle restatement of the reference DeconvModel graph, loss and backward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, as text:
  topology             /root/reference/models/deconvolution.py:101-178
  loss / optimizer     /root/reference/models/basemodel.py:59-70,357-369 (UPDATE_OPS dependency :364-365)
slim defaults that the file relies on: convolution2d / convolution2d_transpose apply bias + ReLU, batch_norm has
decay 0.999, epsilon 0.001, center=True, scale=False and FOLLOWS the ReLU of the layer in front of it (the reference calls
slim.batch_norm on the layer's output); slim.dropout keeps 0.5 and is always on when `bayesian` (is_training defaults True).
Quirks reproduced on purpose: both `bayesian` encoder sites are scoped 'drop1' (:129,144) -- they are two different
masks; infer() runs the TRAINING graph (batch statistics, dropout on) because y_hat is built with training=True (:80,:102)
while test() uses the moving averages (models/basemodel.py:397).
PARITY STATUS: unpinned at the TF boundary like the rest of the oracle (TensorFlow 1.x cannot run here); two independent
implementations (this file, oracle/torch_ref.py deconv_* autograd) agree to 1e-9 in float64 (tests/test_oracle.py).
The dropout mask generator is build-defined (counter-based, oracle.np_ops.dropout_mask).
"""
import numpy as np
from . import np_ops as ops

WEIGHTED = ['conv1_0', 'conv2_0', 'conv3_0', 'conv4_0', 'deconv1_0', 'deconv2_0', 'deconv2_1', 'deconv3_0', 'conv_out']
BN_OF = {'conv1_0': 'bn1', 'conv2_0': 'bn2', 'conv3_0': 'bn3', 'conv4_0': 'bn4', 'deconv1_0': 'bn5', 'deconv2_0': 'bn6',
         'deconv2_1': 'bn7', 'deconv3_0': 'bn8'}
DROP_SITES = {'bn2': 1, 'bn4': 2, 'bn5': 3}            # after which batch norm a `bayesian` dropout sits -> seed increment


def layer_shapes(n_classes=2, n_kernels=32, input_channel=3):
    """name -> (kind, weight shape, cin, cout, k, stride, padding).  conv: HWIO; deconv: [kh,kw,Cout,Cin]."""
    nk = n_kernels
    return {
        'conv1_0': ('conv', (5, 5, input_channel, nk), input_channel, nk, 5, 2, 'SAME'),
        'conv2_0': ('conv', (3, 3, nk, 2 * nk), nk, 2 * nk, 3, 1, 'VALID'),
        'conv3_0': ('conv', (3, 3, 2 * nk, 4 * nk), 2 * nk, 4 * nk, 3, 1, 'VALID'),
        'conv4_0': ('conv', (3, 3, 4 * nk, 8 * nk), 4 * nk, 8 * nk, 3, 1, 'VALID'),
        'deconv1_0': ('deconv', (5, 5, 2 * nk, 8 * nk), 8 * nk, 2 * nk, 5, 2, 'VALID'),
        'deconv2_0': ('deconv', (5, 5, nk, 2 * nk), 2 * nk, nk, 5, 2, 'VALID'),
        'deconv2_1': ('deconv', (5, 5, nk, nk), nk, nk, 5, 2, 'VALID'),
        'deconv3_0': ('deconv', (2, 2, n_classes, nk), nk, n_classes, 2, 2, 'VALID'),
        'conv_out': ('conv', (3, 3, n_classes, n_classes), n_classes, n_classes, 3, 1, 'SAME'),
    }


def init_params(n_classes=2, n_kernels=32, input_channel=3, seed=5555):
    """xavier-uniform weights, zero biases, zero betas; moving_mean 0, moving_variance 1 (slim defaults)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name in WEIGHTED:
        kind, shp, ci, co, k, s, pad = layer_shapes(n_classes, n_kernels, input_channel)[name]
        p[name] = {'weights': ops.xavier_uniform(shp, rng, k * k * ci, k * k * co), 'biases': np.zeros((co,), np.float32)}
        if name in BN_OF:
            p[BN_OF[name]] = {'beta': np.zeros((co,), np.float32), 'moving_mean': np.zeros((co,), np.float32),
                              'moving_variance': np.ones((co,), np.float32)}
    return p


def forward(p, x, n_classes=None, training=True, bayesian=False, dropout=None, dt=np.float64):
    """Returns (logits, cache, new_moving {bn: (mean, var)}).  dropout = {'keep':, 'seed':, 'offset':} (bayesian only)."""
    c, newmov = {}, {}
    x = np.asarray(x, dt)
    H, W = x.shape[1:3]
    c['x'] = x

    def act_bn(name, a):
        """ReLU output `a` of layer `name` -> batch norm (+ dropout)"""
        bn = BN_OF[name]
        y, cache, nm, nv = ops.batch_norm(a, p[bn]['beta'], p[bn]['moving_mean'], p[bn]['moving_variance'], training, dt=dt)
        c[name] = a; c[bn] = y; c[bn + '/cache'] = cache
        if training:
            newmov[bn] = (nm, nv)
        if bayesian and bn in DROP_SITES:
            d = dropout or {'keep': 0.5, 'seed': 5555, 'offset': 0}
            m = ops.dropout_mask(y.shape, d['keep'], d['seed'] + DROP_SITES[bn], d['offset'])
            c[bn + '/mask'] = m
            y = y * m * dt(np.float32(1.0) / np.float32(d['keep']))
            c[bn + '/drop'] = y
        return y

    conv = lambda t, n, s, pad: ops.conv2d(t, p[n]['weights'], p[n]['biases'], pad, s, True, dt)
    dec = lambda t, n, s: ops.conv2d_transpose(t, p[n]['weights'], p[n]['biases'], s, 'VALID', True, dt)
    net = act_bn('conv1_0', conv(x, 'conv1_0', 2, 'SAME'))
    net, c['idx1'] = ops.max_pool_k(net, 2); c['pool1'] = net
    net = act_bn('conv2_0', conv(net, 'conv2_0', 1, 'VALID'))
    c['pool2_in'] = net
    net, c['idx2'] = ops.max_pool_k(net, 3); c['pool2'] = net
    net = act_bn('conv3_0', conv(net, 'conv3_0', 1, 'VALID'))
    net, c['idx3'] = ops.max_pool_k(net, 3); c['pool3'] = net
    net = act_bn('conv4_0', conv(net, 'conv4_0', 1, 'VALID'))
    c['dec_in'] = net
    net = act_bn('deconv1_0', dec(net, 'deconv1_0', 2)); c['d1'] = net
    net = act_bn('deconv2_0', dec(net, 'deconv2_0', 2)); c['d2'] = net
    net = act_bn('deconv2_1', dec(net, 'deconv2_1', 2)); c['d3'] = net
    net = ops.resize_bilinear(net, (H // 2, W // 2), dt); c['resize'] = net
    net = act_bn('deconv3_0', dec(net, 'deconv3_0', 2)); c['d4'] = net
    net = ops.crop_or_pad(net, H, W); c['force_resize'] = net
    c['logits'] = ops.conv2d(net, p['conv_out']['weights'], p['conv_out']['biases'], 'SAME', 1, False, dt)
    return c['logits'], c, newmov


def loss_and_grads(p, x, y, bayesian=False, dropout=None, dt=np.float64):
    """Training-mode forward + backward.  Returns (loss, grads {layer: {weights, biases}, bn: {beta}}, cache, new_moving)."""
    logits, c, newmov = forward(p, x, training=True, bayesian=bayesian, dropout=dropout, dt=dt)
    loss, _, d = ops.softmax_xent(logits, y, dt)
    g = {}
    W = lambda n: np.asarray(p[n]['weights'], dt)
    H, Wd = c['x'].shape[1:3]

    def bn_bwd(name, dy):
        """gradient at the batch-norm output of layer `name` -> masked gradient dz of that layer's pre-activation"""
        bn = BN_OF[name]
        if bayesian and bn in DROP_SITES:
            keep = (dropout or {'keep': 0.5})['keep']
            dy = dy * c[bn + '/mask'] * dt(np.float32(1.0) / np.float32(keep))
        da, dbeta = ops.batch_norm_bwd(dy, c[bn + '/cache'])
        g[bn] = {'beta': dbeta}
        return da * (c[name] > 0)

    def conv_bwd(name, xin, dz, s, pad, need_dx=True):
        dw, db = ops.conv2d_wgrad(xin, dz, W(name).shape[:2], pad, s, dt)
        g[name] = {'weights': dw, 'biases': db}
        return ops.conv2d_dgrad(dz, W(name), xin.shape[1:3], pad, s, dt) if need_dx else None

    def dec_bwd(name, xin, dz, s):
        k = W(name).shape[:2]
        dw, db = ops.conv2d_transpose_wgrad(xin, dz, k, s, 'VALID', dt)
        g[name] = {'weights': dw, 'biases': db}
        return ops.conv2d_transpose_dgrad(dz, W(name), xin.shape[1:3], s, 'VALID', dt)

    d = conv_bwd('conv_out', c['force_resize'], d, 1, 'SAME')
    d = ops.crop_or_pad_bwd(d, c['d4'].shape[1:3])
    d = dec_bwd('deconv3_0', c['resize'], bn_bwd('deconv3_0', d), 2)
    d = ops.resize_bilinear_bwd(d, c['d3'].shape[1:3], dt)
    d = dec_bwd('deconv2_1', c['d2'], bn_bwd('deconv2_1', d), 2)
    d = dec_bwd('deconv2_0', c['d1'], bn_bwd('deconv2_0', d), 2)
    d = dec_bwd('deconv1_0', c['dec_in'], bn_bwd('deconv1_0', d), 2)
    d = conv_bwd('conv4_0', c['pool3'], bn_bwd('conv4_0', d), 1, 'VALID')
    d = ops.max_pool_k_bwd(d, c['idx3'], c['bn3'].shape[1:3], 3)
    d = conv_bwd('conv3_0', c['pool2'], bn_bwd('conv3_0', d), 1, 'VALID')
    d = ops.max_pool_k_bwd(d, c['idx2'], c['pool2_in'].shape[1:3], 3)
    d = conv_bwd('conv2_0', c['pool1'], bn_bwd('conv2_0', d), 1, 'VALID')
    d = ops.max_pool_k_bwd(d, c['idx1'], c['bn1'].shape[1:3], 2)
    conv_bwd('conv1_0', c['x'], bn_bwd('conv1_0', d), 2, 'SAME', need_dx=False)
    return loss, g, c, newmov


def trainable(p):
    """tensors the optimizer updates (moving averages are not trained)"""
    out = {}
    for n, t in p.items():
        out[n] = {k: v for k, v in t.items() if k in ('weights', 'biases', 'beta')}
    return out


def train_step(p, m, v, step, x, y, lr=1e-4, bayesian=False, dropout=None, dt=np.float64):
    """fwd + mean x-entropy + bwd + TF-Adam on weights / biases / betas, and the UPDATE_OPS moving-average assignment
    (models/basemodel.py:364-366).  m, v: Adam slots keyed like trainable(p).  Returns (loss, p', m', v')."""
    loss, g, _, newmov = loss_and_grads(p, x, y, bayesian, dropout, dt)
    p2 = {n: dict(t) for n, t in p.items()}
    m2 = {n: dict(t) for n, t in m.items()}
    v2 = {n: dict(t) for n, t in v.items()}
    for n, t in trainable(p).items():
        for k in t:
            p2[n][k], m2[n][k], v2[n][k] = ops.adam_tf(p[n][k], g[n][k], m[n][k], v[n][k], step, lr, dt=dt)
    for bn, (nm, nv) in newmov.items():
        p2[bn]['moving_mean'], p2[bn]['moving_variance'] = nm, nv
    return loss, p2, m2, v2


def init_opt_state(p):
    z = lambda: {n: {k: np.zeros_like(a, dtype=np.float64) for k, a in t.items()} for n, t in trainable(p).items()}
    return z(), z()


def n_params(p):
    return int(sum(a.size for t in trainable(p).values() for a in t.values()))


def infer(p, x, bayesian=False, dropout=None, dt=np.float64):
    """BaseModel.infer on the reference DeconvModel: the TRAINING graph (batch statistics; dropout on when bayesian)."""
    logits, _, _ = forward(p, x, training=True, bayesian=bayesian, dropout=dropout, dt=dt)
    return ops.sigmoid_argmax(logits.astype(np.float32))
