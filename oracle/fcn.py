"""Oracle restatement of the reference FCN graph (encode_model + fcn32s/16s/8s heads).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, as text,
/root/reference/models/fcn.py:106-130 (encoder), :133-145 (32s), :148-176 (16s),
:179-220 (8s).  Reproduced on purpose: one 3x3 SAME conv per stage,
widths nk*{1,2,4,8,8}, conv6/conv7 1x1 of width 32*nk, and slim's default ReLU
kept on conv_fr / pool3_score / pool4_score.  The bilinear filters are
tf.constant (not trained) and use the dense channel-diagonal [k,k,C,C] bank of
utils/upsampling.py:27-46, SAME padding (tf.nn.conv2d_transpose default).
"""
import numpy as np
from . import np_ops as ops

ENC = ['conv1', 'conv2', 'conv3', 'conv4', 'conv5']


def layer_names(fcn_type='8s'):
    n = ENC + ['conv6', 'conv7', 'conv_fr']
    if fcn_type == '16s':
        n += ['pool4_score']
    elif fcn_type == '8s':
        n += ['pool3_score', 'pool4_score']
    return n


def layer_shapes(n_classes=2, n_kernels=32, input_channel=3, fcn_type='8s'):
    nk = n_kernels
    s = {}

    def conv(name, ci, co, k):
        s[name] = ((k, k, ci, co), k * k * ci, k * k * co)
    conv('conv1', input_channel, nk, 3); conv('conv2', nk, 2 * nk, 3); conv('conv3', 2 * nk, 4 * nk, 3)
    conv('conv4', 4 * nk, 8 * nk, 3); conv('conv5', 8 * nk, 8 * nk, 3)
    conv('conv6', 8 * nk, 32 * nk, 1); conv('conv7', 32 * nk, 32 * nk, 1); conv('conv_fr', 32 * nk, n_classes, 1)
    if fcn_type in ('8s',):
        conv('pool3_score', 4 * nk, n_classes, 1)
    if fcn_type in ('16s', '8s'):
        conv('pool4_score', 8 * nk, n_classes, 1)
    return s


def init_params(n_classes=2, n_kernels=32, input_channel=3, fcn_type='8s', seed=5555):
    rng = np.random.default_rng(seed)
    shapes = layer_shapes(n_classes, n_kernels, input_channel, fcn_type)
    p = {}
    for name in layer_names(fcn_type):
        shp, fi, fo = shapes[name]
        p[name] = {'weights': ops.xavier_uniform(shp, rng, fi, fo), 'biases': np.zeros((shp[3],), np.float32)}
    return p


def _up(t, factor, dt):
    C = t.shape[-1]
    return ops.conv2d_transpose(t, ops.bilinear_upsample_weights(factor, C), None, factor, 'SAME', False, dt)


def _up_bwd(d, factor, in_hw, dt):
    C = d.shape[-1]
    return ops.conv2d_transpose_dgrad(d, ops.bilinear_upsample_weights(factor, C), in_hw, factor, 'SAME', dt)


def forward(p, x, fcn_type='8s', dt=np.float64):
    c = {'x': np.asarray(x, dt)}
    W = lambda n: p[n]['weights']
    b = lambda n: p[n]['biases']
    t = c['x']
    for i, n in enumerate(ENC):
        c[n] = ops.conv2d(t, W(n), b(n), 'SAME', 1, True, dt)
        c['pool%d' % (i + 1)], c['idx%d' % (i + 1)] = ops.max_pool2x2(c[n])
        t = c['pool%d' % (i + 1)]
    c['conv6'] = ops.conv2d(t, W('conv6'), b('conv6'), 'SAME', 1, True, dt)
    c['conv7'] = ops.conv2d(c['conv6'], W('conv7'), b('conv7'), 'SAME', 1, True, dt)
    c['conv_fr'] = ops.conv2d(c['conv7'], W('conv_fr'), b('conv_fr'), 'SAME', 1, True, dt)
    H, Wd = c['x'].shape[1:3]
    if fcn_type == '32s':
        c['up_final'] = _up(c['conv_fr'], 32, dt)
    elif fcn_type == '16s':
        c['pool4_score'] = ops.conv2d(c['pool4'], W('pool4_score'), b('pool4_score'), 'SAME', 1, True, dt)
        c['up4'] = _up(c['conv_fr'], 2, dt)
        h4 = c['pool4_score'].shape[1]
        c['fuse4'] = c['pool4_score'] + ops.crop_or_pad(c['up4'], h4, h4)
        c['up_final'] = _up(c['fuse4'], 16, dt)
    else:
        c['pool3_score'] = ops.conv2d(c['pool3'], W('pool3_score'), b('pool3_score'), 'SAME', 1, True, dt)
        c['pool4_score'] = ops.conv2d(c['pool4'], W('pool4_score'), b('pool4_score'), 'SAME', 1, True, dt)
        c['up4'] = _up(c['conv_fr'], 2, dt)
        h4, w4 = c['pool4_score'].shape[1:3]
        c['fuse4'] = c['pool4_score'] + ops.crop_or_pad(c['up4'], h4, w4)
        c['up3'] = _up(c['fuse4'], 2, dt)
        h3, w3 = c['pool3_score'].shape[1:3]
        c['fuse3'] = c['pool3_score'] + ops.crop_or_pad(c['up3'], h3, w3)
        c['up_final'] = _up(c['fuse3'], 8, dt)
    c['logits'] = ops.crop_or_pad(c['up_final'], H, Wd)
    return c['logits'], c


def loss_and_grads(p, x, y, fcn_type='8s', dt=np.float64):
    logits, c = forward(p, x, fcn_type, dt)
    loss, _, dlog = ops.softmax_xent(logits, y, dt)
    g = {}
    W = lambda n: np.asarray(p[n]['weights'], dt)

    def conv_bwd(name, xin, yout, dy, need_dx=True):
        dz = dy * (yout > 0)                         # every FCN conv keeps the ReLU
        dw, db = ops.conv2d_wgrad(xin, dz, W(name).shape[:2], 'SAME', 1, dt)
        g[name] = {'weights': dw, 'biases': db}
        return ops.conv2d_dgrad(dz, W(name), xin.shape[1:3], 'SAME', 1, dt) if need_dx else None

    d = ops.crop_or_pad_bwd(dlog, c['up_final'].shape[1:3])
    d_pool3 = d_pool4 = 0.0
    if fcn_type == '32s':
        d = _up_bwd(d, 32, c['conv_fr'].shape[1:3], dt)
    elif fcn_type == '16s':
        d = _up_bwd(d, 16, c['fuse4'].shape[1:3], dt)
        d_pool4 = conv_bwd('pool4_score', c['pool4'], c['pool4_score'], d)
        d = _up_bwd(ops.crop_or_pad_bwd(d, c['up4'].shape[1:3]), 2, c['conv_fr'].shape[1:3], dt)
    else:
        d = _up_bwd(d, 8, c['fuse3'].shape[1:3], dt)
        d_pool3 = conv_bwd('pool3_score', c['pool3'], c['pool3_score'], d)
        d = _up_bwd(ops.crop_or_pad_bwd(d, c['up3'].shape[1:3]), 2, c['fuse4'].shape[1:3], dt)
        d_pool4 = conv_bwd('pool4_score', c['pool4'], c['pool4_score'], d)
        d = _up_bwd(ops.crop_or_pad_bwd(d, c['up4'].shape[1:3]), 2, c['conv_fr'].shape[1:3], dt)
    d = conv_bwd('conv_fr', c['conv7'], c['conv_fr'], d)
    d = conv_bwd('conv7', c['conv6'], c['conv7'], d)
    d = conv_bwd('conv6', c['pool5'], c['conv6'], d)
    for i in (5, 4, 3, 2, 1):
        n = 'conv%d' % i
        d = ops.max_pool2x2_bwd(d, c['idx%d' % i], c[n].shape[1:3])
        xin = c['pool%d' % (i - 1)] if i > 1 else c['x']
        d = conv_bwd(n, xin, c[n], d, need_dx=(i > 1))
        if i == 5:
            d = d + d_pool4
        if i == 4:
            d = d + d_pool3
    return loss, g, c


def infer(p, x, fcn_type='8s', dt=np.float64):
    logits, _ = forward(p, x, fcn_type, dt)
    sig, out = ops.sigmoid_argmax(logits.astype(np.float32))
    return [sig, out]
