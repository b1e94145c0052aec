"""Oracle restatement of the reference U-Net graph, loss, backward and Adam step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, as text:
  topology            /root/reference/models/unet.py:109-175
  label crop          /root/reference/models/unet.py:71-72,171-174
  outputs             /root/reference/models/unet.py:75-79
  loss / optimizer    /root/reference/models/basemodel.py:59-70,357-369
  step contract       /root/reference/models/basemodel.py:477-489
Quirks reproduced on purpose: all-VALID padding; pool1 consumes
conv1_1 (``net``), conv1_2 only feeds the last skip; concat is skip-first.
"""
import numpy as np
from . import np_ops as ops

CONV_ORDER = ['conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv4_1', 'conv4_2',
              'conv5_1', 'conv5_2', 'upconv1', 'conv6_1', 'conv6_2', 'upconv2', 'conv7_1', 'conv7_2',
              'upconv3', 'conv8_1', 'conv8_2', 'upconv4', 'conv9_1', 'conv9_2', 'output']


def layer_shapes(n_classes=2, n_kernels=32, input_channel=3):
    """name -> (weight shape, fan_in, fan_out).  conv: HWIO; upconv: [2,2,Cout,Cin]."""
    nk = n_kernels
    s = {}

    def conv(name, ci, co, k=3):
        s[name] = ((k, k, ci, co), k * k * ci, k * k * co)

    def up(name, ci, co):
        # slim.conv2d_transpose creates [kh,kw,Cout,Cin]; xavier fans computed by TF from the
        # shape as (receptive*shape[-2], receptive*shape[-1]) = (4*Cout, 4*Cin); the limit is
        # symmetric in the two so the order is immaterial.
        s[name] = ((2, 2, co, ci), 4 * co, 4 * ci)

    conv('conv1_1', input_channel, nk); conv('conv1_2', nk, nk)
    conv('conv2_1', nk, 2 * nk); conv('conv2_2', 2 * nk, 2 * nk)
    conv('conv3_1', 2 * nk, 4 * nk); conv('conv3_2', 4 * nk, 4 * nk)
    conv('conv4_1', 4 * nk, 8 * nk); conv('conv4_2', 8 * nk, 8 * nk)
    conv('conv5_1', 8 * nk, 16 * nk); conv('conv5_2', 16 * nk, 16 * nk)
    up('upconv1', 16 * nk, 8 * nk); conv('conv6_1', 16 * nk, 8 * nk); conv('conv6_2', 8 * nk, 8 * nk)
    up('upconv2', 8 * nk, 4 * nk); conv('conv7_1', 8 * nk, 4 * nk); conv('conv7_2', 4 * nk, 4 * nk)
    up('upconv3', 4 * nk, 2 * nk); conv('conv8_1', 4 * nk, 2 * nk); conv('conv8_2', 2 * nk, 2 * nk)
    up('upconv4', 2 * nk, nk); conv('conv9_1', 2 * nk, nk); conv('conv9_2', nk, nk)
    conv('output', nk, n_classes, k=1)
    return s


def init_params(n_classes=2, n_kernels=32, input_channel=3, seed=5555):
    """xavier-uniform weights, zero biases (slim defaults), drawn in CONV_ORDER from one
    numpy Generator so that the HIP side and the oracle can be given identical values."""
    rng = np.random.default_rng(seed)
    shapes = layer_shapes(n_classes, n_kernels, input_channel)
    p = {}
    for name in CONV_ORDER:
        shp, fi, fo = shapes[name]
        co = shp[2] if name.startswith('upconv') else shp[3]
        p[name] = {'weights': ops.xavier_uniform(shp, rng, fi, fo), 'biases': np.zeros((co,), np.float32)}
    return p


def n_params(p):
    return int(sum(v['weights'].size + v['biases'].size for v in p.values()))


def output_size(n):
    """spatial size of the logits for a square input n (raises if infeasible: SURVEY F11)."""
    def cc(v):
        if v < 5:
            raise ValueError('U-Net all-VALID graph infeasible at input %d' % n)
        return v - 4
    v = cc(n) + 2                      # after conv1_1
    v = (v) // 2                       # pool1 of conv1_1
    for _ in range(3):
        v = cc(v); v //= 2
    v = cc(v)
    for _ in range(4):
        v = cc(v * 2)
    return v


MC_SITES = {'conv2_2': 2, 'conv5_2': 5, 'conv6_2': 10}      # site -> seed increment (build-defined, a19)


def forward(p, x, dt=np.float64, dropout=None):
    """Returns (logits, cache).  dropout = {'keep':, 'seed':, 'offset':} applies the build-defined MC-dropout masks
    (ops.dropout) after conv2_2, conv5_2 and conv6_2 with per-site seeds seed + MC_SITES[site]."""
    c = {}
    W = lambda n: p[n]['weights']
    b = lambda n: p[n]['biases']
    conv = lambda t, n, relu=True: ops.conv2d(t, W(n), b(n), 'VALID', 1, relu, dt)
    up = lambda t, n: ops.conv2d_transpose(t, W(n), b(n), 2, 'VALID', True, dt)
    c['x'] = np.asarray(x, dt)
    c['conv1_1'] = conv(c['x'], 'conv1_1')
    c['conv1_2'] = conv(c['conv1_1'], 'conv1_2')
    c['pool1'], c['idx1'] = ops.max_pool2x2(c['conv1_1'])          # F12: pool of conv1_1
    c['conv2_1'] = conv(c['pool1'], 'conv2_1')
    c['conv2_2'] = conv(c['conv2_1'], 'conv2_2')
    drop = (lambda t, site: t) if dropout is None else \
        (lambda t, site: ops.dropout(t, dropout['keep'], dropout['seed'] + MC_SITES[site], dropout['offset'], dt))
    c['conv2_2'] = drop(c['conv2_2'], 'conv2_2')
    c['pool2'], c['idx2'] = ops.max_pool2x2(c['conv2_2'])
    c['conv3_1'] = conv(c['pool2'], 'conv3_1')
    c['conv3_2'] = conv(c['conv3_1'], 'conv3_2')
    c['pool3'], c['idx3'] = ops.max_pool2x2(c['conv3_2'])
    c['conv4_1'] = conv(c['pool3'], 'conv4_1')
    c['conv4_2'] = conv(c['conv4_1'], 'conv4_2')
    c['pool4'], c['idx4'] = ops.max_pool2x2(c['conv4_2'])
    c['conv5_1'] = conv(c['pool4'], 'conv5_1')
    c['conv5_2'] = drop(conv(c['conv5_1'], 'conv5_2'), 'conv5_2')
    prev = c['conv5_2']
    for lvl, (upn, skip, ca, cb) in enumerate([('upconv1', 'conv4_2', 'conv6_1', 'conv6_2'),
                                               ('upconv2', 'conv3_2', 'conv7_1', 'conv7_2'),
                                               ('upconv3', 'conv2_2', 'conv8_1', 'conv8_2'),
                                               ('upconv4', 'conv1_2', 'conv9_1', 'conv9_2')]):
        c[upn] = up(prev, upn)
        t = c[upn].shape[1]
        crop = ops.crop_or_pad(c[skip], t, t)
        c['cat%d' % (lvl + 1)] = np.concatenate([crop, c[upn]], axis=-1)     # skip first
        c[ca] = conv(c['cat%d' % (lvl + 1)], ca)
        c[cb] = conv(c[ca], cb)
        if cb in MC_SITES:
            c[cb] = drop(c[cb], cb)
        prev = c[cb]
    c['logits'] = conv(prev, 'output', relu=False)
    return c['logits'], c


def crop_labels(y, target):
    """models/unet.py:171-174: resize_image_with_crop_or_pad(input_y, target, target)."""
    y = np.asarray(y)
    if y.ndim == 3:
        y = y[..., None]
    return ops.crop_or_pad(y, target, target)


def loss_and_grads(p, x, y, dt=np.float64):
    """Returns (loss, grads{name:{weights,biases}}, cache).  y: uint8 [B,H,W,1] full-size."""
    logits, c = forward(p, x, dt)
    yc = crop_labels(y, logits.shape[1])
    loss, _, dlog = ops.softmax_xent(logits, yc, dt)
    g = {}
    W = lambda n: np.asarray(p[n]['weights'], dt)

    def conv_bwd(name, xin, yout, dy, relu=True, need_dx=True):
        dz = dy * (yout > 0) if relu else dy
        dw, db = ops.conv2d_wgrad(xin, dz, W(name).shape[:2], 'VALID', 1, dt)
        g[name] = {'weights': dw, 'biases': db}
        return ops.conv2d_dgrad(dz, W(name), xin.shape[1:3], 'VALID', 1, dt) if need_dx else None

    def up_bwd(name, xin, yout, dy):
        dz = dy * (yout > 0)
        dw, db = ops.conv2d_transpose_wgrad(xin, dz, (2, 2), 2, 'VALID', dt)
        g[name] = {'weights': dw, 'biases': db}
        return ops.conv2d_transpose_dgrad(dz, W(name), xin.shape[1:3], 2, 'VALID', dt)

    d = conv_bwd('output', c['conv9_2'], c['logits'], dlog, relu=False)
    skip_grads = {}
    prev_names = ['conv5_2', 'conv6_2', 'conv7_2', 'conv8_2']
    levels = [('upconv1', 'conv4_2', 'conv6_1', 'conv6_2'), ('upconv2', 'conv3_2', 'conv7_1', 'conv7_2'),
              ('upconv3', 'conv2_2', 'conv8_1', 'conv8_2'), ('upconv4', 'conv1_2', 'conv9_1', 'conv9_2')]
    for lvl in (3, 2, 1, 0):
        upn, skip, ca, cb = levels[lvl]
        d = conv_bwd(cb, c[ca], c[cb], d)
        dcat = conv_bwd(ca, c['cat%d' % (lvl + 1)], c[ca], d)
        cs = c[skip].shape[-1]
        skip_grads[skip] = ops.crop_or_pad_bwd(dcat[..., :cs], c[skip].shape[1:3])
        d = up_bwd(upn, c[prev_names[lvl]], c[upn], dcat[..., cs:])
    # encoder
    d = conv_bwd('conv5_2', c['conv5_1'], c['conv5_2'], d)
    d = conv_bwd('conv5_1', c['pool4'], c['conv5_1'], d)
    d = ops.max_pool2x2_bwd(d, c['idx4'], c['conv4_2'].shape[1:3]) + skip_grads['conv4_2']
    d = conv_bwd('conv4_2', c['conv4_1'], c['conv4_2'], d)
    d = conv_bwd('conv4_1', c['pool3'], c['conv4_1'], d)
    d = ops.max_pool2x2_bwd(d, c['idx3'], c['conv3_2'].shape[1:3]) + skip_grads['conv3_2']
    d = conv_bwd('conv3_2', c['conv3_1'], c['conv3_2'], d)
    d = conv_bwd('conv3_1', c['pool2'], c['conv3_1'], d)
    d = ops.max_pool2x2_bwd(d, c['idx2'], c['conv2_2'].shape[1:3]) + skip_grads['conv2_2']
    d = conv_bwd('conv2_2', c['conv2_1'], c['conv2_2'], d)
    d = conv_bwd('conv2_1', c['pool1'], c['conv2_1'], d)
    d11 = ops.max_pool2x2_bwd(d, c['idx1'], c['conv1_1'].shape[1:3])
    d11 = d11 + conv_bwd('conv1_2', c['conv1_1'], c['conv1_2'], skip_grads['conv1_2'])
    conv_bwd('conv1_1', c['x'], c['conv1_1'], d11, need_dx=False)
    return loss, g, c


def init_opt_state(p):
    return {n: {k: np.zeros_like(v, dtype=np.float64) for k, v in t.items()} for n, t in p.items()}, \
           {n: {k: np.zeros_like(v, dtype=np.float64) for k, v in t.items()} for n, t in p.items()}


def train_step(p, m, v, step, x, y, lr=1e-4, dt=np.float64):
    """One intended BaseModel.train_step(): fwd + mean x-entropy + bwd + TF Adam; step is the
    1-based Adam timestep (= global_step after the increment).  Returns (loss, p', m', v')."""
    loss, g, _ = loss_and_grads(p, x, y, dt)
    p2, m2, v2 = {}, {}, {}
    for n in p:
        p2[n], m2[n], v2[n] = {}, {}, {}
        for k in ('weights', 'biases'):
            p2[n][k], m2[n][k], v2[n][k] = ops.adam_tf(p[n][k], g[n][k], m[n][k], v[n][k], step, lr, dt=dt)
    return loss, p2, m2, v2


def infer(p, x, dt=np.float64):
    """BaseModel.infer: returns [sigmoid(logits) f32, float32(argmax(sigmoid)) [B,h,w,1]]."""
    logits, _ = forward(p, x, dt)
    sig, out = ops.sigmoid_argmax(logits.astype(np.float32))
    return [sig, out]


def infer_mc(p, x, passes=30, keep=0.5, seed=5555, dt=np.float64):
    """Build-defined MC-dropout inference (a19): `passes` stochastic forwards, pass t (0-based) uses the counter offset
    (t + 1) << 40; returns [mean sigmoid, variance of sigmoid, float32 argmax of the mean, per-pass float32 sigmoids]."""
    sigs = []
    for t in range(passes):
        logits, _ = forward(p, x, dt, dropout={'keep': keep, 'seed': seed, 'offset': (t + 1) << 40})
        sig, _ = ops.sigmoid_argmax(logits.astype(np.float32))
        sigs.append(sig)
    s = np.stack(sigs).astype(np.float64)
    mean = s.mean(0)
    var = np.maximum((s * s).mean(0) - mean * mean, 0)
    return [mean, var, np.argmax(mean, -1).astype(np.float32)[..., None], sigs]
