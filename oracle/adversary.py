"""Oracle restatement of the reference's adversarial segmentation training (Luc et al. 2016).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, as text:
  adversary network      /root/reference/models/basemodel.py:215-262
  adversarial losses     /root/reference/models/basemodel.py:278-305
  training objectives    /root/reference/models/basemodel.py:323-355
slim defaults the file relies on: convolution2d / fully_connected apply bias + ReLU (adv_output: activation_fn=None),
batch_norm decay 0.999, epsilon 0.001, beta only, training mode, applied to the layer's (post-ReLU) output; max_pool2d VALID;
flatten in NHWC order; xavier weights, zero biases.

The reference branch does not run at HEAD (`_adversarial_net_fn` and `adversarial_lr` are never set on BaseModel,
`l_bce_fake_one` is a local, `minimize()` has no var_list, the adversary is applied with reuse=True to tensors of different
channel counts).  What is restated is therefore the INTENDED construction, with these build-defined resolutions
(parity unpinned, stated in DESIGN.md):
  * real input  = one_hot(labels) over the output window, float;  fake input = softmax(logits)   (Luc et al. section 3; the
    reference comment about "absolute values (1.0, vs 0.0 < x < 1.0)" describes exactly this pair);
  * seg optimizer updates the segmentation variables with d/d(seg) of  mean_pixels xent + lambda * mean_batch bce(a(fake), 1);
    adversary optimizer updates the adversary variables with d/d(adv) of mean_batch [bce(a(real), 1) + bce(a(fake), 0)]
    (variable lists separated as the reference's own GAN models do, models/gan.py:198-232);
  * one batch per train_step(): both gradients are taken at the same (pre-update) weights from one forward pass;
  * the batch-norm moving averages see the real pass, then the fake pass (two updates per step);
  * lambda = 2 (basemodel.py:279), adversarial_lr default 1e-5 ("a LOW learning rate", :325-328).
"""
import numpy as np
from . import np_ops as ops

N_KERNELS = 36          # basemodel.py:216
DADV = 4                # basemodel.py:217: the input is bilinearly down-sampled by 4
ADV_LAMBDA = 2.0        # basemodel.py:279
LAYERS = ['adv_conv1', 'adv_bn1', 'adv_conv2', 'adv_bn2', 'adv_bn3', 'adv_fc1', 'adv_bn4', 'adv_output']


def sizes(h, w):
    """(h, w) of the adversary input -> ladder {name: (h, w)}; raises when a map collapses."""
    s = {'in': (h, w), 'resize': (h // DADV, w // DADV)}
    cur = s['resize']
    for name, k, st in [('conv1', 3, 2), ('pool1', 2, 2), ('conv2', 3, 2), ('pool2', 2, 2)]:
        cur = tuple((n - k) // st + 1 if n >= k else 0 for n in cur)
        if min(cur) < 1:
            raise ValueError('adversary: a %dx%d input collapses at %s' % (h, w, name))
        s[name] = cur
    return s


def init_params(in_channels, h, w, seed=7777):
    rng = np.random.default_rng(seed)
    nk = N_KERNELS
    ph, pw = sizes(h, w)['pool2']
    F = ph * pw * 2 * nk
    p = {}

    def conv(name, shp, fi, fo):
        p[name] = {'weights': ops.xavier_uniform(shp, rng, fi, fo), 'biases': np.zeros((shp[-1],), np.float32)}

    def bn(name, c):
        p[name] = {'beta': np.zeros((c,), np.float32), 'moving_mean': np.zeros((c,), np.float32), 'moving_variance': np.ones((c,), np.float32)}
    conv('adv_conv1', (3, 3, in_channels, nk), 9 * in_channels, 9 * nk); bn('adv_bn1', nk)
    conv('adv_conv2', (3, 3, nk, 2 * nk), 9 * nk, 18 * nk); bn('adv_bn2', 2 * nk)
    bn('adv_bn3', F)
    conv('adv_fc1', (F, 1024), F, 1024); bn('adv_bn4', 1024)
    conv('adv_output', (1024, 2), 1024, 2)
    return p


def forward(p, x, dt=np.float64, moving=None):
    """x [B,h,w,C] -> (logits [B,2], cache, moving') in training mode.  moving = {bn: (mean, var)} to chain the two passes."""
    x = np.asarray(x, dt)
    c = {'x_hw': x.shape[1:3]}
    B = x.shape[0]
    mv = {} if moving is None else dict(moving)

    def bn(name, a):
        mm, mvv = mv.get(name, (p[name]['moving_mean'], p[name]['moving_variance']))
        y, cache, nm, nv = ops.batch_norm(a, p[name]['beta'], mm, mvv, True, dt=dt)
        c[name] = cache
        mv[name] = (nm, nv)
        return y
    r = ops.resize_bilinear(x, (x.shape[1] // DADV, x.shape[2] // DADV), dt); c['r'] = r
    a1 = ops.conv2d(r, p['adv_conv1']['weights'], p['adv_conv1']['biases'], 'VALID', 2, True, dt); c['a1'] = a1
    b1 = bn('adv_bn1', a1)
    p1, c['i1'] = ops.max_pool_k(b1, 2); c['b1_hw'] = b1.shape[1:3]; c['p1'] = p1
    a2 = ops.conv2d(p1, p['adv_conv2']['weights'], p['adv_conv2']['biases'], 'VALID', 2, True, dt); c['a2'] = a2
    b2 = bn('adv_bn2', a2)
    p2, c['i2'] = ops.max_pool_k(b2, 2); c['b2_hw'] = b2.shape[1:3]; c['p2_shape'] = p2.shape
    f = p2.reshape(B, 1, 1, -1)
    b3 = bn('adv_bn3', f); c['b3'] = b3
    h = np.maximum(b3.reshape(B, -1) @ np.asarray(p['adv_fc1']['weights'], dt) + np.asarray(p['adv_fc1']['biases'], dt), 0); c['h'] = h
    b4 = bn('adv_bn4', h.reshape(B, 1, 1, -1)); c['b4'] = b4
    logits = b4.reshape(B, -1) @ np.asarray(p['adv_output']['weights'], dt) + np.asarray(p['adv_output']['biases'], dt)
    return logits, c, mv


def backward(p, c, dlogits, dt=np.float64, need_dx=True):
    """gradient at the logits -> ({layer: {...}}, dx at the adversary input)."""
    g = {}
    B = dlogits.shape[0]
    W = lambda n: np.asarray(p[n]['weights'], dt)
    b4 = c['b4'].reshape(B, -1)
    g['adv_output'] = {'weights': b4.T @ dlogits, 'biases': dlogits.sum(0)}
    d = (dlogits @ W('adv_output').T).reshape(B, 1, 1, -1)
    d, db = ops.batch_norm_bwd(d, c['adv_bn4']); g['adv_bn4'] = {'beta': db}
    d = d.reshape(B, -1) * (c['h'] > 0)
    b3 = c['b3'].reshape(B, -1)
    g['adv_fc1'] = {'weights': b3.T @ d, 'biases': d.sum(0)}
    d = (d @ W('adv_fc1').T).reshape(B, 1, 1, -1)
    d, db = ops.batch_norm_bwd(d, c['adv_bn3']); g['adv_bn3'] = {'beta': db}
    d = d.reshape(c['p2_shape'])
    d = ops.max_pool_k_bwd(d, c['i2'], c['b2_hw'], 2)
    d, db = ops.batch_norm_bwd(d, c['adv_bn2']); g['adv_bn2'] = {'beta': db}
    d = d * (c['a2'] > 0)
    dw, dbias = ops.conv2d_wgrad(c['p1'], d, (3, 3), 'VALID', 2, dt); g['adv_conv2'] = {'weights': dw, 'biases': dbias}
    d = ops.conv2d_dgrad(d, W('adv_conv2'), c['p1'].shape[1:3], 'VALID', 2, dt)
    d = ops.max_pool_k_bwd(d, c['i1'], c['b1_hw'], 2)
    d, db = ops.batch_norm_bwd(d, c['adv_bn1']); g['adv_bn1'] = {'beta': db}
    d = d * (c['a1'] > 0)
    dw, dbias = ops.conv2d_wgrad(c['r'], d, (3, 3), 'VALID', 2, dt); g['adv_conv1'] = {'weights': dw, 'biases': dbias}
    if not need_dx:
        return g, None
    d = ops.conv2d_dgrad(d, W('adv_conv1'), c['r'].shape[1:3], 'VALID', 2, dt)
    return g, ops.resize_bilinear_bwd(d, c['x_hw'], dt)


def bce(logits, label, dt=np.float64):
    """softmax_cross_entropy_with_logits against one_hot(label) per row -> (per-row loss, d(mean loss)/dlogits)."""
    z = np.asarray(logits, dt)
    m = z.max(1, keepdims=True)
    e = np.exp(z - m); s = e.sum(1, keepdims=True)
    l = np.log(s[:, 0]) - (z[:, label] - m[:, 0])
    d = e / s
    d[:, label] -= 1.0
    return l, d / z.shape[0]


def softmax(z, dt=np.float64):
    z = np.asarray(z, dt)
    e = np.exp(z - z.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def softmax_bwd(prob, dprob):
    return prob * (dprob - (dprob * prob).sum(-1, keepdims=True))


def one_hot(y, n_classes, dt=np.float64):
    return (np.asarray(y)[..., 0][..., None] == np.arange(n_classes)).astype(dt)


def adversarial_terms(p_adv, seg_logits, y_win, n_classes, lam=ADV_LAMBDA, dt=np.float64):
    """Everything the adversary contributes to one train step.  seg_logits [B,h,w,C]; y_win [B,h,w,1] labels over the same
    window.  Returns dict: l_bce_real / l_bce_fake / l_bce_fake_one (batch means), adv_loss, d_seg_logits (= lambda *
    d mean(bce_fake_one) / d seg_logits, to be ADDED to the x-entropy gradient), adv_grads, moving (after both passes),
    logits_real, logits_fake."""
    real = one_hot(y_win, n_classes, dt)
    prob = softmax(seg_logits, dt)
    lr_, cr, mv = forward(p_adv, real, dt)
    lf_, cf, mv = forward(p_adv, prob, dt, moving=mv)
    l_real, d_real = bce(lr_, 1, dt)
    l_fake, d_fake = bce(lf_, 0, dt)
    l_one, d_one = bce(lf_, 1, dt)
    g_real, _ = backward(p_adv, cr, d_real, dt, need_dx=False)
    g_fake, _ = backward(p_adv, cf, d_fake, dt, need_dx=False)
    _, dprob = backward(p_adv, cf, d_one, dt, need_dx=True)
    adv_grads = {n: {k: g_real[n][k] + g_fake[n][k] for k in g_real[n]} for n in g_real}
    return {'l_bce_real': l_real.mean(), 'l_bce_fake': l_fake.mean(), 'l_bce_fake_one': l_one.mean(),
            'adv_loss': l_real.mean() + l_fake.mean(), 'd_seg_logits': lam * softmax_bwd(prob, dprob), 'adv_grads': adv_grads,
            'moving': mv, 'logits_real': lr_, 'logits_fake': lf_}


def trainable(p):
    out = []
    for n in LAYERS:
        for k in ('weights', 'biases', 'beta'):
            if k in p[n]:
                out.append((n, k))
    return out


def n_params(p):
    return int(sum(np.asarray(p[n][k]).size for n, k in trainable(p)))
