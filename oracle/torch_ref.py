"""Independent second CPU implementation: torch-CPU functional ops + autograd.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Used (a) to cross-check the numpy
restatement (two independent implementations must agree before either is trusted,
SURVEY §8(c)), and (b) as the optimised multi-threaded stand-in for "the TF-CPU
path" in bench.py's cpu_baseline (kind "port"; TensorFlow itself is never run).

Graph per /root/reference/models/unet.py:109-175 and models/fcn.py:93-220.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _w(p, name, dtype):
    w = torch.as_tensor(np.asarray(p[name]['weights']), dtype=dtype)
    b = torch.as_tensor(np.asarray(p[name]['biases']), dtype=dtype)
    return w, b


def to_torch_params(p, dtype=torch.float64, requires_grad=True):
    tp = {}
    for n, t in p.items():
        w, b = _w(p, n, dtype)
        tp[n] = {'weights': w.clone().requires_grad_(requires_grad), 'biases': b.clone().requires_grad_(requires_grad)}
    return tp


def _conv(x, w_hwio, b, relu=True, same=False):
    # HWIO -> OIHW
    w = w_hwio.permute(3, 2, 0, 1)
    if same:
        k = w.shape[-1]
        x = F.pad(x, ((k - 1) // 2, k // 2, (k - 1) // 2, k // 2))
    y = F.conv2d(x, w, b)
    return F.relu(y) if relu else y


def _upconv(x, w_hwoi, b):
    # TF [kh,kw,Cout,Cin] -> torch conv_transpose2d weight [Cin, Cout, kh, kw]
    w = w_hwoi.permute(3, 2, 0, 1)
    return F.relu(F.conv_transpose2d(x, w, b, stride=2))


def _center_crop(t, target):
    h = t.shape[-1]
    o = (h - target) // 2
    return t[..., o:o + target, o:o + target]


def unet_forward(tp, x_nhwc):
    """x: torch [B,H,W,C]; returns logits NHWC."""
    x = x_nhwc.permute(0, 3, 1, 2)
    P = lambda n: (tp[n]['weights'], tp[n]['biases'])
    n11 = _conv(x, *P('conv1_1'))
    n12 = _conv(n11, *P('conv1_2'))
    t = F.max_pool2d(n11, 2)                      # F12
    t = _conv(t, *P('conv2_1')); n22 = _conv(t, *P('conv2_2'))
    t = F.max_pool2d(n22, 2)
    t = _conv(t, *P('conv3_1')); n32 = _conv(t, *P('conv3_2'))
    t = F.max_pool2d(n32, 2)
    t = _conv(t, *P('conv4_1')); n42 = _conv(t, *P('conv4_2'))
    t = F.max_pool2d(n42, 2)
    t = _conv(t, *P('conv5_1')); t = _conv(t, *P('conv5_2'))
    for upn, skip, ca, cb in [('upconv1', n42, 'conv6_1', 'conv6_2'), ('upconv2', n32, 'conv7_1', 'conv7_2'),
                              ('upconv3', n22, 'conv8_1', 'conv8_2'), ('upconv4', n12, 'conv9_1', 'conv9_2')]:
        u = _upconv(t, *P(upn))
        t = torch.cat([_center_crop(skip, u.shape[-1]), u], dim=1)
        t = _conv(t, *P(ca)); t = _conv(t, *P(cb))
    y = _conv(t, *P('output'), relu=False)
    return y.permute(0, 2, 3, 1)


def xent_mean(logits_nhwc, labels):
    """mean over pixels of softmax x-entropy against integer labels [B,h,w]."""
    C = logits_nhwc.shape[-1]
    return F.cross_entropy(logits_nhwc.reshape(-1, C), labels.reshape(-1).long(), reduction='mean')


def unet_loss_and_grads(p, x, y_full, dtype=torch.float64):
    tp = to_torch_params(p, dtype)
    xt = torch.as_tensor(np.asarray(x), dtype=dtype)
    logits = unet_forward(tp, xt)
    t = logits.shape[1]
    y = np.asarray(y_full).reshape(np.asarray(y_full).shape[:3])
    o = (y.shape[1] - t) // 2
    yc = torch.as_tensor(y[:, o:o + t, o:o + t].astype(np.int64))
    loss = xent_mean(logits, yc)
    loss.backward()
    g = {n: {k: tp[n][k].grad.numpy() for k in ('weights', 'biases')} for n in tp}
    return float(loss.detach()), g, logits.detach().numpy()


def adam_tf_(params, grads, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """In-place TF-style Adam on lists of tensors (eps outside the bias correction)."""
    lr_t = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
    for p, g, mm, vv in zip(params, grads, m, v):
        mm.mul_(b1).add_(g, alpha=1 - b1)
        vv.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.sub_(lr_t * mm / (vv.sqrt() + eps))


class TorchUNetStepper:
    """float32, multi-threaded train-step used as the timed CPU baseline (bench.py)."""

    def __init__(self, p, lr=1e-4, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.tp = to_torch_params(p, torch.float32)
        self.flat = [t[k] for t in self.tp.values() for k in ('weights', 'biases')]
        self.m = [torch.zeros_like(t) for t in self.flat]
        self.v = [torch.zeros_like(t) for t in self.flat]
        self.step = 0
        self.lr = lr

    def train_step(self, x, y_full):
        xt = torch.as_tensor(x, dtype=torch.float32)
        logits = unet_forward(self.tp, xt)
        t = logits.shape[1]
        y = np.asarray(y_full).reshape(np.asarray(y_full).shape[:3])
        o = (y.shape[1] - t) // 2
        yc = torch.as_tensor(y[:, o:o + t, o:o + t].astype(np.int64))
        loss = xent_mean(logits, yc)
        grads = torch.autograd.grad(loss, self.flat)
        self.step += 1
        with torch.no_grad():
            adam_tf_(self.flat, grads, self.m, self.v, self.step, self.lr)
        return float(loss)


# ------------------------------- FCN ---------------------------------------
def _bilinear_up(x, factor):
    """tf.nn.conv2d_transpose(x, bilinear_upsample_weights(factor, C), SAME) as a depthwise
    transposed conv (the filter bank is channel-diagonal)."""
    C = x.shape[1]
    k = 2 * factor - factor % 2
    f = (k + 1) // 2
    c = f - 1 if k % 2 == 1 else f - 0.5
    og = torch.arange(k, dtype=x.dtype)
    tent = 1 - (og - c).abs() / f
    w = (tent[:, None] * tent[None, :]).to(torch.float32).to(x.dtype)     # float32 cast as the reference
    w = w[None, None].repeat(C, 1, 1, 1)
    total = max(k - factor, 0)       # (in-1)*s + k - in*s
    pb = total // 2
    full = F.conv_transpose2d(x, w, stride=factor, groups=C)
    n = x.shape[-1] * factor
    return full[..., pb:pb + n, pb:pb + n]


def _crop_or_pad(t, th, tw):
    h, w = t.shape[-2:]
    if h >= th:
        o = (h - th) // 2; t = t[..., o:o + th, :]
    else:
        a = (th - h) // 2; t = F.pad(t, (0, 0, a, th - h - a))
    if w >= tw:
        o = (w - tw) // 2; t = t[..., o:o + tw]
    else:
        a = (tw - w) // 2; t = F.pad(t, (a, tw - w - a))
    return t


def fcn_forward(tp, x_nhwc, fcn_type='8s'):
    x = x_nhwc.permute(0, 3, 1, 2)
    H, W = x.shape[-2:]
    P = lambda n: (tp[n]['weights'], tp[n]['biases'])
    t = F.max_pool2d(_conv(x, *P('conv1'), same=True), 2)
    t = F.max_pool2d(_conv(t, *P('conv2'), same=True), 2)
    pool3 = F.max_pool2d(_conv(t, *P('conv3'), same=True), 2)
    pool4 = F.max_pool2d(_conv(pool3, *P('conv4'), same=True), 2)
    pool5 = F.max_pool2d(_conv(pool4, *P('conv5'), same=True), 2)
    t = _conv(pool5, *P('conv6')); t = _conv(t, *P('conv7')); t = _conv(t, *P('conv_fr'))   # ReLU kept (F14)
    if fcn_type == '32s':
        up = _bilinear_up(t, 32)
    elif fcn_type == '16s':
        s4 = _conv(pool4, *P('pool4_score'))
        up = _bilinear_up(t, 2)
        up = s4 + _crop_or_pad(up, s4.shape[-2], s4.shape[-2])      # pool4_h twice (reference typo :166)
        up = _bilinear_up(up, 16)
    else:
        s3 = _conv(pool3, *P('pool3_score'))
        s4 = _conv(pool4, *P('pool4_score'))
        up = _bilinear_up(t, 2)
        up = s4 + _crop_or_pad(up, s4.shape[-2], s4.shape[-1])
        up = _bilinear_up(up, 2)
        up = s3 + _crop_or_pad(up, s3.shape[-2], s3.shape[-1])
        up = _bilinear_up(up, 8)
    up = _crop_or_pad(up, H, W)
    return up.permute(0, 2, 3, 1)


def fcn_loss_and_grads(p, x, y, fcn_type='8s', dtype=torch.float64):
    tp = to_torch_params(p, dtype)
    xt = torch.as_tensor(np.asarray(x), dtype=dtype)
    logits = fcn_forward(tp, xt, fcn_type)
    yy = np.asarray(y).reshape(np.asarray(y).shape[:3]).astype(np.int64)
    loss = xent_mean(logits, torch.as_tensor(yy))
    loss.backward()
    g = {n: {k: (tp[n][k].grad.numpy() if tp[n][k].grad is not None else np.zeros_like(np.asarray(p[n][k]), dtype=np.float64))
             for k in ('weights', 'biases')} for n in tp}
    return float(loss.detach()), g, logits.detach().numpy()


# ---------------------------------------------------------------------------------------------------------------------
# DeconvModel (models/deconvolution.py:101-178): independent torch-autograd composition of the graph in oracle/deconv.py
# ---------------------------------------------------------------------------------------------------------------------
def _conv_s(x, w_hwio, b, stride, same, relu=True):
    """NHWC conv with TF padding rules (SAME pads more at the bottom / right when the total is odd)"""
    k = w_hwio.shape[0]
    xc = x.permute(0, 3, 1, 2)
    if same:
        H, W = xc.shape[2:]
        oh, ow = -(-H // stride), -(-W // stride)
        ph, pw = max((oh - 1) * stride + k - H, 0), max((ow - 1) * stride + k - W, 0)
        xc = F.pad(xc, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    y = F.conv2d(xc, w_hwio.permute(3, 2, 0, 1), b, stride=stride)
    y = y.permute(0, 2, 3, 1)
    return torch.relu(y) if relu else y


def _deconv_s(x, w_hwoi, b, stride):
    """slim.convolution2d_transpose VALID + bias + ReLU; filter [kh,kw,Cout,Cin]"""
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w_hwoi.permute(3, 2, 0, 1), b, stride=stride)
    return torch.relu(y.permute(0, 2, 3, 1))


def _bn_train(a, beta, eps=1e-3):
    mean = a.mean((0, 1, 2)); var = a.var((0, 1, 2), unbiased=False)
    return (a - mean) / torch.sqrt(var + eps) + beta, mean, var


def _resize_tf(x, out_hw):
    """tf.image.resize_bilinear, align_corners=False (source = dst * in/out, no half-pixel shift), gather formulation"""
    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        f = (np.arange(n_out, dtype=np.float32) * scale).astype(np.float32)
        lo = np.minimum(np.floor(f).astype(np.int64), n_in - 1); hi = np.minimum(lo + 1, n_in - 1)
        return torch.from_numpy(lo), torch.from_numpy(hi), torch.from_numpy((f - np.floor(f)).astype(np.float64)).to(x.dtype)
    y0, y1, ly = axis(x.shape[1], out_hw[0]); x0, x1, lx = axis(x.shape[2], out_hw[1])
    r0, r1 = x.index_select(1, y0), x.index_select(1, y1)
    lx_ = lx[None, None, :, None]; ly_ = ly[None, :, None, None]
    top = r0.index_select(2, x0) * (1 - lx_) + r0.index_select(2, x1) * lx_
    bot = r1.index_select(2, x0) * (1 - lx_) + r1.index_select(2, x1) * lx_
    return top * (1 - ly_) + bot * ly_


def deconv_forward(tp, x, masks=None, keep=0.5):
    """tp: {layer: {weights, biases}, bn: {beta}} torch tensors; masks: {bn: bool ndarray} dropout keep-masks (bayesian)"""
    H, W = x.shape[1:3]
    stats = {}

    def bn(name, a):
        y, mean, var = _bn_train(a, tp[name]['beta'])
        stats[name] = (mean.detach(), var.detach())
        if masks is not None and name in masks:
            y = y * torch.from_numpy(masks[name]).to(y.dtype) * float(np.float32(1.0) / np.float32(keep))
        return y
    net = bn('bn1', _conv_s(x, tp['conv1_0']['weights'], tp['conv1_0']['biases'], 2, True))
    net = F.max_pool2d(net.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    net = bn('bn2', _conv_s(net, tp['conv2_0']['weights'], tp['conv2_0']['biases'], 1, False))
    net = F.max_pool2d(net.permute(0, 3, 1, 2), 3, 3).permute(0, 2, 3, 1)
    net = bn('bn3', _conv_s(net, tp['conv3_0']['weights'], tp['conv3_0']['biases'], 1, False))
    net = F.max_pool2d(net.permute(0, 3, 1, 2), 3, 3).permute(0, 2, 3, 1)
    net = bn('bn4', _conv_s(net, tp['conv4_0']['weights'], tp['conv4_0']['biases'], 1, False))
    net = bn('bn5', _deconv_s(net, tp['deconv1_0']['weights'], tp['deconv1_0']['biases'], 2))
    net = bn('bn6', _deconv_s(net, tp['deconv2_0']['weights'], tp['deconv2_0']['biases'], 2))
    net = bn('bn7', _deconv_s(net, tp['deconv2_1']['weights'], tp['deconv2_1']['biases'], 2))
    net = _resize_tf(net, (H // 2, W // 2))
    net = bn('bn8', _deconv_s(net, tp['deconv3_0']['weights'], tp['deconv3_0']['biases'], 2))
    net = _crop_or_pad(net.permute(0, 3, 1, 2), H, W).permute(0, 2, 3, 1)      # (that helper is NCHW)
    return _conv_s(net, tp['conv_out']['weights'], tp['conv_out']['biases'], 1, True, relu=False), stats


def deconv_loss_and_grads(p, x, y, masks=None, keep=0.5, dtype=torch.float64):
    tp = {}
    for n, t in p.items():
        tp[n] = {k: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True) for k, a in t.items() if k in ('weights', 'biases', 'beta')}
    logits, stats = deconv_forward(tp, torch.as_tensor(np.asarray(x), dtype=dtype), masks, keep)
    loss = xent_mean(logits, torch.as_tensor(np.asarray(y)[..., 0].astype(np.int64)))
    loss.backward()
    g = {n: {k: a.grad.numpy() for k, a in t.items()} for n, t in tp.items()}
    return float(loss.detach()), g, logits.detach().numpy(), {n: (m.numpy(), v.numpy()) for n, (m, v) in stats.items()}


# Adversarial training terms (models/basemodel.py:215-355): independent torch-autograd composition of oracle/adversary.py
# ---------------------------------------------------------------------------------------------------------------------
def adversary_forward(tp, x):
    """tp: {layer: {weights, biases | beta}} torch tensors; x [B,h,w,C] -> logits [B,2] (training-mode batch norms)"""
    B = x.shape[0]
    net = _resize_tf(x, (x.shape[1] // 4, x.shape[2] // 4))
    net = _bn_train(_conv_s(net, tp['adv_conv1']['weights'], tp['adv_conv1']['biases'], 2, False), tp['adv_bn1']['beta'])[0]
    net = F.max_pool2d(net.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    net = _bn_train(_conv_s(net, tp['adv_conv2']['weights'], tp['adv_conv2']['biases'], 2, False), tp['adv_bn2']['beta'])[0]
    net = F.max_pool2d(net.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    f = net.reshape(B, 1, 1, -1)
    f = _bn_train(f, tp['adv_bn3']['beta'])[0].reshape(B, -1)
    h = torch.relu(f @ tp['adv_fc1']['weights'] + tp['adv_fc1']['biases'])
    h = _bn_train(h.reshape(B, 1, 1, -1), tp['adv_bn4']['beta'])[0].reshape(B, -1)
    return h @ tp['adv_output']['weights'] + tp['adv_output']['biases']


def adversarial_terms(p_adv, seg_logits, y_win, n_classes, lam=2.0, dtype=torch.float64):
    tp = {n: {k: torch.tensor(np.asarray(a), dtype=dtype, requires_grad=True) for k, a in t.items() if k in ('weights', 'biases', 'beta')}
          for n, t in p_adv.items()}
    z = torch.tensor(np.asarray(seg_logits), dtype=dtype, requires_grad=True)
    real = F.one_hot(torch.as_tensor(np.asarray(y_win)[..., 0].astype(np.int64)), n_classes).to(dtype)
    fake = torch.softmax(z, -1)
    lr_, lf_ = adversary_forward(tp, real), adversary_forward(tp, fake)
    B = z.shape[0]
    ones, zeros = torch.ones(B, dtype=torch.int64), torch.zeros(B, dtype=torch.int64)
    l_real, l_fake, l_one = F.cross_entropy(lr_, ones), F.cross_entropy(lf_, zeros), F.cross_entropy(lf_, ones)
    flat = [a for t in tp.values() for a in t.values()]
    ga = torch.autograd.grad(l_real + l_fake, flat, retain_graph=True)
    gz = torch.autograd.grad(l_one, z)[0]
    it = iter(ga)
    adv_grads = {n: {k: next(it).numpy() for k in t} for n, t in tp.items()}
    return {'l_bce_real': float(l_real.detach()), 'l_bce_fake': float(l_fake.detach()), 'l_bce_fake_one': float(l_one.detach()), 'd_seg_logits': lam * gz.numpy(),
            'adv_grads': adv_grads, 'logits_real': lr_.detach().numpy(), 'logits_fake': lf_.detach().numpy()}
