"""ctypes binding of oracle/seg_cpu.c (libseg_cpu.so) and the U-Net train step composed of its ops.

TEST INFRASTRUCTURE (see oracle/__init__.py): the plain-C restatement named by SURVEY 8(b),(d)(i).  tests/test_oracle.py checks
every op and a whole train step against the numpy restatement; bench.py's `cpu_baseline` leg times CUNetStepper on the host
cores of the GPU box (kind "port").  The product (segmentation_amd/) never imports this module.

Graph followed: /root/reference/models/unet.py:109-175 (all-VALID U-Net, pool1 of conv1_1, skip-first concat), loss
models/basemodel.py:59-70,360, TF-Adam models/basemodel.py:321 -- the same restatement as oracle/unet.py, in float32."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'seg_cpu.c')
LIB = os.path.join(HERE, '_build', 'libseg_cpu.so')
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')
i64, i32, vp = C.c_int64, C.c_int, C.c_void_p


def build(force=False):
    """gcc -O3 -mavx2 -mfma -fopenmp (no -march=native: the library is built in one container and run on another host)"""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    tmp = LIB + '.tmp.%d' % os.getpid()
    cmd = ['gcc', '-O3', '-mavx2', '-mfma', '-fopenmp', '-fPIC', '-shared', '-std=c99', '-Wall', '-o', tmp, SRC, '-lm']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('gcc failed on seg_cpu.c:\n' + r.stderr[-3000:])
    os.replace(tmp, LIB)
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.segcpu_conv2d_fwd.argtypes = [vp, i64, i64, i64, f32p, vp, f32p] + [i32] * 12
        lib.segcpu_conv2d_dgrad.argtypes = [f32p, f32p, f32p] + [i32] * 10
        lib.segcpu_conv2d_wgrad.argtypes = [vp, i64, i64, i64, f32p, f32p, vp] + [i32] * 11
        lib.segcpu_convT2x2_fwd.argtypes = [f32p, f32p, vp, f32p] + [i32] * 6
        lib.segcpu_convT2x2_bwd.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p] + [i32] * 5
        lib.segcpu_maxpool2x2_fwd.argtypes = [f32p, f32p, u8p] + [i32] * 4
        lib.segcpu_maxpool2x2_bwd.argtypes = [f32p, u8p, f32p] + [i32] * 4
        lib.segcpu_softmax_xent.argtypes = [f32p, u8p, f32p, i64, i32]; lib.segcpu_softmax_xent.restype = C.c_double
        lib.segcpu_adam.argtypes = [f32p, f32p, f32p, f32p, i64, i32, C.c_float, C.c_float, C.c_float, C.c_float]
        lib.segcpu_relu_grad.argtypes = [f32p, f32p, f32p, i64]
        lib.segcpu_set_threads.argtypes = [i32]
        lib.segcpu_threads.restype = i32
        lib.segcpu_version.restype = i32
        _lib = lib
    return _lib


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def conv2d(x, w, b=None, padding='VALID', stride=1, relu=True, window=None):
    """x [B,H,W,Cin] float32; window = (oy, ox, h, w): convolve only that window of x (crop by view, no copy)"""
    lib = load()
    x = _f(x); w = _f(w)
    B, Hb, Wb, Cin = x.shape
    oy, ox, H, W = window if window is not None else (0, 0, Hb, Wb)
    k, _, _, Cout = w.shape
    if padding == 'VALID':
        Ho, Wo, pt, pl = (H - k) // stride + 1, (W - k) // stride + 1, 0, 0
    else:
        Ho, Wo = -(-H // stride), -(-W // stride)
        pt = max((Ho - 1) * stride + k - H, 0) // 2; pl = max((Wo - 1) * stride + k - W, 0) // 2
    y = np.empty((B, Ho, Wo, Cout), np.float32)
    base = x.ctypes.data + 4 * ((oy * Wb + ox) * Cin)
    bias = _f(b) if b is not None else None
    lib.segcpu_conv2d_fwd(base, Hb * Wb * Cin, Wb * Cin, Cin, w, bias.ctypes.data if bias is not None else None, y, B, H, W, Cin, Cout, k, stride,
                          pt, pl, Ho, Wo, 1 if relu else 0)
    return y


def conv2d_dgrad(dz, w, in_hw, padding='VALID', stride=1):
    lib = load()
    dz = _f(dz); w = _f(w)
    B, Ho, Wo, Cout = dz.shape
    k, _, Cin, _ = w.shape
    H, W = in_hw
    pt = pl = 0
    if padding != 'VALID':
        pt = max((Ho - 1) * stride + k - H, 0) // 2; pl = max((Wo - 1) * stride + k - W, 0) // 2
    dx = np.empty((B, H, W, Cin), np.float32)
    lib.segcpu_conv2d_dgrad(dz, w, dx, B, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo)
    return dx


def conv2d_wgrad(x, dz, k, padding='VALID', stride=1, window=None):
    lib = load()
    x = _f(x); dz = _f(dz)
    B, Hb, Wb, Cin = x.shape
    oy, ox, H, W = window if window is not None else (0, 0, Hb, Wb)
    _, Ho, Wo, Cout = dz.shape
    pt = pl = 0
    if padding != 'VALID':
        pt = max((Ho - 1) * stride + k - H, 0) // 2; pl = max((Wo - 1) * stride + k - W, 0) // 2
    dw = np.empty((k, k, Cin, Cout), np.float32); db = np.empty((Cout,), np.float32)
    base = x.ctypes.data + 4 * ((oy * Wb + ox) * Cin)
    lib.segcpu_conv2d_wgrad(base, Hb * Wb * Cin, Wb * Cin, Cin, dz, dw, db.ctypes.data, B, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo)
    return dw, db


def convT2x2(x, w, b, relu=True):
    lib = load()
    x = _f(x); w = _f(w)
    B, H, W, Cin = x.shape
    Cout = w.shape[2]
    y = np.empty((B, 2 * H, 2 * W, Cout), np.float32)
    bias = _f(b)
    lib.segcpu_convT2x2_fwd(x, w, bias.ctypes.data, y, B, H, W, Cin, Cout, 1 if relu else 0)
    return y


def convT2x2_bwd(x, w, dz):
    lib = load()
    x = _f(x); w = _f(w); dz = _f(dz)
    B, H, W, Cin = x.shape
    Cout = w.shape[2]
    dx = np.empty_like(x); dw = np.empty_like(w); db = np.empty((Cout,), np.float32)
    lib.segcpu_convT2x2_bwd(x, w, dz, dx, dw, db, B, H, W, Cin, Cout)
    return dx, dw, db


def maxpool2x2(x):
    lib = load()
    x = _f(x)
    B, H, W, Cc = x.shape
    y = np.empty((B, H // 2, W // 2, Cc), np.float32); idx = np.empty(y.shape, np.uint8)
    lib.segcpu_maxpool2x2_fwd(x, y, idx, B, H, W, Cc)
    return y, idx


def maxpool2x2_bwd(dy, idx, in_hw):
    lib = load()
    dy = _f(dy)
    B, Ho, Wo, Cc = dy.shape
    dx = np.empty((B, in_hw[0], in_hw[1], Cc), np.float32)
    lib.segcpu_maxpool2x2_bwd(dy, np.ascontiguousarray(idx, np.uint8), dx, B, in_hw[0], in_hw[1], Cc)
    return dx


def softmax_xent(z, labels):
    lib = load()
    z = _f(z)
    lab = np.ascontiguousarray(np.asarray(labels).reshape(z.shape[:-1]), np.uint8)
    dz = np.empty_like(z)
    loss = lib.segcpu_softmax_xent(z, lab, dz, int(np.prod(z.shape[:-1])), z.shape[-1])
    return float(loss), dz


def relu_grad(dy, y):
    lib = load()
    dy = _f(dy); y = _f(y)
    dz = np.empty_like(dy)
    lib.segcpu_relu_grad(dy, y, dz, dy.size)
    return dz


LEVELS = [('upconv1', 'conv4_2', 'conv6_1', 'conv6_2'), ('upconv2', 'conv3_2', 'conv7_1', 'conv7_2'),
          ('upconv3', 'conv2_2', 'conv8_1', 'conv8_2'), ('upconv4', 'conv1_2', 'conv9_1', 'conv9_2')]


class CUNetStepper(object):
    """fwd + mean x-entropy + bwd + TF-Adam of the reference U-Net on libseg_cpu.so (float32).  conv1_2 is evaluated on the window
    that survives the last skip crop, like the HIP path (bit-identical to the dense evaluation followed by the crop)."""

    def __init__(self, p, lr=1e-4, threads=None):
        lib = load()
        if threads:
            lib.segcpu_set_threads(int(threads))
        self.threads = lib.segcpu_threads()
        self.p = {n: {k: _f(v).copy() for k, v in t.items()} for n, t in p.items()}
        self.m = {n: {k: np.zeros_like(v) for k, v in t.items()} for n, t in self.p.items()}
        self.v = {n: {k: np.zeros_like(v) for k, v in t.items()} for n, t in self.p.items()}
        self.lr, self.t = lr, 0

    def loss_and_grads(self, x, y):
        p = self.p
        W_ = lambda n: p[n]['weights']
        b_ = lambda n: p[n]['biases']
        c = {'x': _f(x)}
        conv = lambda t, n, relu=True, window=None: conv2d(t, W_(n), b_(n), 'VALID', 1, relu, window)
        c['conv1_1'] = conv(c['x'], 'conv1_1')
        c['pool1'], c['idx1'] = maxpool2x2(c['conv1_1'])
        prev = c['pool1']
        for i in (2, 3, 4, 5):
            c['conv%d_1' % i] = conv(prev, 'conv%d_1' % i)
            c['conv%d_2' % i] = conv(c['conv%d_1' % i], 'conv%d_2' % i)
            if i < 5:
                c['pool%d' % i], c['idx%d' % i] = maxpool2x2(c['conv%d_2' % i])
                prev = c['pool%d' % i]
        prev = c['conv5_2']
        offs = {}
        for lvl, (upn, skip, ca, cb) in enumerate(LEVELS):
            c[upn] = convT2x2(prev, W_(upn), b_(upn), True)
            t = c[upn].shape[1]
            if skip == 'conv1_2':
                o = (c['conv1_1'].shape[1] - 2 - t) // 2                 # offset of the surviving window inside conv1_2's output
                c['conv1_2'] = conv(c['conv1_1'], 'conv1_2', window=(o, o, t + 2, t + 2))
                crop = c['conv1_2']; offs[skip] = o
            else:
                o = (c[skip].shape[1] - t) // 2; offs[skip] = o
                crop = c[skip][:, o:o + t, o:o + t, :]
            c['cat%d' % lvl] = np.ascontiguousarray(np.concatenate([crop, c[upn]], axis=-1))
            c[ca] = conv(c['cat%d' % lvl], ca)
            c[cb] = conv(c[ca], cb)
            prev = c[cb]
        c['logits'] = conv(prev, 'output', relu=False)
        oh = c['logits'].shape[1]
        H = c['x'].shape[1]
        lo = (H - oh) // 2
        yc = np.asarray(y).reshape(c['x'].shape[:3])[:, lo:lo + oh, lo:lo + oh]
        loss, d = softmax_xent(c['logits'], yc)
        g = {}

        def conv_bwd(name, xin, yout, dy, relu=True, need_dx=True, window=None):
            dz = relu_grad(dy, yout) if relu else _f(dy)
            dw, db = conv2d_wgrad(xin, dz, W_(name).shape[0], 'VALID', 1, window)
            g[name] = {'weights': dw, 'biases': db}
            hw = xin.shape[1:3] if window is None else window[2:]
            return conv2d_dgrad(dz, W_(name), hw) if need_dx else None

        d = conv_bwd('output', c['conv9_2'], c['logits'], d, relu=False)
        prevs = ['conv5_2', 'conv6_2', 'conv7_2', 'conv8_2']
        dskip = {}
        for lvl in (3, 2, 1, 0):
            upn, skip, ca, cb = LEVELS[lvl]
            d = conv_bwd(cb, c[ca], c[cb], d)
            dcat = conv_bwd(ca, c['cat%d' % lvl], c[ca], d)
            cs = c[skip].shape[-1]
            dskip[skip] = np.ascontiguousarray(dcat[..., :cs])
            dz = relu_grad(np.ascontiguousarray(dcat[..., cs:]), c[upn])
            d, dw, db = convT2x2_bwd(c[prevs[lvl]], W_(upn), dz)
            g[upn] = {'weights': dw, 'biases': db}

        def pad_skip(name):
            o, t = offs[name], dskip[name].shape[1]
            full = np.zeros_like(c[name]); full[:, o:o + t, o:o + t, :] = dskip[name]
            return full
        d = conv_bwd('conv5_2', c['conv5_1'], c['conv5_2'], d)
        d = conv_bwd('conv5_1', c['pool4'], c['conv5_1'], d)
        for i in (4, 3, 2):
            d = maxpool2x2_bwd(d, c['idx%d' % i], c['conv%d_2' % i].shape[1:3]) + pad_skip('conv%d_2' % i)
            d = conv_bwd('conv%d_2' % i, c['conv%d_1' % i], c['conv%d_2' % i], d)
            d = conv_bwd('conv%d_1' % i, c['pool%d' % (i - 1)], c['conv%d_1' % i], d)
        d11 = maxpool2x2_bwd(d, c['idx1'], c['conv1_1'].shape[1:3])
        o, t = offs['conv1_2'], dskip['conv1_2'].shape[1]
        dwin = conv_bwd('conv1_2', c['conv1_1'], c['conv1_2'], dskip['conv1_2'], window=(o, o, t + 2, t + 2))
        d11[:, o:o + t + 2, o:o + t + 2, :] += dwin
        conv_bwd('conv1_1', c['x'], c['conv1_1'], d11, need_dx=False)
        return loss, g, c

    def train_step(self, x, y):
        loss, g, _ = self.loss_and_grads(x, y)
        lib = load()
        self.t += 1
        for n in self.p:
            for k in ('weights', 'biases'):
                lib.segcpu_adam(self.p[n][k].reshape(-1), _f(g[n][k]).reshape(-1), self.m[n][k].reshape(-1), self.v[n][k].reshape(-1),
                                self.p[n][k].size, self.t, self.lr, 0.9, 0.999, 1e-8)
        return loss
