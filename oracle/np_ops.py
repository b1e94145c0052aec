"""Numpy restatement of the TF-1.x / tf.contrib.slim ops the reference graphs invoke.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Layouts follow TF: activations
NHWC, conv filters HWIO ``[kh,kw,Cin,Cout]``, transposed-conv filters
``[kh,kw,Cout,Cin]``; cross-correlation (no kernel flip).  Every function takes
``dt`` = the arithmetic dtype (float64 for parity checks, float32 for the timed
CPU baseline).

Reference call sites restated here (paths relative to /root/reference):
  slim.convolution2d            models/unet.py:111-166, models/fcn.py:110-128,192,195
  slim.max_pool2d(x, 2)         models/unet.py:120-132, models/fcn.py:116-126
  slim.convolution2d_transpose  models/unet.py:138-159
  tf.nn.conv2d_transpose        models/fcn.py:199-216
  resize_image_with_crop_or_pad models/unet.py:72,140-173, models/fcn.py:202-218
  softmax_cross_entropy_with_logits  models/basemodel.py:59-70 (commented spec), :360
  tf.nn.sigmoid / tf.argmax     models/unet.py:75-79, models/fcn.py:74-78
  tf.train.AdamOptimizer        models/basemodel.py:321,366
"""
import numpy as np


# ----------------------------------------------------------------------------
# padding arithmetic (TF rules)
# ----------------------------------------------------------------------------
def same_pads(n_in, k, s):
    """TF SAME: out = ceil(in/s); pad_total = max((out-1)s + k - in, 0); before = total//2."""
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def conv_out_size(n_in, k, s, padding):
    if padding == 'VALID':
        return (n_in - k) // s + 1, 0
    n_out, before, _ = same_pads(n_in, k, s)
    return n_out, before


# ----------------------------------------------------------------------------
# conv2d  (slim.convolution2d = conv + bias + ReLU by default)
# ----------------------------------------------------------------------------
def _pad_hw(x, pt, pb, pl, pr):
    if pt == pb == pl == pr == 0:
        return x
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))


def conv2d(x, w, b=None, padding='VALID', stride=1, relu=True, dt=np.float64):
    """y[b,i,j,o] = act(sum_{u,v,c} x[b, i*s+u-pt, j*s+v-pl, c] * w[u,v,c,o] + b[o])."""
    x = np.asarray(x, dt); w = np.asarray(w, dt)
    B, H, W, C = x.shape
    kh, kw, ci, co = w.shape
    assert ci == C
    Ho, pt = conv_out_size(H, kh, stride, padding)
    Wo, pl = conv_out_size(W, kw, stride, padding)
    pb = max((Ho - 1) * stride + kh - H - pt, 0)
    pr = max((Wo - 1) * stride + kw - W - pl, 0)
    xp = _pad_hw(x, pt, pb, pl, pr)
    y = np.zeros((B, Ho, Wo, co), dt)
    for u in range(kh):
        for v in range(kw):
            xs = xp[:, u:u + (Ho - 1) * stride + 1:stride, v:v + (Wo - 1) * stride + 1:stride, :]
            y += np.matmul(xs.reshape(-1, C), w[u, v]).reshape(B, Ho, Wo, co)
    if b is not None:
        y += np.asarray(b, dt)
    if relu:
        np.maximum(y, 0, out=y)
    return y


def conv2d_dgrad(dz, w, in_hw, padding='VALID', stride=1, dt=np.float64):
    """Gradient of the pre-activation conv wrt its input (Conv2DBackpropInput)."""
    dz = np.asarray(dz, dt); w = np.asarray(w, dt)
    B, Ho, Wo, co = dz.shape
    kh, kw, ci, _ = w.shape
    H, W = in_hw
    _, pt = conv_out_size(H, kh, stride, padding)
    _, pl = conv_out_size(W, kw, stride, padding)
    Hp = max((Ho - 1) * stride + kh, H + pt)
    Wp = max((Wo - 1) * stride + kw, W + pl)
    dxp = np.zeros((B, Hp, Wp, ci), dt)
    flat = dz.reshape(-1, co)
    for u in range(kh):
        for v in range(kw):
            dxp[:, u:u + (Ho - 1) * stride + 1:stride, v:v + (Wo - 1) * stride + 1:stride, :] += \
                np.matmul(flat, w[u, v].T).reshape(B, Ho, Wo, ci)
    return dxp[:, pt:pt + H, pl:pl + W, :]


def conv2d_wgrad(x, dz, k_hw, padding='VALID', stride=1, dt=np.float64):
    """Gradient wrt the HWIO filter (Conv2DBackpropFilter) and the bias (BiasAddGrad)."""
    x = np.asarray(x, dt); dz = np.asarray(dz, dt)
    B, H, W, C = x.shape
    _, Ho, Wo, co = dz.shape
    kh, kw = k_hw
    _, pt = conv_out_size(H, kh, stride, padding)
    _, pl = conv_out_size(W, kw, stride, padding)
    pb = max((Ho - 1) * stride + kh - H - pt, 0)
    pr = max((Wo - 1) * stride + kw - W - pl, 0)
    xp = _pad_hw(x, pt, pb, pl, pr)
    dw = np.zeros((kh, kw, C, co), dt)
    flat = dz.reshape(-1, co)
    for u in range(kh):
        for v in range(kw):
            xs = xp[:, u:u + (Ho - 1) * stride + 1:stride, v:v + (Wo - 1) * stride + 1:stride, :]
            dw[u, v] = np.matmul(xs.reshape(-1, C).T, flat)
    db = flat.sum(0)
    return dw, db


# ----------------------------------------------------------------------------
# max-pool 2x2 stride 2 VALID  (slim.max_pool2d(x, 2): stride defaults to 2)
# ----------------------------------------------------------------------------
def max_pool2x2(x):
    """Returns (y, idx). idx in {0,1,2,3} = first maximum in row-major window order
    (MaxPoolGrad routing).  Odd trailing row/col dropped (VALID)."""
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    xs = x[:, :Ho * 2, :Wo * 2, :].reshape(B, Ho, 2, Wo, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, Ho, Wo, C, 4)
    idx = np.argmax(xs, axis=-1).astype(np.uint8)      # np.argmax returns the first max
    y = np.take_along_axis(xs, idx[..., None].astype(np.int64), axis=-1)[..., 0]
    return y, idx


def max_pool2x2_bwd(dy, idx, in_hw):
    B, Ho, Wo, C = dy.shape
    H, W = in_hw
    g = np.zeros((B, Ho, Wo, C, 4), dy.dtype)
    np.put_along_axis(g, idx[..., None].astype(np.int64), dy[..., None], axis=-1)
    g = g.reshape(B, Ho, Wo, C, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(B, Ho * 2, Wo * 2, C)
    dx = np.zeros((B, H, W, C), dy.dtype)
    dx[:, :Ho * 2, :Wo * 2, :] = g
    return dx


def max_pool_k(x, k):
    """slim.max_pool2d(x, k, k): kernel k, stride k, VALID (models/deconvolution.py:118,131,140).  Returns (y, idx) with idx the
    FIRST maximum in row-major window order (MaxPoolGrad routing)."""
    B, H, W, C = x.shape
    Ho, Wo = H // k, W // k
    xs = x[:, :Ho * k, :Wo * k, :].reshape(B, Ho, k, Wo, k, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, Ho, Wo, C, k * k)
    idx = np.argmax(xs, axis=-1)
    y = np.take_along_axis(xs, idx[..., None], axis=-1)[..., 0]
    return y, idx.astype(np.uint8)


def max_pool_k_bwd(dy, idx, in_hw, k):
    B, Ho, Wo, C = dy.shape
    H, W = in_hw
    g = np.zeros((B, Ho, Wo, C, k * k), dy.dtype)
    np.put_along_axis(g, idx[..., None].astype(np.int64), dy[..., None], axis=-1)
    g = g.reshape(B, Ho, Wo, C, k, k).transpose(0, 1, 4, 2, 5, 3).reshape(B, Ho * k, Wo * k, C)
    dx = np.zeros((B, H, W, C), dy.dtype)
    dx[:, :Ho * k, :Wo * k, :] = g
    return dx


# ----------------------------------------------------------------------------
# slim.batch_norm with its defaults: decay 0.999, center=True (beta), scale=False (no gamma), epsilon 0.001,
# statistics over (B,H,W), population variance (tf.nn.moments); moving averages updated through UPDATE_OPS
# (models/deconvolution.py:116-165, models/basemodel.py:364-365)
# ----------------------------------------------------------------------------
def batch_norm(x, beta, moving_mean=None, moving_var=None, training=True, decay=0.999, eps=1e-3, dt=np.float64):
    """Returns (y, cache, new_moving_mean, new_moving_var)."""
    x = np.asarray(x, dt)
    if training:
        mean = x.mean((0, 1, 2)); var = x.var((0, 1, 2))
        nm = None if moving_mean is None else decay * np.asarray(moving_mean, dt) + (1 - decay) * mean
        nv = None if moving_var is None else decay * np.asarray(moving_var, dt) + (1 - decay) * var
    else:
        mean, var = np.asarray(moving_mean, dt), np.asarray(moving_var, dt)
        nm, nv = moving_mean, moving_var
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat + np.asarray(beta, dt), (xhat, rstd), nm, nv


def batch_norm_bwd(dy, cache):
    """Training-mode gradient wrt the input and beta (no gamma)."""
    xhat, rstd = cache
    dbeta = dy.sum((0, 1, 2))
    dx = rstd * (dy - dy.mean((0, 1, 2)) - xhat * (dy * xhat).mean((0, 1, 2)))
    return dx, dbeta


# ----------------------------------------------------------------------------
# tf.image.resize_bilinear(x, [Hd, Wd]) with align_corners=False (models/deconvolution.py:160): in = out * (Hs / Hd) (the scale
# and the source coordinate are float32 in TF's kernel), lower = floor(in), upper = min(lower + 1, Hs - 1), lerp = in - lower
# ----------------------------------------------------------------------------
def _resize_axis(n_in, n_out):
    scale = np.float32(n_in) / np.float32(n_out)
    f = (np.arange(n_out, dtype=np.float32) * scale).astype(np.float32)
    lo = np.minimum(np.floor(f).astype(np.int64), n_in - 1)
    hi = np.minimum(lo + 1, n_in - 1)
    return lo, hi, (f - np.floor(f)).astype(np.float64)


def resize_bilinear(x, out_hw, dt=np.float64):
    x = np.asarray(x, dt)
    B, H, W, C = x.shape
    y0, y1, ly = _resize_axis(H, out_hw[0]); x0, x1, lx = _resize_axis(W, out_hw[1])
    ly = ly[None, :, None, None]; lx = lx[None, None, :, None]
    top = x[:, y0][:, :, x0] + (x[:, y0][:, :, x1] - x[:, y0][:, :, x0]) * lx
    bot = x[:, y1][:, :, x0] + (x[:, y1][:, :, x1] - x[:, y1][:, :, x0]) * lx
    return top + (bot - top) * ly


def resize_bilinear_bwd(dy, in_hw, dt=np.float64):
    dy = np.asarray(dy, dt)
    B, Hd, Wd, C = dy.shape
    H, W = in_hw
    y0, y1, ly = _resize_axis(H, Hd); x0, x1, lx = _resize_axis(W, Wd)
    My = np.zeros((Hd, H), dt); My[np.arange(Hd), y0] += 1 - ly; My[np.arange(Hd), y1] += ly
    Mx = np.zeros((Wd, W), dt); Mx[np.arange(Wd), x0] += 1 - lx; Mx[np.arange(Wd), x1] += lx
    return np.einsum('byxc,yi,xj->bijc', dy, My, Mx, optimize=True)


# ----------------------------------------------------------------------------
# transposed conv.  w is [kh,kw,Cout,Cin] (TF conv2d_transpose filter layout)
# ----------------------------------------------------------------------------
def convT_out_size(n_in, k, s, padding):
    """VALID: in*s + max(k-s,0); SAME: in*s.  Returns (out, pad_before)."""
    if padding == 'VALID':
        return n_in * s + max(k - s, 0), 0
    n_out = n_in * s
    total = max((n_in - 1) * s + k - n_out, 0)
    return n_out, total // 2


def conv2d_transpose(x, w, b=None, stride=2, padding='VALID', relu=False, dt=np.float64):
    """out[b, i*s+u-pt, j*s+v-pl, o] += x[b,i,j,c] * w[u,v,o,c]."""
    x = np.asarray(x, dt); w = np.asarray(w, dt)
    B, H, W, C = x.shape
    kh, kw, co, ci = w.shape
    assert ci == C
    Ho, pt = convT_out_size(H, kh, stride, padding)
    Wo, pl = convT_out_size(W, kw, stride, padding)
    full = np.zeros((B, (H - 1) * stride + kh, (W - 1) * stride + kw, co), dt)
    flat = x.reshape(-1, C)
    for u in range(kh):
        for v in range(kw):
            full[:, u:u + (H - 1) * stride + 1:stride, v:v + (W - 1) * stride + 1:stride, :] += \
                np.matmul(flat, w[u, v].T).reshape(B, H, W, co)
    y = full[:, pt:pt + Ho, pl:pl + Wo, :].copy()
    if b is not None:
        y += np.asarray(b, dt)
    if relu:
        np.maximum(y, 0, out=y)
    return y


def conv2d_transpose_dgrad(dz, w, in_hw, stride=2, padding='VALID', dt=np.float64):
    """dx[b,i,j,c] = sum_{u,v,o} dz[b, i*s+u-pt, j*s+v-pl, o] * w[u,v,o,c]."""
    dz = np.asarray(dz, dt); w = np.asarray(w, dt)
    B, Ho, Wo, co = dz.shape
    kh, kw, _, ci = w.shape
    H, W = in_hw
    _, pt = convT_out_size(H, kh, stride, padding)
    _, pl = convT_out_size(W, kw, stride, padding)
    fh, fw = (H - 1) * stride + kh, (W - 1) * stride + kw
    full = np.zeros((B, fh, fw, co), dt)
    full[:, pt:pt + Ho, pl:pl + Wo, :] = dz
    dx = np.zeros((B, H, W, ci), dt)
    for u in range(kh):
        for v in range(kw):
            s = full[:, u:u + (H - 1) * stride + 1:stride, v:v + (W - 1) * stride + 1:stride, :]
            dx += np.matmul(s.reshape(-1, co), w[u, v]).reshape(B, H, W, ci)
    return dx


def conv2d_transpose_wgrad(x, dz, k_hw, stride=2, padding='VALID', dt=np.float64):
    x = np.asarray(x, dt); dz = np.asarray(dz, dt)
    B, H, W, C = x.shape
    _, Ho, Wo, co = dz.shape
    kh, kw = k_hw
    _, pt = convT_out_size(H, kh, stride, padding)
    _, pl = convT_out_size(W, kw, stride, padding)
    fh, fw = (H - 1) * stride + kh, (W - 1) * stride + kw
    full = np.zeros((B, fh, fw, co), dt)
    full[:, pt:pt + Ho, pl:pl + Wo, :] = dz
    dw = np.zeros((kh, kw, co, C), dt)
    flat = x.reshape(-1, C)
    for u in range(kh):
        for v in range(kw):
            s = full[:, u:u + (H - 1) * stride + 1:stride, v:v + (W - 1) * stride + 1:stride, :]
            dw[u, v] = np.matmul(s.reshape(-1, co).T, flat)
    db = dz.reshape(-1, co).sum(0)
    return dw, db


# ----------------------------------------------------------------------------
# tf.image.resize_image_with_crop_or_pad (centre crop / zero pad, floor offsets)
# ----------------------------------------------------------------------------
def _cp_axis(n_in, n_t):
    """returns (src_start, dst_start, length)"""
    if n_in >= n_t:
        return (n_in - n_t) // 2, 0, n_t
    return 0, (n_t - n_in) // 2, n_in


def crop_or_pad(x, th, tw):
    B, H, W, C = x.shape
    sy, dy, ly = _cp_axis(H, th)
    sx, dx, lx = _cp_axis(W, tw)
    out = np.zeros((B, th, tw, C), x.dtype)
    out[:, dy:dy + ly, dx:dx + lx, :] = x[:, sy:sy + ly, sx:sx + lx, :]
    return out


def crop_or_pad_bwd(dout, in_hw):
    B, th, tw, C = dout.shape
    H, W = in_hw
    sy, dy, ly = _cp_axis(H, th)
    sx, dx, lx = _cp_axis(W, tw)
    g = np.zeros((B, H, W, C), dout.dtype)
    g[:, sy:sy + ly, sx:sx + lx, :] = dout[:, dy:dy + ly, dx:dx + lx, :]
    return g


# ----------------------------------------------------------------------------
# loss / output ops
# ----------------------------------------------------------------------------
def softmax_xent(logits, labels, dt=np.float64):
    """tf.nn.softmax_cross_entropy_with_logits(labels=one_hot(y), logits) per pixel, then
    tf.reduce_mean (models/basemodel.py:69-70,360).  labels: integer [B,h,w] or [B,h,w,1].
    Returns (mean_loss, per_pixel_loss, dlogits) with dlogits = (softmax - onehot)/(B*h*w)."""
    z = np.asarray(logits, dt)
    y = np.asarray(labels).reshape(z.shape[:-1]).astype(np.int64)
    C = z.shape[-1]
    zm = z - z.max(-1, keepdims=True)
    e = np.exp(zm)
    s = e.sum(-1, keepdims=True)
    logp = zm - np.log(s)
    # tf.one_hot gives an all-zero row for labels outside [0,C)
    valid = (y >= 0) & (y < C)
    yc = np.clip(y, 0, C - 1)
    onehot = np.zeros_like(z)
    np.put_along_axis(onehot, yc[..., None], 1.0, axis=-1)
    onehot *= valid[..., None]
    per = -(onehot * logp).sum(-1)
    n = per.size
    # d/dz of -sum_k onehot_k * logp_k  =  softmax * sum(onehot) - onehot
    dz = ((e / s) * onehot.sum(-1, keepdims=True) - onehot) / n
    return per.mean(), per, dz


def sigmoid_argmax(logits):
    """models/unet.py:75-79: y_hat_sig = sigmoid(y_hat) in float32; output = float32(argmax over C of
    the *sigmoid* values, first maximum wins) with a trailing unit axis."""
    z = np.asarray(logits, np.float32)
    sig = (np.float32(1) / (np.float32(1) + np.exp(-z.astype(np.float64)))).astype(np.float32)
    out = np.argmax(sig, axis=-1).astype(np.float32)[..., None]
    return sig, out


def adam_tf(p, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, dt=np.float64):
    """tf.train.AdamOptimizer step t (1-based): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2; p -= lr_t*m/(sqrt(v)+eps)  (eps outside the correction)."""
    p = np.asarray(p, dt); g = np.asarray(g, dt); m = np.asarray(m, dt); v = np.asarray(v, dt)
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    p = p - lr_t * m / (np.sqrt(v) + eps)
    return p, m, v


def xavier_uniform(shape, rng, fan_in, fan_out):
    """slim default weights_initializer: U(-l, l), l = sqrt(6/(fan_in+fan_out))."""
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def miou(pred, label, n_classes):
    """Build-defined metric (the reference has none): per-class IoU of argmax vs label,
    mean over classes present in either."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    label = np.asarray(label).reshape(-1).astype(np.int64)
    ious = []
    for c in range(n_classes):
        p, l = pred == c, label == c
        union = (p | l).sum()
        if union:
            ious.append((p & l).sum() / union)
    return float(np.mean(ious)) if ious else 0.0


# ----------------------------------------------------------------------------
# slim.dropout-style mask (BUILD-DEFINED for the U-Net; the only reference precedent is
# models/deconvolution.py:128-129,143-144,153-154: slim.dropout, keep_prob 0.5, always on)
# ----------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(k):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    k = np.asarray(k, np.uint64)
    with np.errstate(over='ignore'):
        k = k + np.uint64(0x9E3779B97F4A7C15)
        k = (k ^ (k >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        k = (k ^ (k >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return k ^ (k >> np.uint64(31))


def dropout_mask(shape, keep, seed, offset, c_pad=None):
    """Boolean keep-mask of an NHWC tensor [B,H,W,C] whose buffer pads the channels to c_pad (default: C rounded up to 32).
    Element (b,y,x,c) draws the counter offset + ((b*H + y)*W + x)*c_pad + c from the stream keyed by splitmix64(seed):
    r = high 32 bits of splitmix64(key ^ counter); kept iff r <= uint32(keep * 4294967295.0)."""
    B, H, W, Cc = shape
    cp = c_pad if c_pad is not None else (Cc + 31) // 32 * 32
    idx = (np.arange(B * H * W, dtype=np.uint64)[:, None] * np.uint64(cp) + np.arange(Cc, dtype=np.uint64)[None, :]).reshape(B, H, W, Cc)
    key = splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
    with np.errstate(over='ignore'):
        ctr = idx + np.uint64(offset & 0xFFFFFFFFFFFFFFFF)
    r = (splitmix64(key ^ ctr) >> np.uint64(32)).astype(np.uint32)
    thr = np.uint32(int(np.float64(np.float32(keep)) * 4294967295.0))
    return r <= thr


def dropout(x, keep, seed, offset, dt=np.float64, c_pad=None):
    """y = x * mask / keep (the 1/keep factor is formed in float32 like the kernel's `1.f / keep`)."""
    x = np.asarray(x, dt)
    inv = np.float32(1.0) / np.float32(keep)
    return x * dropout_mask(x.shape, keep, seed, offset, c_pad) * dt(inv)


# ----------------------------------------------------------------------------
# bilinear filter bank  (restates utils/upsampling.py:6-46)
# ----------------------------------------------------------------------------
def get_kernel_size(factor):
    return 2 * factor - factor % 2


def upsample_filt(size):
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    og = np.ogrid[:size, :size]
    return (1 - abs(og[0] - center) / factor) * (1 - abs(og[1] - center) / factor)


def bilinear_upsample_weights(factor, n_classes):
    k = get_kernel_size(factor)
    w = np.zeros((k, k, n_classes, n_classes), np.float32)
    f = upsample_filt(k)
    for i in range(n_classes):
        w[:, :, i, i] = f
    return w
