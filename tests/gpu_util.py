"""Helpers for the -m gpu parity tests: build single-layer plans through the C-ABI and compare with the oracle."""
import numpy as np
import torch

from segmentation_amd import _lib as L
from segmentation_amd import engine as E


def dev():
    return torch.device('cuda', 0)


def round_dtype(a, dtype):
    """value grid of the compute dtype, returned as float64 for the oracle"""
    a = np.asarray(a, np.float32)
    if dtype == L.SEG_BF16:
        return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)
    return a.astype(np.float64)


def fill_act(act, arr):
    """arr: logical [B,H,W,C] -> writes channels [0,C), pad channels zero"""
    t = torch.zeros(act.t.shape, dtype=torch.float32)
    t[..., :arr.shape[-1]] = torch.from_numpy(np.asarray(arr, np.float32))
    act.t.copy_(t.to(act.t.dtype))


def read_act(act, C=None):
    C = act.C if C is None else C
    return act.t[..., :C].to(torch.float32).cpu().numpy().astype(np.float64)


def pad_channels_zero(act):
    if act.Cp == act.C:
        return True
    return bool((act.t[..., act.C:].to(torch.float32) == 0).all().item())


def stream():
    return torch.cuda.current_stream().cuda_stream


def sync():
    torch.cuda.synchronize()


def tol(dtype, f32_tol=2e-5, bf16_tol=2e-2):
    return f32_tol if dtype == L.SEG_F32 else bf16_tol


def tol_sum(dtype, f32_tol=2e-5):
    """Bound for an f32 OUTPUT that is a sum of exact products of the operands given to the oracle (filter / bias gradients: bf16 x
    bf16 products are exact in f32, so only the summation order differs from the float64 oracle): 1e-4 of the tensor maximum in
    bf16 mode -- one dropped window pixel moves such a sum by 4e-3 and more."""
    return f32_tol if dtype == L.SEG_F32 else 1e-4


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / (np.sqrt((b ** 2).sum()) + 1e-30))


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def make_store(layers, dtype, params):
    store = E.ParamStore(layers, dtype, dev(), training=True)
    store.set_params(params)
    net = E.Net(store, 1, dtype, dev())
    p = E.Plan('pack'); net.pack(p); p.run(stream()); sync()
    return store


# ---------------------------------------------------------------------------------------------------------------------------
# Precision attribution (tests/test_precision_gpu.py): run an F32-MODE model's plans with ONE of the roundings of the bf16
# mode switched on at a time.  `arms` is a set of:
#   'w'      the packed MFMA copies of the filters are bf16 values (forward and data-gradient operands)
#   'act'    every forward activation is rounded to bf16 when it is stored (convolutions, transposed convolutions, first layer)
#   'dz'     every gradient tensor of the backward chain is rounded when it is stored (data gradients, pool backward, head)
#   'wgrad'  only the two operands of each filter gradient are rounded (copies; the chains themselves stay float32)
# All of them together reproduce the arithmetic of the bf16 mode (bf16 operands, float32 accumulation, bf16 storage).
# ---------------------------------------------------------------------------------------------------------------------------
def quantized_plans(m, arms):
    """-> (fwd ops, bwd ops) as lists of (name, fn, args) to be launched IN ORDER on one stream (markers dropped)"""
    import ctypes as C
    lib = L.load()
    keep = m.__dict__.setdefault('_quant_keep', [])
    B = m.batch_size

    def rnd(src, dst, H, W, Cc):
        s_, d_ = L.View.from_buffer_copy(src), L.View.from_buffer_copy(dst)
        keep.extend([s_, d_])
        return ('round', lib.seg_round_bf16, (C.byref(s_), C.byref(d_), B, H, W, Cc))

    def view_of(a):
        return getattr(a, '_obj', None)

    def temp_like(v, H, W):
        t = torch.zeros((B, H, W, v.c), dtype=torch.float32, device=dev())
        keep.append(t)
        return L.View(t.data_ptr(), H, W, v.c, 0, 0, 0, v.c)

    out = []
    for plan, is_fwd in ((m.fwd_plan, True), (m.bwd_plan, False)):
        ops = []
        for (name, fn, args), meta in zip(plan.ops, plan.meta):
            if fn is None:
                continue
            fname = fn.__name__
            pre, post = [], []
            d = meta.get('desc')
            if d is None and fname == 'seg_conv2d_wgrad':
                d = getattr(args[0], '_obj', None)            # (the slab reduction's descriptor: phase 2)
            if fname == 'seg_conv2d':
                sc = 2 if d.up2 else 1
                nch = d.up_cout if d.up2 else (d.n_split if d.n_split > 0 else d.n_count)
                want = ('act' in arms and is_fwd and not d.out_f32) or ('dz' in arms and not is_fwd)
                if want:
                    post.append(rnd(d.dst, d.dst, sc * d.Ho, sc * d.Wo, nch))
                    if d.n_split > 0:
                        post.append(rnd(d.dst1, d.dst1, d.Ho, d.Wo, d.n_count - d.n_split))
            elif fname in ('seg_conv_first_fwd', 'seg_conv_first_pool_fwd') and 'act' in arms:
                dv = view_of(args[9]); Ho, Wo = args[10], args[11]
                post.append(rnd(dv, dv, Ho, Wo, dv.c))
                if fname == 'seg_conv_first_pool_fwd':
                    pv = view_of(args[13])
                    post.append(rnd(pv, pv, args[14], args[15], pv.c))
            elif fname == 'seg_maxpool2x2_bwd' and 'dz' in arms:
                zv = view_of(args[7])
                post.append(rnd(zv, zv, args[9], args[10], args[11]))
            elif fname == 'seg_head_xent' and 'dz' in arms:
                gv = view_of(args[17])
                post.append(rnd(gv, gv, args[10], args[11], gv.c))
            elif fname in ('seg_pack_weights', 'seg_pack_weights_dual') and 'w' in arms:
                n = m.store.packed.numel()
                post.append(('round_w', lib.seg_round_bf16_flat, (m.store.packed.data_ptr(), m.store.packed.data_ptr(), n)))
            elif fname == 'seg_conv2d_wgrad' and 'wgrad' in arms and d.phase != 2:
                w2 = L.WgradDesc.from_buffer_copy(d)
                keep.append(w2)
                if d.im2col_x:
                    n = B * d.im2col_h * d.im2col_w * d.im2col_cin
                    t = torch.zeros(n, dtype=torch.float32, device=dev()); keep.append(t)
                    pre.append(('round_x', lib.seg_round_bf16_flat, (d.im2col_x, t.data_ptr(), n)))
                    w2.im2col_x = t.data_ptr(); w2.src0.ptr = t.data_ptr()
                else:
                    hs, ws_ = (d.Hi, d.Wi)
                    for fld in ('src0', 'src1'):
                        v = getattr(d, fld)
                        if not v.ptr or v.c == 0:
                            continue
                        win = L.View(v.ptr, v.H, v.W, v.cs, v.coff, v.oy, v.ox, v.c)
                        tv = temp_like(v, hs, ws_)
                        pre.append(rnd(win, tv, hs, ws_, v.c))
                        setattr(w2, fld, tv)
                zv = d.dz
                tz = temp_like(zv, d.Ho, d.Wo)
                pre.append(rnd(zv, tz, d.Ho, d.Wo, zv.c))
                w2.dz = tz
                args = (C.byref(w2),)
            ops += pre + [(name, fn, args)] + post
        out.append(ops)
    return out


def run_ops(ops):
    import ctypes as C
    sp = C.c_void_p(stream())
    for name, fn, args in ops:
        rc = fn(*args, sp)
        if rc != 0:
            L.check(rc, name)
