"""-m gpu parity tests, kernel level: every C-ABI entry point vs the numpy oracle on identical inputs,
at the odd sizes the VALID U-Net actually produces.  f32 mode: <= 2e-5 relative (exact-f32 MFMA vs float64
oracle); bf16 mode: inputs/weights pre-rounded to bf16, <= 2e-2 relative (bf16 output rounding)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import np_ops as ops
from segmentation_amd import _lib as L
from segmentation_amd import engine as E
import gpu_util as U

pytestmark = pytest.mark.gpu
DT = [L.SEG_F32, L.SEG_BF16]


def _rand_params(layer, rng, dtype):
    w = rng.standard_normal(layer.wshape).astype(np.float32) * 0.2
    b = rng.standard_normal(layer.cout).astype(np.float32) * 0.1
    return {'weights': w, 'biases': b}


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('case', [
    # k, padding, segs, cout, H, W, B, relu, cfg
    (3, 'VALID', [32], 64, 13, 15, 2, True, 0),
    (3, 'VALID', [32], 64, 13, 15, 2, True, 1),
    (3, 'VALID', [32], 64, 13, 15, 2, True, 3),
    (3, 'VALID', [64], 32, 21, 19, 1, True, 2),
    (3, 'VALID', [64], 32, 10, 10, 2, False, 4),
    (3, 'VALID', [32], 64, 13, 15, 2, True, 11),
    (3, 'VALID', [64], 32, 21, 19, 1, True, 12),
    (3, 'VALID', [32], 64, 13, 15, 2, True, 13),
    (3, 'VALID', [64], 32, 10, 10, 2, False, 14),
    (3, 'VALID', [64, 32], 64, 35, 37, 2, True, 15),
    (3, 'SAME', [64], 64, 19, 33, 1, True, 15),
    (3, 'SAME', [64], 96, 9, 11, 2, True, 0),
    (3, 'VALID', [64, 32], 64, 12, 12, 2, True, 0),
    (3, 'VALID', [16], 24, 11, 11, 1, True, 0),          # unpadded channel counts (n_kernels=16 style)
    (1, 'SAME', [96], 4, 7, 9, 2, False, 0),
    (1, 'SAME', [64], 160, 16, 16, 1, True, 0),
    (3, 'VALID', [256], 256, 12, 12, 3, True, 0),        # automatic tile class on a deep layer
    # the round-4 kernel (csrc/conv_ring.hip; bf16, 64-channel blocks): cfg 208 = 512-pixel tiles, 204 = 256-pixel tiles; the
    # data gradients of these cases run on it too where they have 64-channel blocks (two-destination form: [64, 64])
    (3, 'VALID', [64, 64], 64, 35, 37, 2, True, 208),
    (3, 'VALID', [64, 64], 64, 35, 37, 2, True, 204),
    (3, 'SAME', [64], 128, 19, 33, 3, True, 208),        # zero padding on every side, ragged tiles
    (3, 'SAME', [128], 64, 9, 11, 2, False, 204),
    (3, 'VALID', [32], 128, 26, 26, 2, True, 208),
    (3, 'VALID', [64], 64, 150, 131, 4, True, 204),      # more tiles than compute units: the persistent walk + tile tickets
    (3, 'SAME', [32, 32], 64, 97, 140, 3, True, 208),
    (3, 'VALID', [64], 128, 70, 200, 2, True, 208),      # wide maps: the 8 x 64 tile shape
    (3, 'VALID', [192], 64, 10, 10, 5, True, 204),       # 8 x 8 maps: most of a tile is padding
    # linearised tiles (conv_fwd_kernel<.., LIN>: cfg 7 = 128 slots x 64 channels, 8 = x 32); the data gradients run on cfg 8
    (3, 'VALID', [64], 64, 12, 12, 3, True, 7),          # 10 x 10 map: one tile, 28 padding slots
    (3, 'VALID', [64, 32], 64, 28, 28, 2, True, 7),      # 26 x 26: 2 x 3 windows of 9 x 13; two-destination data gradient
    (3, 'SAME', [32], 96, 38, 38, 2, True, 8),
    (3, 'VALID', [64], 64, 59, 61, 2, False, 7),         # column strips
    (3, 'SAME', [64], 32, 5, 131, 1, True, 8),           # one-row windows
    (3, 'VALID', [32], 64, 3, 3, 2, True, 8),            # a single output pixel
    (3, 'VALID', [64], 64, 150, 9, 1, True, 7),          # tall and narrow: 18-row windows
    # multi-tile walk of the one-chunk layers (conv_fwd_mt_kernel: cfg 32 / 34 / 38 = 2 / 4 / 8 tiles per workgroup x 32 channels, 62 / 64 = x 64)
    (3, 'VALID', [32], 32, 37, 41, 2, True, 32),
    (3, 'SAME', [32], 32, 40, 75, 3, True, 34),          # 5 x 5 tiles per image: workgroups span images, the last one is short
    (3, 'VALID', [32], 64, 21, 150, 1, False, 62),
    (3, 'SAME', [17], 64, 9, 11, 2, True, 64),           # fewer tiles than a workgroup walks
    (3, 'VALID', [32], 32, 70, 70, 2, True, 38),
    # [32 | 32]-channel concat input: its two-destination data gradient is ONE 64-channel block per tile (conv_fwd_kernel<.., SPLIT>)
    (3, 'VALID', [32, 32], 32, 37, 41, 2, True, 9),
    (3, 'SAME', [32, 32], 64, 18, 23, 3, False, 9),
])
def test_conv_fwd_bwd(dtype, case):
    k, padding, segs, cout, H, W, B, relu, cfg = case
    if cfg > 10 and dtype != L.SEG_BF16 and cfg not in (32, 34, 38, 62, 64, 68):
        pytest.skip('direct-to-LDS / persistent variants are bf16 only')
    rng = np.random.default_rng(k * 7919 + sum(segs) * 31 + cout * 17 + H * 5 + W + cfg)
    layer = E.Layer('c', 'conv', k, segs, cout, padding, relu)
    p = {'c': _rand_params(layer, rng, dtype)}
    p['c']['weights'] = U.round_dtype(p['c']['weights'], dtype).astype(np.float32)
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    # sources live inside larger buffers to exercise window offsets
    srcs, xs = [], []
    for i, c in enumerate(segs):
        a = net.act(H + 3 + i, W + 2, c)
        full = U.round_dtype(rng.standard_normal((B, a.H, a.W, c)), dtype)
        U.fill_act(a, full)
        oy, ox = 1 + i, 2 - i
        srcs.append((a, oy, ox))
        xs.append(full[:, oy:oy + H, ox:ox + W, :])
    x = np.concatenate(xs, -1)
    pad = layer.pad
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    out = net.act(Ho, Wo, cout)
    plan = E.Plan('t')
    net.conv_fwd(plan, layer, srcs, H, W, out, cfg=0 if cfg == 9 else cfg)
    if cfg in (204, 208):
        assert plan.kernel_name(0).startswith('conv_ring_kernel<%s,' % {204: '8,4,2,2', 208: '8,4,1,4'}[cfg]), plan.kernel_name(0)
    if cfg in (7, 8):
        assert ',lin128,%d,' % {7: 64, 8: 32}[cfg] in plan.kernel_name(0), plan.kernel_name(0)
    if cfg in (32, 34, 38, 62, 64):
        assert plan.kernel_name(0).startswith('conv_fwd_mt_kernel<'), plan.kernel_name(0)
    plan.run(U.stream()); U.sync()
    ref = ops.conv2d(x, p['c']['weights'], p['c']['biases'], padding, 1, relu)
    got = U.read_act(out)
    assert U.rel_err(got, ref) < U.tol(dtype), 'fwd'
    assert U.pad_channels_zero(out)

    # backward: dz random (already "masked"), check dW, db, dX (with and without a ReLU-grad mask)
    dzv = U.round_dtype(rng.standard_normal((B, Ho, Wo, cout)) * 0.5, dtype)
    dz = net.act(Ho, Wo, cout); U.fill_act(dz, dzv)
    dsrc_acts, dspecs, masks = [], [], []
    for i, c in enumerate(segs):
        da = net.act(H, W, c)
        if i == 0:
            mk = net.act(H, W, c)
            mv = U.round_dtype(rng.standard_normal((B, H, W, c)), dtype)
            U.fill_act(mk, mv)
            masks.append(mv)
            dspecs.append((da, (0, 0), mk, (0, 0)))
        else:
            masks.append(None)
            dspecs.append((da, (0, 0), None, (0, 0)))
        dsrc_acts.append(da)
    store.g.zero_()
    bplan = E.Plan('b')
    net.conv_bwd(bplan, layer, srcs, H, W, dz, dspecs, cfg=cfg if cfg >= 100 or cfg == 9 else (8 if cfg in (7, 8) else (cfg if cfg in (32, 34, 38) and cout == 32 else 0)))
    net.flush_reduce(bplan)
    if cfg == 9:
        names = [bplan.kernel_name(i) for i, (n, _, _) in enumerate(bplan.ops) if n.endswith('/dx01')]
        assert len(names) == 1 and names[0].endswith(',split>'), names
    bplan.run(U.stream()); U.sync()
    dw_ref, db_ref = ops.conv2d_wgrad(x, dzv, (k, k), padding, 1)
    dx_ref = ops.conv2d_dgrad(dzv, p['c']['weights'], (H, W), padding, 1)
    g = store.get_grads()['c']
    assert U.rel_err(g['weights'], dw_ref) < U.tol_sum(dtype), 'wgrad'
    assert U.rel_err(g['biases'], db_ref) < U.tol_sum(dtype), 'bias grad'
    c0 = 0
    for i, c in enumerate(segs):
        want = dx_ref[..., c0:c0 + c]
        if masks[i] is not None:
            want = want * (masks[i] > 0)
        assert U.rel_err(U.read_act(dsrc_acts[i]), want) < U.tol(dtype), 'dgrad seg %d' % i
        assert U.pad_channels_zero(dsrc_acts[i])
        c0 += c


@pytest.mark.parametrize('H,W,lin', [(12, 12, True), (28, 28, True), (18, 18, False), (26, 26, False), (61, 61, False)])
def test_conv_tile_choice_takes_linearised_tiles_where_fixed_ones_pad_badly(H, W, lin):
    """automatic tile choice (cfg 0): a 10 x 10 / 26 x 26 output runs on the linearised tile, maps that fill the 8 x 8 / 8 x 16 tiles do not"""
    layer = E.Layer('c', 'conv', 3, [64], 64, 'VALID', True)
    store = U.make_store([layer], L.SEG_BF16, {'c': _rand_params(layer, np.random.default_rng(0), L.SEG_BF16)})
    net = E.Net(store, 4, L.SEG_BF16, U.dev())
    a = net.act(H, W, 64); out = net.act(H - 2, W - 2, 64)
    plan = E.Plan('t')
    net.conv_fwd(plan, layer, [(a, 0, 0)], H, W, out)
    assert ('lin128' in plan.kernel_name(0)) == lin, plan.kernel_name(0)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('case', [
    # k, padding, segs, cout, H, W, B, cfg, ksplit     split K (seg_conv_desc.ksplit): the deep small-map layers
    (3, 'VALID', [256], 64, 10, 10, 3, 0, 2), (3, 'VALID', [256], 64, 10, 10, 3, 0, 4), (3, 'SAME', [128, 128], 96, 9, 13, 2, 0, 3),
    (3, 'VALID', [512], 32, 8, 8, 2, 24, 4), (3, 'VALID', [256], 64, 12, 12, 2, 11, 2), (3, 'VALID', [256], 64, 12, 12, 2, 1, 8),
    (1, 'SAME', [256], 64, 7, 9, 2, 0, 4), (3, 'VALID', [96], 32, 6, 6, 1, 4, 3),
])
def test_conv_split_k(dtype, case):
    """ksplit workgroups share a tile's K chunks; the last arriver adds the f32 partials in split order and runs the epilogue:
    forward (bias + ReLU) and data gradient (two-destination form with a ReLU-grad mask) against the oracle, bitwise reproducible
    from run to run, tickets left zero."""
    k, padding, segs, cout, H, W, B, cfg, ks = case
    if cfg > 10 and dtype != L.SEG_BF16:
        pytest.skip('direct-to-LDS variants are bf16 only')
    rng = np.random.default_rng(k * 7 + sum(segs) + cout * 3 + H + ks)
    layer = E.Layer('c', 'conv', k, segs, cout, padding, True)
    p = {'c': _rand_params(layer, rng, dtype)}
    p['c']['weights'] = U.round_dtype(p['c']['weights'], dtype).astype(np.float32)
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    srcs, xs = [], []
    for c in segs:
        a = net.act(H, W, c)
        full = U.round_dtype(rng.standard_normal((B, H, W, c)), dtype)
        U.fill_act(a, full); srcs.append((a, 0, 0)); xs.append(full)
    x = np.concatenate(xs, -1)
    pad = layer.pad
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    out = net.act(Ho, Wo, cout)
    plan = E.Plan('t')
    net.conv_fwd(plan, layer, srcs, H, W, out, cfg=cfg, ksplit=ks)
    d = plan.meta[0]['desc']
    assert d.ksplit == min(ks, sum(E.rup(c) for c in segs) // 32) and d.splitk_ws and d.splitk_tickets
    tickets = [t for t in plan.keep if isinstance(t, torch.Tensor) and t.dtype == torch.int32][0]
    plan.run(U.stream()); U.sync()
    ref = ops.conv2d(x, p['c']['weights'], p['c']['biases'], padding, 1, True)
    got = U.read_act(out)
    assert U.rel_err(got, ref) < U.tol(dtype), 'fwd'
    assert U.pad_channels_zero(out) and int(tickets.abs().sum().item()) == 0
    first = out.t.clone()
    for _ in range(3):
        out.t.zero_()
        plan.run(U.stream()); U.sync()
        assert torch.equal(out.t, first), 'split-K sum must not depend on the arrival order'
    # data gradient through the same kernels (K = cout chunks: only split when there are enough of them)
    if E.rup(cout) // 32 >= 2:
        dzv = U.round_dtype(rng.standard_normal((B, Ho, Wo, cout)) * 0.5, dtype)
        dz = net.act(Ho, Wo, cout); U.fill_act(dz, dzv)
        dspecs, dacts, masks = [], [], []
        for c in segs:
            da = net.act(H, W, c); mk = net.act(H, W, c)
            mv = U.round_dtype(rng.standard_normal((B, H, W, c)), dtype)
            U.fill_act(mk, mv); masks.append(mv); dacts.append(da)
            dspecs.append((da, (0, 0), mk, (0, 0)))
        bplan = E.Plan('b')
        net.conv_bwd(bplan, layer, srcs, H, W, dz, dspecs, dgrad_ksplit=2)
        net.flush_reduce(bplan)
        dd = [m['desc'] for m in bplan.meta if isinstance(m.get('desc'), L.ConvDesc)]
        assert dd and all(q.ksplit == 2 for q in dd)
        bplan.run(U.stream()); U.sync()
        dx_ref = ops.conv2d_dgrad(dzv, p['c']['weights'], (H, W), padding, 1)
        c0 = 0
        for i, c in enumerate(segs):
            want = dx_ref[..., c0:c0 + c] * (masks[i] > 0)
            assert U.rel_err(U.read_act(dacts[i]), want) < U.tol(dtype), 'dgrad seg %d' % i
            c0 += c


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('C_,nlog,H,W,B', [(32, 32, 190, 177, 2), (64, 40, 130, 257, 2), (8, 2, 300, 220, 1), (128, 128, 96, 96, 8)])
def test_bias_grad_two_stage(dtype, C_, nlog, H, W, B):
    """seg_bias_grad_ws (512 partial rows + a fixed-order final pass) against the column sums and the one-stage kernel;
    bitwise reproducible from run to run"""
    rng = np.random.default_rng(C_ + H)
    layer = E.Layer('c', 'conv', 1, [32], 32, 'VALID', True)
    net = E.Net(U.make_store([layer], dtype, {'c': _rand_params(layer, rng, dtype)}), B, dtype, U.dev())
    a = net.act(H, W, nlog, thin=(C_ == 8)) if C_ in (8,) else net.act(H, W, C_)
    v = U.round_dtype(rng.standard_normal((B, H, W, nlog)), dtype)
    U.fill_act(a, v)
    lib = L.load()
    zv = a.view()
    nb = int(lib.seg_bias_grad_ws_bytes(zv.c))
    assert nb == 512 * zv.c * 4
    ws = torch.empty(nb // 4, dtype=torch.float32, device=U.dev())
    db = torch.full((zv.c,), float('nan'), dtype=torch.float32, device=U.dev())
    L.check(lib.seg_bias_grad_ws(C.byref(zv), B, H, W, nlog, db.data_ptr(), ws.data_ptr(), nb, dtype, U.stream()), 'bias_grad_ws'); U.sync()
    want = v.reshape(-1, nlog).sum(0)
    got = db[:nlog].cpu().numpy().astype(np.float64)
    assert np.abs(got - want).max() < 2e-4 * np.abs(v).sum(axis=(0, 1, 2)).max()
    first = db[:nlog].clone()
    ws.fill_(float('nan')); db.fill_(float('nan'))
    L.check(lib.seg_bias_grad_ws(C.byref(zv), B, H, W, nlog, db.data_ptr(), ws.data_ptr(), nb, dtype, U.stream()), 'bias_grad_ws'); U.sync()
    assert torch.equal(db[:nlog], first)
    db1 = torch.zeros(zv.c, dtype=torch.float32, device=U.dev())
    L.check(lib.seg_bias_grad(C.byref(zv), B, H, W, nlog, db1.data_ptr(), dtype, U.stream()), 'bias_grad'); U.sync()
    assert np.abs(db1[:nlog].cpu().numpy() - got).max() < 2e-4 * np.abs(v).sum(axis=(0, 1, 2)).max()


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('valu', ['1', '0'])
@pytest.mark.parametrize('nc,H,W,B', [(2, 19, 23, 2), (5, 12, 40, 3), (8, 33, 9, 1), (4, 10, 14, 2)])
def test_thin_tensors_through_the_mfma_kernels(dtype, nc, H, W, B, valu, monkeypatch):
    """<= 8-channel tensors at a channel stride of 8 (engine.Act(thin=True)): a 2x2/s2 transposed conv INTO a thin tensor, a 3x3
    SAME conv FROM a thin tensor into a thin float tensor, and every gradient of both (thin dZ, thin sources, thin data-gradient
    destinations with a thin ReLU mask) against the oracle -- the DeconvModel's deconv3_0 / conv_out tail.  valu '1': the 3x3
    convolution and its data gradient on the vector-ALU kernel (seg_thin_conv3x3, the default); '0': on the MFMA kernels."""
    monkeypatch.setenv('SEG_THIN_VALU', valu)
    rng = np.random.default_rng(nc * 100 + H)
    up = E.Layer('u', 'up', 2, [32], nc, 'VALID', True)
    cv = E.Layer('c', 'conv', 3, [nc], nc, 'SAME', False)
    p = {'u': _rand_params(up, rng, dtype), 'c': _rand_params(cv, rng, dtype)}
    for n in p:
        p[n]['weights'] = U.round_dtype(p[n]['weights'], dtype).astype(np.float32)
    store = U.make_store([cv, up], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    xs = U.round_dtype(rng.standard_normal((B, H, W, 32)), dtype)
    x = net.act(H, W, 32); U.fill_act(x, xs)
    a = net.act(2 * H, 2 * W, nc, thin=True)
    lg = net.act(2 * H, 2 * W, nc, f32=True, thin=True)
    assert a.thin and a.Cp == 8 and a.t.shape[-1] == 8 and lg.t.dtype == torch.float32
    plan = E.Plan('t')
    net.up_fwd(plan, up, x, H, W, a)
    net.conv_fwd(plan, cv, [(a, 0, 0)], 2 * H, 2 * W, lg, out_f32=True)
    assert plan.meta[1]['kernel' if valu == '1' else 'desc'] is not None and (plan.meta[1].get('kernel') == 'thin_conv3x3_kernel') == (valu == '1')
    plan.run(U.stream()); U.sync()
    a_ref = ops.conv2d_transpose(xs, p['u']['weights'], p['u']['biases'], stride=2, padding='VALID', relu=True)
    assert U.rel_err(U.read_act(a), a_ref) < U.tol(dtype)
    a_got = U.round_dtype(U.read_act(a), dtype)
    lg_ref = ops.conv2d(a_got, p['c']['weights'], p['c']['biases'], 'SAME', 1, False)
    assert U.rel_err(lg.t[..., :nc].cpu().numpy().astype(np.float64), lg_ref) < U.tol(dtype)
    assert bool((a.t[..., nc:].float() == 0).all().item())
    # backward of the conv: thin dZ, thin source, thin destination
    dzv = U.round_dtype(rng.standard_normal((B, 2 * H, 2 * W, nc)) * 0.5, dtype)
    dz = net.act(2 * H, 2 * W, nc, thin=True); U.fill_act(dz, dzv)
    da = net.act(2 * H, 2 * W, nc, thin=True)
    store.g.zero_()
    bp = E.Plan('b')
    net.conv_bwd(bp, cv, [(a, 0, 0)], 2 * H, 2 * W, dz, [(da, (0, 0), a, (0, 0))])
    # ... and of the transposed conv from the (thin) gradient of its output
    dx = net.act(H, W, 32)
    net.up_bwd(bp, up, x, H, W, da, dx, None)
    net.flush_reduce(bp)
    bp.run(U.stream()); U.sync()
    g = store.get_grads()
    dw_ref, db_ref = ops.conv2d_wgrad(a_got, dzv, (3, 3), 'SAME', 1)
    assert U.rel_err(g['c']['weights'], dw_ref) < U.tol_sum(dtype) and U.rel_err(g['c']['biases'], db_ref) < U.tol_sum(dtype)
    da_ref = ops.conv2d_dgrad(dzv, p['c']['weights'], (2 * H, 2 * W), 'SAME', 1) * (a_got > 0)
    assert U.rel_err(U.read_act(da), da_ref) < U.tol(dtype)
    assert bool((da.t[..., nc:].float() == 0).all().item())
    da_got = U.round_dtype(U.read_act(da), dtype)
    uw_ref, ub_ref = ops.conv2d_transpose_wgrad(xs, da_got, (2, 2), stride=2, padding='VALID')
    assert U.rel_err(g['u']['weights'], uw_ref) < U.tol_sum(dtype) and U.rel_err(g['u']['biases'], ub_ref) < U.tol_sum(dtype)
    dx_ref = ops.conv2d_transpose_dgrad(da_got, p['u']['weights'], (H, W), stride=2, padding='VALID')
    assert U.rel_err(U.read_act(dx), dx_ref) < U.tol(dtype)


@pytest.mark.parametrize('case', [
    # k, padding, segs, cout, H, W, B, wcfg            filter-gradient layouts (bf16): 11/14 = 64 ci x 64 co, 12/15 = 32 ci x 64 co
    (3, 'VALID', [64], 64, 37, 35, 3, 11),
    (3, 'VALID', [64], 64, 37, 35, 3, 14),
    (3, 'SAME', [128], 64, 20, 44, 2, 11),              # two X chunks, edge tiles on both axes
    (3, 'VALID', [64, 64], 128, 26, 26, 2, 11),         # channel-concat input, two dZ blocks
    (3, 'VALID', [64], 64, 61, 59, 4, 11),              # long tile walk + K split + slab reduction
    (3, 'VALID', [256], 256, 10, 10, 2, 14),            # deep layer: filter rows split over blockIdx.z (no K split)
    (3, 'VALID', [32], 64, 37, 35, 3, 12),
    (3, 'VALID', [96], 64, 19, 23, 2, 15),
    (3, 'SAME', [32, 32], 64, 33, 17, 2, 12),
    (1, 'SAME', [128], 192, 16, 16, 2, 11),             # 1x1 (FCN conv6/conv7)
    (1, 'SAME', [64], 64, 9, 9, 1, 14),
    (3, 'VALID', [64], 64, 37, 35, 3, 0),               # automatic choice = the 64 x 64 layout
    (3, 'VALID', [48, 16], 40, 21, 19, 2, 0),           # unpadded channel counts -> 64 / 32 padded: automatic choice
])
def test_wgrad_layouts(case):
    k, padding, segs, cout, H, W, B, wcfg = case
    dtype = L.SEG_BF16
    rng = np.random.default_rng(k * 131 + sum(segs) * 7 + cout * 3 + H + W + wcfg)
    layer = E.Layer('c', 'conv', k, segs, cout, padding, True)
    p = {'c': _rand_params(layer, rng, dtype)}
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    srcs, xs = [], []
    for i, c in enumerate(segs):
        a = net.act(H + 2, W + 3, c)
        full = U.round_dtype(rng.standard_normal((B, a.H, a.W, c)), dtype)
        U.fill_act(a, full)
        srcs.append((a, 1, 2 - i)); xs.append(full[:, 1:1 + H, 2 - i:2 - i + W, :])
    x = np.concatenate(xs, -1)
    Ho, Wo = H + 2 * layer.pad - k + 1, W + 2 * layer.pad - k + 1
    dzv = U.round_dtype(rng.standard_normal((B, Ho, Wo, cout)) * 0.5, dtype)
    dz = net.act(Ho, Wo, cout); U.fill_act(dz, dzv)
    store.g.fill_(float('nan'))
    bplan = E.Plan('b')
    net.conv_bwd(bplan, layer, srcs, H, W, dz, [None] * len(segs), wcfg=wcfg)
    net.flush_reduce(bplan)
    name = bplan.kernel_name(0)
    if wcfg in (11, 14):
        assert ',4,1,1,4,' in name, name              # (wcfg 0: the cost model picks layout and tile by the amount of work)
    bplan.run(U.stream()); U.sync()
    dw_ref, db_ref = ops.conv2d_wgrad(x, dzv, (k, k), padding, 1)
    g = store.get_grads()['c']
    assert np.isfinite(g['weights']).all() and np.isfinite(g['biases']).all()
    assert U.rel_err(g['weights'], dw_ref) < U.tol_sum(dtype), 'wgrad ' + name
    assert U.rel_err(g['biases'], db_ref) < U.tol_sum(dtype), 'bias grad ' + name
    # same bits on a second run (fixed reduction order)
    g1 = store.g.clone(); store.g.fill_(0); bplan.run(U.stream()); U.sync()
    assert torch.equal(g1, store.g)


@pytest.mark.parametrize('case', [
    # padding, segs, cout, H, W, B, wcfg (200 + 10 * tap groups + window class; 0 = automatic), ksplit (0 = automatic)
    ('VALID', [64], 64, 37, 35, 3, 210, 0),             # 64 ci x 64 co, all 9 taps, 4-K-step windows, 3-stage ring
    ('VALID', [64], 64, 37, 35, 3, 211, 0),             # ... 8-K-step windows, 2-stage ring
    ('VALID', [64], 64, 37, 35, 3, 230, 1),             # one filter row per workgroup, every workgroup sweeps all pixels (no slabs)
    ('VALID', [64], 64, 37, 35, 3, 291, 1),             # one tap per workgroup
    ('VALID', [64], 64, 37, 35, 3, 290, 3),             # one tap per workgroup AND a K split (slabs + reduction)
    ('SAME', [128], 64, 20, 44, 2, 210, 0),             # two X chunks, zero padding: every window is an edge window
    ('SAME', [128], 64, 20, 44, 2, 231, 2),
    ('VALID', [64, 64], 128, 26, 26, 2, 210, 0),        # channel-concat input, two dZ blocks
    ('VALID', [64], 64, 61, 59, 4, 210, 5),             # long window walk + K split + slab reduction
    ('VALID', [256], 256, 10, 10, 5, 211, 1),           # deep layer: 8 x 8 maps, several images per window, direct store
    ('VALID', [256], 256, 10, 10, 5, 291, 1),
    ('VALID', [128], 192, 12, 12, 3, 230, 1),           # whole 10 x 10 maps as windows (100 pixels -> 4 K steps)
    ('VALID', [128], 128, 16, 16, 2, 211, 1),           # 14 x 14 = 196 pixels: one window per image at 7 K steps
    ('VALID', [32], 64, 37, 35, 3, 211, 0),             # 32 ci x 64 co (forced: by default inputs in 32-channel chunks run on the register-staged kernel, r04)
    ('VALID', [96], 32, 19, 23, 2, 211, 0),             # 32-channel layouts on unpadded / ragged channel counts
    ('SAME', [32, 32], 64, 33, 17, 2, 211, 3),
    ('VALID', [64], 32, 41, 23, 2, 0, 0),               # 64 ci x 32 co
    ('VALID', [32], 32, 41, 23, 2, 211, 4),             # 32 x 32
    ('VALID', [48, 16], 40, 21, 19, 2, 211, 0),         # unpadded channel counts
    ('VALID', [32], 64, 37, 35, 3, 0, 0),               # ... and the default route of such layers: conv_wgrad_kernel, 8 x 16 tiles
    ('SAME', [32, 32], 32, 170, 165, 2, 0, 0),          # ... 16 x 16 tiles on big maps
    ('VALID', [64], 64, 37, 35, 3, 0, 0),               # automatic choice
    ('VALID', [512], 512, 10, 10, 4, 0, 0),             # automatic choice on a bottleneck layer
])
def test_wgrad_sweep(case):
    """The wave-specialised bf16 3x3 filter gradient (csrc/wgrad_sweep.hip) against the oracle: every tap grouping, both window
    classes, whole-image / multi-image / rectangular windows, edge windows, K splits with slab reduction, all four channel layouts."""
    padding, segs, cout, H, W, B, wcfg, ksplit = case
    k = 3
    dtype = L.SEG_BF16
    rng = np.random.default_rng(sum(segs) * 7 + cout * 3 + H + W + wcfg + ksplit)
    layer = E.Layer('c', 'conv', k, segs, cout, padding, True)
    p = {'c': _rand_params(layer, rng, dtype)}
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    srcs, xs = [], []
    for i, c in enumerate(segs):
        a = net.act(H + 2, W + 3, c)
        full = U.round_dtype(rng.standard_normal((B, a.H, a.W, c)), dtype)
        U.fill_act(a, full)
        srcs.append((a, 1, 2 - i)); xs.append(full[:, 1:1 + H, 2 - i:2 - i + W, :])
    x = np.concatenate(xs, -1)
    Ho, Wo = H + 2 * layer.pad - k + 1, W + 2 * layer.pad - k + 1
    dzv = U.round_dtype(rng.standard_normal((B, Ho, Wo, cout)) * 0.5, dtype)
    dz = net.act(Ho, Wo, cout); U.fill_act(dz, dzv)
    store.g.fill_(float('nan'))
    bplan = E.Plan('b')
    net.conv_bwd(bplan, layer, srcs, H, W, dz, [None] * len(segs), wcfg=wcfg, ksplit=ksplit)
    net.flush_reduce(bplan)
    name = bplan.kernel_name(0)
    if wcfg == 0 and any(E.rup(c) % 64 for c in segs):
        assert name.startswith('conv_wgrad_kernel<bf16,%s,3,3,1,2,2,1,1,' % ('16,16' if Ho * Wo >= 160 * 160 else '8,16')), name
    else:
        assert name.startswith('wgrad_sweep_kernel<'), name
    if wcfg >= 200:
        nu_nv = {1: ',3,3,', 3: ',1,3,', 9: ',1,1,'}[(wcfg - 200) // 10]
        assert nu_nv in name and name.endswith(',4,192>' if wcfg % 10 == 0 else ',8,352>'), name
    bplan.run(U.stream()); U.sync()
    dw_ref, db_ref = ops.conv2d_wgrad(x, dzv, (k, k), padding, 1)
    g = store.get_grads()['c']
    assert np.isfinite(g['weights']).all() and np.isfinite(g['biases']).all(), name
    assert U.rel_err(g['weights'], dw_ref) < U.tol_sum(dtype), 'wgrad ' + name
    assert U.rel_err(g['biases'], db_ref) < U.tol_sum(dtype), 'bias grad ' + name
    # same bits on a second run (fixed summation order), also from a poisoned workspace
    g1 = store.g.clone(); store.g.fill_(float('nan')); bplan.run(U.stream()); U.sync()
    assert torch.equal(g1, store.g), name


@pytest.mark.parametrize('cin,cout,H,W,padding,cfg', [(64, 64, 23, 37, 'VALID', 0), (32, 32, 34, 34, 'SAME', 0), (128, 96, 19, 50, 'VALID', 0),
                                                       (64, 64, 23, 37, 'VALID', 208), (128, 128, 34, 70, 'SAME', 204), (32, 64, 61, 59, 'VALID', 208)])
def test_conv_with_fused_maxpool(cin, cout, H, W, padding, cfg):
    """seg_conv_desc.pool: same activation bits as the plain launch, pooled map == 2x2 max-pool of those bits."""
    dtype = L.SEG_BF16
    rng = np.random.default_rng(cin + H)
    layer = E.Layer('c', 'conv', 3, [cin], cout, padding, True)
    p = {'c': _rand_params(layer, rng, dtype)}
    store = U.make_store([layer], dtype, p)
    B = 3
    net = E.Net(store, B, dtype, U.dev())
    x = net.act(H, W, cin); U.fill_act(x, U.round_dtype(rng.standard_normal((B, H, W, cin)), dtype))
    pad = layer.pad
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    y0 = net.act(Ho, Wo, cout); y1 = net.act(Ho, Wo, cout); pooled = net.act(Ho // 2, Wo // 2, cout)
    plan = E.Plan('t')
    net.conv_fwd(plan, layer, [(x, 0, 0)], H, W, y0, cfg=cfg)          # (the same kernel: another one sums the taps in another order)
    net.conv_fwd(plan, layer, [(x, 0, 0)], H, W, y1, pool=pooled, cfg=cfg)
    assert net.pool_fused and plan.ops[-1][0] == 'c+pool'
    if cfg:
        assert plan.kernel_name(1).startswith('conv_ring_kernel<') and plan.kernel_name(1).endswith('true>'), plan.kernel_name(1)
    plan.run(U.stream()); U.sync()
    assert torch.equal(y0.t, y1.t)
    o = y0.t.to(torch.float32)[:, :Ho // 2 * 2, :Wo // 2 * 2]
    want = torch.maximum(torch.maximum(o[:, 0::2, 0::2], o[:, 0::2, 1::2]), torch.maximum(o[:, 1::2, 0::2], o[:, 1::2, 1::2]))
    assert torch.equal(pooled.t.to(torch.float32), want)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('case', [(64, 32, 8, 8, 2), (32, 32, 36, 36, 1), (128, 64, 5, 7, 2), (24, 8, 6, 6, 1),
                                  # 44 / 20x44 maps pick the 8x16 stride-2 filter-gradient tile (upconv2 of the 512x512 U-Net): interior AND edge tiles
                                  (256, 128, 44, 44, 2), (64, 32, 20, 44, 3)])
def test_upconv_fwd_bwd(dtype, case):
    cin, cout, H, W, B = case
    rng = np.random.default_rng(cin * 1000 + H)
    layer = E.Layer('u', 'up', 2, [cin], cout, 'VALID', True)
    p = {'u': _rand_params(layer, rng, dtype)}
    p['u']['weights'] = U.round_dtype(p['u']['weights'], dtype).astype(np.float32)
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    xa = net.act(H, W, cin)
    xv = U.round_dtype(rng.standard_normal((B, H, W, cin)), dtype); U.fill_act(xa, xv)
    out = net.act(2 * H, 2 * W, cout)
    plan = E.Plan('t'); net.up_fwd(plan, layer, xa, H, W, out); plan.run(U.stream()); U.sync()
    ref = ops.conv2d_transpose(xv, p['u']['weights'], p['u']['biases'], 2, 'VALID', True)
    assert U.rel_err(U.read_act(out), ref) < U.tol(dtype), 'up fwd'
    assert U.pad_channels_zero(out)
    dzv = U.round_dtype(rng.standard_normal((B, 2 * H, 2 * W, cout)) * 0.5, dtype)
    dz = net.act(2 * H, 2 * W, cout); U.fill_act(dz, dzv)
    dx = net.act(H, W, cin)
    store.g.zero_()
    bplan = E.Plan('b'); net.up_bwd(bplan, layer, xa, H, W, dz, dx, xa); net.flush_reduce(bplan); bplan.run(U.stream()); U.sync()
    dw_ref, db_ref = ops.conv2d_transpose_wgrad(xv, dzv, (2, 2), 2, 'VALID')
    dx_ref = ops.conv2d_transpose_dgrad(dzv, p['u']['weights'], (H, W), 2, 'VALID') * (xv > 0)
    g = store.get_grads()['u']
    assert U.rel_err(g['weights'], dw_ref) < U.tol_sum(dtype), 'up wgrad'
    assert U.rel_err(g['biases'], db_ref) < U.tol_sum(dtype), 'up bias grad'
    assert U.rel_err(U.read_act(dx), dx_ref) < U.tol(dtype), 'up dgrad'


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('kind', ['conv3', 'conv1', 'up2'])
@pytest.mark.parametrize('wcfg', [3, 6, 9, 11, 12, 14, 15])
def test_wgrad_every_instance_on_a_long_tile_walk(dtype, kind, wcfg):
    """Every instantiated filter-gradient layout, both dtypes, with few K splits so that each workgroup walks > 10 tiles:
    the kernel keeps one or two tiles of asm-issued global loads in flight across its MFMA section and waits for them by
    hand, so a register-allocation accident (a staging register parked elsewhere before the wait -- seen once with a
    deeper LDS look-ahead in the f32 256-pixel kernel) shows up as a garbage gradient in exactly this situation."""
    if wcfg >= 9 and dtype != L.SEG_BF16:
        pytest.skip('bf16-only layouts')
    B, H, W, cin, cout = 2, 70, 66, 64, 128
    rng = np.random.default_rng(wcfg * 17 + len(kind))
    store_g = None
    if kind == 'up2':
        layer = E.Layer('u', 'up', 2, [cin], cout, 'VALID', True)
        p = {'u': _rand_params(layer, rng, dtype)}
        store = U.make_store([layer], dtype, p)
        net = E.Net(store, B, dtype, U.dev())
        Hs, Ws = H // 2, W // 2
        xa = net.act(Hs, Ws, cin); xv = U.round_dtype(rng.standard_normal((B, Hs, Ws, cin)), dtype); U.fill_act(xa, xv)
        dzv = U.round_dtype(rng.standard_normal((B, 2 * Hs, 2 * Ws, cout)) * 0.5, dtype)
        dz = net.act(2 * Hs, 2 * Ws, cout); U.fill_act(dz, dzv)
        dx = net.act(Hs, Ws, cin)
        bplan = E.Plan('b')
        try:
            net.up_bwd(bplan, layer, xa, Hs, Ws, dz, dx, xa, wcfg=wcfg, ksplit=2)
        except L.SegError as e:
            pytest.skip('layout not offered for this kernel: %s' % e)
        dw_ref, db_ref = ops.conv2d_transpose_wgrad(xv, dzv, (2, 2), 2, 'VALID')
        key = 'u'
    else:
        k = 3 if kind == 'conv3' else 1
        layer = E.Layer('c', 'conv', k, [cin], cout, 'VALID', True)
        p = {'c': _rand_params(layer, rng, dtype)}
        store = U.make_store([layer], dtype, p)
        net = E.Net(store, B, dtype, U.dev())
        a = net.act(H, W, cin); xv = U.round_dtype(rng.standard_normal((B, H, W, cin)), dtype); U.fill_act(a, xv)
        Ho, Wo = H - k + 1, W - k + 1
        dzv = U.round_dtype(rng.standard_normal((B, Ho, Wo, cout)) * 0.5, dtype)
        dz = net.act(Ho, Wo, cout); U.fill_act(dz, dzv)
        bplan = E.Plan('b')
        try:
            net.conv_bwd(bplan, layer, [(a, 0, 0)], H, W, dz, [None], wcfg=wcfg, ksplit=2)
        except L.SegError as e:
            pytest.skip('layout not offered for this kernel: %s' % e)
        dw_ref, db_ref = ops.conv2d_wgrad(xv, dzv, (k, k), 'VALID', 1)
        key = 'c'
    net.flush_reduce(bplan)
    store.g.fill_(float('nan'))
    bplan.run(U.stream()); U.sync()
    g = store.get_grads()[key]
    name = bplan.kernel_name(0)
    assert np.isfinite(g['weights']).all() and np.isfinite(g['biases']).all(), name
    assert U.rel_err(g['weights'], dw_ref) < U.tol_sum(dtype, 5e-5), 'wgrad ' + name
    assert U.rel_err(g['biases'], db_ref) < U.tol_sum(dtype, 5e-5), 'bias grad ' + name


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('k,stride,padding,cin,cout,H', [(5, 2, 'SAME', 3, 64, 64), (5, 2, 'SAME', 3, 32, 37), (5, 2, 'SAME', 1, 40, 50), (3, 2, 'SAME', 3, 16, 33),
                                                          (3, 1, 'VALID', 2, 64, 21), (7, 2, 'SAME', 3, 64, 70), (5, 2, 'VALID', 3, 64, 41)])
def test_conv_first_gen(k, stride, padding, cin, cout, H, relu):
    """seg_conv_first_gen (the DeconvModel's conv1_0, models/deconvolution.py:44-46: 5x5 / stride 2 SAME on the raw image) against the
    oracle's convolution; the layer keeps the filter as the 1x1 layer over the im2col that the filter gradient uses."""
    dtype = L.SEG_BF16
    B, W = 2, H + 5
    rng = np.random.default_rng(k * 100 + cout + H)
    layer = E.Layer('f', 'conv', 1, [k * k * cin], cout, 'VALID', relu)
    w = (rng.standard_normal((k, k, cin, cout)) * 0.2).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    p = {'f': {'weights': w.reshape(1, 1, k * k * cin, cout), 'biases': b}}
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    x = rng.uniform(-1, 1, (B, H, W, cin)).astype(np.float32)
    xt = torch.from_numpy(x).to(U.dev())
    Ho, pt = ops.conv_out_size(H, k, stride, padding)
    Wo, pl = ops.conv_out_size(W, k, stride, padding)
    out = net.act(Ho, Wo, cout)
    out.t.fill_(float('nan'))
    plan = E.Plan('t'); net.first_gen_fwd(plan, layer, xt, H, W, cin, k, k, stride, pt, pl, out); plan.run(U.stream()); U.sync()
    ref = ops.conv2d(U.round_dtype(x, dtype), U.round_dtype(w, dtype), b, padding, stride, relu)
    got = U.read_act(out)
    assert np.isfinite(got).all()
    assert U.rel_err(got, ref) < 6e-3                  # bf16 operands, f32 accumulation, one rounding of the result
    assert U.pad_channels_zero(out)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('nc,H,W,B', [(2, 21, 30, 2), (4, 16, 9, 3), (7, 12, 25, 1)])
def test_thin_conv_normalises_its_source_on_load(dtype, nc, H, W, B):
    """seg_thin_conv3x3_bn / seg_thin_wgrad3x3_bn (DeconvModel bn8 -> conv_out, models/deconvolution.py:168-170): the convolution and its
    filter gradient read the pre-batch-norm activation and normalise on load -- bit-identical logits and filter gradient to running
    seg_bn_fwd first and reading its output, which therefore need not exist."""
    rng = np.random.default_rng(nc * 31 + H)
    cv = E.Layer('c', 'conv', 3, [nc], nc, 'SAME', False)
    bnl = E.Layer('bn', 'bn', 1, [nc], nc); bnl.cout_p = 8
    p = {'c': _rand_params(cv, rng, dtype), 'bn': {'beta': (rng.standard_normal(nc) * 0.2).astype(np.float32)}}
    store = U.make_store([cv, bnl], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    av = np.maximum(U.round_dtype(rng.standard_normal((B, H, W, nc)) + 0.2, dtype), 0)
    a = net.act(H, W, nc, thin=True); U.fill_act(a, av)
    dzv = U.round_dtype(rng.standard_normal((B, H, W, nc)), dtype)
    dz = net.act(H, W, nc, thin=True); U.fill_act(dz, dzv)
    res = []
    for fused in (False, True):
        st = net.bn_state(bnl)
        lg = net.act(H, W, nc, f32=True, thin=True); lg.t.fill_(float('nan'))
        da = net.act(H, W, nc, thin=True)
        plan = E.Plan('t')
        store.g.fill_(float('nan'))
        if fused:
            net.bn_stats(plan, bnl, st, a, training=True, update_moving=True)
            net.conv_fwd(plan, cv, [(a, 0, 0)], H, W, lg, out_f32=True, src_bn=(st, bnl))
            net.conv_bwd(plan, cv, [(a, 0, 0)], H, W, dz, [(da, (0, 0), None, (0, 0))], src_bn=(st, bnl))
        else:
            y = net.act(H, W, nc, thin=True)
            net.bn_fwd(plan, bnl, st, a, y, training=True, update_moving=True)
            net.conv_fwd(plan, cv, [(y, 0, 0)], H, W, lg, out_f32=True)
            net.conv_bwd(plan, cv, [(y, 0, 0)], H, W, dz, [(da, (0, 0), None, (0, 0))])
        net.flush_reduce(plan)
        plan.run(U.stream()); U.sync()
        g = store.get_grads()['c']
        res.append((lg.t.clone(), st['stats'].clone(), st['moving'].clone(), da.t.clone(), g['weights'].copy(), g['biases'].copy()))
    for x0, x1 in zip(res[0][:4], res[1][:4]):
        assert torch.equal(x0, x1)
    assert np.isfinite(res[1][4]).all() and np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])


@pytest.mark.parametrize('padding,cin,cout,H,B', [('SAME', 3, 32, 64, 3), ('SAME', 3, 64, 37, 2), ('SAME', 1, 20, 50, 4), ('VALID', 3, 32, 41, 2), ('SAME', 2, 40, 130, 1)])
def test_conv_first_gen_wgrad(padding, cin, cout, H, B):
    """seg_conv_first_gen_wgrad (filter + bias gradient of the DeconvModel's conv1_0 straight from the image, models/deconvolution.py:
    44-46) against the oracle's Conv2DBackpropFilter on the same bf16-rounded operands."""
    dtype = L.SEG_BF16
    W = H + 7
    rng = np.random.default_rng(cout + H)
    layer = E.Layer('f', 'conv', 1, [25 * cin], cout, 'VALID', True)
    p = {'f': {'weights': np.zeros((1, 1, 25 * cin, cout), np.float32), 'biases': np.zeros(cout, np.float32)}}
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    x = rng.uniform(-1, 1, (B, H, W, cin)).astype(np.float32)
    xt = torch.from_numpy(x).to(U.dev())
    Ho, pt = ops.conv_out_size(H, 5, 2, padding); Wo, pl = ops.conv_out_size(W, 5, 2, padding)
    dzv = U.round_dtype(rng.standard_normal((B, Ho, Wo, cout)), dtype)
    dz = net.act(Ho, Wo, cout); U.fill_act(dz, dzv)
    store.g.fill_(float('nan'))
    plan = E.Plan('b'); net.first_gen_bwd(plan, layer, xt, H, W, cin, 5, 5, 2, pt, pl, dz); plan.run(U.stream()); U.sync()
    dw_ref, db_ref = ops.conv2d_wgrad(U.round_dtype(x, dtype), dzv, (5, 5), padding, 2)
    g = store.get_grads()['f']
    got = g['weights'].reshape(5, 5, cin, cout)
    assert np.isfinite(got).all() and np.isfinite(g['biases']).all()
    assert U.rel_err(got, dw_ref) < 1e-4                     # exact products of bf16 operands, f32 accumulation: only the summation order differs
    assert U.rel_err(g["biases"], db_ref) < 1e-4


@pytest.mark.parametrize('which,cout,H,B', [('first', 32, 64, 3), ('first', 64, 37, 2), ('first', 20, 130, 4), ('up', 2, 40, 3), ('up', 5, 23, 2), ('up', 8, 150, 2)])
def test_batch_norm_statistics_from_the_producing_launch(which, cout, H, B):
    """seg_conv_first_gen_bn / seg_thin_up2x2_bn + seg_bn_fwd_rows (the DeconvModel's conv1_0 -> bn1 and deconv3_0 -> bn8,
    models/deconvolution.py:44-50,166-168): the same activation bits, the same statistics (another summation order) and the same
    normalised output as the producer followed by the three-stage seg_bn_fwd."""
    dtype = L.SEG_BF16
    rng = np.random.default_rng(cout * 7 + H)
    W = H + 6
    thin = which == 'up'
    if which == 'first':
        prod = E.Layer('f', 'conv', 1, [75], cout, 'VALID', True)
        w = (rng.standard_normal((1, 1, 75, cout)) * 0.2).astype(np.float32)
        Ho, pt = ops.conv_out_size(H, 5, 2, 'SAME'); Wo, pl = ops.conv_out_size(W, 5, 2, 'SAME')
    else:
        prod = E.Layer('f', 'up', 2, [32], cout, 'VALID', True)
        w = (rng.standard_normal((2, 2, cout, 32)) * 0.2).astype(np.float32)
        Ho, Wo = 2 * H, 2 * W
    bnl = E.Layer('bn', 'bn', 1, [cout], cout)
    if thin:
        bnl.cout_p = 8
    p = {'f': {'weights': w, 'biases': (rng.standard_normal(cout) * 0.1).astype(np.float32)}, 'bn': {'beta': (rng.standard_normal(cout) * 0.1).astype(np.float32)}}
    store = U.make_store([prod, bnl], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    if which == 'first':
        xt = torch.from_numpy(rng.uniform(-1, 1, (B, H, W, 3)).astype(np.float32)).to(U.dev())
    else:
        xs = net.act(H, W, 32); U.fill_act(xs, U.round_dtype(rng.standard_normal((B, H, W, 32)), dtype))
    outs = []
    for fused in (True, False):
        a = net.act(Ho, Wo, cout, thin=thin); a.t.fill_(float('nan'))
        y = net.act(Ho, Wo, cout, thin=thin)
        st = net.bn_state(bnl)
        st['ws'].fill_(float('nan'))
        plan = E.Plan('t')
        if which == 'first':
            rows = net.first_gen_fwd(plan, prod, xt, H, W, 3, 5, 5, 2, pt, pl, a, bn_st=st if fused else None)
        else:
            rows = net.up_fwd(plan, prod, xs, H, W, a, bn_st=st if fused else None)
        assert (rows > 0) == fused
        net.bn_fwd(plan, bnl, st, a, y, training=True, update_moving=True, rows=rows)
        assert [o[0] for o in plan.ops] == ['f', 'bn']
        plan.run(U.stream()); U.sync()
        outs.append((a.t.clone(), y.t.float().cpu().numpy(), st['stats'].cpu().numpy().copy(), st['moving'].cpu().numpy().copy()))
    (a1, y1, s1, m1), (a0, y0, s0, m0) = outs
    assert torch.equal(a1, a0)
    assert np.isfinite(s1).all() and np.isfinite(y1).all()
    Cp = s0.size // 2
    assert np.abs(s1[:Cp] - s0[:Cp]).max() < 1e-5 * max(1.0, np.abs(s0[:Cp]).max())          # means
    assert np.abs(s1[Cp:] / s0[Cp:] - 1).max() < 1e-4                                        # 1 / sqrt(var + eps)
    assert np.abs(m1 - m0).max() < 1e-6
    assert np.abs(y1 - y0).max() < 2e-2 * max(1.0, np.abs(y0).max())


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('impl,relu', [('win', True), ('old', True), ('win', False), ('old', False)])
@pytest.mark.parametrize('pad,cin,cout,H', [(0, 3, 32, 21), (1, 3, 16, 18), (0, 1, 40, 33), (1, 2, 64, 20), (0, 3, 32, 64)])
def test_conv_first(dtype, pad, cin, cout, H, impl, relu, monkeypatch):
    if dtype != L.SEG_BF16 and (impl == 'old' or not relu):
        pytest.skip('the two MFMA forms of the first layer are bf16 kernels')
    monkeypatch.setenv('SEG_FIRST_IMPL', impl)         # bf16: the bf16-staged window kernel / the float-staged one
    L.load().seg_dbg_reload_env()                      # (the library reads its switches once)
    B, W = 2, H + 3
    rng = np.random.default_rng(cout + H)
    layer = E.Layer('f', 'first', 3, [cin], cout, 'VALID' if pad == 0 else 'SAME', relu)
    p = {'f': _rand_params(layer, rng, dtype)}
    store = U.make_store([layer], dtype, p)
    net = E.Net(store, B, dtype, U.dev())
    x = rng.uniform(0, 1, (B, H, W, cin)).astype(np.float32)
    xt = torch.from_numpy(x).to(U.dev())
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    out = net.act(Ho, Wo, cout)
    plan = E.Plan('t'); net.first_fwd(plan, layer, xt, H, W, out); plan.run(U.stream()); U.sync()
    ref = ops.conv2d(x, p['f']['weights'], p['f']['biases'], layer.padding, 1, relu)
    assert U.rel_err(U.read_act(out), ref) < U.tol(dtype, 2e-5, 1e-2)
    assert U.pad_channels_zero(out)
    if dtype == L.SEG_BF16 and cin <= 3 and cout <= 64:
        # fused conv + 2x2 max-pool: same activation bits, pooled map == max-pool of those bits (first-max irrelevant for values)
        out2 = net.act(Ho, Wo, cout); pooled = net.act(Ho // 2, Wo // 2, cout)
        pl = E.Plan('p'); assert net.first_fwd(pl, layer, xt, H, W, out2, pool=pooled); pl.run(U.stream()); U.sync()
        assert torch.equal(out2.t, out.t)
        o = out.t.to(torch.float32)[:, :Ho // 2 * 2, :Wo // 2 * 2]
        want = torch.maximum(torch.maximum(o[:, 0::2, 0::2], o[:, 0::2, 1::2]), torch.maximum(o[:, 1::2, 0::2], o[:, 1::2, 1::2]))
        assert torch.equal(pooled.t.to(torch.float32), want)
    if cin <= 3 and relu and impl == 'win':            # (the filter gradient does not depend on the forward kernel's form)
        dzv = U.round_dtype(rng.standard_normal((B, Ho, Wo, cout)), dtype)
        dz = net.act(Ho, Wo, cout); U.fill_act(dz, dzv)
        store.g.zero_()
        bp = E.Plan('b'); net.first_bwd(bp, layer, xt, H, W, dz); net.flush_reduce(bp); bp.run(U.stream()); U.sync()
        # bf16 mode rounds the im2col'd input to bf16 (the MFMA operand type): the oracle gets the same rounded image, so the
        # filter gradient is an f32 sum of exact products
        dw_ref, db_ref = ops.conv2d_wgrad(U.round_dtype(x, dtype), dzv, (3, 3), layer.padding, 1)
        g = store.get_grads()['f']
        assert U.rel_err(g['weights'], dw_ref) < U.tol_sum(dtype)
        assert U.rel_err(g['biases'], db_ref) < U.tol_sum(dtype)
        assert [o[0] for o in bp.ops if o[1] is not None][0] == 'f/dw'          # no im2col launch: the rows are gathered while staging
        store.g.zero_()
        bl = E.Plan('bl'); net.first_bwd(bl, layer, xt, H, W, dz, ksplit=2); net.flush_reduce(bl); bl.run(U.stream()); U.sync()     # long tile walks
        gl = store.get_grads()['f']
        assert U.rel_err(gl['weights'], dw_ref) < U.tol_sum(dtype) and U.rel_err(gl['biases'], db_ref) < U.tol_sum(dtype)
        # the explicit form (im2col tensor + the same 1x1 walk): the same rounded operands, another summation order at most
        monkeypatch.setenv('SEG_FIRST_IM2COL', '1')
        store.g.zero_()
        bp2 = E.Plan('b2'); col = net.first_im2col(bp2, layer, xt, H, W); assert col is not None
        net.first_bwd(bp2, layer, xt, H, W, dz, col=col); net.flush_reduce(bp2); bp2.run(U.stream()); U.sync()
        g2 = store.get_grads()['f']
        assert U.rel_err(g2['weights'], g['weights']) < 2e-5 and U.rel_err(g2['biases'], g['biases']) < 2e-5
        monkeypatch.delenv('SEG_FIRST_IM2COL')
        if net.fuses_first_pool_bwd():
            # dZ rebuilt inside the kernel from the 2x2 max-pool that consumes the layer (+ a second consumer's gradient in a
            # window, as the U-Net's conv1_2) against the separate pool backward launch feeding the same filter gradient
            Hp, Wp = Ho // 2, Wo // 2
            dpv = U.round_dtype(rng.standard_normal((B, Hp, Wp, cout)), dtype)
            dpa = net.act(Hp, Wp, cout); U.fill_act(dpa, dpv)
            # (ksplit 2: every workgroup walks many tiles -- the staging registers are reused from tile to tile)
            for add_on, ks in ((False, 0), (True, 0), (False, 2), (True, 2)):
                ah, aw, ay0, ax0 = (max(Ho - 7, 1), max(Wo - 5, 1), 3, 2) if add_on else (0, 0, 0, 0)
                adda = None
                if add_on:
                    adda = net.act(ah, aw, cout); U.fill_act(adda, U.round_dtype(rng.standard_normal((B, ah, aw, cout)), dtype))
                dzr = net.act(Ho, Wo, cout)
                store.g.zero_()
                pr = E.Plan('pr'); net.pool_bwd(pr, out, dpa, adda, (ah, aw), (ay0, ax0), dzr, Ho, Wo)
                net.first_bwd(pr, layer, xt, H, W, dzr, ksplit=ks); net.flush_reduce(pr); pr.run(U.stream()); U.sync()
                gr = store.get_grads()['f']
                store.g.zero_()
                pf = E.Plan('pf'); net.first_bwd(pf, layer, xt, H, W, None, pool=(out, dpa, adda, (ah, aw), (ay0, ax0)), ksplit=ks)
                net.flush_reduce(pf); pf.run(U.stream()); U.sync()
                gf = store.get_grads()['f']
                assert [o[0] for o in pf.ops if o[1] is not None][0] == 'f/dw'
                assert U.rel_err(gf['weights'], gr['weights']) < 2e-5 and U.rel_err(gf['biases'], gr['biases']) < 2e-5, (add_on, ks)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('H,W,Ch', [(13, 11, 32), (8, 8, 64), (7, 9, 16)])
def test_maxpool_fwd_bwd(dtype, H, W, Ch):
    B = 2
    rng = np.random.default_rng(H * 100 + Ch)
    lib = L.load()
    net = E.Net(None, B, dtype, U.dev())
    ya = net.act(H, W, Ch)
    yv = np.maximum(U.round_dtype(rng.standard_normal((B, H, W, Ch)), dtype), 0)      # post-ReLU, many exact-zero ties
    U.fill_act(ya, yv)
    Ho, Wo = H // 2, W // 2
    pa = net.act(Ho, Wo, Ch)
    idx = torch.zeros((B, Ho, Wo, ya.Cp), dtype=torch.uint8, device=U.dev())
    sv, dv = ya.view(), pa.view()
    L.check(lib.seg_maxpool2x2_fwd(C.byref(sv), C.byref(dv), idx.data_ptr(), B, Ho, Wo, ya.Cp, dtype, U.stream()))
    U.sync()
    pref, iref = ops.max_pool2x2(yv)
    assert np.array_equal(U.read_act(pa), pref)
    assert np.array_equal(idx.cpu().numpy()[..., :Ch], iref)
    # backward with a skip-gradient add window
    dpv = U.round_dtype(rng.standard_normal((B, Ho, Wo, Ch)), dtype)
    dpa = net.act(Ho, Wo, Ch); U.fill_act(dpa, dpv)
    ah, aw, ay, ax = 4, 5, 2, 1
    addv = U.round_dtype(rng.standard_normal((B, ah, aw, Ch)), dtype)
    adda = net.act(ah, aw, Ch); U.fill_act(adda, addv)
    dza = net.act(H, W, Ch)
    plan = E.Plan('b'); net.pool_bwd(plan, ya, dpa, adda, (ah, aw), (ay, ax), dza, H, W); plan.run(U.stream()); U.sync()
    ref = ops.max_pool2x2_bwd(dpv, iref, (H, W))
    pad = np.zeros_like(ref); pad[:, ay:ay + ah, ax:ax + aw, :] = addv
    ref = (ref + pad) * (yv > 0)
    got = U.read_act(dza)
    assert U.rel_err(got, ref) < U.tol(dtype, 1e-6, 1e-2)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('nc', [2, 4, 21])
def test_softmax_xent_and_sigmoid_argmax(dtype, nc):
    B, H, W, LH, LW = 2, 9, 7, 15, 12
    rng = np.random.default_rng(nc)
    lib = L.load()
    net = E.Net(None, B, dtype, U.dev())
    lg = net.act(H, W, nc, f32=True)
    z = (rng.standard_normal((B, H, W, nc)) * 3).astype(np.float32)
    z[0, 0, 0, :] = 25.0           # saturated tie -> index 0
    z[0, 0, 1, -1] = 30.0; z[0, 0, 1, 0] = 18.0
    U.fill_act(lg, z)
    labels = rng.integers(0, nc, (B, LH, LW)).astype(np.uint8)
    lt = torch.from_numpy(labels).to(U.dev())
    loss = torch.zeros(1, dtype=torch.float32, device=U.dev())
    dl = net.act(H, W, nc)
    plan = E.Plan('x'); net.softmax_xent(plan, lg, lt, LH, LW, (3, 2), H, W, nc, loss, dl); plan.run(U.stream()); U.sync()
    lref, _, dref = ops.softmax_xent(z, labels[:, 3:3 + H, 2:2 + W])
    assert abs(float(loss.item()) - lref) < 1e-5 * max(1, abs(lref))
    assert U.rel_err(U.read_act(dl), dref) < U.tol(dtype, 1e-5, 1e-2)
    assert U.pad_channels_zero(dl)
    sig = torch.zeros((B, H, W, nc), dtype=torch.float32, device=U.dev())
    out = torch.zeros((B, H, W, 1), dtype=torch.float32, device=U.dev())
    p2 = E.Plan('s'); net.sigmoid_argmax(p2, lg, H, W, nc, sig, out); p2.run(U.stream()); U.sync()
    sref, oref = ops.sigmoid_argmax(z)
    assert np.array_equal(sig.cpu().numpy(), sref)          # bit-exact float32 sigmoid
    assert np.array_equal(out.cpu().numpy(), oref)          # bit-exact argmax incl. saturated ties
    assert out[0, 0, 0, 0].item() == 0.0


def test_batched_slab_reduction_is_bitwise_the_per_layer_one():
    """seg_wgrad_reduce_batch (one launch for several layers) vs phase-2 launches per layer: identical bits,
    including layers whose wgrad stores directly (ksplit 1) and ragged channel counts (scalar form)."""
    dtype = L.SEG_BF16
    rng = np.random.default_rng(99)
    specs = [('a', 3, [32], 64, 40), ('b', 3, [64], 64, 38), ('c', 3, [24], 10, 21), ('d', 3, [512], 512, 6), ('e', 1, [64], 4, 30)]
    layers = [E.Layer(n, 'conv', k, segs, cout, 'VALID', True) for n, k, segs, cout, H in specs]
    p = {l.name: _rand_params(l, rng, dtype) for l in layers}
    store = U.make_store(layers, dtype, p)
    B = 3
    got = []
    for batched in (True, False):
        net = E.Net(store, B, dtype, U.dev())
        net.batch_reduce = batched
        rs = np.random.default_rng(5)
        plan = E.Plan('b')
        for l, (n, k, segs, cout, H) in zip(layers, specs):
            x = net.act(H, H, segs[0]); U.fill_act(x, rs.standard_normal((B, H, H, segs[0])))
            dz = net.act(H - k + 1, H - k + 1, cout); U.fill_act(dz, rs.standard_normal((B, H - k + 1, H - k + 1, cout)))
            net.conv_bwd(plan, l, [(x, 0, 0)], H, H, dz, [None])
        net.flush_reduce(plan)
        names = [o[0] for o in plan.ops]
        assert any(n.startswith('dw/reduce[') for n in names) == batched
        store.g.fill_(float('nan'))
        plan.run(U.stream(), flavor='batched' if batched else 'per_layer'); U.sync()
        assert bool(torch.isfinite(store.g).all())
        got.append(store.g.clone())
    assert torch.equal(got[0], got[1])


@pytest.mark.parametrize('dtype', DT)
def test_pack_of_several_layers_equals_packing_each_alone(dtype):
    """The re-pack launch owns four 32x32 tiles per block and finds each tile's table entry itself: a store of several
    layers whose tile count is not a multiple of four must give, slice by slice, the bytes each layer packs to alone."""
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(11)
    specs = [('a', 'conv', 3, [32], 32), ('b', 'conv', 1, [64], 32), ('u', 'up', 2, [64], 32), ('c', 'conv', 3, [32, 32], 64), ('d', 'conv', 1, [32], 32), ('e', 'conv', 3, [32], 32)]

    def mk(spec):
        name, kind, k, cin, cout = spec
        return E.Layer(name, kind, k, cin if kind == 'conv' else cin, cout, 'VALID', True)

    def packed_of(layers, params):
        store = E.ParamStore(layers, dtype, dev, training=True)
        store.set_params({l.name: params[l.name] for l in layers})
        net = E.Net(store, 1, dtype, dev)
        plan = E.Plan('pack'); net.pack(plan); plan.run(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return store, store.packed.float().cpu().numpy()

    layers = [mk(sp) for sp in specs]
    params = {l.name: _rand_params(l, rng, dtype) for l in layers}
    store, whole = packed_of(layers, params)
    assert store.pack_blocks % 4 != 0, 'the case must leave a partial block of tiles'
    for sp in specs:
        one = mk(sp)
        s1, alone = packed_of([one], params)
        full = store.layers[sp[0]]
        for attr in ('pk_fwd', 'pk_dgrad'):
            o1 = getattr(one, attr, None)
            if o1 is None:
                continue
            o = getattr(full, attr)
            nxt = min([x for l in s1.layers.values() for x in (getattr(l, 'pk_fwd', None), getattr(l, 'pk_dgrad', None)) if x is not None and x > o1] + [alone.size])
            n = nxt - o1
            assert np.array_equal(whole[o:o + n], alone[o1:o1 + n]), (sp[0], attr)


def test_adam_matches_tf_variant():
    n = 1003
    rng = np.random.default_rng(1)
    lib = L.load()
    p0, g0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    p = torch.from_numpy(p0.copy()).to(U.dev()); g = torch.from_numpy(g0).to(U.dev())
    m = torch.zeros(n, device=U.dev()); v = torch.zeros(n, device=U.dev())
    step = torch.zeros(2, dtype=torch.int64, device=U.dev())      # {global_step, completed}; Adam reads &step[1]
    loss = torch.full((1,), 7.0, device=U.dev())
    pr, mr, vr = p0.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in (1, 2, 3):
        L.check(lib.seg_step_begin(step.data_ptr(), loss.data_ptr(), U.stream()))
        L.check(lib.seg_adam(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 0.5, step.data_ptr() + 8, U.stream()))
        pr, mr, vr = ops.adam_tf(pr, g0.astype(np.float64) * 0.5, mr, vr, t, lr=1e-3)
    L.check(lib.seg_step_increment(step.data_ptr() + 8, U.stream()))      # the stand-alone assign_add still works
    U.sync()
    assert step.tolist() == [3, 3] and float(loss.item()) == 0.0
    assert np.allclose(p.cpu().numpy(), pr, atol=1e-6)
    assert np.allclose(m.cpu().numpy(), mr, atol=1e-6) and np.allclose(v.cpu().numpy(), vr, atol=1e-7)


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('f,Hs,Hd,add', [(2, 4, 8, True), (8, 4, 30, False), (2, 5, 9, True), (8, 3, 26, False)])
def test_bilinear_up(dtype, f, Hs, Hd, add):
    B, Cc = 2, 5
    rng = np.random.default_rng(f * 10 + Hs)
    lib = L.load()
    net = E.Net(None, B, dtype, U.dev())
    sa = net.act(Hs, Hs, Cc); sv = U.round_dtype(rng.standard_normal((B, Hs, Hs, Cc)), dtype); U.fill_act(sa, sv)
    filt = ops.upsample_filt(ops.get_kernel_size(f)).astype(np.float32)
    ft = torch.from_numpy(filt).to(U.dev())
    full = ops.conv2d_transpose(sv, ops.bilinear_upsample_weights(f, Cc), None, f, 'SAME')
    ref = ops.crop_or_pad(full, Hd, Hd)
    cy = (Hs * f - Hd) // 2 if Hs * f >= Hd else -((Hd - Hs * f) // 2)
    da = net.act(Hd, Hd, Cc, f32=not add)
    s_v, d_v = sa.view(), da.view()
    if add:
        aa = net.act(Hd, Hd, Cc); av = U.round_dtype(rng.standard_normal((B, Hd, Hd, Cc)), dtype); U.fill_act(aa, av)
        a_v = aa.view(); ref = ref + av
        L.check(lib.seg_bilinear_up_fwd(C.byref(s_v), Hs, Hs, f, ft.data_ptr(), C.byref(a_v), C.byref(d_v), Hd, Hd, cy, cy, B, sa.Cp, 0, dtype, U.stream()))
    else:
        L.check(lib.seg_bilinear_up_fwd(C.byref(s_v), Hs, Hs, f, ft.data_ptr(), None, C.byref(d_v), Hd, Hd, cy, cy, B, sa.Cp, 1, dtype, U.stream()))
    U.sync()
    assert U.rel_err(U.read_act(da), ref) < U.tol(dtype, 1e-6, 1e-2)
    # adjoint
    gv = U.round_dtype(rng.standard_normal((B, Hd, Hd, Cc)), dtype)
    ga = net.act(Hd, Hd, Cc); U.fill_act(ga, gv)
    dsa = net.act(Hs, Hs, Cc)
    g_v, ds_v = ga.view(), dsa.view()
    # ... with the gradient behind a ReLU as a second output (mask = the source activation here)
    dza = net.act(Hs, Hs, Cc); m_v, dz_v = sa.view(), dza.view()
    L.check(lib.seg_bilinear_up_bwd(C.byref(g_v), Hd, Hd, cy, cy, f, ft.data_ptr(), C.byref(ds_v), Hs, Hs, B, sa.Cp, 0, C.byref(m_v), C.byref(dz_v), dtype, U.stream()))
    U.sync()
    got_ds = U.read_act(dsa)
    assert np.array_equal(U.read_act(dza), got_ds * (U.read_act(sa) > 0))
    gref = ops.conv2d_transpose_dgrad(ops.crop_or_pad_bwd(gv, (Hs * f, Hs * f)), ops.bilinear_upsample_weights(f, Cc), (Hs, Hs), f, 'SAME')
    assert U.rel_err(U.read_act(dsa), gref) < U.tol(dtype, 1e-6, 1e-2)
    # the separable form of the adjoint (horizontal pass into a float workspace, then vertical)
    nb = int(lib.seg_bilinear_up_bwd_ws_bytes(B, Hd, Hs, sa.Cp))
    ws = torch.empty(nb // 4, dtype=torch.float32, device=U.dev())
    dsa.t.fill_(3.0)
    dza.t.fill_(5.0)
    L.check(lib.seg_bilinear_up_bwd_sep(C.byref(g_v), Hd, Hd, cy, cy, f, ft.data_ptr(), C.byref(ds_v), Hs, Hs, B, sa.Cp, 0, ws.data_ptr(), nb, C.byref(m_v), C.byref(dz_v), dtype, U.stream()))
    U.sync()
    assert U.rel_err(U.read_act(dsa), gref) < U.tol(dtype, 2e-6, 1e-2) and U.pad_channels_zero(dsa)
    assert np.array_equal(U.read_act(dza), U.read_act(dsa) * (U.read_act(sa) > 0)) and U.pad_channels_zero(dza)
    L.check(lib.seg_bilinear_up_bwd_sep(C.byref(g_v), Hd, Hd, cy, cy, f, ft.data_ptr(), C.byref(ds_v), Hs, Hs, B, sa.Cp, 0, ws.data_ptr(), nb, None, None, dtype, U.stream()))
    U.sync()


@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('f,Hs,Hd,nc,keep', [(8, 4, 30, 21, True), (8, 5, 40, 3, False), (32, 2, 60, 2, True), (16, 3, 50, 11, False)])
def test_bilinear_xent_fused_head(dtype, f, Hs, Hd, nc, keep):
    """seg_bilinear_xent = seg_bilinear_up_fwd (float logits) + seg_softmax_xent, in one launch"""
    B = 2
    rng = np.random.default_rng(f * 7 + nc)
    lib = L.load()
    net = E.Net(None, B, dtype, U.dev())
    sa = net.act(Hs, Hs, nc); sv = U.round_dtype(rng.standard_normal((B, Hs, Hs, nc)) * 2, dtype); U.fill_act(sa, sv)
    filt = ops.upsample_filt(ops.get_kernel_size(f)).astype(np.float32)
    ft = torch.from_numpy(filt).to(U.dev())
    logits_ref = ops.crop_or_pad(ops.conv2d_transpose(sv, ops.bilinear_upsample_weights(f, nc), None, f, 'SAME'), Hd, Hd)
    LH, LW, off = Hd + 3, Hd + 2, (2, 1)
    y = rng.integers(0, nc, (B, LH, LW, 1)).astype(np.uint8)
    yd = torch.from_numpy(y).to(U.dev())
    loss_ref, _, d_ref = ops.softmax_xent(logits_ref, y[:, off[0]:off[0] + Hd, off[1]:off[1] + Hd])
    cy = (Hs * f - Hd) // 2 if Hs * f >= Hd else -((Hd - Hs * f) // 2)
    dl = net.act(Hd, Hd, nc); dl.t.fill_(0.0)
    lo = net.act(Hd, Hd, nc, f32=True) if keep else None
    loss = torch.zeros(1, device=U.dev())
    s_v, d_v = sa.view(), dl.view()
    l_v = lo.view() if keep else None
    L.check(lib.seg_bilinear_xent(C.byref(s_v), Hs, Hs, f, ft.data_ptr(), cy, cy, yd.data_ptr(), LH, LW, off[0], off[1], B, Hd, Hd, nc,
                                  1.0 / (B * Hd * Hd), 1.0, loss.data_ptr(), C.byref(d_v), C.byref(l_v) if keep else None, dtype, U.stream()))
    U.sync()
    assert abs(float(loss) - loss_ref) < (1e-5 if dtype == L.SEG_F32 else 1e-4) * max(1.0, abs(loss_ref))
    assert np.abs(U.read_act(dl) - d_ref).max() < (1e-6 if dtype == L.SEG_F32 else 1e-2) * np.abs(d_ref).max() and U.pad_channels_zero(dl)
    if keep:
        assert np.abs(U.read_act(lo) - logits_ref).max() < 1e-5 * max(1.0, np.abs(logits_ref).max())


def test_error_paths():
    lib = L.load()
    d = L.ConvDesc()
    assert lib.seg_conv2d(C.byref(d), None) != 0
    assert b'null' in lib.seg_last_error()
    assert lib.seg_conv2d(None, None) != 0
    assert lib.seg_adam(None, None, None, None, 0, 0.0, 0.0, 0.0, 0.0, 0.0, None, None) != 0
