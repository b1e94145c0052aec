"""-m gpu parity tests AT THE BASELINE CONFIGS, full size and full batch (BASELINE.json configs[1..4]):

  C2  U-Net 256x256x3 -> 4-class, B=16, bf16, eager and hipGraph step modes (the bench.py headline workload)
  C3  FCN-8s 512x512 -> 21-class, B=8, n_kernels 32
  C4  U-Net 512x512 -> 4-class, B=16 (the per-GPU shard of the global-128 data-parallel config)
  C5  U-Net 256x256, B=32, 30 stochastic forward passes (build-defined MC dropout, a19)

Three checks per config (VERDICT r01 "next" item 1):
  (a) f32-mode HIP at full size vs the numpy oracle: logits <= 1e-4 (north_star), argmax bit-exact wherever the oracle's
      top-2 margin exceeds the logit tolerance; where the oracle is affordable (C2, C3: seconds) ALSO the full-batch loss and
      every gradient tensor; C4 checks the gradients of a one-image 512x512 step (same 256-pixel filter-gradient tiles);
  (b) bf16-mode HIP vs f32-mode HIP at full size and batch: logits, loss, every gradient tensor (relative L2 error and
      cosine), argmax pixel-disagreement rate and mIoU against the labels; measured values are written to
      gpurun_out/parity_measured.jsonl (committed copy: profiles/r02_parity_measured.jsonl) and the bounds below are
      ~2x the values measured on MI355X;
  (c) the kernel instances and K splits that run here are the ones bench.py times (same batch, size, dtype, step modes).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import np_ops as ops
from oracle import unet as ounet
from oracle import fcn as ofcn
from segmentation_amd.datasets import ArrayDataSet
from segmentation_amd.unet import UNetModel
from segmentation_amd.fcn import FCNModel

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32_LOGIT_TOL = 1e-4            # north_star: fp32 logits within 1e-4 of the (TF-)CPU path

# bf16-mode vs f32-mode bounds per config = 1.3x the values measured on MI355X (profiles/r03_parity_measured.jsonl; cosines
# 1 - 1.3 (1 - measured)).  NOTE what gl2 / gcos of the U-Net rows measure: uniform-random images with random labels at the xavier /
# zero-bias init put the pre-activations around zero, where a 2^-9 perturbation flips ReLU gates -- tests/test_precision_gpu.py
# attributes the gap (forward roundings only; the gradient chain contributes < 1 %) and measures 0.047 / 0.9989 on image-like data.
#   logit  max |dlogit| / max |logit|          loss   relative loss error
#   gl2    worst relative L2 error of a gradient tensor (weights)       gcos   worst cosine of a gradient tensor
#   dis0   argmax pixel-disagreement rate at the xavier init (tiny top-2 margins)
#   dis1   the same after 20 bf16 train steps on the batch              miou1  |mIoU_bf16 - mIoU_f32| against the labels
BOUNDS = {
    # measured (MI355X, r02): logit .0092  loss 1e-6  gl2 .230 (conv4_1)  gcos .976  dis0 .0028  dis1 .0063  |dmIoU| 7e-5
    'C2': dict(logit=0.012, loss=1e-4, gl2=0.30, gcos=0.969, dis0=0.0037, dis1=0.0082, miou1=0.005),
    # measured: logit .0067  loss 8e-6  gl2 .0123 (conv5)  gcos .99992  dis0 2e-6  dis1 .0200  |dmIoU| 1e-4
    'C3': dict(logit=0.0088, loss=1e-4, gl2=0.016, gcos=0.9999, dis0=1e-4, dis1=0.026, miou1=0.005),
    # measured: logit .0104  loss <1e-6  gl2 .190 (conv5_1)  gcos .982  dis0 .0033  dis1 .0085  |dmIoU| 1e-5
    'C4': dict(logit=0.0136, loss=1e-4, gl2=0.25, gcos=0.9766, dis0=0.0043, dis1=0.0111, miou1=0.005),
}


def _record(cfg, vals):
    vals = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}
    line = json.dumps(dict(config=cfg, **vals))
    print('\n[parity-measured] ' + line)
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_measured.jsonl'), 'a') as fh:
            fh.write(line + '\n')
    except OSError:
        pass


def _data(B, S, nc, seed=5555):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0, 1, (1, B, S, S, 3)).astype(np.float32), rng.integers(0, nc, (1, B, S, S, 1)).astype(np.uint8))


def _unet(x, y, nc, S, dtype, **kw):
    kw.setdefault('learning_rate', 1e-3)
    return UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=S, log_dir=None,
                     save_dir=None, load_snapshot=False, dtype=dtype, n_kernels=32, seed=5555, **kw)


def _fcn8s(x, y, nc, S, dtype, **kw):
    kw.setdefault('learning_rate', 1e-3)
    kw.setdefault('keep_logits', True)
    return FCNModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=nc, input_dims=S, fcn_type='8s', n_kernels=32,
                    log_dir=None, save_dir=None, load_snapshot=False, dtype=dtype, seed=5555, **kw)


def _fwd_bwd(m):
    m._load_batch(m.dataset, m.input_x, m.input_y)
    m.store.g.fill_(float('nan'))                  # every gradient entry must be (over)written
    m._run_fwd_bwd()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.store.g).all())
    nc = m.n_classes
    return m.acts['logits'].t[..., :nc].float().cpu().numpy(), m.last_loss(), m.store.get_grads()


def _argmax_check(logits_hip, logits_ref, tol=F32_LOGIT_TOL):
    """argmax (sigmoid-then-argmax, F17) bit-exact wherever the oracle's top-2 LOGIT margin exceeds twice the tolerance"""
    _, a = ops.sigmoid_argmax(logits_hip)
    _, b = ops.sigmoid_argmax(logits_ref.astype(np.float32))
    srt = np.sort(logits_ref, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 2 * tol
    assert decided.mean() > 0.5
    assert np.array_equal(a[..., 0][decided], b[..., 0][decided])
    return float(decided.mean())


def _grads_vs_ref(g, g_ref, rtol):
    worst = 0.0
    for n in g_ref:
        for k in ('weights', 'biases'):
            ref = np.asarray(g_ref[n][k]); got = g[n][k]
            err = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-20))
            assert err < rtol, (n, k, err)
            worst = max(worst, err)
    return worst


def _mode_gap(gb, gf):
    """worst relative-L2 error and worst cosine over the weight-gradient tensors (bf16 vs f32)"""
    gl2, gcos, who = 0.0, 1.0, None
    for n in gf:
        a, b = gb[n]['weights'].ravel().astype(np.float64), gf[n]['weights'].ravel().astype(np.float64)
        l2 = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        if l2 > gl2:
            gl2, who = l2, n
        gcos = min(gcos, cos)
    return gl2, gcos, who


def _labels_window(y, off, hw):
    return y[:, off[0]:off[0] + hw[0], off[1]:off[1] + hw[1], :]


def _bf16_vs_f32(cfg, mb, mf, x, y, steps=20):
    """check (b): both models hold identical (seed-5555 xavier) weights"""
    assert torch.equal(mb.store.p, mf.store.p)
    nc = mb.n_classes
    lb, lossb, gb = _fwd_bwd(mb)
    lf, lossf, gf = _fwd_bwd(mf)
    B = BOUNDS[cfg]
    scale = float(np.abs(lf).max())
    logit = float(np.abs(lb - lf).max() / scale)
    loss = abs(lossb - lossf) / abs(lossf)
    gl2, gcos, who = _mode_gap(gb, gf)
    yl = _labels_window(y[0], mb.label_off, mb.out_hw)
    ab, af = ops.sigmoid_argmax(lb)[1], ops.sigmoid_argmax(lf)[1]
    dis0 = float((ab != af).mean())
    # after `steps` bf16 train steps on the batch the predictions are decisive: same weights into both, compare infer()
    for _ in range(steps):
        mb.train_step()
    torch.cuda.synchronize()
    mf.set_weights(mb.store.get_params())
    sb, ob = mb.infer(x[0])
    sf, of_ = mf.infer(x[0])
    dis1 = float((ob != of_).mean())
    miou_b, miou_f = ops.miou(ob, yl, nc), ops.miou(of_, yl, nc)
    vals = dict(check='bf16_vs_f32', logit_rel=logit, logit_scale=scale, loss_bf16=lossb, loss_f32=lossf, loss_rel=loss, grad_rel_l2_worst=gl2,
                grad_worst_tensor=who, grad_cos_worst=gcos, disagree_init=dis0, disagree_trained=dis1, sigmoid_maxdiff_trained=float(np.abs(sb - sf).max()),
                miou_bf16=miou_b, miou_f32=miou_f, train_steps=steps, loss_after=mb.last_loss())
    _record(cfg, vals)
    assert logit < B['logit'], vals
    assert loss < B['loss'], vals
    assert gl2 < B['gl2'] and gcos > B['gcos'], vals
    assert dis0 < B['dis0'] and dis1 < B['dis1'], vals
    assert abs(miou_b - miou_f) < B['miou1'], vals
    assert mb.last_loss() < lossb                     # 20 steps on one batch reduce its loss


# --------------------------------------------------------------------------------------------- C2
def test_c2_unet256_b16_f32_vs_oracle_full_batch():
    Bn, S, nc = 16, 256, 4
    x, y = _data(Bn, S, nc)
    mf = _unet(x, y, nc, S, 'f32', use_graph=False)
    p = mf.store.get_params()
    assert sum(v['weights'].size + v['biases'].size for v in p.values()) == 7760196
    lf, lossf, gf = _fwd_bwd(mf)
    assert lf.shape == (Bn, 68, 68, nc) and mf.label_off == (94, 94)
    loss_ref, g_ref, c = ounet.loss_and_grads(p, x[0], y[0])           # the whole batch, float64 (~20 s of host time)
    err = float(np.abs(lf - c['logits']).max())
    assert err < F32_LOGIT_TOL
    assert abs(lossf - loss_ref) < 1e-5
    decided = _argmax_check(lf, c['logits'])
    for name in ('conv1_1', 'conv2_2', 'conv3_2', 'conv5_2', 'upconv1', 'conv6_1', 'conv8_1', 'conv9_2'):
        a = mf.acts[name]
        if name == 'conv1_1' or a.H == c[name].shape[1]:
            assert np.abs(a.t[..., :a.C].cpu().numpy() - c[name]).max() < 1e-4, name
    # (16 x 68 x 68 .. 16 x 254 x 254 terms per filter tap accumulate in f32: 3.8e-4 of the tensor maximum measured on upconv4)
    gerr = _grads_vs_ref(gf, g_ref, 1e-3)
    _record('C2', dict(check='f32_vs_oracle', logit_maxabs=err, loss_abs=abs(lossf - loss_ref), grad_rel_worst=gerr, argmax_decided_frac=decided))


@pytest.mark.parametrize('use_graph', [False, True])
def test_c2_unet256_b16_bf16_vs_f32(use_graph):
    Bn, S, nc = 16, 256, 4
    x, y = _data(Bn, S, nc)
    mb = _unet(x, y, nc, S, 'bf16', use_graph=use_graph)
    mf = _unet(x, y, nc, S, 'f32', use_graph=False)
    _bf16_vs_f32('C2', mb, mf, x, y)


def test_c2_step_modes_are_bitwise_identical():
    """the bench picks eager (high-priority stream) or hipGraph replay at run time: same bits either way, at B=16 / 256"""
    Bn, S, nc = 16, 256, 4
    x, y = _data(Bn, S, nc)
    m1 = _unet(x, y, nc, S, 'bf16', use_graph=False)
    m2 = _unet(x, y, nc, S, 'bf16', use_graph=True)
    for _ in range(4):
        m1.train_step(); m2.train_step()
    torch.cuda.synchronize()
    assert torch.equal(m1.store.p, m2.store.p) and torch.equal(m1.store.g, m2.store.g)
    # (the reported loss is a sum of per-workgroup partial sums added with float atomics: equal up to their arrival order)
    assert abs(m1.last_loss() - m2.last_loss()) < 2e-6 * abs(m1.last_loss())


# --------------------------------------------------------------------------------------------- C3
def test_c3_fcn8s_512_b8_f32_vs_oracle_full_batch():
    Bn, S, nc = 8, 512, 21
    x, y = _data(Bn, S, nc)
    mf = _fcn8s(x, y, nc, S, 'f32', use_graph=False)
    p = mf.store.get_params()
    assert sum(v['weights'].size + v['biases'].size for v in p.values()) == 2320895
    for n in p:      # non-zero biases so that the ReLU'd score layers are not all dead at init
        p[n]['biases'] = (np.random.default_rng(3).standard_normal(p[n]['biases'].shape) * 0.05 + 0.05).astype(np.float32)
    mf.set_weights(p)
    lf, lossf, gf = _fwd_bwd(mf)
    assert lf.shape == (Bn, S, S, nc)
    loss_ref, g_ref, c = ofcn.loss_and_grads(p, x[0], y[0], '8s')
    err = float(np.abs(lf - c['logits']).max())
    assert err < F32_LOGIT_TOL
    assert abs(lossf - loss_ref) < 1e-5
    decided = _argmax_check(lf, c['logits'], tol=max(err, 1e-6))
    gerr = _grads_vs_ref(gf, g_ref, 1e-3)
    _record('C3', dict(check='f32_vs_oracle', logit_maxabs=err, loss_abs=abs(lossf - loss_ref), grad_rel_worst=gerr, argmax_decided_frac=decided))


def test_c3_fcn8s_512_b8_bf16_vs_f32():
    Bn, S, nc = 8, 512, 21
    x, y = _data(Bn, S, nc)
    mb = _fcn8s(x, y, nc, S, 'bf16', use_graph=True)
    mf = _fcn8s(x, y, nc, S, 'f32', use_graph=False)
    p = mf.store.get_params()
    for n in p:
        p[n]['biases'] = (np.random.default_rng(3).standard_normal(p[n]['biases'].shape) * 0.05 + 0.05).astype(np.float32)
    mf.set_weights(p); mb.set_weights(p)
    _bf16_vs_f32('C3', mb, mf, x, y)


# --------------------------------------------------------------------------------------------- C4 (per-GPU shard)
def test_c4_unet512_b16_f32_vs_oracle():
    Bn, S, nc = 16, 512, 4
    x, y = _data(Bn, S, nc)
    mf = _unet(x, y, nc, S, 'f32', use_graph=False)
    p = mf.store.get_params()
    lf, lossf, gf = _fwd_bwd(mf)
    assert lf.shape == (Bn, 324, 324, nc) and mf.label_off == (94, 94)
    # oracle on images 0 and 15 (first / last of the batch: 2 x ~4 s of float64 numpy)
    worst, decided = 0.0, 1.0
    for b in (0, Bn - 1):
        lr, _ = ounet.forward(p, x[0][b:b + 1])
        worst = max(worst, float(np.abs(lf[b:b + 1] - lr).max()))
        decided = min(decided, _argmax_check(lf[b:b + 1], lr))
    assert worst < F32_LOGIT_TOL
    # gradients: a ONE-image 512x512 step against the oracle's full backward (the 512-size tile / K-split choices)
    x1, y1 = x[:, :1], y[:, :1]
    m1 = _unet(x1, y1, nc, S, 'f32', use_graph=False)
    l1, loss1, g1 = _fwd_bwd(m1)
    loss_ref, g_ref, c = ounet.loss_and_grads(p, x1[0], y1[0])
    assert np.abs(l1 - c['logits']).max() < F32_LOGIT_TOL and abs(loss1 - loss_ref) < 1e-5
    # bound: a one-image 512x512 step flips ReLU gates on round-off, which moves single gradient entries by up to ~1e-2 of their
    # tensor's maximum in ANY float32 evaluation -- the float32 run of the ORACLE ITSELF deviates from its float64 run by 6.0e-3
    # of the maximum (conv6_1) and 1.9e-3 in relative L2 (upconv1); the HIP f32 path measured 1.1e-2 / (below) in these norms
    gerr = _grads_vs_ref(g1, g_ref, 3e-2)
    l2 = max(float(np.linalg.norm(g1[n][k] - g_ref[n][k]) / (np.linalg.norm(g_ref[n][k]) + 1e-30)) for n in g_ref for k in ('weights', 'biases'))
    assert l2 < 5e-3, l2
    _record('C4', dict(check='f32_vs_oracle', logit_maxabs=worst, grad_rel_worst_b1=gerr, grad_rel_l2_worst_b1=l2, argmax_decided_frac=decided, loss_b16=lossf))


def test_c4_unet512_b16_bf16_vs_f32():
    Bn, S, nc = 16, 512, 4
    x, y = _data(Bn, S, nc)
    mb = _unet(x, y, nc, S, 'bf16', use_graph=True)
    mf = _unet(x, y, nc, S, 'f32', use_graph=False)
    _bf16_vs_f32('C4', mb, mf, x, y, steps=10)


# --------------------------------------------------------------------------------------------- C5 (MC dropout, a19)
def test_c5_mc_dropout_256_b32_30_passes_vs_oracle():
    """Build-defined stochastic inference at the BASELINE size: 256x256, batch 32, 30 passes.  The oracle regenerates the
    masks bit for bit (oracle.np_ops.dropout_mask); it is run on image 0 (the mask counters of image 0 do not depend on the
    batch size), every pass and the mean / variance are compared in f32 mode; bf16 mode is compared with f32 mode."""
    Bn, S, nc, T = 32, 256, 4, 30
    x, y = _data(Bn, S, nc)
    mf = UNetModel(sess=None, mode='INFERENCE', n_classes=nc, input_dims=S, save_dir=None, load_snapshot=False, dtype='f32', seed=5555)
    p = mf.store.get_params()
    mean, var, amax, per = mf.infer_mc(x[0], passes=T, keep_prob=0.5, seed=5555, return_passes=True)
    assert mean.shape == (Bn, 68, 68, nc) and var.shape == mean.shape and amax.shape == (Bn, 68, 68, 1) and len(per) == T
    omean, ovar, oamax, osig = ounet.infer_mc(p, x[0][:1], passes=T, keep=0.5, seed=5555)
    e_pass = max(float(np.abs(per[t][:1] - osig[t]).max()) for t in range(T))
    e_mean, e_var = float(np.abs(mean[:1] - omean).max()), float(np.abs(var[:1] - ovar).max())
    assert e_pass < 1e-4 and e_mean < 1e-4 and e_var < 1e-4, (e_pass, e_mean, e_var)
    srt = np.sort(omean, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 2e-4
    assert np.array_equal(amax[:1][..., 0][decided], oamax[..., 0][decided])
    assert var.max() > 0 and float(ovar.max()) > 0
    # passes differ from each other and from the deterministic forward
    assert np.abs(per[0] - per[1]).max() > 1e-3
    # bf16 mode against f32 mode (same masks: they depend only on seed / offset / element index)
    mb = UNetModel(sess=None, mode='INFERENCE', n_classes=nc, input_dims=S, save_dir=None, load_snapshot=False, dtype='bf16', seed=5555)
    mb.set_weights(p)
    bmean, bvar, bamax = mb.infer_mc(x[0], passes=T, keep_prob=0.5, seed=5555)
    d_mean = float(np.abs(bmean - mean).max())
    d_var = float(np.abs(bvar - var).max())
    dis = float((bamax != amax).mean())
    _record('C5', dict(check='mc_dropout', f32_vs_oracle_pass=e_pass, f32_vs_oracle_mean=e_mean, f32_vs_oracle_var=e_var,
                       bf16_vs_f32_mean=d_mean, bf16_vs_f32_var=d_var, bf16_argmax_disagree=dis, var_max=float(var.max())))
    # measured: mean 3.7e-4, variance < 1e-6 (the variance itself is <= 1.6e-5 at the xavier init), argmax disagreement 0.0026
    assert d_mean < 1e-3 and d_var < 1e-5 and dis < 0.006
