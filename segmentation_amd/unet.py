"""UNetModel: the reference's U-Net (/root/reference/models/unet.py:24-175) compiled to HIP launch plans.

Graph facts reproduced on purpose: every conv is 3x3 VALID + bias + ReLU; pool1
consumes conv1_1 while conv1_2 only feeds the last skip; up-convs are 2x2/s2 transposed convs with
ReLU; skips are centre-cropped and concatenated skip-first; `output` is a 1x1 conv without activation;
input_y is centre-cropped to the logits; no BatchNorm, no dropout (`bayesian` is accepted and ignored,
exactly as in the reference).

MI355X design: concat and crop are views (the consuming conv reads two base pointers with window
offsets); conv1_2 is evaluated only on the window that survives the crop when crop_aware=True
(bit-identical results, fewer MACs -- bench.py reports MACs actually executed); ReLU-grad masks are
fused into the producing dgrad epilogues and the crop-grad zero-pad + add into the max-pool backward.
"""
import os

import numpy as np
import torch

from . import _lib as L
from . import engine as E
from .basemodel import BaseModel

LEVELS = [('upconv1', 'conv4_2', 'conv6_1', 'conv6_2'), ('upconv2', 'conv3_2', 'conv7_1', 'conv7_2'),
          ('upconv3', 'conv2_2', 'conv8_1', 'conv8_2'), ('upconv4', 'conv1_2', 'conv9_1', 'conv9_2')]
# gradient arena order = the order backward produces the gradients (all-reduce buckets are prefixes)
BWD_ORDER = ['output', 'conv9_2', 'conv9_1', 'upconv4', 'conv8_2', 'conv8_1', 'upconv3', 'conv7_2', 'conv7_1', 'upconv2',
             'conv6_2', 'conv6_1', 'upconv1', 'conv5_2', 'conv5_1', 'conv4_2', 'conv4_1', 'conv3_2', 'conv3_1',
             'conv2_2', 'conv2_1', 'conv1_2', 'conv1_1']


def unet_layers(n_classes, n_kernels, input_channel):
    nk = n_kernels
    ls = {}

    def conv(name, segs, co, k=3, relu=True, kind='conv'):
        ls[name] = E.Layer(name, kind, k, segs, co, 'VALID', relu)
    conv('conv1_1', [input_channel], nk, kind='first'); conv('conv1_2', [nk], nk)
    conv('conv2_1', [nk], 2 * nk); conv('conv2_2', [2 * nk], 2 * nk)
    conv('conv3_1', [2 * nk], 4 * nk); conv('conv3_2', [4 * nk], 4 * nk)
    conv('conv4_1', [4 * nk], 8 * nk); conv('conv4_2', [8 * nk], 8 * nk)
    conv('conv5_1', [8 * nk], 16 * nk); conv('conv5_2', [16 * nk], 16 * nk)
    for i, w in enumerate((8, 4, 2, 1)):
        ls['upconv%d' % (i + 1)] = E.Layer('upconv%d' % (i + 1), 'up', 2, [2 * w * nk], w * nk, 'VALID', True)
        conv('conv%d_1' % (6 + i), [w * nk, w * nk], w * nk)
        conv('conv%d_2' % (6 + i), [w * nk], w * nk)
    conv('output', [nk], n_classes, k=1, relu=False)
    ls['conv1_1'].need_dgrad = False
    return [ls[n] for n in BWD_ORDER]


def unet_sizes(n):
    """Spatial ladder of the all-VALID graph for a square/rect edge n (raises when infeasible, F11)."""
    s = {}

    def cc(v, name):
        if v < 3:
            raise Exception('U-Net (all VALID) infeasible for input %d: %s would see a %dx%d map' % (n, name, v, v))
        return v - 2
    s['conv1_1'] = cc(n, 'conv1_1'); s['conv1_2'] = cc(s['conv1_1'], 'conv1_2')
    v = s['conv1_1'] // 2; s['pool1'] = v
    for i in (2, 3, 4, 5):
        s['conv%d_1' % i] = cc(v, 'conv%d_1' % i); s['conv%d_2' % i] = cc(s['conv%d_1' % i], 'conv%d_2' % i)
        if i < 5:
            v = s['conv%d_2' % i] // 2; s['pool%d' % i] = v
            if v < 1:
                raise Exception('U-Net (all VALID) infeasible for input %d' % n)
    v = s['conv5_2']
    for i, (upn, skip, ca, cb) in enumerate(LEVELS):
        s[upn] = 2 * v
        if s[skip] < s[upn]:
            raise Exception('U-Net infeasible for input %d: skip %s smaller than %s' % (n, skip, upn))
        s[ca] = cc(s[upn], ca); s[cb] = cc(s[ca], cb); v = s[cb]
    s['output'] = v
    return s


_AUX_PACK = os.environ.get('SEG_PACK_ON_AUX', '0') == '1'


class UNetModel(BaseModel):
    def __init__(self,
                 sess=None,
                 n_classes=2,
                 log_dir=None,
                 dataset=None,
                 save_dir=None,
                 bayesian=False,
                 input_dims=512,
                 mode='TRAINING',
                 input_channel=3,
                 test_dataset=None,
                 learning_rate=1e-4,
                 load_snapshot=None,
                 load_snapshot_from=None,
                 n_kernels=32,
                 adversarial_training=False,
                 **mi355x):
        super(UNetModel, self).__init__(
            sess=sess, mode=mode, log_dir=log_dir, dataset=dataset, bayesian=bayesian, save_dir=save_dir,
            n_classes=n_classes, input_dims=input_dims, test_dataset=test_dataset, input_channel=input_channel,
            load_snapshot=load_snapshot, learning_rate=learning_rate, load_snapshot_from=load_snapshot_from,
            adversarial_training=adversarial_training, **mi355x)
        self.model_name = 'unet'
        self.IN_OUT_CROP = True
        self.n_kernels = n_kernels
        if n_classes > 32:
            raise Exception('n_classes > 32 not supported')
        self._init_input()
        training = self.mode != 'INFERENCE'
        self.layers = unet_layers(n_classes, n_kernels, input_channel)
        self.store = E.ParamStore(self.layers, self.dtype, self.device, training=training)
        self._init_weights()
        self.net = None
        if training:
            self._build_training()
        self._repack_initial()
        self.inference_ops = ['y_hat_sig', 'output']
        self._init_saver(self.model_name)

    def _init_weights(self):
        self._xavier_init(['conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv4_1', 'conv4_2', 'conv5_1',
                           'conv5_2', 'upconv1', 'conv6_1', 'conv6_2', 'upconv2', 'conv7_1', 'conv7_2', 'upconv3', 'conv8_1',
                           'conv8_2', 'upconv4', 'conv9_1', 'conv9_2', 'output'])

    def model(self, input_op=None, reuse=False):
        """The reference's model() declares the TF graph; here the graph is the compiled launch plan."""
        return self.fwd_plan

    # ---- forward graph (shared by training and inference builders) ----
    def _emit_forward(self, net, plan, x_in, H, W, crop_aware, dropout=None, after_first=None, head=True):
        nk, Ly = self.n_kernels, self.store.layers
        if H != W:
            # the reference crops every skip with a SQUARE target taken from the height (models/unet.py:139-140,146-147):
            # a non-square input makes its tf.concat fail, so it is rejected here too
            raise Exception('UNetModel needs square inputs (got %dx%d): skips are cropped with a square target' % (H, W))
        sh = sw = unet_sizes(H)
        A = {}
        A['conv1_1'] = net.act(sh['conv1_1'], sw['conv1_1'], nk, name='conv1_1')
        A['pool1'] = net.act(sh['pool1'], sw['pool1'], nk, name='pool1')      # pool1 is taken over conv1_1 (F11)
        pool1_done = net.first_fwd(plan, Ly['conv1_1'], x_in, H, W, A['conv1_1'], pool=A['pool1'])
        if after_first is not None:
            after_first()
        net.join_aux(plan)                 # packed weights (re-packed on the aux stream in training) are needed from here on
        # conv1_2: only its centre window survives the crop of the last skip
        t4h, t4w = sh['upconv4'], sw['upconv4']
        o4h, o4w = (sh['conv1_2'] - t4h) // 2, (sw['conv1_2'] - t4w) // 2
        # (its only consumer is conv9_1, at the far end of the forward pass: in the small training steps it runs on a side stream
        # beside conv2_1 ..., joined in front of conv9_1 -- one launch less on the critical stream: C2 0.969 against 0.972 ms on
        # one box; at 512^2, where the launch is not what its 75 us are made of, 4.20 against 4.18, so it stays on the main stream)
        px = getattr(net, 'input_pixels', None)
        off_side = 1 if (px is not None and px <= 2500000) else 0
        if crop_aware:
            A['conv1_2'] = net.act(t4h, t4w, nk, name='conv1_2')
            net.conv_fwd(plan, Ly['conv1_2'], [(A['conv1_1'], o4h, o4w)], t4h + 2, t4w + 2, A['conv1_2'], side=off_side)
            skip4_off = (0, 0)
        else:
            A['conv1_2'] = net.act(sh['conv1_2'], sw['conv1_2'], nk, name='conv1_2')
            net.conv_fwd(plan, Ly['conv1_2'], [(A['conv1_1'], 0, 0)], sh['conv1_1'], sw['conv1_1'], A['conv1_2'], side=off_side)
            skip4_off = (o4h, o4w)
        prev, ph, pw = A['conv1_1'], sh['conv1_1'], sw['conv1_1']
        pooled = pool1_done                  # the pool of `prev` has already been written by the launch that produced it
        for i in (2, 3, 4, 5):
            P = A['pool%d' % (i - 1)]
            if not pooled:
                net.pool_fwd(plan, prev, P, P.H, P.W)
            c1, c2 = 'conv%d_1' % i, 'conv%d_2' % i
            A[c1] = net.act(sh[c1], sw[c1], Ly[c1].cout, name=c1)
            net.conv_fwd(plan, Ly[c1], [(P, 0, 0)], P.H, P.W, A[c1])
            A[c2] = net.act(sh[c2], sw[c2], Ly[c2].cout, name=c2)
            drop = dropout is not None and c2 in dropout['sites']
            nxt = None
            if i < 5:                        # conv i_2 feeds pool i: fuse the pool into its epilogue where the tile allows
                nxt = A['pool%d' % i] = net.act(sh['pool%d' % i], sw['pool%d' % i], Ly[c2].cout, name='pool%d' % i)
            net.conv_fwd(plan, Ly[c2], [(A[c1], 0, 0)], sh[c1], sw[c1], A[c2], pool=None if drop else nxt)
            pooled = net.pool_fused
            if drop:
                split = dropout.get('split')
                if split is not None and plan is not split:
                    # everything up to here is deterministic: it stays in `plan` (run once per input); the stochastic rest
                    # goes into dropout['split'] (run once per pass) and reads the un-masked map, so the mask is out of place
                    raw, A[c2 + '/raw'] = A[c2], A[c2]
                    A[c2] = net.act(sh[c2], sw[c2], Ly[c2].cout, name=c2 + '/drop')
                    plan = split
                    net.dropout(plan, raw, dropout['keep'], dropout['seed'] + i, dropout['offset'], dst=A[c2])
                else:
                    net.dropout(plan, A[c2], dropout['keep'], dropout['seed'] + i, dropout['offset'])
            prev = A[c2]
        skip_off = {}
        for i, (upn, skip, ca, cb) in enumerate(LEVELS):
            A[upn] = net.act(sh[upn], sw[upn], Ly[upn].cout, name=upn)
            net.up_fwd(plan, Ly[upn], prev, prev.H, prev.W, A[upn])
            if skip == 'conv1_2':
                off = skip4_off
                if off_side:
                    net.join_wgrad(plan)             # conv1_2 ran on a side stream
            else:
                off = ((sh[skip] - sh[upn]) // 2, (sw[skip] - sw[upn]) // 2)      # crop_or_pad: floor offsets
            skip_off[skip] = off
            A[ca] = net.act(sh[ca], sw[ca], Ly[ca].cout, name=ca)
            net.conv_fwd(plan, Ly[ca], [(A[skip], off[0], off[1]), (A[upn], 0, 0)], sh[upn], sw[upn], A[ca])
            A[cb] = net.act(sh[cb], sw[cb], Ly[cb].cout, name=cb)
            net.conv_fwd(plan, Ly[cb], [(A[ca], 0, 0)], sh[ca], sw[ca], A[cb])
            if dropout is not None and cb in dropout['sites']:
                net.dropout(plan, A[cb], dropout['keep'], dropout['seed'] + 10 + i, dropout['offset'])
            prev = A[cb]
        # (written by the 1x1 output convolution as a tensor of its own -- inference, adversarial training: thin up to 8 classes)
        A['logits'] = net.act(sh['output'], sw['output'], self.n_classes, f32=True, name='logits',
                              thin=head and self.n_classes <= 8 and os.environ.get('SEG_THIN_TAIL', '1') != '0')
        if head:        # (the training plan fuses this 1x1 conv with the loss and its input gradient: Net.head_xent)
            net.conv_fwd(plan, Ly['output'], [(prev, 0, 0)], prev.H, prev.W, A['logits'], out_f32=True)
        return A, sh, sw, skip_off, (o4h, o4w)

    # ---- training plans ----
    def _build_training(self):
        B, (H, W) = self.batch_size, self.input_dims
        net = self.net = E.Net(self.store, B, self.dtype, self.device)
        net.n_wgrad_streams = max(1, len(self._side) - 1) if self._side else 1
        net.input_pixels = B * H * W if not self.pg.tuned else None      # (data parallel keeps the lone-launch filter-gradient target: 1.05 against 1.09 ms at world 1)
        # the last filter gradients of the backward pass outlive the critical stream: they aim for the whole chip (256 workgroups)
        net.tail_layers = ('conv1_2', 'conv2_1')
        Ly = self.store.layers
        fwd = self.fwd_plan = E.Plan('fwd')
        self.loss_buf = self.store.loss_slot()          # (behind the gradient arena: reduced with the last bucket under data parallelism)
        net.step_begin(fwd, self.loss_buf, aux=_AUX_PACK or self.pg.tuned)     # aux stream: global_step += 1, loss accumulator = 0
        # refresh the packed weights after the previous Adam step, beside conv1_1.  (Round 4 tried to keep the first layer -- HBM-bound,
        # 49.9 us in the step against 23 alone -- out of the re-pack's shadow: the few filters conv1_2 .. conv3_2 need packed beside
        # conv1_1, the bulk on a filter-gradient stream beside conv2_1 .. conv3_2.  The step got 0.7 % SLOWER at C2 and 1 % at 512^2
        # (profiles/r04_ab_split_pack_*.txt): the 62 MB have to share the memory with something, and conv2_x are no less
        # bandwidth-hungry than conv1_1.  Removed.)
        net.pack(fwd, aux=_AUX_PACK or self.pg.tuned)
        cols = []       # im2col of the input for conv1_1's filter gradient: side stream, right after conv1_1 (both are
        #                 bandwidth-bound), overlapping the rest of the forward pass
        fuse_head = (E.rup(Ly['output'].cin) in (32, 64) and self.n_classes <= 32 and os.environ.get('SEG_FUSE_HEAD', '1') != '0' and
                     not self.adversarial_training)     # (the adversary adds its term to dlogits before the output layer's backward)
        A, sh, sw, skip_off, o4 = self._emit_forward(net, fwd, self.input_x, H, W, self.crop_aware, head=not fuse_head,
                                                     after_first=lambda: cols.append(net.first_im2col(fwd, Ly['conv1_1'], self.input_x, H, W)))
        col = cols[0]
        self.acts = A
        oh, ow = sh['output'], sw['output']
        self.out_hw = (oh, ow)
        dlog = net.act(oh, ow, self.n_classes, name='dlogits', thin=A['logits'].thin)
        # label crop: resize_image_with_crop_or_pad(input_y, target, target) -> floor offsets (unet.py:171-174)
        self.label_off = ((H - oh) // 2, (W - ow) // 2)
        G = {}                                    # masked gradients dZ (same shape as the activation)
        if fuse_head:
            # 1x1 output conv + loss + its input gradient in one launch
            a92 = A['conv9_2']
            G['conv9_2'] = net.act(a92.H, a92.W, a92.C, name='dconv9_2')
            head_ws = net.head_xent(fwd, Ly['output'], a92, self.input_y, H, W, self.label_off, oh, ow, self.n_classes, self.loss_buf,
                                    A['logits'], dlog, G['conv9_2'], fuse_dw=os.environ.get('SEG_FUSE_HEAD_DW', '1') != '0')
        else:
            probs = self._make_adversary(oh, ow, dlog) if self.adversarial_training else None
            net.softmax_xent(fwd, A['logits'], self.input_y, H, W, self.label_off, oh, ow, self.n_classes, self.loss_buf, dlog, probs=probs)
        if self.adversarial_training:
            self._attach_adversary(A['logits'], oh, ow, H, W, dlog)
        self.dlogits = dlog

        def gz(name):
            a = A[name]
            G[name] = net.act(a.H, a.W, a.C, name='d' + name)
            return G[name]

        seg = E.Plan('bwd0')
        segs = []

        # gradient-bucket boundaries (backward order = arena order): a bucket is a contiguous slice of the flat gradient arena and
        # its all-reduce is issued as soon as the backward segment that completes it has been enqueued.  Default: four buckets
        # of ~10 / 11.5 / 9.2 / 0.26 MB cut after conv6_1, conv5_2 and conv3_1 (several medium messages launched as
        # they complete; the big tensors conv6_1 / conv5_2 / conv5_1 appear in the MIDDLE of backward and the last bucket --
        # conv2_x + conv1_x, the only one with nothing left to hide behind -- is tiny).  Every boundary drains the side streams
        # (17-30 us on one GPU), so SEG_DP_CUTS / dp_cuts= can trade boundaries for message size once an
        # N > 1 run has been measured ('conv3_1' = the two-bucket form; 'conv6_2,upconv1,conv5_2,conv5_1,conv3_1' = 4-9 MB messages);
        # bench.py --dp-cuts auto times the candidates on the actual node and keeps the fastest.
        cuts = getattr(self, 'dp_cuts', None) or os.environ.get('SEG_DP_CUTS', 'conv6_1,conv5_2,conv3_1')
        cuts = [c for c in (cuts.split(',') if isinstance(cuts, str) else list(cuts)) if c]
        for c_ in cuts:
            if c_ not in Ly:
                raise Exception('dp cut %r is not a layer (one of %s)' % (c_, ', '.join(BWD_ORDER)))
        self.dp_cuts_used = cuts

        def close_segment(last_layer):
            nonlocal seg
            if last_layer != 'conv1_1' and last_layer not in cuts:
                return
            l = Ly[last_layer]
            net.flush_reduce(seg)             # one slab-reduction launch per segment
            segs.append((seg, l.b_off + l.nbias))
            seg = E.Plan('bwd%d' % len(segs))

        # output layer
        if fuse_head and head_ws is not None:
            net.head_dw_reduce(seg, Ly['output'], head_ws, oh, ow, A['conv9_2'].Cp, self.n_classes)     # the gradient itself came with the head
        else:
            net.conv_bwd(seg, Ly['output'], [(A['conv9_2'], 0, 0)], A['conv9_2'].H, A['conv9_2'].W, dlog,
                         [None if fuse_head else (gz('conv9_2'), (0, 0), A['conv9_2'], (0, 0))])
        close_segment('output')
        dskip = {}
        prev_of = {'upconv1': 'conv5_2', 'upconv2': 'conv6_2', 'upconv3': 'conv7_2', 'upconv4': 'conv8_2'}
        for lvl in (3, 2, 1, 0):
            upn, skip, ca, cb = LEVELS[lvl]
            # conv?_2
            net.conv_bwd(seg, Ly[cb], [(A[ca], 0, 0)], A[ca].H, A[ca].W, G[cb], [(gz(ca), (0, 0), A[ca], (0, 0))])
            close_segment(cb)
            # conv?_1 on [skip_crop | up]: two dgrad launches (skip half, up half)
            so = skip_off[skip]
            uh, uw = A[upn].H, A[upn].W
            if skip == 'conv1_2':
                # conv1_2's only consumer is this skip: its masked gradient is the skip half itself
                dsk = net.act(uh, uw, A[skip].C, name='dz_conv1_2')
                skip_spec = (dsk, (0, 0), A[skip], so)
            else:
                dsk = net.act(uh, uw, A[skip].C, name='dskip_' + skip)
                skip_spec = (dsk, (0, 0), None, (0, 0))
            dskip[skip] = dsk
            net.conv_bwd(seg, Ly[ca], [(A[skip], so[0], so[1]), (A[upn], 0, 0)], uh, uw, G[ca],
                         [skip_spec, (gz(upn), (0, 0), A[upn], (0, 0))])
            close_segment(ca)
            # transposed conv
            pn = prev_of[upn]
            net.up_bwd(seg, Ly[upn], A[pn], A[pn].H, A[pn].W, G[upn], gz(pn), A[pn])
            close_segment(upn)
        # bottleneck
        net.conv_bwd(seg, Ly['conv5_2'], [(A['conv5_1'], 0, 0)], A['conv5_1'].H, A['conv5_1'].W, G['conv5_2'],
                     [(gz('conv5_1'), (0, 0), A['conv5_1'], (0, 0))])
        close_segment('conv5_2')
        dP = {}
        dP[4] = net.act(A['pool4'].H, A['pool4'].W, A['pool4'].C, name='dpool4')
        net.conv_bwd(seg, Ly['conv5_1'], [(A['pool4'], 0, 0)], A['pool4'].H, A['pool4'].W, G['conv5_1'], [(dP[4], (0, 0), None, (0, 0))])
        close_segment('conv5_1')
        for i in (4, 3, 2):
            c1, c2 = 'conv%d_1' % i, 'conv%d_2' % i
            so = skip_off[c2]
            net.pool_bwd(seg, A[c2], dP[i], dskip[c2], (dskip[c2].H, dskip[c2].W), so, gz(c2), A[c2].H, A[c2].W)
            net.conv_bwd(seg, Ly[c2], [(A[c1], 0, 0)], A[c1].H, A[c1].W, G[c2], [(gz(c1), (0, 0), A[c1], (0, 0))])
            close_segment(c2)
            pin = A['pool%d' % (i - 1)]
            dP[i - 1] = net.act(pin.H, pin.W, pin.C, name='dpool%d' % (i - 1))
            net.conv_bwd(seg, Ly[c1], [(pin, 0, 0)], pin.H, pin.W, G[c1], [(dP[i - 1], (0, 0), None, (0, 0))])
            close_segment(c1)
        # conv1_2 (window of conv1_1 at o4, extent t4+2) then pool1 + skip add -> dZ(conv1_1)
        t4h, t4w = A['upconv4'].H, A['upconv4'].W
        d11s = net.act(t4h + 2, t4w + 2, A['conv1_1'].C, name='d_conv1_1_skip')
        net.conv_bwd(seg, Ly['conv1_2'], [(A['conv1_1'], o4[0], o4[1])], t4h + 2, t4w + 2, dskip['conv1_2'],
                     [(d11s, (0, 0), None, (0, 0))])
        close_segment('conv1_2')
        if net.fuses_first_pool_bwd(col):
            # pool1's backward happens inside the first layer's filter gradient (no launch, no dZ(conv1_1) tensor)
            # (the FCN runs Adam beside its last filter gradient: +1 % there; here that measured -0.4 % at 256^2 and +-0 at 512^2 -- both
            # kernels are bandwidth-bound -- so it is not done)
            net.first_bwd(seg, Ly['conv1_1'], self.input_x, H, W, None, same_stream=not self.pg.enabled,
                          pool=(A['conv1_1'], dP[1], d11s, (t4h + 2, t4w + 2), o4))
        else:
            net.pool_bwd(seg, A['conv1_1'], dP[1], d11s, (t4h + 2, t4w + 2), o4, gz('conv1_1'), A['conv1_1'].H, A['conv1_1'].W)
            net.first_bwd(seg, Ly['conv1_1'], self.input_x, H, W, G['conv1_1'], col=col, same_stream=not self.pg.enabled)
        close_segment('conv1_1')
        self.grads_act = G
        self._finish_training_plans(segs)
        # training-time outputs (y_hat_sig, output) on demand
        self.y_hat = A['logits']

    # ---- inference plan for an arbitrary batch / size ----
    def _build_infer(self, B, H, W, Cin):
        if Cin != self.input_channel:
            raise Exception('infer(): expected %d input channels, got %d' % (self.input_channel, Cin))
        net = E.Net(self.store, B, self.dtype, self.device)
        plan = E.Plan('infer')
        x_in = torch.zeros((B, H, W, Cin), dtype=torch.float32, device=self.device)
        A, sh, sw, _, _ = self._emit_forward(net, plan, x_in, H, W, self.crop_aware)
        oh, ow = sh['output'], sw['output']
        sig = torch.zeros((B, oh, ow, self.n_classes), dtype=torch.float32, device=self.device)
        out = torch.zeros((B, oh, ow, 1), dtype=torch.float32, device=self.device)
        net.sigmoid_argmax(plan, A['logits'], oh, ow, self.n_classes, sig, out)
        plan.net = net
        plan.acts = A
        return plan, x_in, sig, out

    # ---- Monte-Carlo dropout inference (BASELINE config 5; BUILD-DEFINED, parity unpinned) ----
    MC_SITES = ('conv2_2', 'conv5_2', 'conv6_2')      # stage-2 encoder, bottleneck, first decoder

    def infer_mc(self, imgs, passes=30, keep_prob=0.5, seed=5555, return_passes=False):
        """The reference U-Net accepts `bayesian` and ignores it; no Kendall-Gal head or dropout exists in it (SURVEY
        F13).  This method DEFINES the stochastic variant: slim.dropout-style masks x*Bernoulli(keep)/keep, always on,
        after conv2_2, conv5_2 and conv6_2 (the placement pattern of models/deconvolution.py:128-154), `passes`
        forward passes with fresh masks (pass t draws its counters from offset (t+1) << 40); returns [mean sigmoid,
        variance of sigmoid, float32 argmax of the mean] (+ the per-pass sigmoids with return_passes=True).
        The layers ahead of the first mask (conv1_1, pool1, conv1_2, conv2_1, conv2_2) are deterministic and are computed
        ONCE per input (SURVEY a19 allows it if stated); every pass runs from the first mask on.  The mean / variance
        accumulate in float64 on the device.  Restated in oracle/unet.py (infer_mc) with bit-identical masks."""
        imgs = np.ascontiguousarray(imgs, np.float32)
        ent = self._mc_entry(imgs.shape, keep_prob, seed)
        ent[2].copy_(torch.from_numpy(imgs))
        per = [] if return_passes else None
        mean, var, amax = self._mc_run(ent, passes, per)
        torch.cuda.synchronize(self.device)
        res = [mean.cpu().numpy(), var.cpu().numpy(), amax.cpu().numpy()]
        if return_passes:
            res.append(per)
        return res

    def _mc_entry(self, shape, keep_prob=0.5, seed=5555):
        """(prefix plan, per-pass plan, input buffer, sigmoid buffer, argmax buffer, counter offset) for an input shape"""
        import ctypes as C
        if self._packed_dirty:
            self._repack()
        key = ('mc',) + tuple(shape) + (keep_prob, seed)
        ent = self._infer_cache.get(key)
        if ent is None:
            B, H, W, Cin = shape
            net = E.Net(self.store, B, self.dtype, self.device)
            prefix, plan = E.Plan('infer_mc/prefix'), E.Plan('infer_mc/pass')
            x_in = torch.zeros((B, H, W, Cin), dtype=torch.float32, device=self.device)
            off = C.c_uint64(0)
            A, sh, sw, _, _ = self._emit_forward(net, prefix, x_in, H, W, self.crop_aware,
                                                 dropout={'sites': self.MC_SITES, 'keep': keep_prob, 'seed': seed, 'offset': off, 'split': plan})
            oh, ow = sh['output'], sw['output']
            sig = torch.zeros((B, oh, ow, self.n_classes), dtype=torch.float32, device=self.device)
            out = torch.zeros((B, oh, ow, 1), dtype=torch.float32, device=self.device)
            net.sigmoid_argmax(plan, A['logits'], oh, ow, self.n_classes, sig, out)
            plan.net, plan.acts = net, A
            ent = (prefix, plan, x_in, sig, out, off)
            self._infer_cache[key] = ent
        return ent

    def _mc_run(self, ent, passes, per=None):
        """Device side of infer_mc: the input is already in ent[2]; returns (mean, variance, argmax) device tensors."""
        prefix, plan, x_in, sig, out, off = ent
        prefix.run(self._stream())
        s1 = torch.zeros(sig.shape, dtype=torch.float64, device=self.device); s2 = torch.zeros_like(s1)
        for t in range(passes):
            off.value = (t + 1) << 40                    # disjoint counter ranges per pass
            plan.run(self._stream())
            d = sig.double()
            s1 += d; s2 += d * d                         # moments of the outputs (post-processing)
            if per is not None:
                per.append(sig.cpu().numpy())
        mean = s1 / passes
        var = torch.clamp(s2 / passes - mean * mean, min=0)
        amax = mean.argmax(dim=-1, keepdim=True).to(torch.float32)
        return mean, var, amax
