"""FCNModel: the reference's FCN-32s/16s/8s (/root/reference/models/fcn.py:25-220) compiled to HIP launch plans.

Graph facts reproduced on purpose: one 3x3 SAME conv + ReLU + 2x2 max-pool per stage, widths
n_kernels*{1,2,4,8,8}; conv6/conv7 are 1x1 convs of width 32*n_kernels; conv_fr, pool3_score and pool4_score keep
slim's default ReLU on the class scores; up-sampling is tf.nn.conv2d_transpose (SAME) with the *constant* bilinear
bank of utils/upsampling.py (not trained); skips are fused by ADDITION after resize_image_with_crop_or_pad; the
final map is crop-or-padded to the input size.  `fcn16s` crops with (pool4_h, pool4_h) -- the reference's typo
(models/fcn.py:166), harmless for square inputs, is kept.

MI355X design: the [k,k,C,C] bilinear bank is channel-diagonal, so the transposed conv runs as a depthwise
kernel (21x fewer MACs than the dense form TF executes) fused with the crop/pad and the skip addition.
"""
import os

import numpy as np
import torch

from . import engine as E
from . import upsampling
from .basemodel import BaseModel

ENC = ['conv1', 'conv2', 'conv3', 'conv4', 'conv5']


def fcn_layers(n_classes, n_kernels, input_channel, fcn_type):
    nk = n_kernels
    ls = {}

    def conv(name, ci, co, k, kind='conv'):
        ls[name] = E.Layer(name, kind, k, [ci], co, 'SAME', True)
    conv('conv1', input_channel, nk, 3, kind='first'); conv('conv2', nk, 2 * nk, 3); conv('conv3', 2 * nk, 4 * nk, 3)
    conv('conv4', 4 * nk, 8 * nk, 3); conv('conv5', 8 * nk, 8 * nk, 3)
    conv('conv6', 8 * nk, 32 * nk, 1); conv('conv7', 32 * nk, 32 * nk, 1); conv('conv_fr', 32 * nk, n_classes, 1)
    order = []                                    # backward-production order
    if fcn_type == '8s':
        conv('pool3_score', 4 * nk, n_classes, 1); conv('pool4_score', 8 * nk, n_classes, 1)
        order += ['pool3_score', 'pool4_score']
    elif fcn_type == '16s':
        conv('pool4_score', 8 * nk, n_classes, 1)
        order += ['pool4_score']
    order += ['conv_fr', 'conv7', 'conv6', 'conv5', 'conv4', 'conv3', 'conv2', 'conv1']
    ls['conv1'].need_dgrad = False
    return [ls[n] for n in order]


_AUX_PACK = os.environ.get('SEG_PACK_ON_AUX', '0') == '1'


class FCNModel(BaseModel):
    SHARE_AUX_STREAM = False     # (its forward starts with the input im2col on a side stream: a fourth stream measured 1-2 % faster)

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
                 fcn_type='32s',
                 autoencoder=False,
                 **mi355x):
        # examples/example_fcn.py:88 passes autoencoder=False, which the reference constructor rejects (F10): accepted here
        super(FCNModel, self).__init__(
            sess=sess, mode=mode, log_dir=log_dir, dataset=dataset, bayesian=bayesian, save_dir=save_dir,
            n_classes=n_classes, input_dims=input_dims, test_dataset=test_dataset, input_channel=input_channel,
            load_snapshot=load_snapshot, learning_rate=learning_rate, load_snapshot_from=load_snapshot_from,
            adversarial_training=adversarial_training, autoencoder=autoencoder, **mi355x)
        self.model_name = 'FCN'
        self.n_kernels = n_kernels
        if fcn_type not in ('32s', '16s', '8s'):
            raise Exception('MODE ERROR: fcn_type must be 32s, 16s or 8s')
        self.fcn_type = fcn_type
        if n_classes > 32:
            raise Exception('n_classes > 32 not supported')
        self._init_input()
        training = self.mode != 'INFERENCE'
        self.layers = fcn_layers(n_classes, n_kernels, input_channel, fcn_type)
        self.store = E.ParamStore(self.layers, self.dtype, self.device, training=training)
        names = ENC + ['conv6', 'conv7', 'conv_fr'] + (['pool4_score'] if fcn_type == '16s' else
                                                       ['pool3_score', 'pool4_score'] if fcn_type == '8s' else [])
        self._xavier_init(names)
        self._filters = {}
        self.net = None
        if training:
            self._build_training()
        self._repack_initial()
        self.inference_ops = ['y_hat_sig', 'output']
        self._init_saver(self.model_name)

    def _filt(self, factor):
        if factor not in self._filters:
            k = upsampling.get_kernel_size(factor)
            # the reference stores the bank as float32 (utils/upsampling.py:35-38)
            self._filters[factor] = torch.from_numpy(upsampling.upsample_filt(k).astype(np.float32)).to(self.device)
        return self._filters[factor]

    def model(self, input_op=None, reuse=False):
        return self.fwd_plan

    # ---- forward graph ----
    def _emit_forward(self, net, plan, x_in, H, W, final_up=True):
        Ly, nc = self.store.layers, self.n_classes
        A = {}
        h, w = H, W
        prev = None
        for i, name in enumerate(ENC):
            A[name] = net.act(h, w, Ly[name].cout, name=name)
            if h // 2 < 1 or w // 2 < 1:
                raise Exception('FCN: input %dx%d too small for five 2x2 pools' % (H, W))
            P = net.act(h // 2, w // 2, Ly[name].cout, name='pool%d' % (i + 1))
            pooled = False
            if i == 0:
                pooled = net.first_fwd(plan, Ly[name], x_in, H, W, A[name], pool=P)
                net.join_aux(plan)         # packed weights needed from here on
            else:
                net.conv_fwd(plan, Ly[name], [(prev, 0, 0)], h, w, A[name], pool=P)      # pool fused where the tile allows
                pooled = net.pool_fused
            h, w = h // 2, w // 2
            if not pooled:
                net.pool_fwd(plan, A[name], P, h, w)
            A['pool%d' % (i + 1)] = P
            prev = P
        for name in ('conv6', 'conv7', 'conv_fr'):
            A[name] = net.act(h, w, Ly[name].cout, name=name)
            net.conv_fwd(plan, Ly[name], [(prev, 0, 0)], h, w, A[name])
            prev = A[name]
        # final_up=False (training): the last up-sampling is fused with the loss (Net.bilinear_xent); geo['final'] names its input
        if final_up or self.keep_logits:
            A['logits'] = net.act(H, W, nc, f32=True, name='logits')
        geo = {}                                        # (src name, Hs, Ws, factor, dst name, Hd, Wd)
        fr = A['conv_fr']
        if self.fcn_type == '32s':
            if final_up:
                net.bilinear_fwd(plan, fr, fr.H, fr.W, 32, self._filt(32), None, A['logits'], H, W, dst_f32=True)
            geo['final'] = ('conv_fr', 32)
        else:
            p4 = A['pool4']
            A['pool4_score'] = net.act(p4.H, p4.W, nc, name='pool4_score')
            net.conv_fwd(plan, Ly['pool4_score'], [(p4, 0, 0)], p4.H, p4.W, A['pool4_score'])
            h4, w4 = p4.H, (p4.H if self.fcn_type == '16s' else p4.W)        # reference typo kept for 16s
            if (h4, w4) != (p4.H, p4.W):
                raise Exception('fcn16s needs square inputs (reference crops with (pool4_h, pool4_h))')
            A['fuse4'] = net.act(p4.H, p4.W, nc, name='fuse4')
            net.bilinear_fwd(plan, fr, fr.H, fr.W, 2, self._filt(2), A['pool4_score'], A['fuse4'], p4.H, p4.W)
            if self.fcn_type == '16s':
                if final_up:
                    net.bilinear_fwd(plan, A['fuse4'], p4.H, p4.W, 16, self._filt(16), None, A['logits'], H, W, dst_f32=True)
                geo['final'] = ('fuse4', 16)
            else:
                p3 = A['pool3']
                A['pool3_score'] = net.act(p3.H, p3.W, nc, name='pool3_score')
                net.conv_fwd(plan, Ly['pool3_score'], [(p3, 0, 0)], p3.H, p3.W, A['pool3_score'])
                A['fuse3'] = net.act(p3.H, p3.W, nc, name='fuse3')
                net.bilinear_fwd(plan, A['fuse4'], p4.H, p4.W, 2, self._filt(2), A['pool3_score'], A['fuse3'], p3.H, p3.W)
                if final_up:
                    net.bilinear_fwd(plan, A['fuse3'], p3.H, p3.W, 8, self._filt(8), None, A['logits'], H, W, dst_f32=True)
                geo['final'] = ('fuse3', 8)
        return A, geo

    # ---- training plans ----
    def _build_training(self):
        B, (H, W) = self.batch_size, self.input_dims
        net = self.net = E.Net(self.store, B, self.dtype, self.device)
        net.n_wgrad_streams = max(1, len(self._side) - 1) if self._side else 1
        net.input_pixels = B * H * W if not self.pg.tuned else None      # (data parallel keeps the lone-launch filter-gradient target: 1.05 against 1.09 ms at world 1)
        net.tail_layers = ('conv2',)      # last tiled filter gradient of the backward pass: aims for the whole chip (see unet.py); +2 % at C3
        Ly, nc = self.store.layers, self.n_classes
        fwd = self.fwd_plan = E.Plan('fwd')
        self.loss_buf = self.store.loss_slot()          # (behind the gradient arena: reduced with the last bucket under data parallelism)
        net.step_begin(fwd, self.loss_buf, aux=_AUX_PACK or self.pg.tuned)     # aux stream: global_step += 1, loss accumulator = 0
        net.pack(fwd, aux=_AUX_PACK or self.pg.tuned)            # refresh the packed weights after the previous Adam step, beside conv1
        col = net.first_im2col(fwd, Ly['conv1'], self.input_x, H, W)         # side stream, overlaps the forward pass
        # The last up-sampling, the loss and dlogits are ONE launch (no 268 MB float logits tensor at 512^2 x 21 classes); the
        # adversary needs the logits as a tensor and keeps the separate launches; keep_logits=True also stores them (y_hat, tests).
        fused_head = not self.adversarial_training and os.environ.get('SEG_FUSE_HEAD', '1') != '0'
        A, geo = self._emit_forward(net, fwd, self.input_x, H, W, final_up=not fused_head)
        self.acts = A
        self.out_hw = (H, W)
        self.label_off = (0, 0)
        dlog = net.act(H, W, nc, name='dlogits')
        if fused_head:
            src = A[geo['final'][0]]
            net.bilinear_xent(fwd, src, src.H, src.W, geo['final'][1], self._filt(geo['final'][1]), self.input_y, H, W, (0, 0), H, W, nc,
                              self.loss_buf, dlog, logits=A.get('logits'))
        else:
            probs = self._make_adversary(H, W, dlog) if self.adversarial_training else None
            net.softmax_xent(fwd, A['logits'], self.input_y, H, W, (0, 0), H, W, nc, self.loss_buf, dlog, probs=probs)
        if self.adversarial_training:
            self._attach_adversary(A['logits'], H, W, H, W, dlog)
        self.dlogits = dlog
        seg = E.Plan('bwd0')
        segs = []

        def act_like(a, name):
            return net.act(a.H, a.W, a.C, name=name)

        def score_bwd(score_name, src_pool, dz, dpool):
            """dz = gradient of (score + cropped up-sampled stream) behind the score conv's own ReLU (written by the bilinear
            adjoint that produced the fused gradient)."""
            net.conv_bwd(seg, Ly[score_name], [(src_pool, 0, 0)], src_pool.H, src_pool.W, dz, [(dpool, (0, 0), None, (0, 0))])

        fr = A['conv_fr']
        dfr = act_like(fr, 'd_conv_fr')
        dzfr = act_like(fr, 'dz_conv_fr')
        dP3 = dP4 = None
        src_name, f_final = geo['final']
        if self.fcn_type == '32s':
            net.bilinear_bwd(seg, dlog, H, W, 32, self._filt(32), dfr, fr.H, fr.W, mask=fr, dz=dzfr)
        else:
            p4 = A['pool4']
            dP4 = act_like(p4, 'dpool4')
            dfuse4 = act_like(A['fuse4'], 'd_fuse4')
            dz4 = act_like(A['pool4_score'], 'dz_pool4_score')
            if self.fcn_type == '16s':
                net.bilinear_bwd(seg, dlog, H, W, 16, self._filt(16), dfuse4, p4.H, p4.W, mask=A['pool4_score'], dz=dz4)
            else:
                p3 = A['pool3']
                dP3 = act_like(p3, 'dpool3')
                dfuse3 = act_like(A['fuse3'], 'd_fuse3')
                dz3 = act_like(A['pool3_score'], 'dz_pool3_score')
                net.bilinear_bwd(seg, dlog, H, W, 8, self._filt(8), dfuse3, p3.H, p3.W, mask=A['pool3_score'], dz=dz3)
                score_bwd('pool3_score', p3, dz3, dP3)
                net.bilinear_bwd(seg, dfuse3, p3.H, p3.W, 2, self._filt(2), dfuse4, p4.H, p4.W, mask=A['pool4_score'], dz=dz4)
            score_bwd('pool4_score', p4, dz4, dP4)
            net.bilinear_bwd(seg, dfuse4, p4.H, p4.W, 2, self._filt(2), dfr, fr.H, fr.W, mask=fr, dz=dzfr)
        dz7 = act_like(A['conv7'], 'dz_conv7')
        net.conv_bwd(seg, Ly['conv_fr'], [(A['conv7'], 0, 0)], fr.H, fr.W, dzfr, [(dz7, (0, 0), A['conv7'], (0, 0))])
        dz6 = act_like(A['conv6'], 'dz_conv6')
        net.conv_bwd(seg, Ly['conv7'], [(A['conv6'], 0, 0)], fr.H, fr.W, dz7, [(dz6, (0, 0), A['conv6'], (0, 0))])
        dP = {5: act_like(A['pool5'], 'dpool5')}
        net.conv_bwd(seg, Ly['conv6'], [(A['pool5'], 0, 0)], fr.H, fr.W, dz6, [(dP[5], (0, 0), None, (0, 0))])
        l = Ly['conv6']
        net.flush_reduce(seg)
        segs.append((seg, l.b_off + l.nbias))
        seg = E.Plan('bwd1')
        for i in (5, 4, 3, 2, 1):
            name = 'conv%d' % i
            a = A[name]
            if i == 1 and net.fuses_first_pool_bwd(col):
                # pool1's backward happens inside the first layer's filter gradient (no launch, no dZ(conv1) tensor)
                aux_tail = not self.pg.enabled and not self.adversarial_training
                if aux_tail:
                    self._aux_tail_layer = name
                net.first_bwd(seg, Ly[name], self.input_x, H, W, None, same_stream=not self.pg.enabled, pool=(a, dP[i], None, (0, 0), (0, 0)),
                              on_aux=aux_tail)
                break
            dz = act_like(a, 'dz_' + name)
            net.pool_bwd(seg, a, dP[i], None, (0, 0), (0, 0), dz, a.H, a.W)
            if i == 1:
                net.first_bwd(seg, Ly[name], self.input_x, H, W, dz, col=col, same_stream=not self.pg.enabled)
                break
            pin = A['pool%d' % (i - 1)]
            # pool4 / pool3 already hold the score-branch gradient: the encoder path accumulates onto it
            shared = dP4 if (i == 5 and dP4 is not None) else dP3 if (i == 4 and dP3 is not None) else None
            dP[i - 1] = shared if shared is not None else act_like(pin, 'dpool%d' % (i - 1))
            net.conv_bwd(seg, Ly[name], [(pin, 0, 0)], pin.H, pin.W, dz, [(dP[i - 1], (0, 0), None, (0, 0), shared is not None)])
        l = Ly['conv1']
        net.flush_reduce(seg)
        segs.append((seg, l.b_off + l.nbias))
        self._finish_training_plans(segs)
        self.y_hat = A.get('logits')          # None unless keep_logits / adversarial training (the fused head does not store them)

    def _build_infer(self, B, H, W, Cin):
        if Cin != self.input_channel:
            raise Exception('infer(): expected %d input channels, got %d' % (self.input_channel, Cin))
        net = E.Net(self.store, B, self.dtype, self.device)
        plan = E.Plan('infer')
        x_in = torch.zeros((B, H, W, Cin), dtype=torch.float32, device=self.device)
        A, _ = self._emit_forward(net, plan, x_in, H, W)
        sig = torch.zeros((B, H, W, self.n_classes), dtype=torch.float32, device=self.device)
        out = torch.zeros((B, H, W, 1), dtype=torch.float32, device=self.device)
        net.sigmoid_argmax(plan, A['logits'], H, W, self.n_classes, sig, out)
        plan.net = net
        plan.acts = A
        return plan, x_in, sig, out
