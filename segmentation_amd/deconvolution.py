"""DeconvModel: the reference's convolution / deconvolution net (/root/reference/models/deconvolution.py:26-178) compiled to
HIP launch plans -- the one segmentation model of the reference with BatchNorm and `bayesian` dropout.

Graph (slim defaults: bias + ReLU on every conv / transposed conv except conv_out; batch_norm decay 0.999, eps 1e-3, beta
only, applied to the ReLU OUTPUT of the layer in front of it):
  conv1_0 5x5/s2 SAME -> bn1 -> pool 2x2 -> conv2_0 3x3 VALID -> bn2 [-> dropout] -> pool 3x3 -> conv3_0 -> bn3 -> pool 3x3
  -> conv4_0 -> bn4 [-> dropout] -> deconv1_0 5x5/s2 VALID -> bn5 [-> dropout] -> deconv2_0 -> bn6 -> deconv2_1 -> bn7
  -> resize_bilinear(H/2, W/2) -> deconv3_0 2x2/s2 -> bn8 -> crop_or_pad(H, W) -> conv_out 3x3 SAME (no activation)
Kernels: the 3x3 and 2x2/s2 layers run on the MFMA tiles (csrc/conv_fwd.hip, conv_wgrad.hip); the 5x5 stride-2 conv and the
three 5x5 stride-2 transposed convs on the direct kernels of csrc/deconv_ops.hip; batch norm = seg_bn_fwd / seg_bn_relu_bwd
(three fixed-order reduction stages, the backward fused with the ReLU-grad of the layer in front).

Reference behaviour kept on purpose: infer() runs the TRAINING graph -- y_hat is built with training=True
(models/deconvolution.py:80,102), so inference normalises with the statistics of the batch it is given and, when
`bayesian`, keeps dropout ON (slim.dropout's is_training defaults to True): every infer() call is one stochastic pass;
test() is the moving-average graph (models/basemodel.py:397).  Moving averages are updated by train_step()
(UPDATE_OPS, models/basemodel.py:364-365).  Dropout masks come from the build's counter-based generator (seed, site, global
step): TF's RNG stream cannot be reproduced, so the masks are build-defined (oracle.np_ops.dropout_mask restates them).
Limits: even input sizes (the reference pads an odd one by a row after deconv3_0; rejected here), input >= 140 pixels.
Data parallel: batch statistics are per rank (like slim towers); rank 0's moving averages are the ones snapshotted."""
import os

import numpy as np
import torch

from . import _lib as L
from . import engine as E
from .basemodel import BaseModel

# (name, kind, k, stride, padding, cin factor, cout factor) in graph order; factors multiply n_kernels (0 = special)
GRAPH = ['conv1_0', 'bn1', 'conv2_0', 'bn2', 'conv3_0', 'bn3', 'conv4_0', 'bn4', 'deconv1_0', 'bn5', 'deconv2_0', 'bn6',
         'deconv2_1', 'bn7', 'deconv3_0', 'bn8', 'conv_out']
BWD_ORDER = list(reversed(GRAPH))
TEST_SEED_INC = 64                                            # test() draws from its own dropout streams
DROP_SITES = {'bn2': 1, 'bn4': 2, 'bn5': 3}                    # batch norm followed by a `bayesian` dropout -> seed increment


def thin_tail(n_classes):
    """The full-resolution tail (deconv3_0 -> bn8 -> conv_out -> logits, their gradients) carries n_classes channels.  Up to 8
    classes the tensors are kept THIN: 8 channels per pixel instead of 32 (engine.Act(thin=True), seg_conv_desc.thin_src /
    n_store) -- at 512^2 x 16 every pass over one of them moves 67 MB instead of 268 MB.  SEG_THIN_TAIL=0: padded to 32."""
    return n_classes <= 8 and os.environ.get('SEG_THIN_TAIL', '1') != '0'


def deconv_layers(n_classes, nk, cin):
    Ly = {}

    def add(name, kind, k, ci, co, padding='VALID', relu=True, stride=1):
        Ly[name] = E.Layer(name, kind, k, [ci], co, padding, relu, stride)
    # conv1_0 (5x5/s2 SAME on the raw image) runs as a 1x1 convolution over the im2col of its input (K = 25*cin, seg_im2col): MFMA
    # forward and filter gradient instead of the direct kernels; its parameter stays the [5,5,cin,nk] HWIO tensor (same memory order)
    add('conv1_0', 'conv', 1, 25 * cin, nk, 'VALID'); Ly['conv1_0'].wshape = (5, 5, cin, nk); add('bn1', 'bn', 1, nk, nk)
    add('conv2_0', 'conv', 3, nk, 2 * nk); add('bn2', 'bn', 1, 2 * nk, 2 * nk)
    add('conv3_0', 'conv', 3, 2 * nk, 4 * nk); add('bn3', 'bn', 1, 4 * nk, 4 * nk)
    add('conv4_0', 'conv', 3, 4 * nk, 8 * nk); add('bn4', 'bn', 1, 8 * nk, 8 * nk)
    # the 5x5/s2 transposed convolutions: on the MFMA kernels as adjoints of strided convolutions (engine.Net.tconv_layer) when their
    # channel counts allow the 16-byte gathers (multiples of 8), else on the direct kernels
    def addt(name, ci, co):
        if co % 8 == 0 and os.environ.get('SEG_TCONV_MFMA', '1') != '0':
            Ly[name] = E.Net.tconv_layer(name, 5, ci, co, 2)
        else:
            add(name, 'dtrans', 5, ci, co, stride=2)
    addt('deconv1_0', 8 * nk, 2 * nk); add('bn5', 'bn', 1, 2 * nk, 2 * nk)
    addt('deconv2_0', 2 * nk, nk); add('bn6', 'bn', 1, nk, nk)
    addt('deconv2_1', nk, nk); add('bn7', 'bn', 1, nk, nk)
    add('deconv3_0', 'up', 2, nk, n_classes); add('bn8', 'bn', 1, n_classes, n_classes)
    add('conv_out', 'conv', 3, n_classes, n_classes, 'SAME', relu=False)
    if thin_tail(n_classes):
        Ly['bn8'].cout_p = 8            # (its statistics vectors follow the 8-channel stride of the tensors it normalises)
    Ly['conv1_0'].need_dgrad = False
    return [Ly[n] for n in BWD_ORDER]


def deconv_sizes(H):
    """spatial ladder for an (even) input edge H; raises when the all-VALID middle collapses"""
    if H % 2:
        raise Exception('DeconvModel needs even input sizes (got %d)' % H)
    s = {'conv1_0': (H + 1) // 2}
    s['pool1'] = s['conv1_0'] // 2
    s['conv2_0'] = s['pool1'] - 2; s['pool2'] = s['conv2_0'] // 3
    s['conv3_0'] = s['pool2'] - 2; s['pool3'] = s['conv3_0'] // 3
    s['conv4_0'] = s['pool3'] - 2
    if min(s.values()) < 1:
        raise Exception('DeconvModel infeasible for input %d (needs >= 140): conv4_0 would see a %dx%d map' % (H, s['pool3'], s['pool3']))
    s['deconv1_0'] = 2 * s['conv4_0'] + 3; s['deconv2_0'] = 2 * s['deconv1_0'] + 3; s['deconv2_1'] = 2 * s['deconv2_0'] + 3
    s['resize'] = H // 2; s['deconv3_0'] = 2 * s['resize']
    return s


_AUX_PACK = os.environ.get('SEG_PACK_ON_AUX', '0') == '1'


class DeconvModel(BaseModel):
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
                 autoencoder=False,
                 adversarial_training=False,
                 **mi355x):
        if adversarial_training:
            raise Exception('adversarial_training is wired for UNetModel and FCNModel only')
        super(DeconvModel, self).__init__(
            sess=sess, mode=mode, log_dir=log_dir, dataset=dataset, bayesian=bayesian, save_dir=save_dir,
            n_classes=n_classes, input_dims=input_dims, autoencoder=autoencoder, test_dataset=test_dataset,
            input_channel=input_channel, load_snapshot=load_snapshot, learning_rate=learning_rate,
            load_snapshot_from=load_snapshot_from, adversarial_training=adversarial_training, **mi355x)
        self.model_name = 'deconvolution'
        self.IN_OUT_EQUAL = True
        self.n_kernels = n_kernels
        if n_classes > 32:
            raise Exception('n_classes > 32 not supported')
        # dropout streams: one generator stream per (seed, site); training draws at counter offset global_step << 40 with
        # seed + site, test() with seed + 64 + site (its own streams: the masks of the last train step are not replayed), infer() at
        # offsets (1 << 62) | call << 40 (disjoint from every training step); data-parallel ranks fold their rank into the seed
        self.keep_prob = 0.5
        self.drop_seed = self.seed + 4096 * self.pg.rank
        self._init_input()
        training = self.mode != 'INFERENCE'
        self.layers = deconv_layers(n_classes, n_kernels, input_channel)
        self.store = E.ParamStore(self.layers, self.dtype, self.device, training=training)
        self._xavier_init(GRAPH)
        self.bn = None
        self.net = None
        if training:
            self._build_training()
        self._repack_initial()
        self.inference_ops = ['y_hat_sig', 'output']
        self._init_saver(self.model_name)

    def model(self, input_op=None, reuse=False, training=True):
        return self.fwd_plan if training else self.test_plan

    # ---- batch-norm state (shared by every plan of this model) ----
    def _bn_states(self, net):
        if self.bn is None:
            self.bn = {n: net.bn_state(self.store.layers[n]) for n in GRAPH if n.startswith('bn')}
        return self.bn

    def _state_blob(self):
        out = {}
        for n, st in (self.bn or {}).items():
            c = self.store.layers[n].cout
            cp = self.store.layers[n].cout_p
            mv = st['moving'].cpu().numpy()
            out[n + '/moving_mean'] = mv[:c].copy(); out[n + '/moving_variance'] = mv[cp:cp + c].copy()
        return out

    def _load_state(self, z):
        if self.bn is None:
            self._bn_states(self.net or E.Net(self.store, 1, self.dtype, self.device))
        for n, st in self.bn.items():
            if n + '/moving_mean' in z:
                c, cp = self.store.layers[n].cout, self.store.layers[n].cout_p
                mv = st['moving'].cpu().numpy()
                mv[:c] = z[n + '/moving_mean']; mv[cp:cp + c] = z[n + '/moving_variance']
                st['moving'].copy_(torch.from_numpy(mv))

    def set_moving(self, moving):
        """moving: {bn: (moving_mean, moving_variance)} (tests / interchange)"""
        blob = {}
        for n, (m_, v_) in moving.items():
            blob[n + '/moving_mean'] = np.asarray(m_, np.float32); blob[n + '/moving_variance'] = np.asarray(v_, np.float32)
        self._load_state(blob)

    def get_moving(self):
        """{bn: (moving_mean, moving_variance)} as numpy (tests / interchange)"""
        return {n: (v[n + '/moving_mean'], v[n + '/moving_variance']) for v in [self._state_blob()] for n in self.bn}

    # ---- forward graph (training / test / inference plans share it) ----
    def _emit_forward(self, net, plan, x_in, H, W, bn_training, update_moving, dropout_step, seed_inc=0, want_col=True):
        """dropout_step: True -> masks keyed by the device-side global step (train / test plans); False -> by a host offset
        (ctypes c_uint64 in self._infer_off: one fresh mask per infer() call)"""
        if H != W:
            raise Exception('DeconvModel plans are built for square inputs (got %dx%d)' % (H, W))
        Ly, nk = self.store.layers, self.n_kernels
        sz = deconv_sizes(H)
        bn = self._bn_states(net)
        A, Y = {}, {}
        pad = max((sz['conv1_0'] - 1) * 2 + 5 - H, 0) // 2           # TF SAME, stride 2: leading pad = total // 2
        # bf16, <= 3 input channels: conv1_0 straight from the image (seg_conv_first_gen); its im2col tensor is then only the filter
        # gradient's operand -- a side stream writes it behind the forward pass (training plan), nobody else asks for it
        mode = os.environ.get('SEG_FIRST_GEN', '1')
        direct = net.dtype == L.SEG_BF16 and self.input_channel <= 3 and nk <= 64 and mode != '0'
        xin = None
        # ... and with the filter gradient taken from the image as well (seg_conv_first_gen_wgrad) there is no im2col at all
        direct_bwd = direct and os.environ.get('SEG_FIRST_GEN_WGRAD', '1') != '0'
        if want_col:
            self._col_late = None
            self._first_direct_bwd = (direct_bwd, pad)
        if (want_col and not direct_bwd) or not direct:
            xin = net.act(sz['conv1_0'], sz['conv1_0'], 25 * self.input_channel, name='x_im2col')
            cv = xin.view()
            plan.keep.append(cv)
            col_args = ('conv1_0/im2col', net.lib.seg_im2col, x_in.data_ptr(), net.B, H, W, self.input_channel, 5, 5, 2, pad, pad, E.C.byref(cv),
                        sz['conv1_0'], sz['conv1_0'], net.dtype)
            if direct and mode != 'fwd' and net.side_enabled:
                # emitted by the backward plan on the filter gradient's stream, half a backward pass ahead of it, off the critical
                # stream (16 x 512^2 step: 1.631 against 1.643 ms with the im2col in the forward pass -- SEG_FIRST_GEN=fwd -- or with
                # im2col + 1x1 convolution -- SEG_FIRST_GEN=0; the plans without a backward pass simply have no im2col: inference
                # 0.73 -> 0.60 ms)
                self._col_late = col_args
            else:
                plan.add(*col_args, kernel='im2col_kernel')
        net.join_aux(plan)                     # packed weights (re-packed on the aux stream in training) are needed from here on

        def act_bn(name, a, rows=0):
            """ReLU output of `name` -> batch norm (-> dropout); rows: statistics rows the producer of `a` left for it"""
            b = 'bn' + str(GRAPH.index(name) // 2 + 1)
            A[name] = a
            Y[b] = net.act(a.H, a.W, a.C, name=b, thin=a.thin)
            net.bn_fwd(plan, Ly[b], bn[b], a, Y[b], training=bn_training, update_moving=update_moving, rows=rows)
            out = Y[b]
            if self.bayesian and b in DROP_SITES:
                Y[b + '/drop'] = net.act(a.H, a.W, a.C, name=b + '/drop', thin=a.thin)
                if dropout_step:
                    net.dropout_step(plan, Y[b], Y[b + '/drop'], self.keep_prob, self.drop_seed + seed_inc + DROP_SITES[b], 0)
                else:
                    v, o = Y[b].view(), Y[b + '/drop'].view()
                    plan.keep += [v, o]
                    plan.add('dropout', net.lib.seg_dropout, E.C.byref(v), E.C.byref(o), net.B, a.H, a.W, a.Cp, float(self.keep_prob),
                             int(self.drop_seed + DROP_SITES[b]), self._infer_off, net.dtype, kernel='dropout_kernel')
                out = Y[b + '/drop']
            return out

        a = net.act(sz['conv1_0'], sz['conv1_0'], nk, name='conv1_0')
        P = {}

        def bn_pool(name, a, pooled, k, rows=0):
            """ReLU output of `name` -> batch norm -> k x k max-pool.  bf16 without a dropout in between: one pass, the normalised
            tensor is never stored and Y[bn + '/poolsrc'] = a tells the backward plan where the pool finds its maxima."""
            b = 'bn' + str(GRAPH.index(name) // 2 + 1)
            if net.dtype == L.SEG_BF16 and not (self.bayesian and b in DROP_SITES) and os.environ.get('SEG_BN_POOL', '1') != '0':
                A[name] = a
                Y[b] = None
                Y[b + '/poolsrc'] = a
                net.bn_pool_fwd(plan, Ly[b], bn[b], a, pooled, k, training=bn_training, update_moving=update_moving, rows=rows)
                return
            t_ = act_bn(name, a, rows)
            Y[b + '/poolsrc'] = t_
            net.pool_k_fwd(plan, t_, pooled, k)

        if direct:
            rows = net.first_gen_fwd(plan, Ly['conv1_0'], x_in, H, W, self.input_channel, 5, 5, 2, pad, pad, a, bn_st=bn['bn1'] if bn_training else None)
        else:
            rows = 0
            net.conv_fwd(plan, Ly['conv1_0'], [(xin, 0, 0)], xin.H, xin.W, a)
        P[1] = net.act(sz['pool1'], sz['pool1'], nk, name='pool1')
        bn_pool('conv1_0', a, P[1], 2, rows)
        for i, (cn, k) in enumerate((('conv2_0', 3), ('conv3_0', 3)), 2):
            a = net.act(sz[cn], sz[cn], Ly[cn].cout, name=cn)
            net.conv_fwd(plan, Ly[cn], [(P[i - 1], 0, 0)], P[i - 1].H, P[i - 1].W, a)
            P[i] = net.act(sz['pool%d' % i], sz['pool%d' % i], Ly[cn].cout, name='pool%d' % i)
            bn_pool(cn, a, P[i], k)
        a = net.act(sz['conv4_0'], sz['conv4_0'], Ly['conv4_0'].cout, name='conv4_0')
        net.conv_fwd(plan, Ly['conv4_0'], [(P[3], 0, 0)], P[3].H, P[3].W, a)
        t = act_bn('conv4_0', a)
        for dn in ('deconv1_0', 'deconv2_0', 'deconv2_1'):
            tc = getattr(Ly[dn], 'tconv', None)
            a = net.act(sz[dn], sz[dn], tc[2] if tc else Ly[dn].cout, name=dn)
            if tc:
                net.tconv_fwd(plan, Ly[dn], t, a)
            else:
                net.dlayer_fwd(plan, Ly[dn], t, a)
            t = act_bn(dn, a)
        R = net.act(sz['resize'], sz['resize'], t.C, name='resize')
        net.resize_fwd(plan, t, R)
        thin = thin_tail(self.n_classes)
        a = net.act(sz['deconv3_0'], sz['deconv3_0'], self.n_classes, name='deconv3_0', thin=thin)
        rows = net.up_fwd(plan, Ly['deconv3_0'], R, R.H, R.W, a, bn_st=bn['bn8'] if bn_training else None)
        A['logits'] = net.act(H, W, self.n_classes, f32=True, name='logits', thin=thin)
        if thin and net.dtype == L.SEG_BF16 and os.environ.get('SEG_THIN_VALU', '1') != '0' and os.environ.get('SEG_BN_ON_LOAD', '1') != '0':
            # bn8 is never materialised: its statistics alone, conv_out (and its filter gradient) normalise deconv3_0's output on load
            A['deconv3_0'] = a
            Y['bn8'] = None
            net.bn_stats(plan, Ly['bn8'], bn['bn8'], a, training=bn_training, update_moving=update_moving, rows=rows)
            net.conv_fwd(plan, Ly['conv_out'], [(a, 0, 0)], H, W, A['logits'], out_f32=True, src_bn=(bn['bn8'], Ly['bn8']))
        else:
            t = act_bn('deconv3_0', a, rows)             # (resize_image_with_crop_or_pad to (H, W) is the identity for even H)
            net.conv_fwd(plan, Ly['conv_out'], [(t, 0, 0)], H, W, A['logits'], out_f32=True)
        A['x'], A['resize'] = xin, R
        return A, Y, P, sz

    # ---- training plans ----
    def _build_training(self):
        B, (H, W) = self.batch_size, self.input_dims
        net = self.net = E.Net(self.store, B, self.dtype, self.device)
        net.n_wgrad_streams = max(1, len(self._side) - 1) if self._side else 1
        net.input_pixels = B * H * W if not self.pg.tuned else None      # (data parallel keeps the lone-launch filter-gradient target: 1.05 against 1.09 ms at world 1)
        Ly, nc = self.store.layers, self.n_classes
        fwd = self.fwd_plan = E.Plan('fwd')
        self.loss_buf = self.store.loss_slot()          # (behind the gradient arena: reduced with the last bucket under data parallelism)
        net.step_begin(fwd, self.loss_buf, aux=_AUX_PACK or self.pg.tuned)
        net.pack(fwd, aux=_AUX_PACK or self.pg.tuned)
        A, Y, P, sz = self._emit_forward(net, fwd, self.input_x, H, W, bn_training=True, update_moving=True, dropout_step=True)
        self.acts, self.bn_out = A, Y
        self.out_hw, self.label_off = (H, W), (0, 0)
        dlog = net.act(H, W, nc, name='dlogits', thin=thin_tail(nc))
        net.softmax_xent(fwd, A['logits'], self.input_y, H, W, (0, 0), H, W, nc, self.loss_buf, dlog)
        self.dlogits = dlog
        # the test() graph: moving averages, no update; dropout (if bayesian) stays on, exactly as in the reference's test graph
        tnet = self.test_net = E.Net(self.store, B, self.dtype, self.device)
        self.test_plan = E.Plan('test')
        TA, _, _, _ = self._emit_forward(tnet, self.test_plan, self.input_x, H, W, bn_training=False, update_moving=False, dropout_step=True, want_col=False,
                                         seed_inc=TEST_SEED_INC)
        tdl = tnet.act(H, W, nc, name='dlogits_test', thin=thin_tail(nc))
        tnet.softmax_xent(self.test_plan, TA['logits'], self.input_y, H, W, (0, 0), H, W, nc, self.loss_buf, tdl)

        seg = E.Plan('bwd0')

        def like(a, name):
            return net.act(a.H, a.W, a.C, name=name, thin=a.thin)

        G = {}

        def bn_bwd(b, name, dy):
            """gradient at the (dropout) output behind batch norm `b` -> masked pre-activation gradient of layer `name`"""
            if self.bayesian and b in DROP_SITES:
                d2 = like(Y[b], 'd_' + b)
                net.dropout_step(seg, dy, d2, self.keep_prob, self.drop_seed + DROP_SITES[b], 0)
                dy = d2
            G[name] = like(A[name], 'dz_' + name)
            net.bn_relu_bwd(seg, Ly[b], self.bn[b], A[name], dy, G[name])
            return G[name]

        last = lambda b: Y[b + '/drop'] if (self.bayesian and b in DROP_SITES) else Y[b]

        def pool_bn_bwd(b, cn, dpool, k):
            """pooled gradient -> masked pre-activation gradient of layer `cn` through the max-pool and batch norm `b`"""
            src = Y[b + '/poolsrc']                  # (the normalised tensor, its dropout, or -- fused forward -- the pre-BN activation)
            if Y[b] is None and os.environ.get('SEG_BN_POOL_BWD', '1') != '0':
                G[cn] = like(A[cn], 'dz_' + cn)
                net.bn_pool_relu_bwd(seg, Ly[b], self.bn[b], A[cn], dpool, G[cn], k)
                return G[cn]
            d_ = like(src, 'd_' + b + '_out')
            net.pool_k_bwd(seg, src, dpool, d_, k)
            return bn_bwd(b, cn, d_)
        # conv_out
        if Y['bn8'] is None:                      # (conv_out read deconv3_0's output through bn8 on load: so does its filter gradient)
            dY8 = like(A['deconv3_0'], 'd_bn8')
            net.conv_bwd(seg, Ly['conv_out'], [(A['deconv3_0'], 0, 0)], H, W, dlog, [(dY8, (0, 0), None, (0, 0))], src_bn=(self.bn['bn8'], Ly['bn8']))
        else:
            dY8 = like(Y['bn8'], 'd_bn8')
            net.conv_bwd(seg, Ly['conv_out'], [(Y['bn8'], 0, 0)], H, W, dlog, [(dY8, (0, 0), None, (0, 0))])
        dz = bn_bwd('bn8', 'deconv3_0', dY8)
        dR = like(A['resize'], 'd_resize')
        net.up_bwd(seg, Ly['deconv3_0'], A['resize'], A['resize'].H, A['resize'].W, dz, dR, None)
        d = like(Y['bn7'], 'd_bn7')
        net.resize_bwd(seg, dR, d)
        srcs = {'deconv2_1': 'bn6', 'deconv2_0': 'bn5', 'deconv1_0': 'bn4'}
        for dn, b in (('deconv2_1', 'bn7'), ('deconv2_0', 'bn6'), ('deconv1_0', 'bn5')):
            dz = bn_bwd(b, dn, d)
            src = last(srcs[dn])
            d = like(src, 'd_' + srcs[dn] + '_out')
            if getattr(Ly[dn], 'tconv', None):
                net.tconv_bwd(seg, Ly[dn], src, dz, dsrc=d)
            else:
                net.dlayer_bwd(seg, Ly[dn], src, dz, dsrc=d, mask=None)
        dz = bn_bwd('bn4', 'conv4_0', d)
        col_sid = None
        if self._col_late is not None:
            col_sid = 2 if net.n_wgrad_streams >= 2 else 1
            seg.add(*self._col_late, kernel='im2col_kernel', side=col_sid)
        dP = {3: like(P[3], 'dpool3')}
        net.conv_bwd(seg, Ly['conv4_0'], [(P[3], 0, 0)], P[3].H, P[3].W, dz, [(dP[3], (0, 0), None, (0, 0))])
        for i, (cn, b, k) in ((3, ('conv3_0', 'bn3', 3)), (2, ('conv2_0', 'bn2', 3))):
            dz = pool_bn_bwd(b, cn, dP[i], k)
            dP[i - 1] = like(P[i - 1], 'dpool%d' % (i - 1))
            net.conv_bwd(seg, Ly[cn], [(P[i - 1], 0, 0)], P[i - 1].H, P[i - 1].W, dz, [(dP[i - 1], (0, 0), None, (0, 0))])
        dz = pool_bn_bwd('bn1', 'conv1_0', dP[1], 2)
        if self._first_direct_bwd[0]:
            net.first_gen_bwd(seg, Ly['conv1_0'], self.input_x, H, W, self.input_channel, 5, 5, 2, self._first_direct_bwd[1], self._first_direct_bwd[1], dz)
        else:
            net.conv_bwd(seg, Ly['conv1_0'], [(A['x'], 0, 0)], A['x'].H, A['x'].W, dz, [None], wgrad_sid=col_sid)
        self.grads_act = G
        net.flush_reduce(seg)
        l = Ly['conv1_0']
        self._finish_training_plans([(seg, l.b_off + l.nbias)])
        self.y_hat = A['logits']

    def test(self):
        """moving-average graph on a held-out batch, mean x-entropy (models/basemodel.py:397,420-421,506-518)"""
        if self.mode == 'INFERENCE':
            print('test() with INFERENCE mode invalid')
            return
        ds = self.test_dataset if self.test_dataset is not None else self.dataset
        train_loss = self.loss_buf.clone()
        if self._packed_dirty:
            self._repack()                 # the MFMA layers read the packed copy, which train_step refreshes at the head of the NEXT step
        self._load_batch(ds, self.input_x, self.input_y)           # (the test plan reads the model's own input buffers)
        self.loss_buf.zero_()
        self.test_plan.run(self._stream())
        self.last_test_loss = float(self.loss_buf.item())
        self.loss_buf.copy_(train_loss)
        print('TEST LOSS', self.last_test_loss, self.global_step)
        self.write_summary({'test_loss': self.last_test_loss})

    # ---- inference: the reference feeds the TRAINING graph (batch statistics; dropout on when bayesian) ----
    def _build_infer(self, B, H, W, Cin):
        if Cin != self.input_channel:
            raise Exception('infer(): expected %d input channels, got %d' % (self.input_channel, Cin))
        import ctypes as C
        net = E.Net(self.store, B, self.dtype, self.device)
        plan = E.Plan('infer')
        x_in = torch.zeros((B, H, W, Cin), dtype=torch.float32, device=self.device)
        if not hasattr(self, '_infer_off'):
            self._infer_off = C.c_uint64(0)
        plan.keep.append(self._infer_off)
        A, _, _, _ = self._emit_forward(net, plan, x_in, H, W, bn_training=True, update_moving=False, dropout_step=False, want_col=False)
        sig = torch.zeros((B, H, W, self.n_classes), dtype=torch.float32, device=self.device)
        out = torch.zeros((B, H, W, 1), dtype=torch.float32, device=self.device)
        net.sigmoid_argmax(plan, A['logits'], H, W, self.n_classes, sig, out)
        plan.net, plan.acts = net, A
        return plan, x_in, sig, out

    def infer(self, imgs, dropout_offset=None):
        """[sigmoid, argmax] of ONE pass of the training graph; with `bayesian` every call draws fresh dropout masks (counter
        offset = (1 << 62) | call number << 40 -- disjoint from the training steps' -- or `dropout_offset` when given: tests replay a known mask)."""
        if not hasattr(self, '_infer_off'):
            import ctypes as C
            self._infer_off = C.c_uint64(0)
            self._infer_calls = 0
        self._infer_calls = getattr(self, '_infer_calls', 0) + 1
        self._infer_off.value = ((1 << 62) | (self._infer_calls << 40)) if dropout_offset is None else int(dropout_offset)
        return super(DeconvModel, self).infer(imgs)
