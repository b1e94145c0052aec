"""Adversarial segmentation training (Luc et al. 2016) behind `adversarial_training=True`:
/root/reference/models/basemodel.py:215-262 (adversary network), :278-305 (losses), :323-355 (objectives, two optimizers).

The reference branch cannot run at HEAD, so this is the intended construction with the resolutions listed in
DESIGN.md section 9 (real = one_hot(labels), fake = softmax(logits); separated variable lists; one batch per step; both
gradients taken at the pre-update weights; lambda = 2; adversarial_lr = 1e-5).

Device layout: the real and the fake batch run through the adversary as ONE batch of 2B images ([0,B) real, [B,2B) fake):
convolutions, pools, resize, flatten and the dense layers are per-image ops and take the 2B batch in one launch, and a
filter gradient over the 2B batch IS the sum of the two passes' gradients (no second arena, no add).  Only the batch norms,
whose statistics are per pass (two calls of the network in the reference), run per half.  The third backward -- fake images
against the label "real", for the segmentation network -- is a data-gradient-only pass over the fake half.
"""
import ctypes as C
import os
import numpy as np
import torch
from . import _lib as L
from . import engine as E

N_KERNELS = 36      # basemodel.py:216
DADV = 4            # basemodel.py:217
SCOPE = 'adversary'


def ladder(h, w):
    """map sizes of the adversary for an h x w input (VALID convs 3x3/s2, VALID pools 2x2/s2); raises when a map collapses"""
    s = {'in': (h, w), 'resize': (h // DADV, w // DADV)}
    cur = s['resize']
    for name, k, st in [('conv1', 3, 2), ('pool1', 2, 2), ('conv2', 3, 2), ('pool2', 2, 2)]:
        cur = tuple((n - k) // st + 1 if n >= k else 0 for n in cur)
        if min(cur) < 1:
            raise Exception('adversarial_training: the %dx%d output map is too small for the adversary (collapses at %s; '
                            'the 4x down-sampled map must reach pool2 with at least one pixel: >= 84 pixels)' % (h, w, name))
        s[name] = cur
    return s


def _sub(act, b0, B):
    """the images [b0, b0+B) of an activation as an Act of their own (a view: same memory)"""
    s = object.__new__(E.Act)
    s.B, s.H, s.W, s.C, s.Cp, s.name = B, act.H, act.W, act.C, act.Cp, act.name + '[%d:%d]' % (b0, b0 + B)
    s.t = act.t[b0:b0 + B]
    s.thin = getattr(act, 'thin', False)       # (a thin half is followed by the other half or by the parent's slack)
    return s


class Adversary(object):
    def __init__(self, B, h, w, n_classes, dtype, device, step_ptr, lr=1e-5, lam=2.0, seed=7777, thin=None):
        self.lib = L.load()
        self.B, self.h, self.w, self.nc, self.dtype, self.device = B, h, w, n_classes, dtype, device
        self.lr, self.lam, self.step_ptr = float(lr), float(lam), step_ptr
        # class-probability maps as THIN tensors (8 channels per pixel): the same kind as the dlogits tensor the model hands over
        # (None: up to 8 classes, unless SEG_THIN_TAIL=0)
        self.thin_maps = (n_classes <= 8 and os.environ.get('SEG_THIN_TAIL', '1') != '0') if thin is None else (bool(thin) and n_classes <= 8)
        self.sz = ladder(h, w)
        nk = N_KERNELS
        ph, pw = self.sz['pool2']
        self.F = F = ph * pw * 2 * nk
        mk = E.Layer
        # the two 3x3/s2 convolutions run on the MFMA kernels as 1x1 convolutions over a strided im2col (engine.Net.sconv_*; the
        # direct VALU kernels took 0.84 of the 2.82 ms adversarial FCN-8s step); SEG_ADV_SCONV=0 keeps the direct kernels
        self.sconv = os.environ.get('SEG_ADV_SCONV', '1') != '0'
        conv = (lambda n, ci, co: E.Net.sconv_layer(n, 3, ci, co, 2)) if self.sconv else (lambda n, ci, co: mk(n, 'direct', 3, [ci], co, 'VALID', True, stride=2))
        self.layers = Ly = {
            'adv_conv1': conv('adv_conv1', n_classes, nk),
            'adv_bn1': mk('adv_bn1', 'bn', 1, [nk], nk),
            'adv_conv2': conv('adv_conv2', nk, 2 * nk),
            'adv_bn2': mk('adv_bn2', 'bn', 1, [2 * nk], 2 * nk),
            'adv_bn3': mk('adv_bn3', 'bn', 1, [F], F),
            'adv_fc1': mk('adv_fc1', 'direct', 1, [F], 1024, 'VALID', True),
            'adv_bn4': mk('adv_bn4', 'bn', 1, [1024], 1024),
            'adv_output': mk('adv_output', 'direct', 1, [1024], 2, 'VALID', False),
        }
        order = ['adv_output', 'adv_bn4', 'adv_fc1', 'adv_bn3', 'adv_bn2', 'adv_conv2', 'adv_bn1', 'adv_conv1']     # backward order
        self.store = E.ParamStore([Ly[n] for n in order], dtype, device, training=True)
        self._init_weights(seed)
        self.net2 = E.Net(self.store, 2 * B, dtype, device)
        self.net1 = E.Net(self.store, B, dtype, device)
        self.losses = torch.zeros(4, dtype=torch.float32, device=device)       # l_bce_real, l_bce_fake, l_bce_fake_one
        self._alloc()

    # ---- parameters ----
    def _init_weights(self, seed):
        """slim defaults: xavier-uniform weights, zero biases, zero betas"""
        rng = np.random.default_rng(seed)
        p = {}
        for n, l in self.layers.items():
            if l.kind == 'bn':
                p[n] = {'beta': np.zeros(l.wshape, np.float32)}
            else:
                sc = getattr(l, 'sconv', None)
                k2, cin = (sc[0] * sc[0], sc[2]) if sc else (l.k * l.k, l.cin)
                lim = np.sqrt(6.0 / (k2 * cin + k2 * l.cout))
                p[n] = {'weights': rng.uniform(-lim, lim, l.wshape).astype(np.float32), 'biases': np.zeros((l.cout,), np.float32)}
        self.store.set_params(p)

    def set_params(self, p):
        """{layer: {weights | beta, biases}} in TF layouts (adv_fc1 / adv_output weights as [F, 1024] / [1024, 2])"""
        q = {}
        for n, t in p.items():
            q[n] = {k: np.asarray(v, np.float32).reshape(self.layers[n].wshape) if k in ('weights', 'beta') else np.asarray(v, np.float32)
                    for k, v in t.items() if k in ('weights', 'biases', 'beta')}
        self.store.set_params(q)

    def get_params(self):
        return self._squeeze(self.store.get_params())

    def get_grads(self):
        return self._squeeze(self.store.get_grads())

    def _squeeze(self, d):
        for n in ('adv_fc1', 'adv_output'):
            d[n]['weights'] = d[n]['weights'].reshape(d[n]['weights'].shape[2:])
        return d

    def get_moving(self):
        out = {}
        for n, st in self.bn.items():
            Cp, c = st['Cp'], self.layers[n].cout
            m = st['moving'].cpu().numpy()
            out[n] = (m[:c].copy(), m[Cp:Cp + c].copy())
        return out

    def set_moving(self, mv):
        for n, (mean, var) in mv.items():
            st = self.bn[n]; Cp, c = st['Cp'], self.layers[n].cout
            m = np.zeros(2 * Cp, np.float32); m[Cp:] = 1.0
            m[:c] = mean; m[Cp:Cp + c] = var
            st['moving'].copy_(torch.from_numpy(m))

    def state_blob(self):
        """arrays for BaseModel.snapshot(): '<scope>/<layer>/<name>'"""
        out = {}
        for n, t in self.get_params().items():
            for k, v in t.items():
                out['%s/%s/%s' % (SCOPE, n, k)] = v
        for n, (m, v) in self.get_moving().items():
            out['%s/%s/moving_mean' % (SCOPE, n)] = m
            out['%s/%s/moving_variance' % (SCOPE, n)] = v
        out[SCOPE + '/adam_m'] = self.store.m.cpu().numpy()
        out[SCOPE + '/adam_v'] = self.store.v.cpu().numpy()
        return out

    def load_state(self, z):
        p, mv = {}, {}
        for n, l in self.layers.items():
            names = ['beta'] if l.kind == 'bn' else ['weights', 'biases']
            if all('%s/%s/%s' % (SCOPE, n, k) in z for k in names):
                p[n] = {k: z['%s/%s/%s' % (SCOPE, n, k)] for k in names}
            if l.kind == 'bn' and '%s/%s/moving_mean' % (SCOPE, n) in z:
                mv[n] = (z['%s/%s/moving_mean' % (SCOPE, n)], z['%s/%s/moving_variance' % (SCOPE, n)])
        if len(p) == len(self.layers):
            self.set_params(p)
        self.set_moving(mv)
        if SCOPE + '/adam_m' in z and z[SCOPE + '/adam_m'].shape[0] == self.store.n:
            self.store.m.copy_(torch.from_numpy(z[SCOPE + '/adam_m']))
            self.store.v.copy_(torch.from_numpy(z[SCOPE + '/adam_v']))

    # ---- buffers ----
    def _alloc(self):
        n2, Ly, sz = self.net2, self.layers, self.sz
        nk, F = N_KERNELS, self.F
        A = self.A = {}
        # (up to 8 classes the class-probability maps are thin tensors: 8 channels per pixel instead of 32)
        thin = self.thin_maps
        A['x'] = n2.act(self.h, self.w, self.nc, name='adv_in', thin=thin)
        A['r'] = n2.act(*sz['resize'], self.nc, name='adv_resize', thin=thin)
        if self.sconv:                  # im2col of each convolution's input (kept for its filter gradient); G[...] = its gradient
            A['c1'] = n2.act(*sz['conv1'], 9 * self.nc, name='adv_col1'); A['c2'] = n2.act(*sz['conv2'], 9 * nk, name='adv_col2')
        A['a1'] = n2.act(*sz['conv1'], nk, name='adv_conv1'); A['y1'] = n2.act(*sz['conv1'], nk, name='adv_bn1')
        A['p1'] = n2.act(*sz['pool1'], nk, name='adv_pool1')
        A['a2'] = n2.act(*sz['conv2'], 2 * nk, name='adv_conv2'); A['y2'] = n2.act(*sz['conv2'], 2 * nk, name='adv_bn2')
        A['p2'] = n2.act(*sz['pool2'], 2 * nk, name='adv_pool2')
        A['f'] = n2.act(1, 1, F, name='adv_flat'); A['y3'] = n2.act(1, 1, F, name='adv_bn3')
        A['h'] = n2.act(1, 1, 1024, name='adv_fc1'); A['y4'] = n2.act(1, 1, 1024, name='adv_bn4')
        A['lg'] = n2.act(1, 1, 2, name='adv_logits')
        G = self.G = {k: n2.act(a.H, a.W, a.C, name='d' + a.name, thin=a.thin) for k, a in A.items() if k != 'x'}
        G['x'] = self.net1.act(self.h, self.w, self.nc, name='dadv_in', thin=thin)          # only the fake half has an input gradient
        B = self.B
        self.half = [{k: _sub(a, b0, B) for k, a in A.items()} for b0 in (0, B)]
        self.ghalf = [{k: _sub(a, b0, B) for k, a in G.items() if k != 'x'} for b0 in (0, B)]
        self.ghalf[1]['x'] = G['x']
        self.bn = {}
        for n in ('adv_bn1', 'adv_bn2'):
            st = n2.bn_state(Ly[n])
            st['Cp'] = Ly[n].cout_p
            st['half'] = [{'moving': st['moving'], 'stats': torch.zeros_like(st['stats']), 'ws': torch.zeros_like(st['ws'])} for _ in range(2)]
            self.bn[n] = st
        for n in ('adv_bn3', 'adv_bn4'):
            Cp = Ly[n].cout_p
            mov = torch.zeros(2 * Cp, dtype=torch.float32, device=self.device); mov[Cp:] = 1.0
            self.bn[n] = {'moving': mov, 'Cp': Cp, 'half': [{'stats': torch.zeros(2 * Cp, dtype=torch.float32, device=self.device)} for _ in range(2)]}
        self.scratch = torch.zeros(max(l.cout_p for l in Ly.values()), dtype=torch.float32, device=self.device)

    # ---- plan emission ----
    def _bn_rows_fwd(self, plan, name, a, y, half):
        l, st = self.layers[name], self.bn[name]
        av, yv = a.view(), y.view()
        plan.keep += [av, yv, st]
        plan.add(name, self.lib.seg_bn_rows_fwd, C.byref(av), C.byref(yv), self.store.p_ptr(l.w_off), st['moving'].data_ptr(),
                 st['half'][half]['stats'].data_ptr(), self.B, l.cout, 0.999, 1e-3, self.dtype, kernel='bn_rows_fwd_kernel')

    def _bn_rows_bwd(self, plan, name, a, dy, dz, half, relu_mask, to_scratch, add):
        l, st = self.layers[name], self.bn[name]
        av, gv, zv = a.view(), dy.view(), dz.view()
        plan.keep += [av, gv, zv, st]
        dst = self.scratch.data_ptr() if to_scratch else self.store.g_ptr(l.w_off)
        plan.add(name + '/bwd', self.lib.seg_bn_rows_bwd, C.byref(av), C.byref(gv), C.byref(zv), st['half'][half]['stats'].data_ptr(), dst,
                 1 if add else 0, self.B, l.cout, 1 if relu_mask else 0, self.dtype, kernel='bn_rows_bwd_kernel')

    def _flatten(self, plan, net, a, f, backward):
        av, fv = a.view(), f.view()
        plan.keep += [av, fv]
        plan.add('adv_flat' + ('/bwd' if backward else ''), self.lib.seg_flatten, C.byref(av), net.B, a.H, a.W, a.C, C.byref(fv), 1 if backward else 0,
                 self.dtype, kernel='flatten_kernel')

    def _bce(self, plan, lg, label, slot, dlg):
        lv, dv = lg.view(), dlg.view()
        plan.keep += [lv, dv]
        plan.add('adv_bce[%d]' % slot, self.lib.seg_bce2, C.byref(lv), self.B, label, 1.0, self.losses.data_ptr() + 4 * slot, C.byref(dv), self.dtype,
                 kernel='bce2_kernel')

    def emit(self, plan, logits, labels_u8, LH, LW, loff, dlogits, probs_given=False):
        """Appends the adversary's part of a train step to `plan`, behind the x-entropy launch that filled `dlogits`:
        real / fake maps, the forward pass of both, the adversary's own gradients (into its arena), and
        dlogits += lambda * d l_bce_fake_one / d logits."""
        n2, n1, Ly, A, G, H2, GH = self.net2, self.net1, self.layers, self.A, self.G, self.half, self.ghalf
        B, h, w = self.B, self.h, self.w
        # inputs: [0,B) one_hot(labels), [B,2B) softmax(logits)
        rv, fv, lv = H2[0]['x'].view(), H2[1]['x'].view(), logits.view()
        plan.keep += [rv, fv, lv]
        plan.add('adv_onehot', self.lib.seg_onehot, labels_u8.data_ptr(), LH, LW, loff[0], loff[1], B, h, w, C.byref(rv), self.dtype, kernel='onehot_kernel')
        if not probs_given:              # (else the x-entropy launch has written softmax(logits) into the fake half already)
            plan.add('adv_softmax', self.lib.seg_softmax_probs, C.byref(lv), B, h, w, self.nc, C.byref(fv), self.dtype, kernel='softmax_probs_kernel')
        # forward, 2B images at once; batch norms per half (real first: the order the moving averages see, basemodel.py:283-285)
        n2.resize_fwd(plan, A['x'], A['r'])
        if self.sconv:
            n2.pack(plan)                # the MFMA-operand copies of the two filters (a few KB; whoever changed them last, they are current)
            n2.sconv_fwd(plan, Ly['adv_conv1'], A['r'], A['a1'], A['c1'])
        else:
            n2.dlayer_fwd(plan, Ly['adv_conv1'], A['r'], A['a1'])
        for hf in (0, 1):
            n1.bn_fwd(plan, Ly['adv_bn1'], self.bn['adv_bn1']['half'][hf], H2[hf]['a1'], H2[hf]['y1'])
        n2.pool_k_fwd(plan, A['y1'], A['p1'], 2)
        if self.sconv:
            n2.sconv_fwd(plan, Ly['adv_conv2'], A['p1'], A['a2'], A['c2'])
        else:
            n2.dlayer_fwd(plan, Ly['adv_conv2'], A['p1'], A['a2'])
        for hf in (0, 1):
            n1.bn_fwd(plan, Ly['adv_bn2'], self.bn['adv_bn2']['half'][hf], H2[hf]['a2'], H2[hf]['y2'])
        n2.pool_k_fwd(plan, A['y2'], A['p2'], 2)
        self._flatten(plan, n2, A['p2'], A['f'], False)
        for hf in (0, 1):
            self._bn_rows_fwd(plan, 'adv_bn3', H2[hf]['f'], H2[hf]['y3'], hf)
        n2.dlayer_fwd(plan, Ly['adv_fc1'], A['y3'], A['h'])
        for hf in (0, 1):
            self._bn_rows_fwd(plan, 'adv_bn4', H2[hf]['h'], H2[hf]['y4'], hf)
        n2.dlayer_fwd(plan, Ly['adv_output'], A['y4'], A['lg'])
        # adversary objective: mean_b bce(real, 1) + mean_b bce(fake, 0)   (basemodel.py:291-294,337)
        self._bce(plan, H2[0]['lg'], 1, 0, GH[0]['lg'])
        self._bce(plan, H2[1]['lg'], 0, 1, GH[1]['lg'])
        self._backward(plan, n2, A, G, weights=True, halves=(0, 1))
        # segmentation objective's adversarial term: lambda * mean_b bce(fake, 1)   (basemodel.py:296,334)
        self._bce(plan, H2[1]['lg'], 1, 2, GH[1]['lg'])
        self._backward(plan, n1, H2[1], GH[1], weights=False, halves=(1,))
        n1.resize_bwd(plan, GH[1]['r'], GH[1]['x'])
        gv, dv = GH[1]['x'].view(), dlogits.view()
        plan.keep += [gv, dv]
        plan.add('adv_softmax/bwd', self.lib.seg_softmax_bwd_add, C.byref(lv), C.byref(gv), B, h, w, self.nc, self.lam, C.byref(dv), self.dtype,
                 kernel='softmax_bwd_add_kernel')

    def _backward(self, plan, net, A, G, weights, halves):
        """weights=True: every filter / bias / beta gradient (beta: the halves add up) and no input gradient below conv1;
        weights=False: data gradients only, down to the resized input."""
        Ly, H2, GH, n1 = self.layers, self.half, self.ghalf, self.net1
        sc = None if weights else self.scratch.data_ptr()
        net.dlayer_bwd(plan, Ly['adv_output'], A['y4'], G['lg'], dsrc=G['y4'], wgrad=weights)
        for i, hf in enumerate(halves):
            self._bn_rows_bwd(plan, 'adv_bn4', H2[hf]['h'], GH[hf]['y4'], GH[hf]['h'], hf, True, not weights, i > 0)
        net.dlayer_bwd(plan, Ly['adv_fc1'], A['y3'], G['h'], dsrc=G['y3'], wgrad=weights)
        for i, hf in enumerate(halves):
            self._bn_rows_bwd(plan, 'adv_bn3', H2[hf]['f'], GH[hf]['y3'], GH[hf]['f'], hf, False, not weights, i > 0)
        self._flatten(plan, net, G['p2'], G['f'], True)
        net.pool_k_bwd(plan, A['y2'], G['p2'], G['y2'], 2)
        for i, hf in enumerate(halves):
            n1.bn_relu_bwd(plan, Ly['adv_bn2'], self.bn['adv_bn2']['half'][hf], H2[hf]['a2'], GH[hf]['y2'], GH[hf]['a2'], dbeta_ptr=sc, dbeta_add=i > 0)
        if self.sconv:
            net.sconv_bwd(plan, Ly['adv_conv2'], A['c2'], G['a2'], dcol=G['c2'], dsrc=G['p1'], wgrad=weights)
        else:
            net.dlayer_bwd(plan, Ly['adv_conv2'], A['p1'], G['a2'], dsrc=G['p1'], wgrad=weights)
        net.pool_k_bwd(plan, A['y1'], G['p1'], G['y1'], 2)
        for i, hf in enumerate(halves):
            n1.bn_relu_bwd(plan, Ly['adv_bn1'], self.bn['adv_bn1']['half'][hf], H2[hf]['a1'], GH[hf]['y1'], GH[hf]['a1'], dbeta_ptr=sc, dbeta_add=i > 0)
        if self.sconv:
            net.sconv_bwd(plan, Ly['adv_conv1'], A['c1'], G['a1'], dcol=G['c1'], dsrc=None if weights else G['r'], wgrad=weights)
        else:
            net.dlayer_bwd(plan, Ly['adv_conv1'], A['r'], G['a1'], dsrc=None if weights else G['r'], wgrad=weights)

    def emit_update(self, plan):
        """advAdam: the adversary's own Adam at adversarial_lr (basemodel.py:325-328,343), same step count as the model's"""
        s = self.store
        plan.add('advAdam', self.lib.seg_adam, s.p.data_ptr(), s.g.data_ptr(), s.m.data_ptr(), s.v.data_ptr(), s.n, self.lr, 0.9, 0.999, 1e-8, 1.0,
                 self.step_ptr, kernel='adam_kernel')
