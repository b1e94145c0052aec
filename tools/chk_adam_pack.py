#!/usr/bin/env python
"""Diagnostic: seg_adam_pack against seg_adam + seg_pack_weights on a small store; prints where they differ."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from segmentation_amd import _lib as L, engine as E
dt = L.SEG_BF16 if len(sys.argv) < 2 else int(sys.argv[1]); dev = torch.device('cuda', 0)
rng = np.random.default_rng(23)
layers = [E.Layer('f', 'first', 3, [3], 32, 'VALID', True), E.Layer('a', 'conv', 3, [32], 64, 'VALID', True),
          E.Layer('u', 'up', 2, [64], 32, 'VALID', True), E.Layer('c', 'conv', 3, [32, 32], 40, 'VALID', True),
          E.Layer('o', 'conv', 1, [40], 4, 'SAME', False)]
params = {l.name: {'weights': rng.standard_normal(l.wshape).astype(np.float32) * 0.2, 'biases': rng.standard_normal(l.cout).astype(np.float32) * 0.1} for l in layers}
def fresh():
    st = E.ParamStore(layers, dt, dev, training=True); st.set_params(params)
    g = torch.Generator(device='cpu'); g.manual_seed(5)
    st.g.copy_(torch.randn(st.n, generator=g)); st.m.copy_(torch.randn(st.n, generator=g) * 0.1); st.v.copy_(torch.rand(st.n, generator=g) * 0.01)
    st.step.fill_(3)
    return st, E.Net(st, 1, dt, dev)
s0, n0 = fresh(); s1, n1 = fresh()
stream = torch.cuda.current_stream().cuda_stream
a = E.Plan('a'); n0.adam(a, 1e-3, grad_scale=0.5); n0.pack(a); a.run(stream)
b = E.Plan('b'); n1.adam_pack(b, 1e-3, grad_scale=0.5); b.run(stream)
torch.cuda.synchronize()
for name in ('p', 'm', 'v'):
    x, y = getattr(s0, name), getattr(s1, name)
    d = (x != y).nonzero().flatten().cpu().numpy()
    print(name, 'differing', len(d), 'of', x.numel(), 'max abs', float((x - y).abs().max()))
    if len(d):
        for l in layers:
            k = ((d >= l.w_off) & (d < l.w_off + l.wsize)).sum(); kb = ((d >= l.b_off) & (d < l.b_off + l.cout)).sum()
            print('   layer', l.name, 'weights', int(k), 'of', l.wsize, 'biases', int(kb), 'of', l.cout)
x, y = s0.packed.float(), s1.packed.float()
d = (x != y).nonzero().flatten().cpu().numpy()
print('packed differing', len(d), 'of', x.numel())
for l in layers:
    for attr in ('pk_fwd', 'pk_dgrad'):
        o = getattr(l, attr, None)
        if o is not None and len(d):
            print('   ', l.name, attr, 'offset', o, 'first diffs', d[d >= o][:4])
