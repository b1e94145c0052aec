"""Which layers' gradients are non-finite after one forward/backward (fault hunting)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from segmentation_amd.datasets import SyntheticDataSet, ArrayDataSet
from segmentation_amd.unet import UNetModel
size, B, dt = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 16, (sys.argv[2] if len(sys.argv) > 2 else 'f32')
rng = np.random.default_rng(5555)
ds = ArrayDataSet(rng.uniform(0, 1, (1, B, size, size, 3)).astype(np.float32), rng.integers(0, 4, (1, B, size, size, 1)).astype(np.uint8))
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=size, log_dir=None, save_dir=None, load_snapshot=False, dtype=dt, use_graph=False, n_kernels=32, seed=5555, learning_rate=1e-3)
p_ = m.store.get_params()
m._load_batch(m.dataset, m.input_x, m.input_y)
m.store.g.fill_(float('nan'))
m._run_fwd_bwd(); torch.cuda.synchronize()
g = m.store.get_grads()
for n, v in g.items():
    for k in ('weights', 'biases'):
        a = np.asarray(v[k]); bad = ~np.isfinite(a)
        if bad.any():
            idx = np.argwhere(bad)
            print(n, k, a.shape, 'bad', int(bad.sum()), 'first', idx[0].tolist(), 'last', idx[-1].tolist(), 'nan' if np.isnan(a[bad]).all() else 'inf/mixed')
print('loss', m.last_loss())
gflat = m.store.g.detach().cpu().numpy()
bad = np.argwhere(~np.isfinite(gflat)).ravel()
print('flat arena: size', gflat.size, 'bad', bad.size, 'first', bad[:5].tolist(), 'last', bad[-5:].tolist())
for n, l in m.store.layers.items():
    lo, hi = l.w_off, l.b_off + l.nbias
    k = int(((bad >= lo) & (bad < hi)).sum())
    if k:
        kb = int(((bad >= l.b_off) & (bad < hi)).sum())
        print(' ', n, 'w_off', l.w_off, 'b_off', l.b_off, 'bad in weights', k - kb, 'bad in biases', kb, 'values', gflat[bad[(bad >= lo) & (bad < hi)][:4]])
