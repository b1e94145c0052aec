import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from segmentation_amd.datasets import SyntheticDataSet
from segmentation_amd.unet import UNetModel
ds = SyntheticDataSet(16, 256, 4, seed=5555, n_batches=2)
m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=256, learning_rate=1e-4, log_dir=None, save_dir=None,
              load_snapshot=False, dtype='bf16', use_graph=False, seed=5555, wgrad_streams=0)
m._load_batch(m.dataset, m.input_x, m.input_y)
m.store.g.fill_(float('nan'))
m._run_fwd_bwd(); torch.cuda.synchronize()
g = m.store.get_grads()['conv1_1']
print('loss', m.last_loss(), 'finite', np.isfinite(g['weights']).all(), np.isfinite(g['biases']).all(), 'absmax', np.nanmax(np.abs(g['weights'])))
print(g['weights'].reshape(27, -1)[:, :4])
