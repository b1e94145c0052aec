import os, sys, time, torch
sys.path.insert(0, '.')
from segmentation_amd.datasets import SyntheticDataSet
from segmentation_amd.unet import UNetModel
for ws in (2, 3, 4):
    ds = SyntheticDataSet(16, 256, 4, n_batches=2)
    m = UNetModel(sess=None, dataset=ds, n_classes=4, input_dims=256, save_dir=None, log_dir=None, load_snapshot=False, dtype='bf16', wgrad_streams=ws)
    for _ in range(10): m.train_step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): m.train_step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print('wgrad_streams', ws, 'ms/step %.3f img/s %.0f' % (dt * 1e3, 16 / dt))
    del m
