"""FCN-8s with adversarial training (Luc et al. 2016).

The reference ships examples/example_adversarial.py as an empty file and its adversarial branch does not run at
HEAD (F9); this driver follows the shape of /root/reference/examples/example_fcn.py:53-143 with `adversarial_training=True`
(that file's own setting, :89): dataset -> model -> train_step() loop printing the losses the reference logs
(models/basemodel.py:299-301,349-351) -> test() -> snapshot() -> infer().  Trains on the synthetic dataset.

    python examples/example_adversarial.py [--crop 128] [--batch 4] [--classes 2]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from segmentation_amd.fcn import FCNModel                                                 # noqa: E402
from segmentation_amd.datasets import Session, SyntheticDataSet                          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--crop', type=int, default=128)
ap.add_argument('--classes', type=int, default=2)
ap.add_argument('--outer', type=int, default=2)
ap.add_argument('--inner', type=int, default=20)
ap.add_argument('--test-iter', type=int, default=10)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--experiment', default='fcn_adversarial')
args = ap.parse_args()

log_dir = 'examples/{}/logs'.format(args.experiment)
save_dir = 'examples/{}/snapshots'.format(args.experiment)

with Session() as sess:
    dataset = SyntheticDataSet(args.batch, args.crop, args.classes, seed=5555)
    test_dataset = SyntheticDataSet(args.batch, args.crop, args.classes, seed=5556)
    network = FCNModel(sess=sess, dataset=dataset, test_dataset=test_dataset, n_classes=args.classes, input_dims=args.crop,
                       save_dir=save_dir, log_dir=log_dir, load_snapshot=False, learning_rate=1e-4, n_kernels=16, fcn_type='8s',
                       bayesian=False, autoencoder=False, adversarial_training=True, dtype=args.dtype)
    tstart = time.time()
    for _ in range(args.outer):
        for k in range(args.inner):
            network.train_step()
            if (k + 1) % args.test_iter == 0:
                l = network.last_losses()
                print('step {:5d}  seg_xentropy {:.4f}  l_bce_real {:.4f}  l_bce_fake {:.4f}  l_bce_fake_one {:.4f}  seg_loss {:.4f}  adv_loss {:.4f}'.format(
                    network.global_step, l['seg_xentropy'], l['l_bce_real'], l['l_bce_fake'], l['l_bce_fake_one'], l['seg_loss'], l['adv_loss']))
                network.test()
        network.snapshot()
    print('Time: {:.2f}s'.format(time.time() - tstart))
    imgs = np.random.default_rng(0).uniform(0, 1, (args.batch, args.crop, args.crop, 3)).astype(np.float32)
    sig, out = network.infer(imgs)
    print('infer:', sig.shape, out.shape)
    print('Done')
