"""U-Net training / inference driver.

The reference ships examples/example_unet.py as an empty file; this one follows the shape of
/root/reference/examples/example_fcn.py:53-143 and Readme.md:44-66: dataset -> model -> train_step() loop with
test() every `test_iter` steps -> snapshot() per outer loop -> infer().  With no data directories it trains on the
synthetic dataset (BASELINE config C1: 2-class, batch 2, 188x188 -- the all-VALID graph is infeasible at 128x128).

    python examples/example_unet.py                      # synthetic C1 plumbing run
    python examples/example_unet.py --feat-dir data/feature --mask-dir data/label --crop 256 --batch 16
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from segmentation_amd.unet import UNetModel                                               # noqa: E402
from segmentation_amd.datasets import (Session, Coordinator, start_queue_runners,        # noqa: E402
                                       SyntheticDataSet, ThreadedImageMaskDataSet)

ap = argparse.ArgumentParser()
ap.add_argument('--feat-dir', default=None)
ap.add_argument('--mask-dir', default=None)
ap.add_argument('--image-ext', default='jpg')
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--crop', type=int, default=188)
ap.add_argument('--classes', type=int, default=2)
ap.add_argument('--outer', type=int, default=2)
ap.add_argument('--inner', type=int, default=20)
ap.add_argument('--test-iter', type=int, default=10)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--experiment', default='unet')
args = ap.parse_args()

log_dir = 'examples/{}/logs'.format(args.experiment)
save_dir = 'examples/{}/snapshots'.format(args.experiment)

with Session() as sess:
    if args.feat_dir:
        dataset = ThreadedImageMaskDataSet(args.feat_dir, args.mask_dir, image_ext=args.image_ext, n_classes=args.classes,
                                           batch_size=args.batch, crop_size=args.crop, capacity=8, threads=4)
        test_dataset = dataset
    else:
        dataset = SyntheticDataSet(args.batch, args.crop, args.classes, seed=5555)
        test_dataset = SyntheticDataSet(args.batch, args.crop, args.classes, seed=5556)

    network = UNetModel(sess=sess, dataset=dataset, test_dataset=test_dataset, n_classes=args.classes, input_dims=args.crop,
                        save_dir=save_dir, log_dir=log_dir, load_snapshot=False, learning_rate=1e-4, n_kernels=32,
                        bayesian=False, dtype=args.dtype)

    coord = Coordinator()
    threads = start_queue_runners(coord=coord, datasets=[d for d in (dataset,) if hasattr(d, 'start')])

    tstart = time.time()
    for _ in range(args.outer):
        t_outer_loop = time.time()
        for k in range(args.inner):
            network.train_step()
            if k % args.test_iter == 0:
                network.test()
        print('Time: {}  loss: {:.5f}'.format(time.time() - t_outer_loop, network.last_loss()))
        network.snapshot()

    img_tensor, _ = test_dataset.get_batch()
    output = network.infer(np.asarray(img_tensor, np.float32))
    print('infer:', output[0].shape, output[1].shape, 'class histogram', np.bincount(output[1].astype(np.int64).ravel()))
    print('Time: {}'.format(time.time() - tstart))
    print('Done')
    coord.request_stop()
    coord.join(threads)
