#!/usr/bin/env python
"""bench.py -- train-step images/sec of the U-Net 256x256x3 -> 4-class config (BASELINE.json configs[1]):
B=16 images per GPU, bf16 storage / fp32 accumulate, fwd + mean softmax-x-entropy + bwd + TF-Adam + weight repack,
synthetic data resident in HBM, n_kernels=32, lr 1e-4.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Rank 0 prints ONE JSON line.  Weak scaling: 16 images per rank, gradients SUM-all-reduced over RCCL.
Extra objects: "roofline" (dominant kernel family, HIP-event timed on the launch stream, algorithmic FLOPs of the
MACs actually executed) and "cpu_baseline" (the oracle's torch-CPU port timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--classes', type=int, default=4)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--model', default='unet', choices=['unet', 'fcn8s'], help='fcn8s = BASELINE config 3 (use --size 512 --classes 21 --batch 8)')
    ap.add_argument('--nk', type=int, default=32, help='n_kernels')
    ap.add_argument('--dense', action='store_true', help='evaluate conv1_2 densely (no crop-aware window)')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly (no hipGraph)')
    ap.add_argument('--graph', action='store_true', help='always replay the captured hipGraph (default: time both modes during warm-up, keep the faster)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--per-op', action='store_true', help='also print the per-op table to stderr')
    ap.add_argument('--streams', type=int, default=2, help='side streams for the filter gradients (0 = everything on one stream)')
    ap.add_argument('--force-dist', action='store_true', help='diagnostic: take the data-parallel code path (RCCL group of size 1) on one GPU')
    ap.add_argument('--host-data', action='store_true', help='feed from host memory through the pinned-buffer prefetcher (PCIe-inclusive rate; not the headline value)')
    return ap.parse_args()


def kernel_table(model, reps=5):
    """Per-launch durations with HIP events recorded on the stream each kernel is launched on (main or the side
    streams the filter gradients are forked onto -- the same overlap as in the timed region), aggregated by kernel
    template instance (the names rocprofv3 --kernel-trace --stats prints)."""
    stream = torch.cuda.current_stream().cuda_stream
    agg = {}
    ops = []
    for rep in range(reps + 1):
        model.loss_buf.zero_()
        rows = []
        for plan in (model.step_plan,):                         # the plan the timed step runs
            side = model._side if not os.environ.get('SEG_BENCH_SERIAL') else None
            rows += plan.run_profiled(stream, torch, side, flavor=model._flavor())
        if rep == 0:
            continue               # warm-up
        for i, (op, kern, ms, fl) in enumerate(rows):
            a = agg.setdefault(kern, {'ms': 0.0, 'launches': 0, 'flops': 0})
            a['ms'] += ms; a['launches'] += 1; a['flops'] += fl
            if rep == 1:
                ops.append([op, kern, ms, fl])
            else:
                ops[i][2] += ms
    for o in ops:
        o[2] /= reps
    return agg, ops


def cpu_baseline(args):
    """The oracle's torch-CPU port (oracle/torch_ref.py: same graph, TF-Adam, float32, all host threads) on a
    bounded sample of the same workload.  TensorFlow itself cannot be run here."""
    from oracle import unet as ounet
    from oracle import torch_ref
    # the GPU box gives each job a 16-CPU share (os.cpu_count() reports the whole host): more threads only oversubscribe
    threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    bs = 2
    p = ounet.init_params(args.classes, 32, 3, seed=5555)
    st = torch_ref.TorchUNetStepper(p, lr=1e-4, threads=threads)
    rng = np.random.default_rng(5555)
    x = rng.uniform(0, 1, (bs, args.size, args.size, 3)).astype(np.float32)
    y = rng.integers(0, args.classes, (bs, args.size, args.size, 1)).astype(np.uint8)
    st.train_step(x, y)                      # warm-up
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < 12.0 and n < 50):
        st.train_step(x, y); n += 1
    dt = time.time() - t0
    return {'value': round(bs * n / dt, 3), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'sample': '%d train steps of batch %d at %dx%d (oracle/torch_ref.py, float32, oneDNN/torch-CPU stand-in for '
                      'the TF-CPU path, which cannot run here)' % (n, bs, args.size, args.size)}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        # single-process invocation asked for several GPUs: the contract launches us through torch.distributed.run
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    torch.cuda.set_device(local)
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

    from segmentation_amd import _build
    _build.build(verbose=False)
    from segmentation_amd.datasets import SyntheticDataSet
    from segmentation_amd.unet import UNetModel

    ds = SyntheticDataSet(args.batch, args.size, args.classes, seed=5555 + rank, n_batches=2)
    if args.host_data:
        from segmentation_amd.datasets import ArrayDataSet, DevicePrefetcher
        ds = DevicePrefetcher(ArrayDataSet(ds.images, ds.masks), depth=6, threads=4)
    if args.model == 'unet':
        model = UNetModel(sess=None, dataset=ds, n_classes=args.classes, input_dims=args.size, learning_rate=1e-4,
                          log_dir=None, save_dir=None, load_snapshot=False, n_kernels=args.nk,
                          dtype=args.dtype, use_graph=not args.no_graph, crop_aware=not args.dense, seed=5555, wgrad_streams=args.streams)
    else:
        from segmentation_amd.fcn import FCNModel
        model = FCNModel(sess=None, dataset=ds, n_classes=args.classes, input_dims=args.size, learning_rate=1e-4, fcn_type='8s',
                         log_dir=None, save_dir=None, load_snapshot=False, n_kernels=args.nk, dtype=args.dtype,
                         use_graph=not args.no_graph, seed=5555)
    if world > 1:
        model.pg.broadcast_(model.store.p)          # identical replicas (same seed anyway)
        model._repack()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    probe = None
    if not args.no_graph and not args.graph:
        probe = model.autotune_step_mode()          # graph replay vs eager launches: real train steps, part of the warm-up
    for _ in range(args.warmup):
        model.train_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = model.last_loss()

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        flops_step = model.fwd_plan.flops + model.bwd_plan.flops      # (bwd_upd_plan = bwd_plan + Adam)
        out = {
            'metric': 'train-step images/sec', 'value': round(world * args.batch * args.steps / dt, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic' if not args.host_data else 'synthetic, fed from host memory over PCIe (pinned ring + async H2D)',
            'config': {'workload': '%s %dx%dx3 %d-class batch=%d/GPU %s train step (fwd+xent+bwd+Adam+repack), n_kernels=%d'
                                   % ('U-Net' if args.model == 'unet' else 'FCN-8s', args.size, args.size, args.classes, args.batch, args.dtype, args.nk),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world,
                       'conv1_2': 'dense' if args.dense else 'crop-aware (only the window that survives the last skip crop is computed)',
                       'hip_graph': bool(model.use_graph),
                       'step_mode_probe_ms': None if not probe else {k: round(v, 4) for k, v in probe.items()},
                       'executed_gflop_per_step_per_gpu': round(flops_step / 1e9, 2),
                       'step_tflops_per_gpu': round(flops_step / (ms * 1e-3) / 1e12, 2),
                       'final_loss': round(loss, 5)},
        }
    # roofline / per-kernel table: eager, instrumented, after the timed region (rank 0 only does the reporting)
    if not args.no_roofline and world == 1:
        agg, ops = kernel_table(model)
        fam = {}
        for k, a in agg.items():
            f = k.split('<')[0]
            fa = fam.setdefault(f, {'ms': 0.0, 'flops': 0, 'launches': 0})
            fa['ms'] += a['ms']; fa['flops'] += a['flops']; fa['launches'] += a['launches']
        # dominant kernel = the template instance with the largest total time
        # (among the kernels that do arithmetic: the slab reductions / pools / Adam are HBM movers with no FLOP count)
        # Two instances of the tiled convolution are within a few % of each other in total time and swap places from run to
        # run: among the instances within 15 % of the largest total time the one that does the most arithmetic is reported, so
        # that the line names the same kernel every time.
        arith = [kv for kv in agg.items() if kv[1]['flops'] > 0]
        top = max(kv[1]['ms'] for kv in arith)
        dom = max((kv for kv in arith if kv[1]['ms'] >= 0.85 * top), key=lambda kv: kv[1]['flops'])
        name, a = dom
        avg_ms = a['ms'] / a['launches']
        ach = a['flops'] / a['launches'] / (avg_ms * 1e-3) / 1e12 if a['flops'] else 0.0
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(name, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        total_ms = sum(v['ms'] for v in agg.values())
        out['roofline'] = {
            'bound': 'mfma', 'achieved': round(ach, 2), 'peak': BF16_DENSE_PEAK_TFLOPS if args.dtype == 'bf16' else 157.3,
            'unit': 'TFLOP/s', 'frac': round(ach / (BF16_DENSE_PEAK_TFLOPS if args.dtype == 'bf16' else 157.3), 4),
            'traffic': traffic, 'kernel': name, 'avg_launch_us': round(avg_ms * 1e3, 2), 'launches_per_step': a['launches'] // 5,
            'share_of_step_kernel_time': round(a['ms'] / total_ms, 3),
            'families': {f: {'ms_per_step': round(v['ms'] / 5, 4), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1) if v['flops'] else None}
                         for f, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])},
        }
        if args.per_op:
            for op, kern, ms_, fl in ops:
                sys.stderr.write('%-16s %-52s %9.2f us %8.1f TF/s\n' % (op, kern, ms_ * 1e3, fl / (ms_ * 1e-3) / 1e12 if fl else 0))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or args.force_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
