#!/usr/bin/env python
"""bench.py -- train-step images/sec of the U-Net 256x256x3 -> 4-class config (BASELINE.json configs[1]):
B=16 images per GPU, bf16 storage / fp32 accumulate, fwd + mean softmax-x-entropy + bwd + TF-Adam + weight repack,
synthetic data resident in HBM, n_kernels=32, lr 1e-4.

  python bench.py --gpus N --steps K --warmup W
  (N>1 without a torch.distributed.run environment: bench.py starts
   `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`
   itself AS A CHILD PROCESS, before anything touches the GPU, and relays rank 0's JSON line.)

Rank 0 prints ONE JSON line.  Weak scaling: 16 images per rank, gradients SUM-all-reduced over RCCL.
Extra objects: "roofline" (dominant kernel instance, HIP-event timed on the stream it is launched on, algorithmic FLOPs
of the MACs actually executed; "layers" = every arithmetic launch of the step against min(MFMA, HBM)) and
"cpu_baseline" (the oracle's torch-CPU port timed on this box's host cores, N=1 only).

Other workloads (not the headline; same JSON shape):
  --model fcn8s --size 512 --classes 21 --batch 8         BASELINE config 3
  --size 512                                              the per-GPU shard of config 4
  --mode infer                                            forward + sigmoid/argmax (BaseModel.infer's device side)
  --mode mc --batch 32 [--passes 30]                      config 5: MC-dropout inference, a step = 30 stochastic passes
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3         # f32-input MFMA = the f32 vector rate
HBM_PEAK_GBS = 8000.0                # HBM3E spec (6.3 TB/s achievable by a float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--classes', type=int, default=4)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--model', default='unet', choices=['unet', 'fcn8s', 'deconv'], help='fcn8s = BASELINE config 3 (use --size 512 --classes 21 --batch 8); deconv = the reference DeconvModel (SURVEY N3)')
    ap.add_argument('--mode', default='train', choices=['train', 'infer', 'mc'], help='train step (headline) / inference forward / MC-dropout inference (config 5)')
    ap.add_argument('--passes', type=int, default=30, help='--mode mc: stochastic forward passes per step')
    ap.add_argument('--nk', type=int, default=32, help='n_kernels')
    ap.add_argument('--dense', action='store_true', help='evaluate conv1_2 densely (no crop-aware window)')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly (no hipGraph)')
    ap.add_argument('--graph', action='store_true', help='always replay the captured hipGraph (default: time both modes during warm-up, keep the faster)')
    ap.add_argument('--adversarial', action='store_true', help='train step with adversarial_training=True (SURVEY N4; unet / fcn8s, output map >= 84 pixels)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--per-op', action='store_true', help='also print the per-op table to stderr')
    ap.add_argument('--streams', type=int, default=2, help='side streams for the filter gradients (0 = everything on one stream)')
    ap.add_argument('--force-dist', action='store_true', help='diagnostic: take the data-parallel code path (RCCL group of size 1) on one GPU; the one-rank '
                    'collectives themselves are skipped, as in any world-1 run, unless --force-collectives')
    ap.add_argument('--force-collectives', action='store_true', help='with --force-dist: issue the (identity) all-reduces of a one-rank group anyway (SEG_DP_FORCE=1)')
    ap.add_argument('--dp-cuts', default='auto', help="N>1: gradient-bucket boundaries (layer names, backward order); 'auto' times three bucket plans "
                    "(2, 4 and 6 buckets) on the actual node during warm-up and keeps the fastest; 'default' = the model's 4-bucket plan")
    ap.add_argument('--windows', type=int, default=5, help='timed windows of --steps steps each; the FIRST one is `value`, all of them are reported as config.ms_per_step_windows')
    ap.add_argument('--host-data', action='store_true', help='feed from host memory through the pinned-buffer prefetcher (PCIe-inclusive rate; not the headline value)')
    return ap.parse_args()


def spawn_ranks(args):
    """`bench.py --gpus N` from a plain shell: start the launcher as a CHILD process (never exec: this process has not touched
    the GPU and must not be replaced once it has) and relay rank 0's JSON line and the exit code."""
    import socket
    port = os.environ.get('MASTER_PORT')
    if not port:
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0)); port = str(s_.getsockname()[1])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line)
    else:
        sys.stderr.write(r.stdout[-4000:])
    sys.exit(r.returncode if r.returncode else (0 if line is not None else 1))


def kernel_table(plans_run, reps=5):
    """Per-launch durations with HIP events recorded on the stream each kernel is launched on (main or the side
    streams the filter gradients are forked onto -- the same overlap as in the timed region), aggregated by kernel
    template instance (the names rocprofv3 --kernel-trace --stats prints).  plans_run() -> rows of one instrumented pass."""
    agg, ops = {}, []
    for rep in range(reps + 1):
        rows = plans_run()
        if rep == 0:
            continue               # warm-up
        for i, (op, kern, ms, fl, by) in enumerate(rows):
            a = agg.setdefault(kern, {'ms': 0.0, 'launches': 0, 'flops': 0, 'bytes': 0})
            a['ms'] += ms; a['launches'] += 1; a['flops'] += fl; a['bytes'] += by
            if rep == 1:
                ops.append([op, kern, ms, fl, by])
            else:
                ops[i][2] += ms
    for o in ops:
        o[2] /= reps
    return agg, ops, reps


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(args):
    """The oracle's torch-CPU port (oracle/torch_ref.py: same graph, TF-Adam, float32, all host threads of this job's CPU
    share) on a bounded sample of the same workload.  TensorFlow itself cannot be run here."""
    import numpy as np
    import torch
    from oracle import unet as ounet
    from oracle import torch_ref
    # the GPU box gives each job a 16-CPU share (os.cpu_count() reports the whole host): more threads only oversubscribe
    threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    bs = args.batch if args.size <= 256 else max(1, args.batch // 8)       # bounded sample: 512x512 steps are 8x the work per image
    p = ounet.init_params(args.classes, 32, 3, seed=5555)
    rng = np.random.default_rng(5555)
    x = rng.uniform(0, 1, (bs, args.size, args.size, 3)).astype(np.float32)
    y = rng.integers(0, args.classes, (bs, args.size, args.size, 1)).astype(np.uint8)
    if args.mode == 'train':
        st = torch_ref.TorchUNetStepper(p, lr=1e-4, threads=threads)
        fn = lambda: st.train_step(x, y)
        what, per = 'U-Net train steps', 1
    else:
        tp = torch_ref.to_torch_params(p, torch.float32, requires_grad=False)
        xt = torch.from_numpy(x)

        def fn():
            with torch.no_grad():
                torch.sigmoid(torch_ref.unet_forward(tp, xt)).argmax(-1)
        what = 'U-Net inference forwards (+ sigmoid/argmax)' + (
            '; an MC-dropout image = %d such passes (the port has no dropout: same arithmetic minus the masks, prefix not cached)' % args.passes if args.mode == 'mc' else '')
        per = args.passes if args.mode == 'mc' else 1
    fn()                                     # warm-up
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < 15.0 and n < 50):
        fn(); n += 1
    dt = time.time() - t0
    plain_c = None
    if args.mode == 'train' and args.size <= 256:
        # the plain-C restatement (oracle/seg_cpu.c, OpenMP): the same step without a vendor library underneath
        from oracle import c_ops
        cst = c_ops.CUNetStepper(p, lr=1e-4, threads=threads)
        cb = min(bs, 2)
        cst.train_step(x[:cb], y[:cb])
        cn, c0 = 0, time.time()
        while cn < 1 or (time.time() - c0 < 6.0 and cn < 10):
            cst.train_step(x[:cb], y[:cb]); cn += 1
        plain_c = {'value': round(cb * cn / (time.time() - c0), 3), 'unit': 'images/s', 'cores': cst.threads, 'batch': cb,
                   'sample': '%d train steps of batch %d on libseg_cpu.so (gcc -O3 -mavx2 -mfma -fopenmp loop nests)' % (cn, cb)}
    return {'plain_c': plain_c, 'value': round(bs * n / dt / per, 3), 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'cpu_model': cpu_model(),
            'batch': bs,
            'sample': '%d %s of batch %d at %dx%d, %d-class (oracle/torch_ref.py: the same graph%s in float32 on '
                      'torch-CPU/oneDNN, the stand-in for the TF-CPU path, which cannot run here)'
                      % (n, what, bs, args.size, args.size, args.classes, ', loss and TF-Adam' if args.mode == 'train' else '')}


def roofline_report(agg, ops, reps, dtype, total_steps_ms, pmc_tag=None):
    """roofline of the dominant kernel instance + the per-layer table (every launch that does arithmetic)."""
    peak = BF16_DENSE_PEAK_TFLOPS if dtype == 'bf16' else F32_MFMA_PEAK_TFLOPS
    fam = {}
    for k, a in agg.items():
        f = k.split('<')[0]
        fa = fam.setdefault(f, {'ms': 0.0, 'flops': 0, 'launches': 0})
        fa['ms'] += a['ms']; fa['flops'] += a['flops']; fa['launches'] += a['launches']
    # dominant kernel = the template instance with the largest total time among the kernels that do arithmetic (slab
    # reductions / pools / Adam are HBM movers without a FLOP count; they are in "families")
    arith = [kv for kv in agg.items() if kv[1]['flops'] > 0]
    name, a = max(arith, key=lambda kv: kv[1]['ms'])
    avg_ms = a['ms'] / a['launches']
    ach = a['flops'] / a['launches'] / (avg_ms * 1e-3) / 1e12
    traffic, tsrc = None, None
    # HBM bytes per launch come from the committed rocprofv3 --pmc passes OF THIS WORKLOAD (tools/collect_profiles.sh); a workload
    # without a committed counter file reports null rather than another workload's number
    pmc_file = None
    if pmc_tag == '':
        pmc_file = 'pmc_summary.json'
    elif pmc_tag:                                  # newest round that holds this workload's counter passes
        import glob
        c = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s_pmc_summary.json' % pmc_tag)))
        pmc_file = os.path.basename(c[-1]) if c else None
    pmc = os.path.join(ROOT, 'profiles', pmc_file) if pmc_file else None
    if pmc and os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            traffic = j.get(name, {}).get('hbm_bytes_per_launch')
            tsrc = 'profiles/%s@%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not measured in this run)' % (pmc_file, j.get('_commit', 'r01'))
        except Exception:
            traffic = None
    total_ms = sum(v['ms'] for v in agg.values())
    layers = []
    for op, kern, ms_, fl, by in ops:
        if not fl:
            continue
        t_mfma = fl / (peak * 1e12)
        t_hbm = by / (HBM_PEAK_GBS * 1e9) if by else 0.0
        bound = 'mfma' if t_mfma >= t_hbm else 'hbm'
        sec = ms_ * 1e-3
        layers.append({'op': op, 'kernel': kern, 'us': round(ms_ * 1e3, 2), 'gflop': round(fl / 1e9, 3), 'mbytes': round(by / 1e6, 2),
                       'flop_per_byte': round(fl / by, 1) if by else None, 'bound': bound,
                       'achieved': round(fl / sec / 1e12, 1) if bound == 'mfma' else round(by / sec / 1e9, 1),
                       'unit': 'TFLOP/s' if bound == 'mfma' else 'GB/s', 'frac': round(max(t_mfma, t_hbm) / sec, 4)})
    # the dominant kernel's own bound: MFMA unless its algorithmic bytes at the HBM peak take longer than its FLOPs at the MFMA peak
    # (the vector-ALU kernels of the <= 8-channel tensors, the first layer): then achieved / peak are bytes per second
    own = [(fl, by) for _, kern, _, fl, by in ops if kern == name and fl]
    head = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4)}
    if own:
        fl1 = sum(o_[0] for o_ in own) / len(own); by1 = sum(o_[1] for o_ in own) / len(own)
        if by1 / (HBM_PEAK_GBS * 1e9) > fl1 / (peak * 1e12):
            gbs = by1 / (avg_ms * 1e-3) / 1e9
            head = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4)}
    # peaks measured on a box of this pool with reference kernels (tools/calibrate.py -> profiles/r04_calibration.json): a bare MFMA
    # loop on random register operands, hipBLASLt's 8192^3 bf16 GEMM, a float4 copy beyond the Infinity Cache.  `frac` stays
    # against the guide's 2.5 PF / 8 TB/s.
    cal = None
    try:
        cj = json.load(open(os.path.join(ROOT, 'profiles', 'r04_calibration.json')))['peak_calibrated']
        cpk = cj['hbm_gbs'] if head['bound'] == 'hbm' else (cj['mfma_bf16_tflops'] if dtype == 'bf16' else F32_MFMA_PEAK_TFLOPS)
        cal = {'peak': cpk, 'unit': head['unit'], 'frac': round(head['achieved'] / cpk, 4), 'mfma_bf16_tflops': cj['mfma_bf16_tflops'],
               'gemm_bf16_tflops': cj['gemm_bf16_tflops'], 'hbm_gbs': cj['hbm_gbs'], 'source': 'profiles/r04_calibration.json'}
    except Exception:                                  # noqa
        cal = None
    return {
        **head,
        'peak_calibrated': cal,
        'traffic': traffic, 'traffic_source': tsrc, 'kernel': name, 'avg_launch_us': round(avg_ms * 1e3, 2),
        'launches_per_step': a['launches'] // reps, 'share_of_step_kernel_time': round(a['ms'] / total_ms, 3),
        'algorithmic_gflop_per_launch': round(a['flops'] / a['launches'] / 1e9, 3),
        'families': {f: {'ms_per_step': round(v['ms'] / reps, 4), 'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1) if v['flops'] else None}
                     for f, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])},
        'layers': layers,
    }


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        if 'WORLD_SIZE' in os.environ:
            raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
        spawn_ranks(args)                          # does not return

    # stdout carries exactly ONE line, the JSON record: whatever the library prints while it builds the model (the reference's
    # own 'Training from scratch...' notice among it) goes to stderr -- at the file-descriptor level too, because RCCL prints its
    # version banner from C code straight to descriptor 1
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(json_fd, 'w')
    sys.stdout = sys.stderr

    import numpy as np
    import torch
    # SEG_BENCH_BACKEND=gloo: REHEARSAL of the N > 1 control flow (launcher child, bucket-plan probe, exposure report, one JSON
    # line) with the ranks sharing whatever GPUs exist -- the collectives then travel through the host, so its numbers mean nothing
    backend = os.environ.get('SEG_BENCH_BACKEND', 'nccl')
    local = local % max(1, torch.cuda.device_count()) if backend != 'nccl' else local
    torch.cuda.set_device(local)
    if args.force_collectives:
        os.environ['SEG_DP_FORCE'] = '1'
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    from segmentation_amd import _build
    _build.build(verbose=False)
    from segmentation_amd.datasets import SyntheticDataSet
    from segmentation_amd.unet import UNetModel

    training = args.mode == 'train'
    ds = SyntheticDataSet(args.batch, args.size, args.classes, seed=5555 + rank, n_batches=2)
    if args.host_data:
        from segmentation_amd.datasets import ArrayDataSet, DevicePrefetcher
        ds = DevicePrefetcher(ArrayDataSet(ds.images, ds.masks), depth=6, threads=4)
    common = dict(sess=None, n_classes=args.classes, input_dims=args.size, learning_rate=1e-4, log_dir=None, save_dir=None,
                  load_snapshot=False, n_kernels=args.nk, dtype=args.dtype, seed=5555)
    if not training:
        common.update(mode='INFERENCE')
    else:
        common.update(dataset=ds, use_graph=not args.no_graph)
        if args.adversarial:
            common.update(adversarial_training=True)
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def make_model(dp_cuts=None):
        if args.model == 'unet':
            m = UNetModel(crop_aware=not args.dense, wgrad_streams=args.streams, dp_cuts=dp_cuts, **common)
        elif args.model == 'deconv':
            from segmentation_amd.deconvolution import DeconvModel
            m = DeconvModel(**common)
        else:
            from segmentation_amd.fcn import FCNModel
            m = FCNModel(fcn_type='8s', **common)
        if not training:
            m.weights_restored = True             # random-init weights of the named architecture, as the contract says (synthetic)
        if world > 1 and training:
            m.pg.broadcast_(m.store.p)            # identical replicas (same seed anyway)
            m._repack()
        return m

    dp = training and (world > 1 or args.force_dist) and args.model == 'unet'
    cuts_probe = None
    cuts = None if args.dp_cuts in ('auto', 'default') else args.dp_cuts
    if dp and args.dp_cuts == 'auto':
        # the bucket plan is a function of MEASURED overlap: time real data-parallel steps with 2 / 4 / 6 gradient buckets on
        # this node (part of the warm-up; the decision is the max over ranks, so every rank builds the same plan)
        cuts_probe = {}
        for cand in ('conv3_1', 'conv6_1,conv5_2,conv3_1', 'conv6_2,upconv1,conv5_2,conv5_1,conv3_1'):
            m = make_model(cand)
            for _ in range(6):
                m.train_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                m.train_step()
            barrier()
            t = torch.tensor([(time.perf_counter() - t0) / 20 * 1e3], dtype=torch.float64, device='cuda')
            if world > 1:
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            cuts_probe[cand] = round(float(t.item()), 4)
            del m
            import gc
            gc.collect(); torch.cuda.empty_cache()
        cuts = min(cuts_probe, key=cuts_probe.get)
    model = make_model(cuts)

    probe = None
    flops_step = 0
    if training:
        step = model.train_step
        if not args.no_graph and not args.graph:
            probe = model.autotune_step_mode()          # graph replay vs eager launches: real train steps, part of the warm-up
        flops_step = model.fwd_plan.flops + model.bwd_plan.flops      # (bwd_upd_plan = bwd_plan + Adam)
        what = 'train step (fwd+xent+bwd+Adam+repack)' + (' with adversarial training (adversary fwd on real+fake, its gradients, advAdam)' if args.adversarial else '')
    else:
        x_dev = ds.get_device_batch()[0]
        shape = tuple(x_dev.shape)
        stream = lambda: torch.cuda.current_stream().cuda_stream
        if args.mode == 'infer':
            if model._packed_dirty:
                model._repack()
            ent = model._build_infer(*shape)
            ent[1].copy_(x_dev)

            def step():
                ent[0].run(stream())
            flops_step = ent[0].flops
            what = 'inference forward + sigmoid/argmax (device side of BaseModel.infer; inputs and outputs resident in HBM)'
        else:
            if args.model != 'unet':
                raise SystemExit('--mode mc is the U-Net config 5')
            ent = model._mc_entry(shape, 0.5, 5555)
            ent[2].copy_(x_dev)

            def step():
                model._mc_run(ent, args.passes)
            flops_step = ent[0].flops + args.passes * ent[1].flops
            what = ('MC-dropout inference, %d stochastic passes per step (build-defined dropout after conv2_2/conv5_2/conv6_2, keep 0.5; the '
                    'deterministic prefix conv1_1..conv2_2 is computed once per input), mean/variance accumulated on the device' % args.passes)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # Beside `value` (exactly K steps, above): further windows of K steps each, same bracketing, max over ranks; their median
    # says whether the headline window caught a clock / thermal transient (at K = 20 a window is ~20 ms).  Not part of `value`.
    windows = [dt / args.steps * 1e3]
    for _ in range(0 if args.windows <= 1 else args.windows - 1):
        barrier()
        w0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        wd = time.perf_counter() - w0
        if world > 1:
            t = torch.tensor([wd], dtype=torch.float64, device='cuda')
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            wd = float(t.item())
        windows.append(wd / args.steps * 1e3)
    # per-bucket exposure of the gradient all-reduce: the instrumented steps issue real collectives, so EVERY rank runs them
    # (rank 0 alone would wait for peers that have already left); only rank 0 reports
    dp_rep = None
    if world > 1 and training and dp:
        try:
            dp_rep = model.dp_exposure_report()
        except Exception as e:                           # noqa
            dp_rep = {'error': repr(e)}
    elif args.force_dist and training and dp:
        try:
            dp_rep = model.dp_exposure_report()
        except Exception as e:                           # noqa
            dp_rep = {'error': repr(e)}

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        metric = {'train': 'train-step images/sec', 'infer': 'inference images/sec', 'mc': 'MC-dropout inference images/sec (each image = %d stochastic passes)' % args.passes}[args.mode]
        out = {
            'metric': metric, 'value': round(world * args.batch * args.steps / dt, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic' if not args.host_data else 'synthetic, fed from host memory over PCIe (pinned ring + async H2D)',
            'config': {'workload': '%s %dx%dx3 %d-class batch=%d/GPU %s %s, n_kernels=%d'
                                   % ({'unet': 'U-Net', 'fcn8s': 'FCN-8s', 'deconv': 'DeconvModel (BatchNorm)'}[args.model], args.size, args.size, args.classes, args.batch, args.dtype, what, args.nk),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world,
                       'conv1_2': ('dense' if args.dense else 'crop-aware (only the window that survives the last skip crop is computed)') if args.model == 'unet' else None,
                       'hip_graph': bool(model.use_graph) if training else False,
                       'step_mode_probe_ms': None if not probe else {k: round(v, 4) for k, v in probe.items()},
                       'executed_gflop_per_step_per_gpu': round(flops_step / 1e9, 2),
                       'step_tflops_per_gpu': round(flops_step / (ms * 1e-3) / 1e12, 2)},
        }
        if training:
            out['config']['final_loss'] = round(model.last_loss(), 5)
        sw = sorted(windows)
        out['config']['ms_per_step_windows'] = {'n': len(windows), 'steps_each': args.steps, 'median': round(sw[len(sw) // 2], 4),
                                                'min': round(sw[0], 4), 'max': round(sw[-1], 4), 'all': [round(w, 4) for w in windows]}
        if dp_rep is not None:
            # how long the update waits for each gradient bucket after the last backward segment has been enqueued
            out['config']['allreduce'] = dp_rep
            if isinstance(dp_rep, dict):
                out['config']['allreduce']['bucket_plan_probe_ms'] = cuts_probe
    # roofline / per-kernel table: eager, instrumented, after the timed region (rank 0 only does the reporting)
    if not args.no_roofline and world == 1:
        stream = torch.cuda.current_stream().cuda_stream
        if training:
            side = model._side

            def run():
                model.loss_buf.zero_()
                return model.step_plan.run_profiled(stream, torch, side, flavor=model._flavor())
        elif args.mode == 'infer':
            def run():
                return ent[0].run_profiled(stream, torch, None)
        else:
            def run():
                ent[5].value = 1 << 40
                return ent[0].run_profiled(stream, torch, None) + ent[1].run_profiled(stream, torch, None)
        agg, ops, reps = kernel_table(run)
        # which committed counter file describes this command line ('' = the headline)
        tag = None
        if args.mode == 'train' and not args.adversarial and args.dtype == 'bf16' and args.nk == 32 and not args.host_data:
            if args.model == 'unet' and args.classes == 4 and args.batch == 16:
                tag = '' if args.size == 256 else ('c4' if args.size == 512 else None)
            elif args.model == 'fcn8s' and (args.size, args.classes, args.batch) == (512, 21, 8):
                tag = 'c3'
        elif args.mode == 'mc' and args.model == 'unet' and (args.size, args.batch, args.classes) == (256, 32, 4):
            tag = 'c5'
        out['roofline'] = roofline_report(agg, ops, reps, args.dtype, dt / args.steps * 1e3, pmc_tag=tag)
        if training and 'wgrad' in out['roofline']['kernel']:
            # the filter gradients deliberately run on a SHARE of the chip (they overlap the critical stream's kernels): `frac` above is
            # against the whole chip's peak, this is against the CUs a launch may hold (one workgroup per CU, 153 KB of LDS each)
            from segmentation_amd import engine as _E
            tw = _E._step_wgrad_wgs(getattr(model.net, 'input_pixels', None)) or 128
            out['roofline']['launch_workgroup_target'] = tw
            # (the fraction above is of the WHOLE chip's peak although the launch is sized for `tw` of its 256 CUs: the step runs the
            # filter gradients beside the data gradients on purpose -- DESIGN.md section 5)
        if args.per_op:
            for op, kern, ms_, fl, by in ops:
                sys.stderr.write('%-16s %-52s %9.2f us %8.1f TF/s %8.1f GB/s\n' % (op, kern, ms_ * 1e3, fl / (ms_ * 1e-3) / 1e12 if fl else 0, by / (ms_ * 1e-3) / 1e9 if by else 0))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == 'unet':
        out['cpu_baseline'] = cpu_baseline(args)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = None                  # (the torch-CPU stepper covers the U-Net graph only)
    if rank == 0:
        print(json.dumps(out), file=json_out)
        json_out.flush()
    if world > 1 or args.force_dist:
        barrier()                                   # nobody tears the group down while a peer is still inside a collective
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
