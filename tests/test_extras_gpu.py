"""-m gpu tests: 1-GPU RCCL data-parallel path (segmented backward + bucket all-reduce), MC-dropout inference,
dropout kernel statistics, snapshot/restore."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from segmentation_amd import _lib as L
from segmentation_amd import engine as E
from segmentation_amd.datasets import ArrayDataSet, SyntheticDataSet
from segmentation_amd.unet import UNetModel

pytestmark = pytest.mark.gpu


def _data(B, S, nc, seed=5555, n=1):
    rng = np.random.default_rng(seed)
    return (rng.uniform(0, 1, (n, B, S, S, 3)).astype(np.float32), rng.integers(0, nc, (n, B, S, S, 1)).astype(np.uint8))


def test_dp_world1_rccl_path_equals_plain_step(monkeypatch):
    # (the single-GPU plans ask for 64-workgroup filter gradients, the data-parallel ones keep 128: another K split is another
    # f32 summation order, so the bitwise comparison pins the same target for both)
    monkeypatch.setenv('SEG_WGRAD_WGS', '128')
    x, y = _data(2, 188, 2, n=2)
    kw = dict(sess=None, n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None, load_snapshot=False, dtype='f32')
    plain = UNetModel(dataset=ArrayDataSet(x, y), use_graph=True, **kw)
    for _ in range(4):
        plain.train_step()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        dp = UNetModel(dataset=ArrayDataSet(x, y), use_graph=True, **kw)
        assert dp.pg.enabled and dp.pg.world == 1 and len(dp.bwd_segments) == 4 and dp.dp_cuts_used == ['conv6_1', 'conv5_2', 'conv3_1']
        bounds = [b for _, b in dp.bwd_segments]
        assert bounds[0][0] == 0 and all(bounds[i][1] == bounds[i + 1][0] for i in range(3)) and bounds[3][1] == dp.store.n
        assert all(8e6 < (hi - lo) * 4 < 12e6 for lo, hi in bounds[:3])   # three medium messages in the middle of backward
        assert (bounds[3][1] - bounds[3][0]) * 4 < 300e3            # last (exposed) bucket: conv2_x + conv1_x only
        for _ in range(4):
            dp.train_step()
        torch.cuda.synchronize()
        # filter gradients are reduced in a fixed order -> the two paths agree bit for bit
        assert torch.equal(plain.store.p, dp.store.p)
        assert plain.last_loss() == dp.last_loss()
    finally:
        torch.distributed.destroy_process_group()


def test_dropout_kernel_statistics_and_determinism():
    lib = L.load()
    net = E.Net(None, 2, L.SEG_F32, torch.device('cuda', 0))
    a = net.act(64, 64, 32); a.t.fill_(1.0)
    b = net.act(64, 64, 32); c = net.act(64, 64, 32)
    av, bv, cv = a.view(), b.view(), c.view()
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.seg_dropout(C.byref(av), C.byref(bv), 2, 64, 64, 32, 0.5, 7, 0, L.SEG_F32, s))
    L.check(lib.seg_dropout(C.byref(av), C.byref(cv), 2, 64, 64, 32, 0.5, 7, 0, L.SEG_F32, s))
    torch.cuda.synchronize()
    assert torch.equal(b.t, c.t)                                  # same (seed, offset) -> same mask
    vals = torch.unique(b.t).cpu().numpy().tolist()
    assert vals == [0.0, 2.0]
    assert abs(float((b.t > 0).float().mean()) - 0.5) < 0.01
    L.check(lib.seg_dropout(C.byref(av), C.byref(cv), 2, 64, 64, 32, 0.5, 7, 1 << 40, L.SEG_F32, s))
    torch.cuda.synchronize()
    assert not torch.equal(b.t, c.t)


def test_mc_dropout_inference():
    x, y = _data(2, 188, 4, seed=3)
    m = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=4, input_dims=188, log_dir=None, save_dir=None,
                  load_snapshot=False, dtype='bf16', bayesian=True)
    mean, var, amax = m.infer_mc(x[0], passes=8)
    assert mean.shape == (2, 4, 4, 4) and var.shape == mean.shape and amax.shape == (2, 4, 4, 1)
    assert np.isfinite(mean).all() and (var >= 0).all() and var.max() > 0       # stochastic passes differ
    sig, out = m.infer(x[0])                                                     # `bayesian` is ignored by infer(), as in the reference
    sig2, out2 = m.infer(x[0])
    assert np.array_equal(sig, sig2)


def test_snapshot_restore_roundtrip(tmp_path):
    x, y = _data(1, 188, 2, seed=4)
    kw = dict(sess=None, n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=str(tmp_path / 'logs'), dtype='f32', use_graph=False)
    m = UNetModel(dataset=ArrayDataSet(x, y), save_dir=str(tmp_path / 'snap'), load_snapshot=False, **kw)
    m.train_step(); m.train_step(); m.snapshot(); m.test()
    assert m.last_test_loss is not None
    m2 = UNetModel(dataset=ArrayDataSet(x, y), save_dir=str(tmp_path / 'snap'), load_snapshot=True, **kw)
    assert m2.global_step == 2 and int(m2.store.step[0].item()) == 2
    assert torch.equal(m.store.p, m2.store.p) and torch.equal(m.store.m, m2.store.m)
    m.train_step(); m2.train_step()
    torch.cuda.synchronize()
    assert torch.equal(m.store.p, m2.store.p)                    # resume continues the same trajectory
    names = set(np.load(m2._latest_checkpoint()).files)
    assert {'conv1_1/weights', 'conv1_1/biases', 'output/weights', 'global_step'} <= names


def test_device_prefetcher_feeds_identical_batches():
    """host dataset -> pinned ring -> async H2D on a copy stream -> train_step: same trajectory as the direct feed"""
    from segmentation_amd.datasets import DevicePrefetcher
    x, y = _data(2, 188, 2, seed=8, n=3)
    kw = dict(sess=None, n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None, load_snapshot=False, dtype='f32')
    a = UNetModel(dataset=ArrayDataSet(x, y), **kw)
    pf = DevicePrefetcher(ArrayDataSet(x, y), depth=2)
    b = UNetModel(dataset=pf, **kw)
    try:
        for _ in range(7):
            a.train_step(); b.train_step()
        torch.cuda.synchronize()
        assert torch.equal(a.store.p, b.store.p)
    finally:
        pf.stop()


class _HostOnly(object):
    """a device-resident dataset seen through the host interface only (get_batch -> numpy): the model copies every batch into its own
    input buffers instead of reading the dataset's buffers in place"""

    def __init__(self, ds):
        self.ds = ds
        self.batch_size, self.has_masks, self.use_feed = ds.batch_size, True, False

    def set_tf_sess(self, sess):
        pass

    def get_batch(self):
        x, y = self.ds.get_device_batch()
        return x.cpu().numpy(), y.cpu().numpy()


def test_zero_copy_device_batches_equal_copied_ones():
    """Device-resident batches consumed in place (launch arguments re-pointed, one graph per buffer) vs copied into the
    model's own input buffers: identical parameters after a few steps, in graph and in eager mode."""
    import numpy as np
    from segmentation_amd.datasets import SyntheticDataSet
    from segmentation_amd.unet import UNetModel
    out = {}
    for zc in ('1', '0'):
        for graph in (True, False):
            ds = SyntheticDataSet(2, 188, 3, seed=11, n_batches=3)
            if zc == '0':
                ds = _HostOnly(ds)
            m = UNetModel(sess=None, dataset=ds, n_classes=3, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None,
                          load_snapshot=False, dtype='bf16', use_graph=graph, seed=3)
            for _ in range(7):
                m.train_step()
            torch.cuda.synchronize()
            out[(zc, graph)] = (m.store.p.clone(), m.last_loss())
            if zc == '1':
                assert len(m._slots) == 3          # three dataset buffers were bound in place
            else:
                assert not getattr(m, '_slots', None)
    ref = out[('0', False)]
    for k, v in out.items():
        assert torch.equal(v[0], ref[0]) and v[1] == ref[1], k


class _Recorder(object):
    """dataset proxy that keeps a copy of every batch it hands out (in hand-out order)"""

    def __init__(self, ds):
        self.ds, self.log = ds, []
        self.batch_size, self.has_masks, self.use_feed = ds.batch_size, True, False

    def set_tf_sess(self, sess):
        pass

    def start(self):
        self.ds.start()

    def stop(self):
        self.ds.stop()

    def get_batch(self):
        img, msk = self.ds.get_batch()
        self.log.append((np.array(img), np.array(msk)))
        return img, msk


class _PinnedRecorder(_Recorder):
    """... and passes the loader's pinned ring slots through (DevicePrefetcher then copies H2D straight out of them)"""

    def get_pinned_batch(self):
        img, msk = self.ds.get_pinned_batch()
        self.log.append((img.clone().numpy(), msk.clone().numpy()))
        return img, msk


def _png_folder(tmp_path, n=6, hw=(200, 220)):
    from PIL import Image
    fd, md = tmp_path / 'feature', tmp_path / 'label'
    fd.mkdir(); md.mkdir()
    rng = np.random.default_rng(12)
    for i in range(n):
        im = rng.integers(0, 256, hw + (3,)).astype(np.uint8)
        mk = np.zeros(hw, np.uint8); mk[20 + 5 * i:150, 30:120 + 10 * i] = 255; mk[0, :5] = 128
        Image.fromarray(im).save(fd / ('%02d.png' % i)); Image.fromarray(mk).save(md / ('%02d.png' % i))
    return str(fd), str(md)


def test_image_folder_loader_prefetcher_train_step(tmp_path):
    """N1 end to end: PNG folder -> ThreadedImageMaskDataSet (decode threads, shuffle pool, pinned ring) -> DevicePrefetcher
    (H2D on a copy stream straight out of the ring) -> train_step (batches consumed in place).  The trajectory must equal
    feeding the very same decoded batches directly (utils/datasets.py:136-190 semantics are checked on the CPU side)."""
    from segmentation_amd.datasets import DevicePrefetcher, ThreadedImageMaskDataSet
    fd, md = _png_folder(tmp_path)
    kw = dict(sess=None, n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None, load_snapshot=False, dtype='bf16', seed=9)
    for wrap in (True, False):               # through the prefetcher, and through BaseModel's own host path
        loader = ThreadedImageMaskDataSet(fd, md, batch_size=2, crop_size=188, image_ext='png', threads=3, capacity=8, min_holding=2)
        rec = _PinnedRecorder(loader) if wrap else _Recorder(loader)     # (BaseModel's own host path uses get_batch())
        ds = DevicePrefetcher(rec, depth=3) if wrap else rec
        try:
            b = UNetModel(dataset=ds, use_graph=True, **kw)
            steps = 6
            for _ in range(steps):
                b.train_step()
            torch.cuda.synchronize()
        finally:
            ds.stop()
            loader.stop()
        assert len(rec.log) >= steps
        for img, msk in rec.log[:steps]:
            assert img.dtype == np.float32 and 0 <= img.min() and img.max() <= 1 and set(np.unique(msk)) <= {0, 1} and msk.any()
        x = np.stack([r[0] for r in rec.log[:steps]]); y = np.stack([r[1] for r in rec.log[:steps]])
        a = UNetModel(dataset=ArrayDataSet(x, y), use_graph=True, **kw)
        for _ in range(steps):
            a.train_step()
        torch.cuda.synchronize()
        assert torch.equal(a.store.p, b.store.p), wrap
        assert a.last_loss() == b.last_loss()


@pytest.mark.parametrize('folder', [False, True])
def test_example_unet_script_runs(tmp_path, folder):
    """BASELINE config C1 (the plumbing config, 188x188 because the all-VALID graph is infeasible at 128): the authored
    examples/example_unet.py end to end -- train loop, test(), snapshot(), infer() -- on the synthetic set and on a folder."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'examples', 'example_unet.py'), '--outer', '1', '--inner', '4', '--test-iter', '2', '--dtype', 'bf16']
    if folder:
        fd, md = _png_folder(tmp_path)
        cmd += ['--feat-dir', fd, '--mask-dir', md, '--image-ext', 'png']
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'Done' in r.stdout and 'TEST LOSS' in r.stdout and 'infer: (2, 4, 4, 2) (2, 4, 4, 1)' in r.stdout
    snaps = os.listdir(tmp_path / 'examples' / 'unet' / 'snapshots')
    assert any(f.endswith('-4.npz') for f in snaps), snaps


def test_example_adversarial_script_runs(tmp_path):
    """the authored examples/example_adversarial.py (the reference's is an empty file, F1): FCN-8s with adversarial_training=True"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'examples', 'example_adversarial.py'), '--outer', '1', '--inner', '4', '--test-iter', '2']
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'Done' in r.stdout and 'l_bce_fake_one' in r.stdout and 'infer: (4, 128, 128, 2) (4, 128, 128, 1)' in r.stdout
    assert any(f.endswith('-4.npz') for f in os.listdir(tmp_path / 'examples' / 'fcn_adversarial' / 'snapshots'))


@pytest.mark.gpu
def test_loaded_library_was_linted_for_the_wgrad_register_budget():
    """The register-staged filter-gradient kernels (f32, 1x1, 2x2/s2, first layer) wait by hand for asm loads: sound only while no
    instance is over its register budget (segmentation_amd/_wgrad_regs.py).  _build runs that lint with the compiler that builds the
    library and leaves the verdict next to the .so; here, on the GPU box, the library that is actually loaded must carry a clean
    verdict for exactly the sources in the tree (the guard travels with the build)."""
    import json
    from segmentation_amd import _build, _lib
    assert os.path.exists(_lib.LIB_PATH)
    assert os.path.exists(_build.LINT), 'no register-budget verdict next to the library'
    rec = json.load(open(_build.LINT))
    assert rec['sources'] == _build.source_digest(), 'the verdict belongs to other sources than the ones in the tree'
    assert rec['instances'] >= 30 and not rec['over_budget'], rec
    assert rec['max_arch_vgprs'] < rec['limit']


@pytest.mark.gpu
def test_bench_force_dist_child_process_reports_the_allreduce():
    """`bench.py --gpus 1 --force-dist` in a child process: the data-parallel code path end to end on one GPU (RCCL group of one --
    the launcher / multi-rank part needs an 8-GPU node and is the driver's to run): one JSON line, the bucket plan, every
    bucket's exposed time, a finite loss -- and the line must not hang on a collective only rank 0 joins."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--force-dist', '--force-collectives', '--steps', '5', '--warmup', '3', '--windows', '2',
                        '--no-cpu-baseline', '--no-roofline', '--dp-cuts', 'default'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-500:]
    j = json.loads(lines[0])
    ar = j['config']['allreduce']
    assert ar['world'] == 1 and len(ar['buckets_mb']) == len(ar['cuts']) + 1 == len(ar['exposed_us'])
    assert abs(sum(ar['buckets_mb']) - 31.04) < 0.01                       # the 7 760 196-parameter fp32 arena
    assert all(u >= 0 for u in ar['exposed_us']) and ar['exposed_total_us'] < 2000
    assert j['n_gpus'] == 1 and j['value'] > 0 and np.isfinite(j['config']['final_loss'])
    assert j['config']['ms_per_step_windows']['n'] == 2


def test_bench_two_ranks_rehearsal_over_gloo():
    """`bench.py --gpus 2` as the driver starts it from a plain shell -- launcher child, two ranks, bucket-plan probe, exposure
    report on every rank, ONE JSON line from rank 0 -- rehearsed on this one GPU with SEG_BENCH_BACKEND=gloo (both ranks share
    the card, the collectives travel through the host: the numbers mean nothing, the control flow is what an 8-GPU node runs)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEG_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('MASTER_PORT', None); env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--windows', '2',
                        '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-500:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch'] == 32 and j['config']['parallelism'] == 'dp2' and j['scaling'] == 'weak'
    ar = j['config']['allreduce']
    assert ar['world'] == 2 and len(ar['exposed_us']) == len(ar['buckets_mb']) and len(ar['bucket_plan_probe_ms']) == 3
    assert abs(sum(ar['buckets_mb']) - 31.04) < 0.01 and j['value'] > 0 and np.isfinite(j['config']['final_loss'])


def test_bench_four_ranks_one_image_each_rehearsal_over_gloo():
    """C4's launch shape at the rank count a one-GPU box admits (process guard: 6 on the card): `bench.py --gpus 4 --batch 1 --size 188`
    over gloo with rs_ag -- one JSON line, global batch 4, dp4.  (8 ranks x 1 image == batch 8: tests/test_dist_cpu.py on the CPU.)"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SEG_BENCH_BACKEND='gloo', SEG_DP_ALGO='rs_ag', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('MASTER_PORT', None); env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--batch', '1', '--size', '188', '--steps', '3', '--warmup', '1',
                        '--windows', '1', '--no-cpu-baseline', '--no-roofline', '--dp-cuts', 'default'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-500:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 4 and j['config']['global_batch'] == 4 and j['config']['parallelism'] == 'dp4' and j['scaling'] == 'weak'
    assert j['config']['allreduce']['world'] == 4 and j['value'] > 0 and np.isfinite(j['config']['final_loss'])


@pytest.mark.gpu
def test_compiled_plan_replay_equals_the_interpreter_walk(monkeypatch):
    """seg_plan_run (one host call per train step: the recorded walk of Plan.run -- launches, event forks, signal forks, held side
    launches) enqueues the same launches in the same order as the per-launch Python walk: three bf16 U-Net train steps end in
    bitwise equal parameters and Adam moments, with and without the input re-pointing of a device-resident dataset."""
    from segmentation_amd import engine as E
    from segmentation_amd.datasets import ArrayDataSet
    from segmentation_amd.unet import UNetModel
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 1, (3, 2, 188, 188, 3)).astype(np.float32)
    y = rng.integers(0, 2, (3, 2, 188, 188, 1)).astype(np.uint8)
    outs = []
    for compiled in (True, False):
        monkeypatch.setattr(E, '_PLAN_C', compiled)
        m = UNetModel(sess=None, dataset=ArrayDataSet(x, y), n_classes=2, input_dims=188, learning_rate=1e-3, log_dir=None, save_dir=None,
                      load_snapshot=False, dtype='bf16', use_graph=False, seed=7)
        for _ in range(3):
            m.train_step()
        torch.cuda.synchronize()
        if compiled:
            cps = m.step_plan.__dict__.get('_compiled', {})
            assert len(cps) == 1, 'the eager step did not go through seg_plan_run'
            cp = list(cps.values())[0]
            kinds = [int(cp.ops[k].kind) for k in range(cp.n)]
            assert kinds.count(0) >= 70 and (kinds.count(2) >= 10 or kinds.count(1) >= 10), kinds      # launches + signal (or event) forks
        else:
            assert not m.step_plan.__dict__.get('_compiled')
        outs.append((m.store.p.clone(), m.store.m.clone(), m.store.v.clone(), m.last_loss()))
    (p1, m1, v1, l1), (p0, m0, v0, l0) = outs
    assert torch.equal(p1, p0) and torch.equal(m1, m0) and torch.equal(v1, v0) and abs(l1 - l0) < 1e-5       # (the reported loss is a float-atomic sum)


@pytest.mark.gpu
def test_last_kernel_name_reports_the_first_layer_instance(monkeypatch):
    """seg_last_kernel_name(): the first kernel launched since the previous query, as spelled at its launch site -- bench.py labels the
    first layer with the instance the C side picked (window form for small outputs, float-staged MFMA form otherwise / when forced)."""
    from segmentation_amd import _lib as L, engine as E
    import gpu_util as U
    lib = L.load()
    layer = E.Layer('f', 'first', 3, [3], 32, 'VALID', True)
    rng = np.random.default_rng(3)
    store = U.make_store([layer], L.SEG_BF16, {'f': {'weights': rng.standard_normal(layer.wshape).astype(np.float32), 'biases': np.zeros(32, np.float32)}})
    net = E.Net(store, 2, L.SEG_BF16, U.dev())
    xt = torch.rand(2, 40, 40, 3, device=U.dev())
    out = net.act(38, 38, 32); pooled = net.act(19, 19, 32)
    for impl, want in (('win', 'conv_first_win_kernel<1,true,true>'), ('old', 'conv_first_mfma_kernel<1>')):
        monkeypatch.setenv('SEG_FIRST_IMPL', impl); lib.seg_dbg_reload_env()
        pl = E.Plan('p'); assert net.first_fwd(pl, layer, xt, 40, 40, out, pool=pooled)
        lib.seg_last_kernel_name()
        pl.run(U.stream()); U.sync()
        assert lib.seg_last_kernel_name().decode() == want
        assert lib.seg_last_kernel_name().decode() == ''                   # (the query clears)
        rows = pl.run_profiled(U.stream(), torch)
        assert rows[0][1] == want, rows
    monkeypatch.delenv('SEG_FIRST_IMPL'); lib.seg_dbg_reload_env()
