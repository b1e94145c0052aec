"""CPU tests of the boundary: libseg_hip.so builds/loads, exports every symbol include/seg_hip.h declares, the
ctypes mirrors have the C struct layouts, and the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, 'include', 'seg_hip.h')


def _declared():
    txt = open(HDR).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(seg_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from segmentation_amd import _build, _lib
    _build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    bound = set(_lib.SIGNATURES) | {'seg_last_error', 'seg_version', 'seg_last_kernel_name', 'seg_dbg_reload_env', 'seg_plan_fn_id', 'seg_plan_run', 'seg_plan_destroy_events', 'seg_bn_ws_bytes', 'seg_dconv_wgrad_ws_bytes', 'seg_bilinear_up_bwd_ws_bytes',
             'seg_head_xent_ws_bytes', 'seg_bias_grad_ws_bytes', 'seg_conv_first_gen_rows', 'seg_thin_up2x2_rows', 'seg_thin_wgrad3x3_ws_bytes', 'seg_conv_first_gen_wgrad_ws_bytes'}
    assert bound == set(names)
    assert _lib.load().seg_version() == 100


def test_plan_thunks_are_up_to_date_and_cover_every_launcher():
    """csrc/plan_thunks.inc is generated from _lib.SIGNATURES (segmentation_amd/_gen_thunks.py): the committed file must be what the
    generator writes today, and seg_plan_fn_id must know every launching entry point."""
    from segmentation_amd import _gen_thunks as G, _lib
    assert open(G.OUT).read() == G.render(), 'run python -m segmentation_amd._gen_thunks'
    lib = _lib.load()
    names = [n for n, _ in G.launchers()]
    assert len(names) >= 55 and 'seg_conv2d' in names and 'seg_adam' in names
    ids = [lib.seg_plan_fn_id(n.encode()) for n in names]
    assert sorted(ids) == list(range(len(names)))
    assert lib.seg_plan_fn_id(b'seg_conv2d_kernel_name') == -1 and lib.seg_plan_fn_id(b'nope') == -1
    # argument checks without a GPU: a launch op whose argument count does not match its entry point is refused
    a = (_lib.SegArg * 2)()
    op = (_lib.PlanOp * 1)()
    op[0].kind = _lib.OP_LAUNCH; op[0].fn = lib.seg_plan_fn_id(b'seg_adam'); op[0].stream = 0; op[0].nargs = 2; op[0].args = a
    st = (ctypes.c_void_p * 1)(None)
    bad = ctypes.c_int32(-5)
    assert lib.seg_plan_run(op, 1, st, 1, None, 0, ctypes.byref(bad)) == -1 and bad.value == 0
    assert b'takes 11 arguments' in lib.seg_last_error()
    assert lib.seg_plan_run(op, 0, st, 1, None, 0, ctypes.byref(bad)) == 0


def test_ctypes_struct_layouts_match_c(tmp_path):
    from segmentation_amd import _lib
    src = tmp_path / 's.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "seg_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(seg_view),sizeof(seg_conv_desc),sizeof(seg_wgrad_desc),sizeof(seg_pack_entry),'
                   'offsetof(seg_conv_desc,dst),offsetof(seg_conv_desc,cfg),offsetof(seg_wgrad_desc,dw),'
                   'sizeof(seg_dconv_desc),offsetof(seg_dconv_desc,w),offsetof(seg_dconv_desc,mask),sizeof(seg_arg),sizeof(seg_plan_op),'
                   'offsetof(seg_plan_op,args),offsetof(seg_plan_op,event),offsetof(seg_conv_desc,signal),offsetof(seg_conv_desc,signal_value));return 0;}')
    exe = tmp_path / 's'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    want = [ctypes.sizeof(_lib.View), ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.WgradDesc), ctypes.sizeof(_lib.PackEntry),
            _lib.ConvDesc.dst.offset, _lib.ConvDesc.cfg.offset, _lib.WgradDesc.dw.offset,
            ctypes.sizeof(_lib.DconvDesc), _lib.DconvDesc.w.offset, _lib.DconvDesc.mask.offset,
            ctypes.sizeof(_lib.SegArg), ctypes.sizeof(_lib.PlanOp), _lib.PlanOp.args.offset, _lib.PlanOp.event.offset,
            _lib.ConvDesc.signal.offset, _lib.ConvDesc.signal_value.offset]
    assert got == want


def test_argument_validation_without_gpu():
    from segmentation_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    assert lib.seg_conv2d(ctypes.byref(d), None) == -1
    assert b'null pointer' in lib.seg_last_error()
    w = _lib.WgradDesc()
    assert lib.seg_conv2d_wgrad(ctypes.byref(w), None) == -1
    with pytest.raises(_lib.SegError):
        _lib.check(-1, 'x')


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from segmentation_amd.unet import UNetModel
    from segmentation_amd.datasets import ArrayDataSet
    import numpy as np
    ds = ArrayDataSet(np.zeros((1, 1, 188, 188, 3), np.float32), np.zeros((1, 1, 188, 188, 1), np.uint8))
    with pytest.raises(Exception) as e:
        UNetModel(sess=None, dataset=ds, n_classes=2, input_dims=188, save_dir=None, load_snapshot=False)
    assert 'no GPU' in str(e.value) or 'HIP' in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'segmentation_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, flags=re.M), f


def test_unet_plan_geometry():
    from segmentation_amd.unet import unet_sizes, unet_layers
    s = unet_sizes(256)
    assert (s['conv1_1'], s['conv1_2'], s['pool1'], s['conv5_2'], s['upconv1'], s['upconv4'], s['output']) == (254, 252, 127, 8, 16, 72, 68)
    assert unet_sizes(512)['output'] == 324 and unet_sizes(186)['output'] == 4
    with pytest.raises(Exception):
        unet_sizes(128)
    ls = unet_layers(4, 32, 3)
    assert sum(l.wsize + l.cout for l in ls) == 7760196


def test_signal_forks_are_off_under_any_tool(monkeypatch):
    """hipStreamWaitValue32 waiters have no time-out and deadlock when a tool dispatches one kernel at a time (rocprofv3 --pmc):
    the default is conservative -- events whenever ANY profiler / HSA tool / serialising mode is in sight; only an
    explicit SEG_FORK_SIGNAL=1 (kernel tracing, which leaves the queues concurrent) overrides."""
    from segmentation_amd import engine as E
    for k in list(os.environ):
        if k.startswith(('ROCPROF', 'ROCP_', 'HSA_TOOLS', 'ROCTRACER')) or k in ('AMD_SERIALIZE_KERNEL', 'HIP_LAUNCH_BLOCKING', 'AMD_LOG_LEVEL', 'SEG_FORK_SIGNAL', 'LD_PRELOAD'):
            monkeypatch.delenv(k, raising=False)
    assert E._signals_allowed()
    for var, val in (('ROCPROF_KERNEL_TRACE', '1'), ('ROCPROF_COUNTER_COLLECTION', '1'), ('ROCPROF_COUNTERS', 'pmc: FETCH_SIZE'),
                     ('ROCP_TOOL_LIBRARIES', 'librocprofiler-sdk-tool.so'), ('HSA_TOOLS_LIB', 'librocprofiler64.so'),
                     ('LD_PRELOAD', '/opt/rocm/lib/librocprofiler-sdk-tool.so'), ('AMD_SERIALIZE_KERNEL', '3'), ('HIP_LAUNCH_BLOCKING', '1')):
        monkeypatch.setenv(var, val)
        assert not E._signals_allowed(), var
        monkeypatch.setenv('SEG_FORK_SIGNAL', '1'); assert E._signals_allowed()
        monkeypatch.delenv('SEG_FORK_SIGNAL'); monkeypatch.delenv(var)
    monkeypatch.setenv('SEG_FORK_SIGNAL', '0'); assert not E._signals_allowed()
