"""ctypes binding of libseg_hip.so (C-ABI in include/seg_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails this module
raises.  Nothing here imports the oracle."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SEG_LIB_PATH') or os.path.join(HERE, 'libseg_hip.so')     # (SEG_LIB_PATH: debug builds, tools/wgrad_ablate.sh)

SEG_F32, SEG_BF16 = 0, 1
PACK_CONV_FWD, PACK_CONV_DGRAD, PACK_UP_FWD, PACK_UP_DGRAD = 0, 1, 2, 3


class SegError(RuntimeError):
    pass


class View(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32), ('cs', C.c_int32), ('coff', C.c_int32),
                ('oy', C.c_int32), ('ox', C.c_int32), ('c', C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [('src0', View), ('src1', View), ('B', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32),
                ('KH', C.c_int32), ('KW', C.c_int32), ('stride', C.c_int32), ('pad_t', C.c_int32), ('pad_l', C.c_int32),
                ('Ho', C.c_int32), ('Wo', C.c_int32), ('w_packed', C.c_void_p), ('n_total', C.c_int32),
                ('n_off', C.c_int32), ('n_count', C.c_int32), ('bias', C.c_void_p), ('bias_n', C.c_int32),
                ('dst', View), ('up2', C.c_int32), ('up_cout', C.c_int32), ('mask', View), ('relu', C.c_int32),
                ('out_f32', C.c_int32), ('dtype', C.c_int32), ('cfg', C.c_int32), ('accum', C.c_int32),
                ('n_split', C.c_int32), ('dst1', View), ('mask1', View), ('pool', View), ('pool_h', C.c_int32), ('pool_w', C.c_int32),
                ('signal', C.c_void_p), ('signal_value', C.c_uint32), ('sched', C.c_void_p),
                ('ksplit', C.c_int32), ('n_store', C.c_int32), ('splitk_ws', C.c_void_p), ('splitk_tickets', C.c_void_p),
                ('thin_src', C.c_int32), ('thin_pad_', C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [('src0', View), ('src1', View), ('src0_clog', C.c_int32), ('src1_clog', C.c_int32),
                ('B', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32),
                ('stride', C.c_int32), ('pad_t', C.c_int32), ('pad_l', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32),
                ('dz', View), ('n_log', C.c_int32), ('dw', C.c_void_p), ('dtype', C.c_int32), ('cfg', C.c_int32),
                ('ws', C.c_void_p), ('ws_bytes', C.c_int64), ('ksplit', C.c_int32), ('bias_mode', C.c_int32),
                ('db', C.c_void_p), ('bias_n', C.c_int32), ('phase', C.c_int32),
                ('im2col_x', C.c_void_p), ('im2col_h', C.c_int32), ('im2col_w', C.c_int32), ('im2col_cin', C.c_int32),
                ('im2col_pad', C.c_int32), ('pool_y', View), ('pool_dp', View), ('pool_add', View),
                ('pool_add_h', C.c_int32), ('pool_add_w', C.c_int32), ('pool_add_y0', C.c_int32), ('pool_add_x0', C.c_int32),
                ('target_wgs', C.c_int32), ('thin', C.c_int32)]


class DconvDesc(C.Structure):
    _fields_ = [('x', View), ('xc', C.c_int32), ('y', View), ('yc', C.c_int32), ('B', C.c_int32), ('Hx', C.c_int32), ('Wx', C.c_int32),
                ('Hy', C.c_int32), ('Wy', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32), ('stride', C.c_int32), ('pad_t', C.c_int32),
                ('pad_l', C.c_int32), ('w', C.c_void_p), ('w_su', C.c_int64), ('w_sv', C.c_int64), ('w_sk', C.c_int64),
                ('bias', C.c_void_p), ('bias_n', C.c_int32), ('relu', C.c_int32), ('mask', View), ('dtype', C.c_int32)]


class PackEntry(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst_off', C.c_int64), ('mode', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32),
                ('cin', C.c_int32), ('cout', C.c_int32), ('seg0_c', C.c_int32), ('seg0_cp', C.c_int32),
                ('seg1_c', C.c_int32), ('seg1_cp', C.c_int32), ('cout_pad', C.c_int32), ('k_pad', C.c_int32),
                ('n_total', C.c_int32), ('n_elems', C.c_int64), ('blk_start', C.c_int64)]


class SegArg(C.Union):
    _fields_ = [('i', C.c_int64), ('f32', C.c_float), ('p', C.c_void_p)]


class PlanOp(C.Structure):
    _fields_ = [('kind', C.c_int32), ('fn', C.c_int32), ('stream', C.c_int32), ('stream2', C.c_int32), ('nargs', C.c_int32),
                ('signal_slot', C.c_int32), ('args', C.POINTER(SegArg)), ('event', C.c_void_p)]


OP_LAUNCH, OP_EVENT_FORK, OP_WAIT_VALUE = 0, 1, 2

i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p
PV = C.POINTER(View)

# name -> argtypes (restype is always int except the two getters); mirrors include/seg_hip.h
SIGNATURES = {
    'seg_conv2d': [C.POINTER(ConvDesc), vp],
    'seg_conv2d_wgrad': [C.POINTER(WgradDesc), vp],
    'seg_conv2d_kernel_name': [C.POINTER(ConvDesc), C.c_char_p, i32],
    'seg_conv2d_splitk_plan': [C.POINTER(ConvDesc), C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)],
    'seg_conv2d_wgrad_kernel_name': [C.POINTER(WgradDesc), C.c_char_p, i32],
    'seg_conv2d_wgrad_plan': [C.POINTER(WgradDesc), C.POINTER(i32), C.POINTER(i64)],
    'seg_wgrad_reduce_batch_plan': [C.POINTER(C.POINTER(WgradDesc)), i32, vp, i64, C.POINTER(i32), C.POINTER(i32)],
    'seg_wgrad_reduce_batch': [vp, i32, i32, vp],
    'seg_conv_first_fwd': [vp, i32, i32, i32, i32, vp, vp, i32, i32, PV, i32, i32, i32, i32, vp],
    'seg_conv_first_gen': [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, PV, i32, i32, i32, i32, vp],
    'seg_conv_first_gen_wgrad': [vp, i32, i32, i32, i32, PV, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, i32, vp],
    'seg_conv_first_gen_bn': [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, PV, i32, i32, i32, vp, i32, i32, vp],
    'seg_conv_first_pool_fwd': [vp, i32, i32, i32, i32, vp, vp, i32, i32, PV, i32, i32, i32, PV, i32, i32, i32, vp],
    'seg_im2col3x3': [vp, i32, i32, i32, i32, i32, PV, i32, i32, i32, vp],
    'seg_im2col': [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, PV, i32, i32, i32, vp],
    'seg_im2col_act': [PV, i32, i32, i32, i32, i32, i32, i32, i32, i32, PV, i32, i32, i32, vp],
    'seg_col2im': [PV, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, PV, i32, i32, i32, vp],
    'seg_maxpool2x2_fwd': [PV, PV, vp, i32, i32, i32, i32, i32, vp],
    'seg_maxpool2x2_bwd': [PV, PV, PV, i32, i32, i32, i32, PV, i32, i32, i32, i32, i32, vp],
    'seg_softmax_xent': [PV, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, PV, i32, vp],
    'seg_softmax_xent_probs': [PV, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, PV, PV, i32, vp],
    'seg_head_xent': [PV, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, PV, PV, PV, vp, i64, i32, vp],
    'seg_head_dw_reduce': [vp, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    'seg_sigmoid_argmax': [PV, i32, i32, i32, i32, vp, vp, vp],
    'seg_bias_grad': [PV, i32, i32, i32, i32, vp, i32, vp],
    'seg_bias_grad_ws': [PV, i32, i32, i32, i32, vp, vp, i64, i32, vp],
    'seg_thin_up2x2': [PV, PV, i32, i32, i32, vp, vp, i32, i32, i32, i32, PV, i32, vp],
    'seg_thin_up2x2_bn': [PV, PV, i32, i32, i32, vp, vp, i32, i32, i32, vp, i32, vp],
    'seg_thin_conv3x3': [PV, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, PV, PV, i32, i32, i32, i32, vp],
    'seg_thin_wgrad3x3': [PV, i32, i32, i32, PV, i32, i32, i32, i32, i32, vp, vp, vp, i64, i32, vp],
    'seg_thin_wgrad3x3_bn': [PV, i32, i32, i32, PV, i32, i32, i32, i32, i32, vp, vp, vp, i64, vp, vp, i32, vp],
    'seg_thin_conv3x3_bn': [PV, i32, i32, i32, vp, vp, i32, i32, i32, i32, PV, i32, i32, i32, vp, vp, i32, vp],
    'seg_adam': [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp],
    'seg_step_increment': [vp, vp],
    'seg_step_begin': [vp, vp, vp],
    'seg_pack_weights': [vp, vp, vp, i32, i64, i32, vp],
    'seg_pack_weights_dual': [vp, vp, vp, vp, i32, i64, i32, vp],
    'seg_bilinear_up_fwd': [PV, i32, i32, i32, vp, PV, PV, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    'seg_bilinear_up_bwd': [PV, i32, i32, i32, i32, i32, vp, PV, i32, i32, i32, i32, i32, PV, PV, i32, vp],
    'seg_bilinear_xent': [PV, i32, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, PV, PV, i32, vp],
    'seg_bilinear_up_bwd_sep': [PV, i32, i32, i32, i32, i32, vp, PV, i32, i32, i32, i32, i32, vp, C.c_int64, PV, PV, i32, vp],
    'seg_relu_grad': [PV, PV, PV, i32, i32, i32, i32, i32, vp],
    'seg_dropout': [PV, PV, i32, i32, i32, i32, f32, u64, u64, i32, vp],
    'seg_cast_pad': [vp, i64, i32, PV, i32, vp],
    'seg_round_bf16': [PV, PV, i32, i32, i32, i32, vp],
    'seg_round_bf16_flat': [vp, vp, i64, vp],
    'seg_dconv_fwd': [C.POINTER(DconvDesc), vp],
    'seg_dconv_bwd_data': [C.POINTER(DconvDesc), vp],
    'seg_dconv_wgrad': [C.POINTER(DconvDesc), vp, vp, i32, vp, C.c_int64, vp],
    'seg_maxpool_k_fwd': [PV, PV, i32, i32, i32, i32, i32, i32, vp],
    'seg_maxpool_k_bwd': [PV, PV, PV, i32, i32, i32, i32, i32, i32, vp],
    'seg_bn_fwd': [PV, PV, vp, vp, vp, i32, f32, f32, i32, i32, i32, i32, i32, vp, i32, vp],
    'seg_bn_fwd_rows': [PV, PV, vp, vp, vp, f32, f32, i32, i32, i32, i32, i32, vp, i32, i32, vp],
    'seg_bn_pool_fwd': [PV, PV, vp, vp, vp, i32, f32, f32, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp],
    'seg_bn_pool_relu_bwd': [PV, PV, PV, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp],
    'seg_bn_relu_bwd': [PV, PV, PV, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp],
    'seg_resize_bilinear_fwd': [PV, i32, i32, PV, i32, i32, i32, i32, i32, vp],
    'seg_resize_bilinear_bwd': [PV, i32, i32, PV, i32, i32, i32, i32, i32, vp],
    'seg_dropout_step': [PV, PV, i32, i32, i32, i32, f32, u64, u64, vp, i32, vp],
    'seg_onehot': [vp, i32, i32, i32, i32, i32, i32, i32, PV, i32, vp],
    'seg_softmax_probs': [PV, i32, i32, i32, i32, PV, i32, vp],
    'seg_softmax_bwd_add': [PV, PV, i32, i32, i32, i32, f32, PV, i32, vp],
    'seg_flatten': [PV, i32, i32, i32, i32, PV, i32, i32, vp],
    'seg_bn_rows_fwd': [PV, PV, vp, vp, vp, i32, i32, f32, f32, i32, vp],
    'seg_bn_rows_bwd': [PV, PV, PV, vp, vp, i32, i32, i32, i32, i32, vp],
    'seg_bce2': [PV, i32, i32, f32, vp, PV, i32, vp],
}

_lib = None


def load():
    """Loads the library (once).  Raises SegError if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SegError('libseg_hip.so not found at %s: run `python -c "import __graft_entry__ as g; g.build()"` '
                       '(the HIP extension is mandatory; there is no CPU fallback)' % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Load
    # torch's copy first so that libseg_hip.so binds to the runtime that owns torch's device context and streams.
    import torch
    rt = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(rt):
        C.CDLL(rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = C.c_int
    lib.seg_last_error.restype = C.c_char_p
    lib.seg_last_error.argtypes = []
    lib.seg_version.restype = C.c_int
    lib.seg_version.argtypes = []
    lib.seg_dbg_reload_env.restype = None
    lib.seg_dbg_reload_env.argtypes = []
    lib.seg_last_kernel_name.restype = C.c_char_p
    lib.seg_last_kernel_name.argtypes = []
    lib.seg_plan_fn_id.restype = C.c_int
    lib.seg_plan_fn_id.argtypes = [C.c_char_p]
    lib.seg_plan_run.restype = C.c_int
    lib.seg_plan_run.argtypes = [C.POINTER(PlanOp), C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
    lib.seg_plan_destroy_events.restype = C.c_int
    lib.seg_plan_destroy_events.argtypes = [C.POINTER(PlanOp), C.c_int32]
    lib.seg_bn_ws_bytes.restype = C.c_int64
    lib.seg_bn_ws_bytes.argtypes = [C.c_int32]
    lib.seg_dconv_wgrad_ws_bytes.restype = C.c_int64
    lib.seg_dconv_wgrad_ws_bytes.argtypes = [C.POINTER(DconvDesc)]
    lib.seg_head_xent_ws_bytes.restype = C.c_int64
    lib.seg_head_xent_ws_bytes.argtypes = [C.c_int32] * 5
    lib.seg_bias_grad_ws_bytes.restype = C.c_int64
    lib.seg_bias_grad_ws_bytes.argtypes = [C.c_int32]
    lib.seg_thin_wgrad3x3_ws_bytes.restype = C.c_int64
    lib.seg_thin_wgrad3x3_ws_bytes.argtypes = [C.c_int32] * 2
    lib.seg_conv_first_gen_wgrad_ws_bytes.restype = C.c_int64
    lib.seg_conv_first_gen_wgrad_ws_bytes.argtypes = [C.c_int32]
    lib.seg_conv_first_gen_rows.restype = C.c_int32
    lib.seg_conv_first_gen_rows.argtypes = [C.c_int32] * 4
    lib.seg_thin_up2x2_rows.restype = C.c_int32
    lib.seg_thin_up2x2_rows.argtypes = [C.c_int32] * 3
    lib.seg_bilinear_up_bwd_ws_bytes.restype = C.c_int64
    lib.seg_bilinear_up_bwd_ws_bytes.argtypes = [C.c_int32] * 4
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        raise SegError('%s failed (%d): %s' % (what, rc, load().seg_last_error().decode()))


def null_view():
    return View(None, 0, 0, 0, 0, 0, 0, 0)


_hip = None


def hip_runtime():
    """ctypes handle of the HIP runtime torch uses, with the three event calls Plan.run needs (raw events carry the
    hipEventReleaseToDevice flag torch.cuda.Event cannot express: a device-scope release instead of a system-scope one at
    every fork point)."""
    global _hip
    if _hip is None:
        import torch
        rt = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
        h = C.CDLL(rt if os.path.exists(rt) else 'libamdhip64.so', mode=C.RTLD_GLOBAL)
        h.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]; h.hipEventCreateWithFlags.restype = C.c_int
        h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]; h.hipEventRecord.restype = C.c_int
        h.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]; h.hipStreamWaitEvent.restype = C.c_int
        h.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]; h.hipStreamWaitValue32.restype = C.c_int
        h.hipDeviceGetAttribute.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]; h.hipDeviceGetAttribute.restype = C.c_int
        _hip = h
    return _hip


HIP_EVENT_DISABLE_TIMING = 0x2
HIP_DEVICE_ATTRIBUTE_CAN_USE_STREAM_WAIT_VALUE = 10013     # hipDeviceAttributeCanUseStreamWaitValue (hip_runtime_api.h, AMD-specific block)
HIP_EVENT_RELEASE_TO_DEVICE = 0x40000000
