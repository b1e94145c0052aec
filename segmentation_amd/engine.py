"""Host-side engine: parameter arenas, activation buffers and static launch plans.

A model is compiled once into flat lists of C-ABI launches (forward / backward / update); a
train step replays them on one HIP stream (eagerly, or as a captured hipGraph).  PyTorch only
provides device memory, streams and torch.distributed here -- every arithmetic op is a kernel
of libseg_hip.so.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


def rup(c, m=32):
    return (c + m - 1) // m * m


def torch_dtype(dtype):
    return torch.float32 if dtype == L.SEG_F32 else torch.bfloat16


class Act(object):
    """NHWC activation buffer [B,H,W,Cp]; Cp = channels padded to 32, pad channels stay zero."""

    def __init__(self, B, H, W, C, dtype, device, f32=False, name='', thin=False):
        # thin: <= 8 logical channels kept at a channel stride of 8 (16 bytes per bf16 pixel instead of 64).  The MFMA kernels
        # read such a tensor as 32 channels per pixel (view_wide: the next pixels' values meet zero filter rows) and store only
        # the first 8 (seg_conv_desc.n_store); the buffer carries 64 elements of zeroed slack for the reads past its last pixel.
        self.thin = bool(thin) and C <= 8
        self.B, self.H, self.W, self.C, self.Cp = B, H, W, C, (8 if self.thin else rup(C))
        self.name = name
        n = B * H * W * self.Cp
        self._buf = torch.zeros(n + (64 if self.thin else 0), dtype=torch.float32 if f32 else torch_dtype(dtype), device=device)
        self.t = self._buf[:n].view(B, H, W, self.Cp)

    def view(self, oy=0, ox=0):
        return L.View(self.t.data_ptr(), self.H, self.W, self.Cp, 0, oy, ox, self.Cp)

    def view_wide(self, oy=0, ox=0):
        """a thin activation as a 32-channel operand of the MFMA kernels (thin_src / seg_wgrad_desc.thin)"""
        return L.View(self.t.data_ptr(), self.H, self.W, self.Cp, 0, oy, ox, 32) if self.thin else self.view(oy, ox)

    def nbytes(self):
        return self.t.numel() * self.t.element_size()


class Layer(object):
    """One trainable layer.  kind: 'first' (3x3, raw float input), 'conv' (k x k stride 1, MFMA tiles), 'up' (2x2/s2 transposed,
    MFMA tiles), 'direct' (k x k conv of any stride on the direct kernels), 'dtrans' (k x k transposed conv of any stride on
    the direct kernels, filter [kh,kw,Cout,Cin]), 'bn' (slim.batch_norm with its defaults: a beta vector, no gamma)."""

    def __init__(self, name, kind, k, cin_segs, cout, padding='VALID', relu=True, stride=1):
        self.name, self.kind, self.k, self.cout, self.padding, self.relu, self.stride = name, kind, k, cout, padding, relu, stride
        self.cin_segs = list(cin_segs)                       # logical channels of each concat segment
        self.cin = sum(self.cin_segs)
        self.cin_p = [rup(c) for c in self.cin_segs]
        self.cout_p = rup(cout)
        self.nbias = cout
        self.wname, self.bname = 'weights', 'biases'
        if kind in ('up', 'dtrans'):
            self.wshape = (k, k, cout, self.cin)              # TF conv2d_transpose filter layout
        elif kind == 'bn':
            self.wshape = (cout,)                             # beta
            self.nbias = 0
            self.wname = 'beta'
        else:
            self.wshape = (k, k, self.cin, cout)              # HWIO
        self.pad = 0 if padding == 'VALID' else (k - 1) // 2  # TF SAME, odd k, stride 1: before = (k-1)//2
        self.w_off = self.b_off = -1
        self.pk_fwd = self.pk_dgrad = -1
        self.need_dgrad = True

    @property
    def packed(self):
        """layers with a packed MFMA-operand copy of their weights"""
        return self.kind in ('conv', 'up')

    @property
    def wsize(self):
        return int(np.prod(self.wshape))


class ParamStore(object):
    """Flat fp32 arenas (params, grads, Adam m/v) + the packed MFMA-operand copy in the compute dtype.
    Tensors are laid out in the order given (the models pass backward-production order so that
    gradient buckets for the all-reduce are contiguous prefixes)."""

    def __init__(self, layers, dtype, device, training=True):
        self.layers = {l.name: l for l in layers}
        self.order = [l.name for l in layers]
        self.dtype, self.device, self.training = dtype, device, training
        off = 0
        for l in layers:
            l.w_off = off; off += l.wsize
            l.b_off = off; off += l.nbias
        self.n = off
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        if training:
            # (+1 float behind the gradients: the loss accumulator lives at g[n], so that under data parallelism the LAST gradient
            # bucket carries it through the same all-reduce -- BaseModel._train_step_dp -- and the reported loss is the global mean)
            self.g_full = torch.zeros(off + 1, dtype=torch.float32, device=device)
            self.g = self.g_full[:off]
            self.m = torch.zeros(off, dtype=torch.float32, device=device)
            self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.step = torch.zeros(2, dtype=torch.int64, device=device)     # [global_step, steps completed before the current one]
        # packed arena + table
        entries, poff, blk = [], 0, 0
        for l in layers:
            if not l.packed:
                continue
            seg0, seg1 = l.cin_segs[0], (l.cin_segs[1] if len(l.cin_segs) > 1 else 0)
            seg0p, seg1p = l.cin_p[0], (l.cin_p[1] if len(l.cin_p) > 1 else 0)
            kin = seg0p + seg1p
            modes = []
            if l.kind == 'conv':
                modes.append((L.PACK_CONV_FWD, l.k * l.k, kin, l.cout_p, 'pk_fwd'))
                if (training and l.need_dgrad) or getattr(l, 'tconv', None):      # (a transposed conv's FORWARD is the 1x1 dgrad)
                    modes.append((L.PACK_CONV_DGRAD, l.k * l.k, l.cout_p, kin, 'pk_dgrad'))
            else:
                modes.append((L.PACK_UP_FWD, 1, kin, 4 * l.cout_p, 'pk_fwd'))
                if training and l.need_dgrad:
                    modes.append((L.PACK_UP_DGRAD, 4, l.cout_p, kin, 'pk_dgrad'))
            for mode, taps, kpad, ntot, attr in modes:
                ne = taps * kpad * ntot
                e = L.PackEntry(l.w_off, poff, mode, l.k if l.kind == 'conv' else 2, l.k if l.kind == 'conv' else 2,
                                l.cin, l.cout, seg0, seg0p, seg1, seg1p, l.cout_p, kpad, ntot, ne, blk)
                setattr(l, attr, poff)
                entries.append(e)
                poff += ne
                blk += ne // 1024                 # one block per 32x32 tile
        self.packed = torch.zeros(max(poff, 8), dtype=torch_dtype(dtype), device=device)
        self.n_pack_entries, self.pack_blocks = len(entries), blk
        # one-read re-pack (seg_pack_weights_dual): the forward entries only, 32x32 tiles counted over them, each with the packed
        # offset of its dgrad copy (training stores; an inference store has no dgrad copies and uses the table-driven kernel)
        self.pack_dual = None
        self._pack_entries = entries
        if training and entries:
            self.pack_dual = self._dual_table(lambda src_off: True)
        if entries:
            raw = b''.join(bytes(e) for e in entries)
            self.pack_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        else:
            self.pack_table = None

    def _dual_table(self, keep):
        """seg_pack_weights_dual table over the forward entries whose arena offset passes `keep`: (entries, dgrad offsets, n, tiles)"""
        entries = self._pack_entries
        fwd, dg, tiles = [], [], 0
        for i, e in enumerate(entries):
            if e.mode not in (L.PACK_CONV_FWD, L.PACK_UP_FWD) or not keep(e.src_off):
                continue
            f = L.PackEntry.from_buffer_copy(e)
            f.blk_start = tiles
            tiles += e.n_elems // 1024
            nxt = entries[i + 1] if i + 1 < len(entries) else None
            dg.append(nxt.dst_off if nxt is not None and nxt.src_off == e.src_off and nxt.mode in (L.PACK_CONV_DGRAD, L.PACK_UP_DGRAD) else -1)
            fwd.append(f)
        if not fwd:
            return None
        return (torch.frombuffer(bytearray(b''.join(bytes(f) for f in fwd)), dtype=torch.uint8).to(self.device),
                torch.tensor(dg, dtype=torch.int64, device=self.device), len(fwd), tiles)

    # ---- host access (tests, snapshots, weight loading) ----
    def packed_ptr(self, off):
        return self.packed.data_ptr() + off * self.packed.element_size()

    def p_ptr(self, off):
        return self.p.data_ptr() + off * 4

    def g_ptr(self, off):
        return self.g.data_ptr() + off * 4

    def set_params(self, params):
        """params: {name: {'weights': ndarray, 'biases': ndarray}} in TF layouts ({'beta': ndarray} for a batch norm)."""
        host = self.p.cpu().numpy().copy()
        for name, l in self.layers.items():
            w = np.asarray(params[name][l.wname], np.float32)
            if tuple(w.shape) != tuple(l.wshape):
                raise ValueError('shape mismatch for %s: %s vs %s' % (name, w.shape, l.wshape))
            host[l.w_off:l.w_off + l.wsize] = w.reshape(-1)
            if l.nbias:
                b = np.asarray(params[name][l.bname], np.float32)
                if b.shape != (l.nbias,):
                    raise ValueError('shape mismatch for %s biases: %s vs %s' % (name, b.shape, (l.nbias,)))
                host[l.b_off:l.b_off + l.nbias] = b
        self.p.copy_(torch.from_numpy(host))

    def _unflatten(self, flat):
        host = flat.detach().cpu().numpy()
        out = {}
        for name, l in self.layers.items():
            out[name] = {l.wname: host[l.w_off:l.w_off + l.wsize].reshape(l.wshape).copy()}
            if l.nbias:
                out[name][l.bname] = host[l.b_off:l.b_off + l.nbias].copy()
        return out

    def get_params(self):
        return self._unflatten(self.p)

    def get_grads(self):
        return self._unflatten(self.g)

    def loss_slot(self):
        """the float behind the gradient arena that models use as their loss accumulator"""
        return self.g_full[self.n:self.n + 1]


def _step_wgrad_wgs(input_pixels):
    """Workgroups a filter gradient aims for INSIDE a training step (0 = the library's default for a lone launch, 128).  It shares
    the chip with the data gradient on the critical stream and the other side stream's filter gradient.  Small steps (C2: 16 x 256^2
    = 1 M input pixels, C3: 8 x 512^2 = 2 M) are bound by the critical stream: 64 workgroups per filter gradient leave it half of
    the CUs (C2 on one box: 48 / 64 / 80 / 96 / 128 -> 0.986 / 0.960 / 0.974 / 0.976 / 0.987 ms; C3 0.797 against 0.817).  At 4 M
    input pixels (C4's shard, DeconvModel 512^2) the side streams are the tail and 128 stays (4.09 against 4.12 ms, 2.81 against
    2.92) -- profiles/r03_step_structure_ab.txt.  SEG_WGRAD_WGS overrides."""
    if os.environ.get('SEG_WGRAD_WGS'):
        return int(os.environ['SEG_WGRAD_WGS'])
    return 64 if input_pixels is not None and input_pixels <= 2500000 else 0


_DEBUG_SKIP = tuple(x for x in os.environ.get('SEG_DEBUG_SKIP', '').split(',') if x)


def _fork(src, dst):
    """cross-stream dependency: `dst waits for what src holds now` (under stream capture this becomes a graph edge)"""
    ev = torch.cuda.Event(); ev.record(src); dst.wait_event(ev)


# Signal forks (eager launches only).  An event recorded on the main stream between two kernels costs that stream 4-7 us (a
# barrier packet with a completion signal; tools/micro/sigwait.hip) and a backward pass forks a filter gradient off before
# almost every data gradient: ~28 records, ~0.15 ms of a 1.17 ms step.  Instead the NEXT convolution on the main stream
# stores a sequence number when it starts (seg_conv_desc.signal: on an in-order stream that means everything before it is
# complete and released) and the side stream waits for that number with hipStreamWaitValue32: nothing is added to the main
# queue.  One flag per main stream (the numbers must reach it in launch order); SEG_FORK_SIGNAL=0 falls back to events.
_SIGNALS = {}
def _signals_allowed():
    """Signal forks need the waiting kernel and the kernel it waits for to be free to run at the same time, and the wait has no
    time-out: a tool or runtime mode that dispatches one kernel at a time across queues (rocprofv3 --pmc serialises dispatches,
    thread trace and PC sampling likewise) can pick the waiter first and never dispatch its signaller -- an unrecoverable hang.
    So the default is conservative: signals only when NO profiler / HSA tool is in sight (any ROCPROF_* / ROCP_* / HSA_TOOLS_*
    variable, rocprof or roctracer libraries in LD_PRELOAD), no serialising debug mode is set, and the device reports
    hipDeviceAttributeCanUseStreamWaitValue (checked in _signal_state).  SEG_FORK_SIGNAL=0 forces events, =1 forces signals
    (kernel tracing alone leaves the queues concurrent: tools/refresh_lines.sh sets it for rocprofv3 --kernel-trace)."""
    e = os.environ
    v = e.get('SEG_FORK_SIGNAL')
    if v == '0':
        return False
    if v == '1':
        return True
    if e.get('AMD_SERIALIZE_KERNEL', '0') not in ('', '0') or e.get('HIP_LAUNCH_BLOCKING', '0') not in ('', '0') or e.get('AMD_LOG_LEVEL', '0') not in ('', '0'):
        return False
    for k, val in e.items():
        if val in ('', '0'):
            continue
        if k.startswith(('ROCPROF', 'ROCP_', 'HSA_TOOLS', 'ROCTRACER', 'ROCPROFILER')):
            return False
    pre = e.get('LD_PRELOAD', '')
    if any(t in pre for t in ('rocprof', 'roctracer', 'rocprofiler', 'omnitrace', 'rocsys')):
        return False
    return True


_SIGNAL_ON = _signals_allowed()


def _signal_state(main):
    key = (main.device.index, main.cuda_stream)
    st = _SIGNALS.get(key)
    if st is None:
        hip = L.hip_runtime()
        st = _SIGNALS[key] = {'flag': torch.zeros(16, dtype=torch.int32, device=main.device), 'n': 0, 'hip': hip}
        # hipDeviceAttributeCanUseStreamWaitValue: a device / driver without stream memory operations gets event forks
        can = C.c_int(0)
        try:
            rc = hip.hipDeviceGetAttribute(C.byref(can), L.HIP_DEVICE_ATTRIBUTE_CAN_USE_STREAM_WAIT_VALUE, main.device.index or 0)
        except Exception:                              # noqa
            rc = 1
        if rc != 0 or can.value == 0:
            _SIGNALS['off'] = True
        torch.cuda.synchronize(main.device)
    return st


_CFG_OV = None


def _cfg_overrides():
    global _CFG_OV
    if _CFG_OV is None:
        _CFG_OV = {}
        for kv in os.environ.get('SEG_CFG_OVERRIDE', '').split(','):
            if '=' in kv:
                k, v = kv.split('=')
                _CFG_OV[k.strip()] = int(v)
    return _CFG_OV


def _tail_layers(default=()):
    """Layers whose filter gradients run after the critical stream has finished (the models name the last ones of their backward
    pass); SEG_WGRAD_TAIL="a,b" overrides, "-" = none."""
    v = os.environ.get('SEG_WGRAD_TAIL')
    if v is None:
        return set(default)
    return set(x for x in v.split(',') if x and x != '-')


def _touches(args, lo_ptr, hi_ptr):
    """does any launch argument (a raw pointer, or a pointer field of a descriptor passed by reference) point into
    [lo_ptr, hi_ptr)?"""
    for a in args:
        if isinstance(a, int):
            if lo_ptr <= a < hi_ptr:
                return True
            continue
        d = getattr(a, '_obj', None)
        if d is None:
            continue
        for fld in ('dw', 'db', 'dst', 'dst1'):
            v = getattr(d, fld, None)
            v = getattr(v, 'ptr', v)                      # (a seg_view field: its base pointer)
            if isinstance(v, int) and lo_ptr <= v < hi_ptr:
                return True
    return False


def mark_bucket_main_writers(plan, lo_ptr, hi_ptr):
    """Data-parallel plans: a bucket's all-reduce is issued from a side stream that has waited for the other side streams, never for
    the main stream.  Gradients written ON the main stream (bias gradients of the transposed convolutions, filter gradients forced
    onto stream 0, the loss word) are covered only if a side launch behind them forks from the main stream before the marker.
    This walks the plan once and sets marker['main_event'] where that does not hold: Plan.run's caller then records an event on the
    main stream for that bucket and the issuing stream waits for it (enforced, not assumed).  Conservative: any
    main-stream launch with a pointer into the arena counts as a writer."""
    uncovered = False
    n = 0
    for (name, fn, args), md in zip(plan.ops, plan.meta):
        if fn is None:
            if md.get('marker') == 'bucket':
                md['main_event'] = uncovered
                n += int(uncovered)
                uncovered = False
            continue
        tag = md.get('side', 0)
        if tag and tag != 'aux':
            uncovered = False          # a filter-gradient stream forks from (or is already ordered behind) everything on main so far
        elif not tag and _touches(args, lo_ptr, hi_ptr):
            uncovered = True
    return n


# SEG_PLAN_C=0: the per-launch Python walk of Plan.run (debugging; bitwise the same launches in the same order)
_PLAN_C = os.environ.get('SEG_PLAN_C', '1') != '0'


class CompiledPlan(object):
    """A recorded multi-stream walk of a Plan as seg_plan_op records (include/seg_hip.h): replayed by ONE call of seg_plan_run.
    Launch arguments are marshalled once into seg_arg arrays (Plan.rebind patches them in place); descriptors passed by reference
    stay the plan's own ctypes objects, so a descriptor field changed on the Python side is seen by the next replay."""

    def __init__(self, plan, rec, nslots, stream, main, side):
        lib = L.load()
        self.lib, self.plan, self.nslots = lib, plan, nslots
        streams, sidx = [stream], {None: 0, id(main): 0}
        for o_ in side:
            if id(o_) not in sidx:
                sidx[id(o_)] = len(streams)
                streams.append(o_.cuda_stream)
        self.streams = (C.c_void_p * len(streams))(*streams)
        self.n = len(rec)
        self.ops = (L.PlanOp * max(1, self.n))()
        self.args_of = {}                          # plan op index -> its seg_arg array (Plan.rebind)
        self.names = []
        self.markers = []                          # (position in ops, marker meta, side streams to hand to the callback)
        self._keep = []
        self.main = main
        for k, r in enumerate(rec):
            op = self.ops[k]
            if r[0] == 'L':
                _, i_, name, fn, args, st, slot = r
                at = L.SIGNATURES[fn.__name__][:-1]
                if len(at) != len(args):
                    raise L.SegError('%s/%s: %d arguments for %s' % (plan.name, name, len(args), fn.__name__))
                arr = (L.SegArg * max(1, len(args)))()
                for j, (t, a) in enumerate(zip(at, args)):
                    self._marshal(arr[j], t, a)
                fid = lib.seg_plan_fn_id(fn.__name__.encode())
                if fid < 0:
                    raise L.SegError('seg_plan_run has no entry point %s' % fn.__name__)
                op.kind, op.fn, op.stream, op.stream2, op.nargs = L.OP_LAUNCH, fid, sidx[None if st is None else id(st)], 0, len(args)
                op.signal_slot = slot or 0
                op.args = arr
                self.args_of[i_] = (arr, at)
                self._keep.append(arr)
                self.names.append(name)
            elif r[0] == 'F':
                op.kind, op.stream, op.stream2 = L.OP_EVENT_FORK, sidx[id(r[1])], sidx[id(r[2])]
                self.names.append('fork')
            elif r[0] == 'M':
                # a data-parallel marker: the replay stops in front of it, the caller's callback runs, the replay goes on (the op
                # itself is a no-op wait on slot 0 of the main stream that is never executed: stretches exclude it)
                op.kind, op.stream, op.signal_slot = L.OP_WAIT_VALUE, 0, 0
                self.markers.append((k, r[1], r[2]))
                self.names.append('marker')
            else:
                op.kind, op.stream, op.signal_slot = L.OP_WAIT_VALUE, sidx[id(r[1])], r[2]
                self.names.append('wait')
        self.failed = C.c_int32(-1)

    @staticmethod
    def _marshal(slot, t, a):
        if t is C.c_float:
            slot.f32 = float(a)
        elif t in (C.c_int32, C.c_int64, C.c_uint64):
            v = int(a)
            slot.i = v - (1 << 64) if v >= (1 << 63) else v           # (uint64 values travel as their two's-complement bits)
        else:                                          # a pointer: raw address, None, byref(ctypes object) or a ctypes array / pointer
            if a is None:
                slot.p = None
            elif isinstance(a, int):
                slot.p = a
            else:
                o_ = getattr(a, '_obj', None)
                slot.p = C.addressof(o_) if o_ is not None else C.cast(a, C.c_void_p).value

    def patch(self, i, j, value):
        ent = self.args_of.get(i)
        if ent is not None:
            self._marshal(ent[0][j], ent[1][j], value)

    def run(self, on_marker=None):
        sig = _signal_state(self.main) if self.nslots else None
        base, flag = 0, None
        if sig is not None:
            base = sig['n']
            sig['n'] += self.nslots
            flag = sig['flag'].data_ptr()
        lo = 0
        for pos, md, streams in self.markers + [(self.n, None, None)]:
            if pos > lo:
                rc = self.lib.seg_plan_run(C.cast(C.byref(self.ops, lo * C.sizeof(L.PlanOp)), C.POINTER(L.PlanOp)), pos - lo, self.streams, len(self.streams),
                                           flag, base & 0x7fffffff, C.byref(self.failed))
                if rc != 0:
                    k = lo + self.failed.value
                    L.check(rc, '%s/%s' % (self.plan.name, self.names[k] if 0 <= k < len(self.names) else '?'))
            if md is not None:
                on_marker(md, streams)
            lo = pos + 1

    def __del__(self):
        try:
            self.lib.seg_plan_destroy_events(self.ops, self.n)
        except Exception:                              # noqa
            pass


class Plan(object):
    """Ordered list of C-ABI launches.  Every entry is (name, fn, args-without-stream)."""

    def __init__(self, name=''):
        self.name = name
        self.ops = []
        self.keep = []          # ctypes objects that must outlive the plan
        self.meta = []          # per-op {'flops':, 'bytes':, 'desc':} for the roofline report
        self.flops = 0

    def add(self, name, fn, *args, **meta):
        ov = _cfg_overrides()
        if ov and name in ov and meta.get('desc') is not None:
            meta['desc'].cfg = ov[name]                   # (tile-choice experiments: SEG_CFG_OVERRIDE="conv5_1/dx0=1,conv5_2=11")
        self.ops.append((name, fn, args))
        self.meta.append(meta)

    def run(self, stream, side=None, skip=(), flavor='per_layer', join_into=None, on_marker=None):
        """Launches every op in order.  Ops tagged side=1 (filter gradients: nothing on the backward critical path
        consumes them) go round-robin onto the `side` torch streams; each first waits for an event recorded on the
        main stream at its program position (so everything launched before it is its dependency) and the side
        streams are joined back into the main stream at the end -- under hipGraph capture this becomes a fork/join."""
        sp = C.c_void_p(stream)
        if _DEBUG_SKIP:
            skip = tuple(skip) + _DEBUG_SKIP       # (timing experiments: what would the step cost without these launches?  results are garbage)
        if not side:
            dbg = os.environ.get('SEG_DEBUG_SYNC')        # name every launch on stderr and synchronise after it (fault hunting)
            for i_, (name, fn, args) in enumerate(self.ops):
                if fn is None and on_marker is not None and self.meta[i_].get('marker'):
                    on_marker(self.meta[i_], [])
                    continue
                if fn is None or name in skip or self.meta[i_].get('flavor', flavor) != flavor:
                    continue
                d_ = self.meta[i_].get('desc')
                if isinstance(d_, L.ConvDesc):
                    d_.signal = None                      # (a multi-stream run of this plan may have left a signal request)
                if dbg:
                    import sys
                    sys.stderr.write('[launch] %s/%s\n' % (self.name, name)); sys.stderr.flush()
                rc = fn(*args, sp)
                if rc != 0:
                    L.check(rc, '%s/%s' % (self.name, name))
                if dbg:
                    torch.cuda.synchronize()
            return
        main = torch.cuda.current_stream()
        capturing = torch.cuda.is_current_stream_capturing()
        if _PLAN_C and not capturing and join_into is None and (on_marker is None or getattr(on_marker, 'compiled_ok', False)):
            # the whole walk below was recorded once (forks, signal forks, held launches and all) and is replayed by ONE call into
            # the library (seg_plan_run): ~130 interpreter iterations + ctypes calls per train step become one -- one call per
            # stretch between two data-parallel markers when there is a marker callback
            return self._run_compiled(stream, main, side, tuple(skip), flavor, on_marker)
        return self._walk(stream, main, side, skip, flavor, join_into, on_marker, None)

    def _walk(self, stream, main, side, skip, flavor, join_into, on_marker, rec):
        """The multi-stream walk of run().  rec is None: launches, forks and waits are issued; rec is a list: they are RECORDED as
        ('L', op index, name, fn, args, stream or None, signal slot) / ('F', src stream, dst stream) / ('W', stream, slot) with
        signal slots 1 .. K instead of absolute signal numbers (CompiledPlan adds a per-run base) and nothing touches the GPU."""
        sp = C.c_void_p(stream)
        recording = rec is not None
        if recording:
            def fork(a_, b_):
                rec.append(('F', a_, b_))
            if on_marker is not None:                  # (markers become entries of the recording: CompiledPlan.run calls back between two replays)
                def on_marker(md_, streams_):          # noqa: F811
                    rec.append(('M', md_, list(streams_)))
        else:
            fork = _fork
        used = {}
        aux = side[-1]                   # dedicated stream for side='aux' ops (weight re-pack), joined by a 'join_aux' marker
        aux_used = False
        main_epoch, forked_at = 0, {}             # launches on the main stream so far; per side stream: the count at its last fork
        capturing = (not recording) and torch.cuda.is_current_stream_capturing()
        dirty = set()                             # side streams with launches the main stream has not waited for yet
        sig = _signal_state(main) if (_SIGNAL_ON and not capturing and not _SIGNALS.get('off')) else None
        nslot = [0]                               # (recording) signal slots handed out so far
        sig_at = {}                               # op index of a main-stream convolution -> the number it has to announce
        held = {}                                 # side stream id -> (op index it waits for, [(name, fn, args)] to launch behind it)
        by_id = {id(o_): o_ for o_ in side}

        def launch(i_, name_, fn_, args_, st_, slot_=None):
            """st_: a side stream, or None for the main stream"""
            if recording:
                rec.append(('L', i_, name_, fn_, args_, st_, slot_))
                return 0
            return fn_(*args_, sp if st_ is None else C.c_void_p(st_.cuda_stream))

        def next_main_conv(i0):
            """the next op after i0 that will launch on the main stream, if it is a convolution (it can carry a signal)"""
            for j in range(i0 + 1, len(self.ops)):
                nm, f_, _ = self.ops[j]
                md = self.meta[j]
                if nm in skip or md.get('flavor', flavor) != flavor or md.get('side', 0):
                    continue
                if f_ is None:
                    return None                    # a join comes first: the launch must be on its stream before the main stream waits for it
                return j if isinstance(md.get('desc'), L.ConvDesc) else None
            return None

        def join(st_):
            # (eagerly a stream that has launched nothing since the main stream last waited for it needs no second wait)
            nonlocal main_epoch
            assert id(st_) not in held, 'a signal fork was still pending at a join (the op it waits for lies behind the join)'
            if capturing or id(st_) in dirty:
                fork(st_, main)
                dirty.discard(id(st_))
                # the main stream now holds st_'s work: a side stream that forked at the current epoch has NOT seen it, so the
                # fork elision below must not treat it as up to date (a later op on it may consume st_'s output)
                main_epoch += 1

        pending_markers = []                      # data-parallel bucket markers waiting for held (signal-forked) side launches

        def fire_markers():
            # a marker's callback (the bucket's all-reduce, issued from a communication stream) may only see the side streams once
            # every side launch in front of the marker is really enqueued: launches held back for a signal fork are not yet
            if held or not pending_markers:
                return
            for md in pending_markers:
                on_marker(md, [by_id[k] for k in dirty if k in by_id])
            del pending_markers[:]

        for i, (name, fn, args) in enumerate(self.ops):
            tag = self.meta[i].get('side', 0)
            if name in skip or self.meta[i].get('flavor', flavor) != flavor:
                continue
            if fn is None and self.meta[i].get('marker'):
                if on_marker is not None:
                    if self.meta[i]['marker'] == 'wait':
                        assert not held
                        fire_markers()
                        on_marker(self.meta[i], [])
                    else:
                        pending_markers.append(self.meta[i])
                        fire_markers()
                continue
            if fn is None and name == 'join_all':      # marker: the main stream waits for every side stream used so far
                probe = getattr(self, 'probe', None) if not recording else None      # (tools/tail_probe.py: how long does the main stream idle here?)
                if probe is not None:
                    e_ = torch.cuda.Event(enable_timing=True); e_.record(main); probe.append(e_)
                for o_ in used.values():
                    join(o_)
                if aux_used:
                    join(aux)
                if probe is not None:
                    e_ = torch.cuda.Event(enable_timing=True); e_.record(main); probe.append(e_)
                continue
            if fn is None and name == 'join_wgrad':    # marker: ... for the filter-gradient streams only (the aux stream keeps running)
                for o_ in used.values():
                    join(o_)
                continue
            if fn is None:               # marker: make the main stream wait for the aux stream
                if aux_used:
                    join(aux)
                    aux_used = False
                continue
            rc = 0
            if tag == 'aux':
                if capturing or forked_at.get(id(aux)) != main_epoch:
                    fork(main, aux)
                    forked_at[id(aux)] = main_epoch
                aux_used = True
                dirty.add(id(aux))
                rc = launch(i, name, fn, args, aux)
            elif tag:
                # side stream ids are assigned when the plan is built (Net._add_wgrad alternates 1, 2): a filter
                # gradient and the reduction of its slabs are then in order on ONE stream
                st = side[(tag - 1) % (len(side) - 1)] if len(side) > 1 else side[0]
                used[id(st)] = st
                # A fork (event recorded on the main stream + wait on the side stream) is only needed if something was launched
                # on the main stream since this side stream last forked from it: a launch right behind its producer on the SAME
                # side stream (the slab reduction behind its filter gradient) is already ordered.  Each redundant fork costs ~20 us
                # of step time when launched eagerly (eleven of them: 1.44 against 1.19 ms), far more than the launch it guards.
                # Under stream capture every edge is kept: the captured graph's split into streams depends on them (1.38 against
                # 1.25 ms without).
                if id(st) in held:
                    held[id(st)][1].append((i, name, fn, args))      # stays in order behind the held launches of its stream
                    continue
                if capturing or forked_at.get(id(st)) != main_epoch:
                    j = next_main_conv(i) if sig is not None else None
                    if j is not None:
                        # signal fork: this launch (and what follows it on its stream) is issued right AFTER main-stream op j, which
                        # announces its start -- the waiter is always enqueued behind its signaller, whatever hardware queue
                        # the two streams share, so the wait cannot block the kernel it waits for
                        if j not in sig_at:
                            if recording:
                                nslot[0] += 1
                                sig_at[j] = nslot[0]
                            else:
                                sig['n'] += 1
                                sig_at[j] = sig['n'] & 0x7fffffff
                        held[id(st)] = (j, [(i, name, fn, args)])
                        forked_at[id(st)] = main_epoch
                        dirty.add(id(st))
                        continue
                    fork(main, st)
                    forked_at[id(st)] = main_epoch
                dirty.add(id(st))
                rc = launch(i, name, fn, args, st)
            else:
                main_epoch += 1
                d_ = self.meta[i].get('desc')
                v = None
                if isinstance(d_, L.ConvDesc):           # (always rewritten: the descriptor is reused by every run of the plan)
                    v = sig_at.pop(i, None)
                    if not recording:
                        d_.signal = sig['flag'].data_ptr() if v is not None else None
                        d_.signal_value = v or 0
                rc = launch(i, name, fn, args, None, v)
                if v is not None and rc == 0:
                    for sid_, (j, lst) in list(held.items()):
                        if j != i:
                            continue
                        st_ = by_id[sid_]
                        if recording:
                            rec.append(('W', st_, v))
                        elif sig['hip'].hipStreamWaitValue32(C.c_void_p(st_.cuda_stream), C.c_void_p(sig['flag'].data_ptr()), v, 0, 0xffffffff) != 0:   # 0 = >=
                            # (a runtime without stream memory operations: an event recorded now orders the held launches just as
                            # well -- op i is already on the main stream -- and later forks of this process use events)
                            _SIGNALS['off'] = True
                            fork(main, st_)
                        for i2_, nm_, f_, a_ in lst:
                            r_ = launch(i2_, nm_, f_, a_, st_)
                            if r_ != 0:
                                L.check(r_, '%s/%s' % (self.name, nm_))
                        del held[sid_]
                    fire_markers()
            if rc != 0:
                L.check(rc, '%s/%s' % (self.name, name))
        if aux_used:
            used[id(aux)] = aux
        if pending_markers and on_marker is not None:
            assert not held
            fire_markers()
        if join_into is not None and not capturing:
            # data-parallel segment: whoever consumes this segment's side-stream results (the bucket's all-reduce, issued from the
            # communication stream `join_into`) waits for them; the main stream does NOT -- it goes straight on to the next segment
            # (each drain of the side streams into the main stream cost 17-30 us of step time)
            assert not held, 'a signal fork was still pending at the end of a segment'
            for st in used.values():
                ev = torch.cuda.Event(); ev.record(st); join_into.wait_event(ev)
            return nslot[0]
        for st in used.values():
            join(st)
        return nslot[0]

    def _run_compiled(self, stream, main, side, skip, flavor, on_marker=None):
        cache = self.__dict__.setdefault('_compiled', {})
        sig_on = bool(_SIGNAL_ON and not _SIGNALS.get('off'))
        key = (stream, main.device.index, tuple(o_.cuda_stream for o_ in side), skip, flavor, sig_on, len(self.ops), on_marker is not None)
        cp = cache.get(key)
        if cp is None:
            rec = []
            nslots = self._walk(stream, main, side, skip, flavor, None, on_marker, rec)
            cp = cache[key] = CompiledPlan(self, rec, nslots, stream, main, side)
        cp.run(on_marker)

    def rebind(self, mapping):
        """Replaces raw device pointers among the launch arguments ({old: new}; the network inputs when a device-resident
        dataset hands over its own buffers).  The argument sites of a pointer are found once and remembered, so the
        per-step re-pointing touches a handful of launches (the data-parallel step is host-bound: a full scan of two
        plans per step made it 20 % slower and erratic).  Returns the number of arguments changed."""
        sites = self.__dict__.setdefault('_ptr_sites', {})
        n = 0
        for old, new in mapping.items():
            where = sites.pop(old, None)
            if where is None:
                where = [(i, j) for i, (_, _, args) in enumerate(self.ops) for j, a in enumerate(args) if isinstance(a, int) and a == old]
                # ... and the descriptors that carry the pointer inside (the first layer's filter gradient reads the input image
                # through seg_wgrad_desc.im2col_x)
                seen = set()
                for _, _, args in self.ops:
                    for a in args:
                        d_ = getattr(a, '_obj', None)
                        if isinstance(d_, L.WgradDesc) and id(d_) not in seen and d_.im2col_x == old:
                            seen.add(id(d_)); where.append(('desc', d_))
            for site in where:
                if site[0] == 'desc':
                    site[1].im2col_x = new; site[1].src0.ptr = new
                    n += 1
                    continue
                i, j = site
                name, fn, args = self.ops[i]
                self.ops[i] = (name, fn, args[:j] + (new,) + args[j + 1:])
                for cp in self.__dict__.get('_compiled', {}).values():
                    cp.patch(i, j, new)
                n += 1
            sites[new] = where
        return n

    def __len__(self):
        return len(self.ops)

    def extend(self, other):
        self.__dict__.pop('_ptr_sites', None)
        self.__dict__.pop('_compiled', None)
        self.ops += other.ops; self.meta += other.meta; self.keep += other.keep; self.flops += other.flops

    def kernel_name(self, i):
        """Name of the kernel instance op i launches (as rocprofv3 prints it, without the namespace)."""
        name, fn, args = self.ops[i]
        lib = L.load()
        d = self.meta[i].get('desc')
        if d is not None:
            buf = C.create_string_buffer(160)
            q = lib.seg_conv2d_kernel_name if isinstance(d, L.ConvDesc) else lib.seg_conv2d_wgrad_kernel_name
            L.check(q(C.byref(d), buf, 160), 'kernel_name')
            return buf.value.decode()
        return self.meta[i].get('kernel', fn.__name__ if fn is not None else 'marker')

    def run_profiled(self, stream, torch_mod, side=None, flavor='per_layer'):
        """Like run(), with a HIP event recorded before and after every launch ON THE STREAM THE KERNEL IS LAUNCHED ON
        (main or side), so that the per-kernel durations include the same cross-stream overlap as the timed region.
        Returns [(op name, kernel name, ms, flops, algorithmic HBM bytes)]."""
        sp = C.c_void_p(stream)
        main = torch_mod.cuda.current_stream()
        E_ = lambda: torch_mod.cuda.Event(enable_timing=True)
        recs = []
        lib_ = L.load()
        launched = {}                              # op index -> the kernel its launch site named (seg_last_kernel_name)
        used, aux_used = {}, False
        aux = side[-1] if side else None
        for i, (name, fn, args) in enumerate(self.ops):
            tag = self.meta[i].get('side', 0) if side else 0
            if self.meta[i].get('flavor', flavor) != flavor:
                continue
            if fn is None and name in ('join_all', 'join_wgrad'):
                for o_ in used.values():
                    ev = torch_mod.cuda.Event(); ev.record(o_); main.wait_event(ev)
                if side and aux_used and name == 'join_all':
                    ev = torch_mod.cuda.Event(); ev.record(aux); main.wait_event(ev)
                continue
            if fn is None:
                if side and aux_used:
                    ev = torch_mod.cuda.Event(); ev.record(aux); main.wait_event(ev); aux_used = False
                continue
            if tag == 'aux':
                st = aux; aux_used = True
            elif tag:
                st = side[(tag - 1) % (len(side) - 1)] if len(side) > 1 else side[0]
                used[id(st)] = st
            else:
                st = main
            if st is not main:
                ev = torch_mod.cuda.Event(); ev.record(main); st.wait_event(ev)
            e0, e1 = E_(), E_()
            e0.record(st)
            d_ = self.meta[i].get('desc')
            if isinstance(d_, L.ConvDesc):
                d_.signal = None                          # (forks are events here)
            lib_.seg_last_kernel_name()                 # (clears: the next answer is the first kernel of THIS launch)
            rc = fn(*args, C.c_void_p(st.cuda_stream) if st is not main else sp)
            if rc != 0:
                L.check(rc, '%s/%s' % (self.name, name))
            launched[i] = lib_.seg_last_kernel_name().decode()
            e1.record(st)
            recs.append((i, e0, e1))
        if side and aux_used:
            used[id(aux)] = aux
        for st in used.values():
            ev = torch_mod.cuda.Event(); ev.record(st); main.wait_event(ev)
        torch_mod.cuda.synchronize()
        # kernel names: what the launch itself reported (first layer, pools ...: the C side picks the instance), the descriptor query
        # for the templated convolution / filter-gradient dispatchers (they launch through a function-pointer variable)
        return [(self.ops[i][0], launched.get(i) or self.kernel_name(i), e0.elapsed_time(e1), self.meta[i].get('flops', 0), self.meta[i].get('bytes', 0))
                for i, e0, e1 in recs]


class Net(object):
    """Emits launches for one network instance (fixed batch / spatial size / dtype)."""

    def __init__(self, store, B, dtype, device):
        self.lib = L.load()
        self.store, self.B, self.dtype, self.device = store, B, dtype, device
        self.acts = []
        # slab reductions of the filter gradients: one launch per layer right behind its wgrad (default: best under
        # hipGraph replay), or batched into one launch per side stream and backward segment (models turn this on for
        # eager execution, where it saves a quarter of the launches; SEG_BATCH_REDUCE=0/1 overrides)
        # None: emit both flavours of the slab reduction (Plan.run chooses); True / False: only the batched / per-layer one
        self.batch_reduce = None if 'SEG_BATCH_REDUCE' not in os.environ else (os.environ['SEG_BATCH_REDUCE'] != '0')
        self._pending_reduce = {}
        self.n_wgrad_streams = 2
        self._wg_rr = 0
        self.side_enabled = True
        self.pool_fused = False
        # tile-ticket words of the persistent convolution launches (seg_conv_desc.sched): two zeroed int32 per launch site
        self._sched = torch.zeros(2 * 1024, dtype=torch.int32, device=device)
        self._sched_n = 0

    @property
    def es(self):
        """bytes per activation element in HBM"""
        return 4 if self.dtype == L.SEG_F32 else 2

    @staticmethod
    def _thin(*acts):
        return any(getattr(a, 'thin', False) for a in acts if a is not None)

    def _splitk(self, d, plan, ksplit=None):
        """Asks the library whether this convolution launch should share its K loop among several workgroups per output tile
        (seg_conv2d_splitk_plan: the deep, small-map layers) and gives the descriptor the workspace and tickets it then needs.
        ksplit: None / 1 = off (no automatic split: measured slower), n = ask for n parts (tests)."""
        d.ksplit = 0 if ksplit is None else ksplit
        d.splitk_ws = None; d.splitk_tickets = None
        if ksplit is None or ksplit == 1:
            d.ksplit = 0                      # (off by default: measured slower, see seg_conv2d_splitk_plan)
            return
        ks, nbytes, nt = C.c_int32(0), C.c_int64(0), C.c_int32(0)
        L.check(self.lib.seg_conv2d_splitk_plan(C.byref(d), C.byref(ks), C.byref(nbytes), C.byref(nt)), 'splitk_plan')
        if ks.value <= 1:
            d.ksplit = 0
            return
        ws = torch.empty(nbytes.value // 4, dtype=torch.float32, device=self.device)
        tk = torch.zeros(nt.value, dtype=torch.int32, device=self.device)
        d.ksplit = ks.value; d.splitk_ws = ws.data_ptr(); d.splitk_tickets = tk.data_ptr()
        self.ws_bytes = getattr(self, 'ws_bytes', 0) + nbytes.value
        plan.keep += [ws, tk]

    def sched_slot(self):
        """device address of a fresh pair of ticket words (None once the pool is used up: the kernel then splits statically)"""
        if 2 * self._sched_n + 2 > self._sched.numel():
            return None
        self._sched_n += 1
        return self._sched.data_ptr() + 8 * (self._sched_n - 1)

    def act(self, H, W, C, f32=False, name='', thin=False):
        a = Act(self.B, H, W, C, self.dtype, self.device, f32=f32, name=name, thin=thin)
        self.acts.append(a)
        return a

    def act_bytes(self):
        return sum(a.nbytes() for a in self.acts)

    # ---------------- forward ----------------
    def first_fwd(self, plan, layer, x_f32, H, W, dst, pool=None):
        """First layer.  With `pool` (the Act of the 2x2 max-pool that consumes it) the bf16 path writes the pooled map
        in the same pass (seg_conv_first_pool_fwd) and True is returned; otherwise the caller emits pool_fwd itself."""
        Ho, Wo = H + 2 * layer.pad - 2, W + 2 * layer.pad - 2
        dv = dst.view()
        plan.keep.append(dv)
        fl = 2 * self.B * Ho * Wo * 9 * layer.cin * layer.cout
        plan.flops += fl
        by = self.B * (H * W * layer.cin * 4 + Ho * Wo * layer.cout * self.es + (pool.H * pool.W * layer.cout * self.es if pool is not None else 0))
        mfma = self.dtype == L.SEG_BF16 and layer.cin <= 3 and layer.cout <= 64
        kern = 'conv_first_mfma_kernel' if mfma else 'conv_first_fwd_kernel'
        if pool is not None and mfma and os.environ.get('SEG_FUSE_POOL1', '1') != '0':
            pv = pool.view()
            plan.keep.append(pv)
            plan.add(layer.name + '+pool', self.lib.seg_conv_first_pool_fwd, x_f32.data_ptr(), self.B, H, W, layer.cin,
                     self.store.p_ptr(layer.w_off), self.store.p_ptr(layer.b_off), layer.cout, layer.pad, C.byref(dv), Ho, Wo,
                     1 if layer.relu else 0, C.byref(pv), pool.H, pool.W, self.dtype, kernel=kern, flops=fl, bytes=by)
            return True
        plan.add(layer.name, self.lib.seg_conv_first_fwd, x_f32.data_ptr(), self.B, H, W, layer.cin,
                 self.store.p_ptr(layer.w_off), self.store.p_ptr(layer.b_off), layer.cout, layer.pad, C.byref(dv), Ho, Wo,
                 1 if layer.relu else 0, self.dtype, kernel=kern, flops=fl, bytes=by)
        return False

    def first_gen_fwd(self, plan, layer, x_f32, H, W, cin, KH, KW, stride, pad_t, pad_l, dst, bn_st=None):
        """A KH x KW / stride filter on the raw float image in one pass (seg_conv_first_gen: bf16, cin <= 3, cout <= 64).  `layer`
        holds the filter as [KH*KW*cin][cout] = HWIO (the DeconvModel keeps conv1_0 as a 1x1 layer over its im2col).
        bn_st (the state of the batch norm that consumes dst; 5x5/s2): the launch leaves that batch norm's statistics rows in its
        workspace and the row count is returned -- pass it to bn_fwd(rows=...); else returns 0."""
        dv = dst.view()
        plan.keep.append(dv)
        fl = 2 * self.B * dst.H * dst.W * KH * KW * cin * layer.cout
        plan.flops += fl
        by = self.B * (H * W * cin * 4 + dst.H * dst.W * layer.cout * self.es)
        if bn_st is not None and (KH, KW, stride) == (5, 5, 2) and os.environ.get('SEG_BN_FUSE_STATS', '1') != '0':
            rows = int(self.lib.seg_conv_first_gen_rows(self.B, dst.H, dst.W, layer.cout))
            plan.keep.append(bn_st)
            plan.add(layer.name, self.lib.seg_conv_first_gen_bn, x_f32.data_ptr(), self.B, H, W, cin, self.store.p_ptr(layer.w_off),
                     self.store.p_ptr(layer.b_off), layer.cout, KH, KW, stride, pad_t, pad_l, C.byref(dv), dst.H, dst.W, 1 if layer.relu else 0,
                     bn_st['ws'].data_ptr(), dst.Cp, self.dtype, kernel='conv_first_gen_kernel', flops=fl, bytes=by)
            return rows
        plan.add(layer.name, self.lib.seg_conv_first_gen, x_f32.data_ptr(), self.B, H, W, cin, self.store.p_ptr(layer.w_off),
                 self.store.p_ptr(layer.b_off), layer.cout, KH, KW, stride, pad_t, pad_l, C.byref(dv), dst.H, dst.W, 1 if layer.relu else 0,
                 self.dtype, kernel='conv_first_gen_kernel', flops=fl, bytes=by)
        return 0

    def first_gen_bwd(self, plan, layer, x_f32, H, W, cin, KH, KW, stride, pad_t, pad_l, dz, sid=None):
        """Filter + bias gradient of first_gen_fwd's layer straight from the image (seg_conv_first_gen_wgrad: 5x5 / stride 2, bf16), on a
        filter-gradient stream; no im2col tensor."""
        zv = dz.view()
        nbytes = int(self.lib.seg_conv_first_gen_wgrad_ws_bytes(layer.cout))
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
        plan.keep += [zv, ws]
        fl = 2 * self.B * dz.H * dz.W * KH * KW * cin * layer.cout
        if sid is None:
            sid = self._pick_wgrad_stream(15.0 + fl / 2e8 + 10.0)
        if not self.side_enabled:
            sid = 0
        plan.add(layer.name + '/dw', self.lib.seg_conv_first_gen_wgrad, x_f32.data_ptr(), self.B, H, W, cin, C.byref(zv), dz.H, dz.W, layer.cout,
                 KH, KW, stride, pad_t, pad_l, self.store.g_ptr(layer.w_off), self.store.g_ptr(layer.b_off) if layer.nbias else None,
                 ws.data_ptr(), nbytes, self.dtype, kernel='conv_first_gen_wgrad_kernel', flops=fl,
                 bytes=self.B * (H * W * cin * 4 + dz.H * dz.W * layer.cout * self.es), side=sid)
        plan.flops += fl

    def conv_fwd(self, plan, layer, srcs, Hi, Wi, dst, dst_off=(0, 0), out_f32=False, cfg=0, pool=None, ksplit=None, side=0, src_bn=None):
        """srcs: list of (Act, oy, ox) (1 or 2 concat segments).  pool: Act of the 2x2 max-pool that consumes dst; when the
        layer's tile can carry it the pooled map is written by the same launch and `self.pool_fused` is set (else the
        caller emits pool_fwd)."""
        self.pool_fused = False
        k, pad = layer.k, layer.pad
        Ho, Wo = Hi + 2 * pad - k + 1, Wi + 2 * pad - k + 1
        if (k == 3 and len(srcs) == 1 and srcs[0][0].thin and dst.thin and pool is None and dst_off == (0, 0) and srcs[0][1:] == (0, 0)
                and layer.cin <= 8 and layer.cout <= 8 and os.environ.get('SEG_THIN_VALU', '1') != '0'):
            # thin -> thin: the vector-ALU kernel (seg_thin_conv3x3) instead of 32 x 32 padded channels of MFMA work
            sv, dv = srcs[0][0].view(), dst.view()
            plan.keep += [sv, dv]
            fl = 2 * self.B * Ho * Wo * 9 * layer.cin * layer.cout
            by = self.B * (Hi * Wi + Ho * Wo) * 8 * (self.es if not out_f32 else (self.es + 4) // 2)
            if src_bn is not None:
                # the source is the pre-batch-norm activation, normalised on load (src_bn = (state, layer) of that batch norm, whose
                # statistics bn_stats() has put into state['stats']): bit-identical to reading the normalised tensor
                st_, bl_ = src_bn
                plan.keep.append(st_)
                plan.add(layer.name, self.lib.seg_thin_conv3x3_bn, C.byref(sv), self.B, Hi, Wi, self.store.p_ptr(layer.w_off), self.store.p_ptr(layer.b_off),
                         layer.cin, layer.cout, pad, 1 if layer.relu else 0, C.byref(dv), Ho, Wo, 1 if out_f32 else 0, st_['stats'].data_ptr(),
                         self.store.p_ptr(bl_.w_off), self.dtype, kernel='thin_conv3x3_kernel', flops=fl, bytes=by)
                plan.flops += fl
                return Ho, Wo
            plan.add(layer.name, self.lib.seg_thin_conv3x3, C.byref(sv), self.B, Hi, Wi, self.store.p_ptr(layer.w_off), self.store.p_ptr(layer.b_off),
                     layer.cin, layer.cout, pad, 1 if layer.relu else 0, 0, None, C.byref(dv), Ho, Wo, 1 if out_f32 else 0, self.dtype,
                     kernel='thin_conv3x3_kernel', flops=fl, bytes=by)
            plan.flops += fl
            return Ho, Wo
        if src_bn is not None:
            raise L.SegError('conv_fwd(src_bn=...): only the thin 3x3 vector-ALU route normalises on load')
        d = L.ConvDesc()
        d.src0 = srcs[0][0].view_wide(srcs[0][1], srcs[0][2])
        d.src1 = srcs[1][0].view_wide(srcs[1][1], srcs[1][2]) if len(srcs) > 1 else L.null_view()
        d.thin_src = 1 if self._thin(*[s_[0] for s_ in srcs]) else 0
        assert not d.thin_src or all(s_[0].thin for s_ in srcs), 'thin and wide sources cannot be concatenated'
        d.B, d.Hi, d.Wi = self.B, Hi, Wi
        d.KH = d.KW = k; d.stride = 1; d.pad_t = d.pad_l = pad
        d.Ho, d.Wo = Ho, Wo
        d.w_packed = self.store.packed_ptr(layer.pk_fwd)
        d.n_total = layer.cout_p; d.n_off = 0; d.n_count = layer.cout_p
        d.bias = self.store.p_ptr(layer.b_off); d.bias_n = layer.cout
        d.dst = dst.view(dst_off[0], dst_off[1])
        d.n_store = 8 if dst.thin else 0
        d.up2 = 0; d.up_cout = 0; d.mask = L.null_view()
        d.relu = 1 if layer.relu else 0
        d.out_f32 = 1 if out_f32 else 0
        d.dtype = self.dtype; d.cfg = cfg
        d.sched = self.sched_slot()
        name = layer.name
        if pool is not None and self.dtype == L.SEG_BF16 and dst_off == (0, 0) and os.environ.get('SEG_FUSE_POOL', '1') != '0':
            d.pool = pool.view(); d.pool_h, d.pool_w = pool.H, pool.W
            buf = C.create_string_buffer(160)
            if self.lib.seg_conv2d_kernel_name(C.byref(d), buf, 160) == 0:        # the library accepts the fused form for this layer
                self.pool_fused = True
                name += '+pool'
            else:
                d.pool = L.null_view(); d.pool_h = d.pool_w = 0
        self._splitk(d, plan, ksplit)
        plan.keep.append(d)
        fl = 2 * self.B * Ho * Wo * k * k * layer.cin * layer.cout
        # algorithmic HBM bytes: every tensor touched once (input window, output, filters; + the fused pooled map)
        by = (self.B * (Hi * Wi * layer.cin * self.es + Ho * Wo * layer.cout * (4 if out_f32 else self.es)) + k * k * layer.cin * layer.cout * self.es
              + (self.B * pool.H * pool.W * layer.cout * self.es if self.pool_fused else 0))
        if side and self.side_enabled:
            # a forward launch nothing on the critical stream waits for soon (the U-Net's conv1_2 only feeds the last skip): a
            # filter-gradient stream, idle during the forward pass; the consumer is emitted behind a join_wgrad
            plan.add(name, self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl, bytes=by, side=side)
        else:
            plan.add(name, self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl, bytes=by)
        plan.flops += fl
        return Ho, Wo

    def up_fwd(self, plan, layer, src, Hi, Wi, dst, cfg=0, bn_st=None):
        """bn_st (state of the batch norm consuming dst): where the launch can leave that batch norm's statistics rows (the thin
        vector-ALU kernel) it does and their count is returned for bn_fwd(rows=...); else 0."""
        if dst.thin and not src.thin and layer.cout <= 8 and os.environ.get('SEG_THIN_VALU', '1') != '0':
            # into a thin tensor: the vector-ALU kernel (a thread per input pixel writes its 2x2 block of <= 8-channel records)
            sv, dv = src.view(), dst.view()
            plan.keep += [sv, dv]
            fl = 2 * self.B * Hi * Wi * 4 * layer.cin * layer.cout
            if bn_st is not None and os.environ.get('SEG_BN_FUSE_STATS', '1') != '0':
                rows = int(self.lib.seg_thin_up2x2_rows(self.B, Hi, Wi))
                plan.keep.append(bn_st)
                plan.add(layer.name, self.lib.seg_thin_up2x2_bn, C.byref(sv), C.byref(dv), self.B, Hi, Wi, self.store.p_ptr(layer.w_off),
                         self.store.p_ptr(layer.b_off), layer.cin, layer.cout, 1 if layer.relu else 0, bn_st['ws'].data_ptr(), self.dtype,
                         kernel='thin_up2x2_kernel', flops=fl, bytes=self.B * Hi * Wi * (src.Cp + 4 * 8) * self.es)
                plan.flops += fl
                return rows
            plan.add(layer.name, self.lib.seg_thin_up2x2, C.byref(sv), C.byref(dv), self.B, Hi, Wi, self.store.p_ptr(layer.w_off), self.store.p_ptr(layer.b_off),
                     layer.cin, layer.cout, 1 if layer.relu else 0, 0, None, self.dtype, kernel='thin_up2x2_kernel', flops=fl,
                     bytes=self.B * Hi * Wi * (src.Cp + 4 * 8) * self.es)
            plan.flops += fl
            return 0
        d = L.ConvDesc()
        d.src0 = src.view(); d.src1 = L.null_view()
        d.B, d.Hi, d.Wi = self.B, Hi, Wi
        d.KH = d.KW = 1; d.stride = 1; d.pad_t = d.pad_l = 0
        d.Ho, d.Wo = Hi, Wi
        d.w_packed = self.store.packed_ptr(layer.pk_fwd)
        d.n_total = 4 * layer.cout_p; d.n_off = 0; d.n_count = 4 * layer.cout_p
        d.bias = self.store.p_ptr(layer.b_off); d.bias_n = layer.cout
        d.dst = dst.view(); d.up2 = 1; d.up_cout = layer.cout_p; d.mask = L.null_view()
        d.n_store = 8 if dst.thin else 0
        d.relu = 1 if layer.relu else 0; d.out_f32 = 0; d.dtype = self.dtype; d.cfg = cfg
        self._splitk(d, plan)
        plan.keep.append(d)
        fl = 2 * self.B * Hi * Wi * 4 * layer.cin * layer.cout
        by = self.B * Hi * Wi * (layer.cin + 4 * layer.cout) * self.es + 4 * layer.cin * layer.cout * self.es
        plan.add(layer.name, self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl, bytes=by)
        plan.flops += fl
        return 0

    def pool_fwd(self, plan, src, dst, Ho, Wo):
        sv, dv = src.view(), dst.view()
        plan.keep += [sv, dv]
        plan.add('pool', self.lib.seg_maxpool2x2_fwd, C.byref(sv), C.byref(dv), None, self.B, Ho, Wo, src.Cp, self.dtype, kernel='maxpool_fwd_kernel')

    # ---------------- backward ----------------
    def _wgrad_ws(self, w, plan, ksplit=0):
        """Asks the library for the K split / partial-slab workspace of this wgrad and allocates it."""
        ks, nbytes = C.c_int32(0), C.c_int64(0)
        w.ksplit = ksplit
        if w.target_wgs == 0 and self.side_enabled:
            w.target_wgs = _step_wgrad_wgs(getattr(self, 'input_pixels', None))     # (the models set Net.input_pixels = B * H * W)
        L.check(self.lib.seg_conv2d_wgrad_plan(C.byref(w), C.byref(ks), C.byref(nbytes)), 'wgrad_plan')
        ws = torch.empty(max(nbytes.value // 4, 4), dtype=torch.float32, device=self.device)
        w.ws = ws.data_ptr(); w.ws_bytes = nbytes.value; w.ksplit = ks.value
        self.ws_bytes = getattr(self, 'ws_bytes', 0) + nbytes.value
        plan.keep += [w, ws]

    def _pick_wgrad_stream(self, cost_us):
        """Side stream (1-based tag) of the next filter gradient: the one with the least work so far.  cost_us is the estimate
        the models' measured in-step durations fit (15 us + FLOPs at 200 TFLOP/s + 10 us for a slab reduction); plain alternation
        gave the conv*_2 layers (cin = cout, the heavier of each pair) to one stream: 567 against 478 us at C2.  Only the big steps
        (> 2.5 M input pixels: the side streams are their tail) gain from it -- 512^2 4.04 against 4.07 ms; at C2, bound by the
        critical stream, the balanced assignment measured 0.953 against 0.941 ms and the alternation stays."""
        n = self.n_wgrad_streams
        big = getattr(self, 'input_pixels', None) is not None and self.input_pixels > 2500000
        if n < 2 or not big:
            k = self._wg_rr % n
            self._wg_rr += 1
            return 1 + k
        load = self.__dict__.setdefault('_wg_load', [0.0] * n)
        k = min(range(n), key=lambda i: load[i])
        load[k] += cost_us
        return 1 + k

    def _add_wgrad(self, plan, name, w, fl, sid=None, own_reduce=False):
        """One plan op for the partial-sum kernel and, when K is split, a second one for the slab reduction (same side
        stream): two C-ABI calls so that each kernel is timed on its own."""
        if sid is None:
            sid = self._pick_wgrad_stream(15.0 + fl / 2e8 + (10.0 if w.ksplit > 1 else 0.0))
        if not self.side_enabled:
            sid = 0                                             # (experiments) keep it on the main stream
        if w.ksplit > 1:
            # the slab reduction exists in two flavours and Plan.run(flavor=...) picks one: 'per_layer' right behind the
            # partial sums (best under hipGraph replay) or 'batched' per side stream at the end of the segment (best for
            # eager launches: a quarter fewer launches)
            w.phase = 1
            w2 = L.WgradDesc.from_buffer_copy(w)
            w2.phase = 2
            plan.keep.append(w2)
            plan.add(name, self.lib.seg_conv2d_wgrad, C.byref(w), desc=w, flops=fl, side=sid, bytes=getattr(self, '_wg_bytes', 0))
            if own_reduce:
                # a plan nobody flushes (the adversary's): the reduction right behind the partial sums, whatever flavour the step runs
                plan.add(name + '/reduce', self.lib.seg_conv2d_wgrad, C.byref(w2), kernel='wgrad_reduce_kernel', side=sid)
                return
            if self.batch_reduce is not True:
                plan.add(name + '/reduce', self.lib.seg_conv2d_wgrad, C.byref(w2), kernel='wgrad_reduce_kernel', side=sid, flavor='per_layer')
            if self.batch_reduce is not False:
                self._pending_reduce.setdefault(sid, []).append(w)
        else:
            w.phase = 0
            plan.add(name, self.lib.seg_conv2d_wgrad, C.byref(w), desc=w, flops=fl, side=sid, bytes=getattr(self, '_wg_bytes', 0))

    def flush_reduce(self, plan, on_main=False):
        """Reduces the slabs of every filter gradient emitted since the last flush: one launch per side stream (each
        stream reduces its own layers, so the launch is simply in order behind them).  Call it where the gradients are
        first needed -- the end of a backward segment (all-reduce / Adam).  on_main: ONE launch on the main stream
        behind a join of the side streams (everything pending, whatever stream it ran on)."""
        pending, self._pending_reduce = self._pending_reduce, {}
        if on_main and pending:
            allw = [w for sid in sorted(pending, key=str) for w in pending[sid]]
            pending = {0: allw}
            self.join_all(plan)
        for sid in sorted(pending, key=str):
            ws = pending[sid]
            n = len(ws)
            arr = (C.POINTER(L.WgradDesc) * n)(*[C.pointer(w) for w in ws])
            host = (C.c_uint8 * (96 * n))()
            nj, nb = C.c_int32(0), C.c_int32(0)
            L.check(self.lib.seg_wgrad_reduce_batch_plan(arr, n, C.cast(host, C.c_void_p), 96 * n, C.byref(nj), C.byref(nb)), 'reduce_batch_plan')
            if nj.value == 0:
                continue
            jobs = torch.frombuffer(bytearray(host), dtype=torch.uint8)[:96 * nj.value].clone().to(self.device)
            plan.keep += [jobs, arr]
            plan.add('dw/reduce[%d]' % nj.value, self.lib.seg_wgrad_reduce_batch, jobs.data_ptr(), nj.value, nb.value,
                     kernel='wgrad_reduce_batch_kernel', side=sid, flavor='batched')          # sid 0 = main stream

    def first_im2col(self, plan, layer, x_f32, H, W):
        """im2col of the raw input (27 -> 32 channels) for the first layer's filter gradient as a tensor of its own
        (SEG_FIRST_IM2COL=1: the r01 form; by default the filter gradient gathers the im2col rows itself while it stages
        its tiles and this returns None).  It depends only on the input batch, so the models emit it at the START of the
        forward plan on a side stream (off the critical path)."""
        if layer.cin > 3:
            raise L.SegError('first-layer gradient supports input_channel <= 3')
        if os.environ.get('SEG_FIRST_IM2COL', '0') != '1':
            return None
        Ho, Wo = H + 2 * layer.pad - 2, W + 2 * layer.pad - 2
        col = self.act(Ho, Wo, 9 * layer.cin, name='im2col')
        cv = col.view()
        plan.keep.append(cv)
        plan.add(layer.name + '/im2col', self.lib.seg_im2col3x3, x_f32.data_ptr(), self.B, H, W, layer.cin, layer.pad, C.byref(cv), Ho, Wo,
                 self.dtype, kernel='im2col3x3_kernel', side=1)
        return col

    def fuses_first_pool_bwd(self, col=None):
        """True when first_bwd(pool=...) can rebuild dZ from the max-pool that consumes the first layer (bf16, virtual im2col)."""
        return col is None and self.dtype == L.SEG_BF16 and os.environ.get('SEG_FUSE_POOL1_BWD', '1') != '0'

    def first_bwd(self, plan, layer, x_f32, H, W, dz, col=None, same_stream=True, pool=None, ksplit=0, on_aux=False):
        """First-layer filter/bias gradient = the generic 1x1 MFMA wgrad over the im2col'd input; the [1][9*cin][cout]
        result is exactly the HWIO filter gradient.  col=None: the im2col rows are gathered from the float image inside the
        filter-gradient kernel (seg_wgrad_desc.im2col_x); else `col` is the tensor first_im2col wrote.
        pool = (y_act, dpool, add, add_hw, add_off) with dz=None (fuses_first_pool_bwd()): dZ is rebuilt inside the kernel from
        the layer's activation, the pooled gradient and the other consumer's gradient -- the arguments of pool_bwd, whose launch
        (on the critical stream) and whose dZ tensor then do not exist."""
        Ho, Wo = H + 2 * layer.pad - 2, W + 2 * layer.pad - 2
        if layer.cin > 3:
            raise L.SegError('first-layer gradient supports input_channel <= 3')
        w = L.WgradDesc()
        if col is None:
            # virtual source: a dense [B,Ho,Wo,32] window description whose pointer is never dereferenced
            w.src0 = L.View(ptr=x_f32.data_ptr(), H=Ho, W=Wo, cs=32, oy=0, ox=0, coff=0, c=32)
            w.im2col_x = x_f32.data_ptr(); w.im2col_h, w.im2col_w, w.im2col_cin, w.im2col_pad = H, W, layer.cin, layer.pad
        else:
            w.src0 = col.view()
        w.src1 = L.null_view(); w.src0_clog = 9 * layer.cin; w.src1_clog = 0
        w.B, w.Hi, w.Wi = self.B, Ho, Wo
        w.KH = w.KW = 1; w.stride = 1; w.pad_t = w.pad_l = 0
        w.Ho, w.Wo = Ho, Wo
        if pool is not None:
            assert dz is None and col is None
            y_act, dpool, add, add_hw, add_off = pool
            w.pool_y = y_act.view()
            w.pool_dp = dpool.view() if dpool is not None else L.null_view()
            w.pool_add = add.view() if add is not None else L.null_view()
            w.pool_add_h, w.pool_add_w, w.pool_add_y0, w.pool_add_x0 = add_hw[0], add_hw[1], add_off[0], add_off[1]
            w.dz = y_act.view()                       # (describes the channel padding; never dereferenced)
        else:
            w.dz = dz.view()
        w.n_log = layer.cout
        w.dw = self.store.g_ptr(layer.w_off); w.dtype = self.dtype; w.cfg = 0
        w.bias_mode = 1; w.db = self.store.g_ptr(layer.b_off); w.bias_n = layer.cout
        if self.side_enabled:
            # the last filter gradient of the backward pass and bandwidth-bound: nothing is left on the critical stream to share the chip
            # with (256^2 x 32 images alone: 109 -> 68 us against the 64-workgroup target of the other layers)
            w.target_wgs = 256
        self._wgrad_ws(w, plan, ksplit)
        fl = 2 * self.B * Ho * Wo * 9 * layer.cin * layer.cout
        self._wg_bytes = self.B * (H * W * layer.cin * 4 + Ho * Wo * layer.cout * self.es) + 9 * layer.cin * layer.cout * 4
        if pool is not None:
            self._wg_bytes += self.B * (Ho // 2) * (Wo // 2) * layer.cout * self.es + (self.B * add_hw[0] * add_hw[1] * layer.cout * self.es if add is not None else 0)
        # same side stream as the im2col that feeds it (stream 1): in order behind it, so a plan that runs forward and
        # backward back to back needs no join of the side streams in between
        # (same_stream=False: the data-parallel plans, which join the side streams after the forward anyway, keep the
        # alternating assignment -- pinning changed the shape of their captured segment graph and made it 20 % slower)
        # on_aux: the LAST filter gradient of the backward pass goes to the auxiliary stream, so that Adam for every other layer
        # (basemodel._finish_training_plans) can run beside it
        self._add_wgrad(plan, layer.name + '/dw', w, fl, sid='aux' if on_aux else (1 if (same_stream and col is not None) else None))
        plan.flops += fl

    def conv_bwd(self, plan, layer, srcs, Hi, Wi, dz, dsrcs, dz_off=(0, 0), cfg=0, wcfg=0, ksplit=0, dgrad_ksplit=None, wgrad_sid=None, src_bn=None):
        """Filter + bias gradient, then one dgrad launch per entry of dsrcs.
        dsrcs: list aligned with srcs; each None (no input gradient wanted) or
        (dst_act, (oy,ox), mask_act_or_None, (moy,mox))."""
        k, pad = layer.k, layer.pad
        Ho, Wo = Hi + 2 * pad - k + 1, Wi + 2 * pad - k + 1
        if (k == 3 and len(srcs) == 1 and srcs[0][0].thin and dz.thin and srcs[0][1:] == (0, 0) and dz_off == (0, 0) and layer.cin <= 8 and layer.cout <= 8
                and os.environ.get('SEG_THIN_VALU', '1') != '0' and os.environ.get('SEG_THIN_WGRAD', '1') != '0'):
            self._thin_wgrad(plan, layer, srcs[0][0], Hi, Wi, dz, Ho, Wo, wgrad_sid, src_bn)
            return self._conv_bwd_data(plan, layer, srcs, Hi, Wi, dz, dsrcs, dz_off, cfg, dgrad_ksplit, Ho, Wo)
        if src_bn is not None:
            raise L.SegError('conv_bwd(src_bn=...): only the thin 3x3 vector-ALU route normalises on load')
        w = L.WgradDesc()
        w.src0 = srcs[0][0].view_wide(srcs[0][1], srcs[0][2])
        w.src1 = srcs[1][0].view_wide(srcs[1][1], srcs[1][2]) if len(srcs) > 1 else L.null_view()
        w.src0_clog = layer.cin_segs[0]; w.src1_clog = layer.cin_segs[1] if len(srcs) > 1 else 0
        w.B, w.Hi, w.Wi = self.B, Hi, Wi
        w.KH = w.KW = k; w.stride = 1; w.pad_t = w.pad_l = pad
        w.Ho, w.Wo = Ho, Wo
        w.dz = dz.view_wide(dz_off[0], dz_off[1]); w.n_log = layer.cout
        w.thin = (1 if self._thin(*[s_[0] for s_ in srcs]) else 0) | (2 if dz.thin else 0)
        w.dw = self.store.g_ptr(layer.w_off); w.dtype = self.dtype; w.cfg = wcfg
        w.bias_mode = 1; w.db = self.store.g_ptr(layer.b_off); w.bias_n = layer.cout
        if layer.name in _tail_layers(getattr(self, 'tail_layers', ())):
            w.target_wgs = 256                       # runs after the critical stream has finished: the whole chip is its own
        self._wgrad_ws(w, plan, ksplit)
        fl = 2 * self.B * Ho * Wo * k * k * layer.cin * layer.cout
        self._wg_bytes = self.B * (Hi * Wi * layer.cin + Ho * Wo * layer.cout) * self.es + k * k * layer.cin * layer.cout * 4
        self._add_wgrad(plan, layer.name + '/dw', w, fl, sid=wgrad_sid)
        plan.flops += fl
        return self._conv_bwd_data(plan, layer, srcs, Hi, Wi, dz, dsrcs, dz_off, cfg, dgrad_ksplit, Ho, Wo)

    def _thin_wgrad(self, plan, layer, src, Hi, Wi, dz, Ho, Wo, sid=None, src_bn=None):
        """3x3 filter + bias gradient between thin tensors on the vector ALU (seg_thin_wgrad3x3), on a filter-gradient stream"""
        sv, zv = src.view(), dz.view()
        nbytes = int(self.lib.seg_thin_wgrad3x3_ws_bytes(layer.cin, layer.cout))
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
        plan.keep += [sv, zv, ws]
        fl = 2 * self.B * Ho * Wo * 9 * layer.cin * layer.cout
        if sid is None:
            sid = self._pick_wgrad_stream(15.0 + self.B * Ho * Wo * 32 * self.es / 3e6)
        if not self.side_enabled:
            sid = 0
        if src_bn is not None:
            st_, bl_ = src_bn
            plan.keep.append(st_)
            plan.add(layer.name + '/dw', self.lib.seg_thin_wgrad3x3_bn, C.byref(sv), self.B, Hi, Wi, C.byref(zv), Ho, Wo, layer.cin, layer.cout, layer.pad,
                     self.store.g_ptr(layer.w_off), self.store.g_ptr(layer.b_off) if layer.nbias else None, ws.data_ptr(), nbytes,
                     st_['stats'].data_ptr(), self.store.p_ptr(bl_.w_off), self.dtype,
                     kernel='thin_wgrad3x3_partial_kernel', flops=fl, bytes=self.B * (Hi * Wi + Ho * Wo) * 8 * self.es, side=sid)
            plan.flops += fl
            return
        plan.add(layer.name + '/dw', self.lib.seg_thin_wgrad3x3, C.byref(sv), self.B, Hi, Wi, C.byref(zv), Ho, Wo, layer.cin, layer.cout, layer.pad,
                 self.store.g_ptr(layer.w_off), self.store.g_ptr(layer.b_off) if layer.nbias else None, ws.data_ptr(), nbytes, self.dtype,
                 kernel='thin_wgrad3x3_partial_kernel', flops=fl, bytes=self.B * (Hi * Wi + Ho * Wo) * 8 * self.es, side=sid)
        plan.flops += fl

    def _conv_bwd_data(self, plan, layer, srcs, Hi, Wi, dz, dsrcs, dz_off, cfg, dgrad_ksplit, Ho, Wo):
        k, pad = layer.k, layer.pad
        n_off = 0
        merged = len(dsrcs) == 2 and dsrcs[0] is not None and dsrcs[1] is not None and not any(len(x) > 4 and x[4] for x in dsrcs)
        if merged:
            # both halves of a channel-concat input in ONE launch (seg_conv_desc.n_split): one kernel less per decoder level
            assert not dz.thin and not self._thin(dsrcs[0][0], dsrcs[1][0]), 'thin tensors have one source'
            d = L.ConvDesc()
            d.accum = 0
            d.src0 = dz.view(dz_off[0], dz_off[1]); d.src1 = L.null_view()
            d.B, d.Hi, d.Wi = self.B, Ho, Wo
            d.KH = d.KW = k; d.stride = 1; d.pad_t = d.pad_l = k - 1 - pad
            d.Ho, d.Wo = Hi, Wi
            d.w_packed = self.store.packed_ptr(layer.pk_dgrad)
            d.n_total = sum(layer.cin_p); d.n_off = 0; d.n_count = d.n_total; d.n_split = layer.cin_p[0]
            d.bias = None; d.bias_n = 0; d.up2 = 0; d.up_cout = 0
            (dst0, doff0, mask0, moff0), (dst1, doff1, mask1, moff1) = dsrcs[0][:4], dsrcs[1][:4]
            d.dst = dst0.view(doff0[0], doff0[1]); d.dst1 = dst1.view(doff1[0], doff1[1])
            d.mask = mask0.view(moff0[0], moff0[1]) if mask0 is not None else L.null_view()
            d.mask1 = mask1.view(moff1[0], moff1[1]) if mask1 is not None else L.null_view()
            d.relu = 0; d.out_f32 = 0; d.dtype = self.dtype; d.cfg = cfg
            d.sched = self.sched_slot()
            self._splitk(d, plan, dgrad_ksplit)
            plan.keep.append(d)
            fl = 2 * self.B * Ho * Wo * k * k * layer.cin * layer.cout
            nmask = sum(layer.cin_segs[i] for i, m_ in enumerate((mask0, mask1)) if m_ is not None)
            by = self.B * (Ho * Wo * layer.cout + Hi * Wi * (layer.cin + nmask)) * self.es + k * k * layer.cin * layer.cout * self.es
            plan.add(layer.name + '/dx01', self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl, bytes=by)
            plan.flops += fl
            return
        for i, ds in enumerate(dsrcs):
            if ds is not None:
                dst, doff, mask, moff = ds[:4]
                if (k == 3 and dz.thin and dst.thin and dz_off == (0, 0) and doff == (0, 0) and moff == (0, 0) and len(srcs) == 1 and not (len(ds) > 4 and ds[4])
                        and layer.cin <= 8 and layer.cout <= 8 and (mask is None or mask.thin) and os.environ.get('SEG_THIN_VALU', '1') != '0'):
                    zv2, xv2 = dz.view(), dst.view()
                    mv2 = mask.view() if mask is not None else None
                    plan.keep += [zv2, xv2] + ([mv2] if mv2 is not None else [])
                    fl = 2 * self.B * Ho * Wo * 9 * layer.cin * layer.cout
                    plan.add(layer.name + '/dx%d' % i, self.lib.seg_thin_conv3x3, C.byref(zv2), self.B, Ho, Wo, self.store.p_ptr(layer.w_off), None,
                             layer.cin, layer.cout, k - 1 - pad, 0, 1, C.byref(mv2) if mv2 is not None else None, C.byref(xv2), Hi, Wi, 0, self.dtype,
                             kernel='thin_conv3x3_kernel', flops=fl, bytes=self.B * (Hi * Wi + Ho * Wo) * 8 * self.es)
                    plan.flops += fl
                    n_off += layer.cin_p[i]
                    continue
                d = L.ConvDesc()
                d.accum = 1 if (len(ds) > 4 and ds[4]) else 0
                d.src0 = dz.view_wide(dz_off[0], dz_off[1]); d.src1 = L.null_view()
                d.thin_src = 1 if dz.thin else 0
                d.n_store = 8 if dst.thin else 0
                assert not dst.thin or mask is None or mask.thin
                d.B, d.Hi, d.Wi = self.B, Ho, Wo
                d.KH = d.KW = k; d.stride = 1; d.pad_t = d.pad_l = k - 1 - pad
                d.Ho, d.Wo = Hi, Wi
                d.w_packed = self.store.packed_ptr(layer.pk_dgrad)
                d.n_total = sum(layer.cin_p); d.n_off = n_off; d.n_count = layer.cin_p[i]
                d.bias = None; d.bias_n = 0
                d.dst = dst.view(doff[0], doff[1]); d.up2 = 0; d.up_cout = 0
                d.mask = mask.view(moff[0], moff[1]) if mask is not None else L.null_view()
                d.relu = 0; d.out_f32 = 0; d.dtype = self.dtype; d.cfg = cfg
                d.sched = self.sched_slot()
                self._splitk(d, plan, dgrad_ksplit)
                plan.keep.append(d)
                fl = 2 * self.B * Ho * Wo * k * k * layer.cin_segs[i] * layer.cout
                by = (self.B * (Ho * Wo * layer.cout + Hi * Wi * layer.cin_segs[i] * (1 + (1 if mask is not None else 0) + (1 if d.accum else 0))) * self.es
                      + k * k * layer.cin_segs[i] * layer.cout * self.es)
                plan.add(layer.name + '/dx%d' % i, self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl, bytes=by)
                plan.flops += fl
            n_off += layer.cin_p[i]

    def up_bwd(self, plan, layer, src, Hi, Wi, dzu, dsrc, mask, cfg=0, wcfg=0, ksplit=0):
        """src: input Act [Hi,Wi,cin]; dzu: masked grad of the upsampled output [2Hi,2Wi,cout]."""
        w = L.WgradDesc()
        w.src0 = dzu.view_wide(); w.src1 = L.null_view(); w.src0_clog = layer.cout; w.src1_clog = 0
        w.thin = 1 if dzu.thin else 0
        w.B, w.Hi, w.Wi = self.B, 2 * Hi, 2 * Wi
        w.KH = w.KW = 2; w.stride = 2; w.pad_t = w.pad_l = 0
        w.Ho, w.Wo = Hi, Wi
        w.dz = src.view(); w.n_log = layer.cin
        w.dw = self.store.g_ptr(layer.w_off); w.dtype = self.dtype; w.cfg = wcfg
        w.bias_mode = 2; w.db = self.store.g_ptr(layer.b_off); w.bias_n = layer.cout
        self._wgrad_ws(w, plan, ksplit)
        fl = 2 * self.B * Hi * Wi * 4 * layer.cin * layer.cout
        self._wg_bytes = self.B * Hi * Wi * (layer.cin + 4 * layer.cout) * self.es + 4 * layer.cin * layer.cout * 4
        self._add_wgrad(plan, layer.name + '/dw', w, fl)
        plan.flops += fl
        if dsrc is not None and dzu.thin and not dsrc.thin and layer.cout <= 8 and os.environ.get('SEG_THIN_VALU', '1') != '0':
            xv, zv_ = dsrc.view(), dzu.view()
            mv = mask.view() if mask is not None else None
            plan.keep += [xv, zv_] + ([mv] if mv is not None else [])
            plan.add(layer.name + '/dx', self.lib.seg_thin_up2x2, C.byref(xv), C.byref(zv_), self.B, Hi, Wi, self.store.p_ptr(layer.w_off), None,
                     layer.cin, layer.cout, 0, 1, C.byref(mv) if mv is not None else None, self.dtype, kernel='thin_up2x2_kernel', flops=fl,
                     bytes=self.B * Hi * Wi * (dsrc.Cp * (2 if mask is not None else 1) + 4 * 8) * self.es)
            plan.flops += fl
        elif dsrc is not None:
            d = L.ConvDesc()
            d.src0 = dzu.view_wide(); d.src1 = L.null_view()
            d.thin_src = 1 if dzu.thin else 0
            d.B, d.Hi, d.Wi = self.B, 2 * Hi, 2 * Wi
            d.KH = d.KW = 2; d.stride = 2; d.pad_t = d.pad_l = 0
            d.Ho, d.Wo = Hi, Wi
            d.w_packed = self.store.packed_ptr(layer.pk_dgrad)
            d.n_total = layer.cin_p[0]; d.n_off = 0; d.n_count = layer.cin_p[0]
            d.bias = None; d.bias_n = 0
            d.dst = dsrc.view(); d.up2 = 0; d.up_cout = 0
            d.mask = mask.view() if mask is not None else L.null_view()
            d.relu = 0; d.out_f32 = 0; d.dtype = self.dtype; d.cfg = cfg
            self._splitk(d, plan)
            plan.keep.append(d)
            by = self.B * Hi * Wi * (4 * layer.cout + layer.cin * (2 if mask is not None else 1)) * self.es + 4 * layer.cin * layer.cout * self.es
            plan.add(layer.name + '/dx', self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl, bytes=by)
            plan.flops += fl

    def pool_bwd(self, plan, y_act, dpool, add, add_hw, add_off, dz, H, W):
        yv, zv = y_act.view(), dz.view()
        pv = dpool.view() if dpool is not None else L.null_view()
        av = add.view() if add is not None else L.null_view()
        plan.keep += [yv, zv, pv, av]
        plan.add('pool/bwd', self.lib.seg_maxpool2x2_bwd, C.byref(yv), C.byref(pv), C.byref(av), add_hw[0], add_hw[1],
                 add_off[0], add_off[1], C.byref(zv), self.B, H, W, y_act.Cp, self.dtype, kernel='maxpool_bwd_kernel')

    def dropout(self, plan, act, keep, seed, offset_ref, dst=None):
        """slim.dropout-style mask: dst = act * Bernoulli(keep) / keep (in place when dst is None).  offset_ref is a ctypes
        c_uint64 whose current value is read at every launch (a fresh mask per stochastic pass without rebuilding the plan)."""
        v = act.view()
        o = dst.view() if dst is not None else v
        plan.keep += [v, o, offset_ref]
        plan.add('dropout', self.lib.seg_dropout, C.byref(v), C.byref(o), self.B, act.H, act.W, act.Cp, float(keep), int(seed), offset_ref,
                 self.dtype, kernel='dropout_kernel')

    def relu_grad(self, plan, dy, y_act, dz, H, W):
        a, b, c = dy.view(), y_act.view(), dz.view()
        plan.keep += [a, b, c]
        plan.add('relu_grad', self.lib.seg_relu_grad, C.byref(a), C.byref(b), C.byref(c), self.B, H, W, y_act.Cp, self.dtype,
                 kernel='relu_grad_kernel')

    def bilinear_fwd(self, plan, src, Hs, Ws, factor, filt, add, dst, Hd, Wd, dst_f32=False):
        """dst = crop_or_pad(conv2d_transpose(src, bilinear(factor), SAME), Hd, Wd) (+ add)"""
        cy = (Hs * factor - Hd) // 2 if Hs * factor >= Hd else -((Hd - Hs * factor) // 2)
        cx = (Ws * factor - Wd) // 2 if Ws * factor >= Wd else -((Wd - Ws * factor) // 2)
        sv, dv = src.view(), dst.view()
        av = add.view() if add is not None else None
        plan.keep += [sv, dv, av, filt]
        plan.add('bilinear_up%d' % factor, self.lib.seg_bilinear_up_fwd, C.byref(sv), Hs, Ws, factor, filt.data_ptr(),
                 C.byref(av) if av is not None else None, C.byref(dv), Hd, Wd, cy, cx, self.B, src.Cp, 1 if dst_f32 else 0, self.dtype,
                 kernel='bilinear_fwd_kernel')
        return cy, cx

    def bilinear_bwd(self, plan, ddst, Hd, Wd, factor, filt, dsrc, Hs, Ws, mask=None, dz=None):
        """dsrc = adjoint of bilinear_fwd wrt its source.  mask / dz (Acts shaped like dsrc): also dz = dsrc * (mask > 0), the
        gradient behind the ReLU of the activation `mask` (one launch instead of a relu_grad behind this one)."""
        cy = (Hs * factor - Hd) // 2 if Hs * factor >= Hd else -((Hd - Hs * factor) // 2)
        cx = (Ws * factor - Wd) // 2 if Ws * factor >= Wd else -((Wd - Ws * factor) // 2)
        gv, sv = ddst.view(), dsrc.view()
        mv, zv = (mask.view(), dz.view()) if mask is not None else (None, None)
        plan.keep += [gv, sv, filt, mv, zv]
        mp, zp = (C.byref(mv), C.byref(zv)) if mask is not None else (None, None)
        if factor >= 4:
            # big factors: separable form (2k instead of k*k taps, the gradient map read once instead of four times)
            nb = int(self.lib.seg_bilinear_up_bwd_ws_bytes(self.B, Hd, Ws, dsrc.Cp))
            ws = torch.empty(nb // 4, dtype=torch.float32, device=self.device)
            plan.keep.append(ws)
            plan.add('bilinear_up%d/bwd' % factor, self.lib.seg_bilinear_up_bwd_sep, C.byref(gv), Hd, Wd, cy, cx, factor, filt.data_ptr(),
                     C.byref(sv), Hs, Ws, self.B, dsrc.Cp, 0, ws.data_ptr(), nb, mp, zp, self.dtype, kernel='bilinear_bwd_h_kernel')
            return
        plan.add('bilinear_up%d/bwd' % factor, self.lib.seg_bilinear_up_bwd, C.byref(gv), Hd, Wd, cy, cx, factor, filt.data_ptr(),
                 C.byref(sv), Hs, Ws, self.B, dsrc.Cp, 0, mp, zp, self.dtype, kernel='bilinear_bwd_kernel')

    def bilinear_xent(self, plan, src, Hs, Ws, factor, filt, labels_u8, LH, LW, loff, H, W, n_classes, loss_buf, dlogits, logits=None):
        """logits = crop_or_pad(up_factor(src), H, W); mean softmax x-entropy; dlogits -- one launch (seg_bilinear_xent).  logits:
        float Act to ALSO receive the logits (None: they are never stored)."""
        cy = (Hs * factor - H) // 2 if Hs * factor >= H else -((H - Hs * factor) // 2)
        cx = (Ws * factor - W) // 2 if Ws * factor >= W else -((W - Ws * factor) // 2)
        sv, dv = src.view(), dlogits.view()
        lv = logits.view() if logits is not None else None
        plan.keep += [sv, dv, lv, filt]
        inv_n = 1.0 / float(self.B * H * W)
        plan.add('up%d+xent' % factor, self.lib.seg_bilinear_xent, C.byref(sv), Hs, Ws, factor, filt.data_ptr(), cy, cx, labels_u8.data_ptr(), LH, LW,
                 loff[0], loff[1], self.B, H, W, n_classes, inv_n, 1.0, loss_buf.data_ptr(), C.byref(dv), C.byref(lv) if lv is not None else None,
                 self.dtype, kernel='bilinear_xent_kernel')

    # ---------------- DeconvModel ops (models/deconvolution.py:101-178) ----------------
    def _dconv_desc(self, layer, big, small, mask=None, bias=False, relu=False):
        """descriptor of a direct-kernel layer.  'direct' (conv): x = input (k = Cin), y = output (n = Cout);
        'dtrans' (transposed conv): x = the LARGE map (k = Cout), y = the small map (n = Cin) -- include/seg_hip.h."""
        d = L.DconvDesc()
        d.x = big.view(); d.y = small.view()
        d.B, d.Hx, d.Wx, d.Hy, d.Wy = self.B, big.H, big.W, small.H, small.W
        k, s_ = layer.k, layer.stride
        d.KH = d.KW = k; d.stride = s_
        if layer.kind == 'direct':
            d.xc, d.yc = layer.cin, layer.cout
            if layer.padding == 'SAME':              # TF: out = ceil(in/s); pad_total = max((out-1)s + k - in, 0); before = total // 2
                d.pad_t = max((small.H - 1) * s_ + k - big.H, 0) // 2
                d.pad_l = max((small.W - 1) * s_ + k - big.W, 0) // 2
            else:
                d.pad_t = d.pad_l = 0
            d.w_sk = layer.cout
        else:
            d.xc, d.yc = layer.cout, layer.cin
            d.pad_t = d.pad_l = 0                    # slim.convolution2d_transpose(padding='VALID'): out = in*s + max(k - s, 0)
            d.w_sk = layer.cin
        d.w_sv = layer.wshape[2] * layer.wshape[3]; d.w_su = k * d.w_sv
        d.w = self.store.p_ptr(layer.w_off)
        d.bias = self.store.p_ptr(layer.b_off) if bias else None
        d.bias_n = layer.cout if bias else 0
        d.relu = 1 if relu else 0
        d.mask = mask.view() if mask is not None else L.null_view()
        d.dtype = self.dtype
        return d

    # ---- k x k / stride-s transposed convolutions on the MFMA kernels (Layer.tconv = (k, stride, Cout); see deconv_ops.hip) ----
    @staticmethod
    def tconv_layer(name, k, cin, cout, stride, relu=True):
        """The layer object of a VALID transposed convolution [k,k,Cout,Cin] run as the adjoint of a strided convolution that is
        a 1x1 convolution over an im2col: a 'conv' layer with K = k*k*Cout inputs and Cin outputs over the same parameter memory
        (TF filter layout kept for get/set_params and snapshots; the bias has Cout entries)."""
        l = Layer(name, 'conv', 1, [k * k * cout], cin, 'VALID', relu)
        l.wshape = (k, k, cout, cin)
        l.nbias = cout
        l.tconv = (k, stride, cout)
        return l

    def _conv1x1_desc(self, layer, src_view, dst_view, H, W, dgrad):
        d = L.ConvDesc()
        d.accum = 0
        d.src0 = src_view; d.src1 = L.null_view()
        d.B, d.Hi, d.Wi = self.B, H, W
        d.KH = d.KW = 1; d.stride = 1; d.pad_t = d.pad_l = 0
        d.Ho, d.Wo = H, W
        if dgrad:
            d.w_packed = self.store.packed_ptr(layer.pk_dgrad); d.n_total = layer.cin_p[0]
        else:
            d.w_packed = self.store.packed_ptr(layer.pk_fwd); d.n_total = layer.cout_p
        d.n_off = 0; d.n_count = d.n_total
        d.bias = None; d.bias_n = 0
        d.dst = dst_view; d.up2 = 0; d.up_cout = 0; d.mask = L.null_view()
        d.relu = 0; d.out_f32 = 0; d.dtype = self.dtype; d.cfg = 0
        return d

    def tconv_fwd(self, plan, layer, src, dst):
        """dst = relu(bias + conv2d_transpose(src)): the 1x1 data gradient of the adjoint convolution (MFMA), then the col2im gather"""
        k, s_, co = layer.tconv
        Hi, Wi = src.H, src.W
        col = self.act(Hi, Wi, k * k * co, name=layer.name + '/col')
        sv, cv, ov = src.view(), col.view(), dst.view()
        d = self._conv1x1_desc(layer, sv, cv, Hi, Wi, dgrad=True)
        plan.keep += [d, sv, cv, ov]
        fl = 2 * self.B * Hi * Wi * k * k * co * layer.cout
        plan.add(layer.name, self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl,
                 bytes=self.B * Hi * Wi * (layer.cout + k * k * co) * self.es + k * k * co * layer.cout * self.es)
        plan.add(layer.name + '/col2im', self.lib.seg_col2im, C.byref(cv), self.B, Hi, Wi, co, k, k, s_, 0, 0, self.store.p_ptr(layer.b_off),
                 1 if layer.relu else 0, C.byref(ov), dst.H, dst.W, self.dtype, kernel='col2im_kernel')
        plan.flops += fl

    def tconv_bwd(self, plan, layer, src, dz, dsrc=None):
        """bias, filter and (dsrc given) input gradient of a tconv layer from its masked output gradient dz"""
        k, s_, co = layer.tconv
        Hi, Wi = src.H, src.W
        zv, sv = dz.view(), src.view()
        col = self.act(Hi, Wi, k * k * co, name=layer.name + '/dzcol')
        cv = col.view()
        plan.keep += [zv, sv, cv]
        self.bias_grad(plan, layer.name + '/db', zv, dz.H, dz.W, co, layer.b_off)
        plan.add(layer.name + '/im2col', self.lib.seg_im2col_act, C.byref(zv), self.B, dz.H, dz.W, co, k, k, s_, 0, 0, C.byref(cv), Hi, Wi,
                 self.dtype, kernel='im2col_act_kernel')
        w = L.WgradDesc()
        w.src0 = cv; w.src1 = L.null_view(); w.src0_clog = k * k * co; w.src1_clog = 0
        w.B, w.Hi, w.Wi = self.B, Hi, Wi
        w.KH = w.KW = 1; w.stride = 1; w.pad_t = w.pad_l = 0
        w.Ho, w.Wo = Hi, Wi
        w.dz = sv; w.n_log = layer.cout
        w.dw = self.store.g_ptr(layer.w_off); w.dtype = self.dtype; w.cfg = 0
        w.bias_mode = 0; w.db = None; w.bias_n = 0
        self._wgrad_ws(w, plan)
        fl = 2 * self.B * Hi * Wi * k * k * co * layer.cout
        self._wg_bytes = self.B * Hi * Wi * (k * k * co + layer.cout) * self.es + k * k * co * layer.cout * 4
        self._add_wgrad(plan, layer.name + '/dw', w, fl)
        plan.flops += fl
        if dsrc is not None:
            dv = dsrc.view()
            d = self._conv1x1_desc(layer, cv, dv, Hi, Wi, dgrad=False)
            plan.keep += [d, dv]
            plan.add(layer.name + '/dx', self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl,
                     bytes=self.B * Hi * Wi * (layer.cout + k * k * co) * self.es + k * k * co * layer.cout * self.es)
            plan.flops += fl

    # ---- k x k / stride-s VALID convolutions on the MFMA kernels: a 1x1 convolution over the strided im2col (Layer.sconv) ----
    @staticmethod
    def sconv_layer(name, k, cin, cout, stride, relu=True):
        """The layer object of a VALID k x k / stride-s convolution [k,k,Cin,Cout] run as a 1x1 convolution over its im2col: a
        'conv' layer with K = k*k*Cin inputs over the same parameter memory (HWIO kept for get/set_params and snapshots).  The
        adversary's two 3x3/s2 convolutions (models/basemodel.py:228-246); the direct VALU kernels ran them at 8 TFLOP/s."""
        l = Layer(name, 'conv', 1, [k * k * cin], cout, 'VALID', relu)
        l.wshape = (k, k, cin, cout)
        l.sconv = (k, stride, cin)
        return l

    def sconv_fwd(self, plan, layer, src, dst, col):
        """dst = relu?(bias + conv(src)); col ([B, Ho, Wo, k*k*Cin], kept for the filter gradient) is written first"""
        k, s_, ci = layer.sconv
        sv, cv, dv = src.view(), col.view(), dst.view()
        plan.keep += [sv, cv, dv]
        plan.add(layer.name + '/im2col', self.lib.seg_im2col_act, C.byref(sv), self.B, src.H, src.W, ci, k, k, s_, 0, 0, C.byref(cv), dst.H, dst.W,
                 self.dtype, kernel='im2col_act_kernel')
        d = self._conv1x1_desc(layer, cv, dv, dst.H, dst.W, dgrad=False)
        d.bias = self.store.p_ptr(layer.b_off); d.bias_n = layer.cout; d.relu = 1 if layer.relu else 0
        plan.keep.append(d)
        fl = 2 * self.B * dst.H * dst.W * k * k * ci * layer.cout
        plan.add(layer.name, self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl,
                 bytes=self.B * dst.H * dst.W * (k * k * ci + layer.cout) * self.es + k * k * ci * layer.cout * self.es)
        plan.flops += fl

    def sconv_bwd(self, plan, layer, col, dz, dcol=None, dsrc=None, wgrad=True):
        """filter + bias gradient from (col, dz) and -- dsrc given -- the input gradient: 1x1 data gradient into dcol, then the
        adjoint gather of the im2col (seg_col2im)"""
        k, s_, ci = layer.sconv
        Ho, Wo = dz.H, dz.W
        cv, zv = col.view(), dz.view()
        plan.keep += [cv, zv]
        fl = 2 * self.B * Ho * Wo * k * k * ci * layer.cout
        if wgrad:
            w = L.WgradDesc()
            w.src0 = cv; w.src1 = L.null_view(); w.src0_clog = k * k * ci; w.src1_clog = 0
            w.B, w.Hi, w.Wi = self.B, Ho, Wo
            w.KH = w.KW = 1; w.stride = 1; w.pad_t = w.pad_l = 0
            w.Ho, w.Wo = Ho, Wo
            w.dz = zv; w.n_log = layer.cout
            w.dw = self.store.g_ptr(layer.w_off); w.dtype = self.dtype; w.cfg = 0
            w.bias_mode = 1; w.db = self.store.g_ptr(layer.b_off); w.bias_n = layer.cout
            self._wgrad_ws(w, plan)
            self._wg_bytes = self.B * Ho * Wo * (k * k * ci + layer.cout) * self.es + k * k * ci * layer.cout * 4
            self._add_wgrad(plan, layer.name + '/dw', w, fl, sid=0, own_reduce=True)      # (the adversary's plan: one stream, never flushed)
            plan.flops += fl
        if dsrc is not None:
            gv, xv = dcol.view(), dsrc.view()
            d = self._conv1x1_desc(layer, zv, gv, Ho, Wo, dgrad=True)
            plan.keep += [d, gv, xv]
            plan.add(layer.name + '/dx', self.lib.seg_conv2d, C.byref(d), desc=d, flops=fl,
                     bytes=self.B * Ho * Wo * (k * k * ci + layer.cout) * self.es + k * k * ci * layer.cout * self.es)
            plan.add(layer.name + '/col2im', self.lib.seg_col2im, C.byref(gv), self.B, Ho, Wo, ci, k, k, s_, 0, 0, None, 0, C.byref(xv), dsrc.H, dsrc.W,
                     self.dtype, kernel='col2im_kernel')
            plan.flops += fl

    def dlayer_fwd(self, plan, layer, src, dst):
        """forward of a 'direct' conv (src -> dst) or a 'dtrans' transposed conv (small src -> large dst), bias + ReLU"""
        k = layer.k
        if layer.kind == 'direct':
            d = self._dconv_desc(layer, src, dst, bias=True, relu=layer.relu)
            fn, kern = self.lib.seg_dconv_fwd, 'dconv_fwd_kernel'
            fl = 2 * self.B * dst.H * dst.W * k * k * layer.cin * layer.cout
        else:
            d = self._dconv_desc(layer, dst, src, bias=True, relu=layer.relu)
            fn, kern = self.lib.seg_dconv_bwd_data, 'dconv_bwd_data_kernel'
            fl = 2 * self.B * src.H * src.W * k * k * layer.cin * layer.cout
        plan.keep.append(d)
        plan.add(layer.name, fn, C.byref(d), kernel=kern, flops=fl)
        plan.flops += fl

    def dlayer_bwd(self, plan, layer, src, dz, dsrc=None, mask=None, wgrad=True):
        """filter / bias gradient of a direct-kernel layer from its masked output gradient dz, then (dsrc given) the input
        gradient, optionally masked by `mask` (the ReLU output that produced src)"""
        k = layer.k
        if layer.kind == 'direct':
            d = self._dconv_desc(layer, src, dz)
            d.bias_n = layer.cout
            mode, fl = 1, 2 * self.B * dz.H * dz.W * k * k * layer.cin * layer.cout
        else:
            d = self._dconv_desc(layer, dz, src)
            d.bias_n = layer.cout
            mode, fl = 2, 2 * self.B * src.H * src.W * k * k * layer.cin * layer.cout
        if wgrad:
            plan.keep.append(d)
            nb = int(self.lib.seg_dconv_wgrad_ws_bytes(C.byref(d)))       # pixel-split partial slabs (0: the layer has tiles enough)
            ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=self.device)
            plan.keep.append(ws)
            plan.add(layer.name + '/dw', self.lib.seg_dconv_wgrad, C.byref(d), self.store.g_ptr(layer.w_off), self.store.g_ptr(layer.b_off), mode,
                     ws.data_ptr() if nb else None, nb, kernel='dconv_wgrad_kernel', flops=fl)
            plan.flops += fl
        if dsrc is None:
            return
        if layer.kind == 'direct':
            d2 = self._dconv_desc(layer, dsrc, dz, mask=mask)
            fn, kern = self.lib.seg_dconv_bwd_data, 'dconv_bwd_data_kernel'
        else:
            d2 = self._dconv_desc(layer, dz, dsrc, mask=mask)
            fn, kern = self.lib.seg_dconv_fwd, 'dconv_fwd_kernel'
        plan.keep.append(d2)
        plan.add(layer.name + '/dx', fn, C.byref(d2), kernel=kern, flops=fl)
        plan.flops += fl

    def bn_state(self, layer):
        """device state of one batch norm: moving [mean | variance], batch stats [mean | rstd], reduction workspace"""
        Cp = layer.cout_p
        mov = torch.zeros(2 * Cp, dtype=torch.float32, device=self.device); mov[Cp:] = 1.0
        stats = torch.zeros(2 * Cp, dtype=torch.float32, device=self.device)
        ws = torch.zeros(self.lib.seg_bn_ws_bytes(Cp) // 4, dtype=torch.float32, device=self.device)
        return {'moving': mov, 'stats': stats, 'ws': ws}

    def bn_fwd(self, plan, layer, st, a, y, training=True, update_moving=True, decay=0.999, eps=1e-3, rows=0):
        """rows > 0: the launch that produced `a` has left that many statistics rows in st['ws'] (first_gen_fwd / up_fwd with
        bn_st): the statistics pass over `a` is skipped (training statistics only)."""
        av, yv = a.view(), y.view()
        plan.keep += [av, yv, st]
        mov = st['moving'].data_ptr() if (update_moving or not training) else None
        if rows > 0 and training:
            plan.add(layer.name, self.lib.seg_bn_fwd_rows, C.byref(av), C.byref(yv), self.store.p_ptr(layer.w_off), mov, st['stats'].data_ptr(),
                     decay, eps, self.B, a.H, a.W, a.Cp, layer.cout, st['ws'].data_ptr(), rows, self.dtype, kernel='bn_apply_kernel')
            return
        plan.add(layer.name, self.lib.seg_bn_fwd, C.byref(av), C.byref(yv), self.store.p_ptr(layer.w_off), mov, st['stats'].data_ptr(),
                 1 if training else 0, decay, eps, self.B, a.H, a.W, a.Cp, layer.cout, st['ws'].data_ptr(), self.dtype, kernel='bn_apply_kernel')

    def bn_stats(self, plan, layer, st, a, training=True, update_moving=True, decay=0.999, eps=1e-3, rows=0):
        """the statistics of bn_fwd alone (batch or moving -> st['stats'], moving averages updated): for a consumer that normalises on
        load (conv_fwd / conv_bwd with src_bn=(st, layer))"""
        av = a.view()
        plan.keep += [av, st]
        mov = st['moving'].data_ptr() if (update_moving or not training) else None
        if rows > 0 and training:
            plan.add(layer.name + '/stats', self.lib.seg_bn_fwd_rows, C.byref(av), None, self.store.p_ptr(layer.w_off), mov, st['stats'].data_ptr(),
                     decay, eps, self.B, a.H, a.W, a.Cp, layer.cout, st['ws'].data_ptr(), rows, self.dtype, kernel='bn_final_kernel')
            return
        plan.add(layer.name + '/stats', self.lib.seg_bn_fwd, C.byref(av), None, self.store.p_ptr(layer.w_off), mov, st['stats'].data_ptr(),
                 1 if training else 0, decay, eps, self.B, a.H, a.W, a.Cp, layer.cout, st['ws'].data_ptr(), self.dtype, kernel='bn_final_kernel')

    def bn_pool_fwd(self, plan, layer, st, a, pooled, k, training=True, update_moving=True, decay=0.999, eps=1e-3, rows=0):
        """batch norm + the k x k / stride-k max-pool that consumes it in one pass (seg_bn_pool_fwd): `pooled` is bit for bit
        max_pool(batch_norm(a)); the normalised tensor is not produced -- pool_k_bwd takes `a` as its source."""
        av, pv = a.view(), pooled.view()
        plan.keep += [av, pv, st]
        mov = st['moving'].data_ptr() if (update_moving or not training) else None
        plan.add(layer.name + '+pool%d' % k, self.lib.seg_bn_pool_fwd, C.byref(av), C.byref(pv), self.store.p_ptr(layer.w_off), mov, st['stats'].data_ptr(),
                 1 if training else 0, decay, eps, self.B, a.H, a.W, a.Cp, layer.cout, st['ws'].data_ptr(), rows if training else 0, k, self.dtype,
                 kernel='bn_pool_apply_kernel')

    def bn_pool_relu_bwd(self, plan, layer, st, a, dpool, dz, k):
        """backward of bn_pool_fwd: pool_k_bwd (source `a`) + bn_relu_bwd without the pool's full-resolution gradient tensor"""
        av, pv, zv = a.view(), dpool.view(), dz.view()
        plan.keep += [av, pv, zv, st]
        plan.add(layer.name + '+pool%d/bwd' % k, self.lib.seg_bn_pool_relu_bwd, C.byref(av), C.byref(pv), C.byref(zv), st['stats'].data_ptr(),
                 self.store.g_ptr(layer.w_off), 0, k, self.B, a.H, a.W, a.Cp, layer.cout, st['ws'].data_ptr(), self.dtype, kernel='bn_pool_apply_bwd_kernel')

    def bn_relu_bwd(self, plan, layer, st, a, dy, dz, dbeta_ptr=None, dbeta_add=False):
        """dbeta goes to the layer's slot of the gradient arena (dbeta_add: added to it -- a second batch through the same
        layer) or to dbeta_ptr (a scratch buffer: data-gradient-only passes)"""
        av, gv, zv = a.view(), dy.view(), dz.view()
        plan.keep += [av, gv, zv, st]
        plan.add(layer.name + '/bwd', self.lib.seg_bn_relu_bwd, C.byref(av), C.byref(gv), C.byref(zv), st['stats'].data_ptr(),
                 self.store.g_ptr(layer.w_off) if dbeta_ptr is None else dbeta_ptr, 1 if dbeta_add else 0, self.B, a.H, a.W, a.Cp, layer.cout,
                 st['ws'].data_ptr(), self.dtype, kernel='bn_apply_kernel')

    def pool_k_fwd(self, plan, src, dst, k):
        sv, dv = src.view(), dst.view()
        plan.keep += [sv, dv]
        plan.add('pool%d' % k, self.lib.seg_maxpool_k_fwd, C.byref(sv), C.byref(dv), k, self.B, dst.H, dst.W, src.Cp, self.dtype, kernel='maxpool_k_fwd_kernel')

    def pool_k_bwd(self, plan, src, dpool, dsrc, k):
        sv, pv, dv = src.view(), dpool.view(), dsrc.view()
        plan.keep += [sv, pv, dv]
        plan.add('pool%d/bwd' % k, self.lib.seg_maxpool_k_bwd, C.byref(sv), C.byref(pv), C.byref(dv), k, self.B, src.H, src.W, src.Cp, self.dtype,
                 kernel='maxpool_k_bwd_kernel')

    def resize_fwd(self, plan, src, dst):
        sv, dv = src.view(), dst.view()
        plan.keep += [sv, dv]
        plan.add('resize', self.lib.seg_resize_bilinear_fwd, C.byref(sv), src.H, src.W, C.byref(dv), dst.H, dst.W, self.B, src.Cp, self.dtype,
                 kernel='resize_fwd_kernel')

    def resize_bwd(self, plan, ddst, dsrc):
        gv, sv = ddst.view(), dsrc.view()
        plan.keep += [gv, sv]
        plan.add('resize/bwd', self.lib.seg_resize_bilinear_bwd, C.byref(gv), ddst.H, ddst.W, C.byref(sv), dsrc.H, dsrc.W, self.B, dsrc.Cp, self.dtype,
                 kernel='resize_bwd_kernel')

    def dropout_step(self, plan, src, dst, keep, seed, offset):
        """training-time slim.dropout: the mask is a function of (seed, offset, global step read on the device); the backward
        launch (gradient in, gradient out, same arguments) regenerates it"""
        sv, dv = src.view(), dst.view()
        plan.keep += [sv, dv]
        plan.add('dropout', self.lib.seg_dropout_step, C.byref(sv), C.byref(dv), self.B, src.H, src.W, src.Cp, float(keep), int(seed), int(offset),
                 self.store.step.data_ptr() + 8, self.dtype, kernel='dropout_kernel')

    def cast_pad(self, plan, x_f32, dst):
        """float32 NHWC image -> activation buffer (compute dtype, channels padded)"""
        dv = dst.view()
        plan.keep.append(dv)
        plan.add('cast_pad', self.lib.seg_cast_pad, x_f32.data_ptr(), self.B * dst.H * dst.W, dst.C, C.byref(dv), self.dtype, kernel='cast_pad_kernel')

    # ---------------- loss / outputs / update ----------------
    def softmax_xent(self, plan, logits, labels_u8, LH, LW, loff, H, W, n_classes, loss_buf, dlogits, probs=None):
        """probs (a view of the compute dtype, adversarial training): softmax(logits) leaves the same launch as a tensor -- the
        adversary's "fake" input (models/basemodel.py:285)."""
        lv, dv = logits.view(), dlogits.view()
        plan.keep += [lv, dv]
        inv_n = 1.0 / float(self.B * H * W)
        if probs is not None:
            plan.keep.append(probs)
            plan.add('xent+probs', self.lib.seg_softmax_xent_probs, C.byref(lv), labels_u8.data_ptr(), LH, LW, loff[0], loff[1], self.B, H, W,
                     n_classes, inv_n, 1.0, loss_buf.data_ptr(), C.byref(dv), C.byref(probs), self.dtype, kernel='softmax_xent_kernel')
            return
        plan.add('xent', self.lib.seg_softmax_xent, C.byref(lv), labels_u8.data_ptr(), LH, LW, loff[0], loff[1], self.B, H, W,
                 n_classes, inv_n, 1.0, loss_buf.data_ptr(), C.byref(dv), self.dtype, kernel='softmax_xent_kernel')

    def head_xent(self, plan, layer, act, labels_u8, LH, LW, loff, H, W, n_classes, loss_buf, logits, dlogits, dact, fuse_dw=True):
        """Fused 1x1 output conv + softmax x-entropy + the conv's masked input gradient (seg_head_xent).  fuse_dw (<= 8
        classes): the conv's filter / bias gradient is accumulated in the same pass and dlogits is not written; returns
        the partial-sum workspace for head_dw_reduce (None when the filter gradient stays a separate launch)."""
        av, lv, dv, gv = act.view(), logits.view(), dlogits.view(), dact.view()
        plan.keep += [av, lv, dv, gv]
        inv_n = 1.0 / float(self.B * H * W)
        fl = 2 * 2 * self.B * H * W * layer.cin * layer.cout          # forward + input gradient
        by = self.B * H * W * (2 * layer.cin * self.es + n_classes * (4 + self.es) + 1)
        ws, nbytes = None, int(self.lib.seg_head_xent_ws_bytes(self.B, H, W, act.Cp, n_classes)) if fuse_dw else 0
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            plan.keep.append(ws)
            fl += 2 * self.B * H * W * layer.cin * layer.cout          # + filter gradient
            by = self.B * H * W * (2 * layer.cin * self.es + n_classes * 4 + 1)
        plan.add(layer.name + '+xent+dx' + ('+dw' if ws is not None else ''), self.lib.seg_head_xent, C.byref(av),
                 self.store.p_ptr(layer.w_off), self.store.p_ptr(layer.b_off),
                 layer.cin, labels_u8.data_ptr(), LH, LW, loff[0], loff[1], self.B, H, W, n_classes, inv_n, loss_buf.data_ptr(),
                 C.byref(lv), C.byref(dv), C.byref(gv), ws.data_ptr() if ws is not None else None, nbytes, self.dtype,
                 kernel='head_xent_kernel', flops=fl, bytes=by)
        plan.flops += fl
        return ws

    def head_dw_reduce(self, plan, layer, ws, H, W, cin_pad, n_classes):
        """Workspace rows of head_xent(fuse_dw) -> the output layer's filter and bias gradient; a side stream, like the
        filter gradients it replaces."""
        sid = self._pick_wgrad_stream(7.0)
        if not self.side_enabled:
            sid = 0
        plan.add(layer.name + '/dw', self.lib.seg_head_dw_reduce, ws.data_ptr(), ws.numel(), self.B, H, W, cin_pad, layer.cin, n_classes,
                 self.store.g_ptr(layer.w_off), self.store.g_ptr(layer.b_off), kernel='head_dw_reduce_kernel', side=sid)

    def sigmoid_argmax(self, plan, logits, H, W, n_classes, sig, out):
        lv = logits.view()
        plan.keep.append(lv)
        plan.add('sigmoid_argmax', self.lib.seg_sigmoid_argmax, C.byref(lv), self.B, H, W, n_classes, sig.data_ptr(), out.data_ptr(), kernel='sigmoid_argmax_kernel')

    def pack(self, plan, aux=False):
        """Re-packs the fp32 master weights into the MFMA operand layout.  aux=True: on the auxiliary stream (the
        training forward starts with it, overlapped with the first layer, which reads the fp32 arena directly)."""
        s = self.store
        if s.pack_table is None:
            return
        meta = {'kernel': 'pack_kernel'}
        if aux:
            meta['side'] = 'aux'
        if getattr(s, 'pack_dual', None) is not None:
            # training stores: both packed copies of a tile from ONE read of the arena
            tab, dg, ne, tiles = s.pack_dual
            meta['kernel'] = 'pack_dual_kernel'
            plan.add('pack', self.lib.seg_pack_weights_dual, s.p.data_ptr(), s.packed.data_ptr(), tab.data_ptr(), dg.data_ptr(), ne, tiles,
                     self.dtype, **meta)
            return
        plan.add('pack', self.lib.seg_pack_weights, s.p.data_ptr(), s.packed.data_ptr(), s.pack_table.data_ptr(),
                 s.n_pack_entries, s.pack_blocks, self.dtype, **meta)

    def bias_grad(self, plan, name, zv, H, W, n_log, b_off):
        """BiasAddGrad of a tensor view as a launch of its own: two-stage (512 partial rows + a fixed-order final pass) on maps of
        >= 8 k pixels, where the one-workgroup-per-8-channels kernel is a handful of workgroups (51 -> ~15 us at 256^2 x 16 x 32)"""
        nb = int(self.lib.seg_bias_grad_ws_bytes(zv.c))
        if nb and self.B * H * W >= 8192:
            ws = torch.empty(nb // 4, dtype=torch.float32, device=self.device)
            plan.keep.append(ws)
            plan.add(name, self.lib.seg_bias_grad_ws, C.byref(zv), self.B, H, W, n_log, self.store.g_ptr(b_off), ws.data_ptr(), nb, self.dtype,
                     kernel='bias_grad_partial_kernel')
        else:
            plan.add(name, self.lib.seg_bias_grad, C.byref(zv), self.B, H, W, n_log, self.store.g_ptr(b_off), self.dtype, kernel='bias_grad_kernel')

    def dp_marker(self, plan, kind, **kw):
        """data-parallel marker: 'bucket' (lo, hi: this slice of the gradient arena is complete once everything in front of the
        marker has run) or 'wait' (every bucket must have landed before what follows).  Plan.run(on_marker=...) calls back; a run
        without a callback ignores them."""
        plan.ops.append(('dp_' + kind, None, ()))
        plan.meta.append(dict(kernel='marker', marker=kind, **kw))

    def join_all(self, plan):
        """main stream waits for every side stream (the filter gradients) -- e.g. before Adam inside the same plan"""
        plan.ops.append(('join_all', None, ()))
        plan.meta.append({'kernel': 'marker'})

    def join_aux(self, plan):
        plan.ops.append(('join_aux', None, ()))
        plan.meta.append({'kernel': 'marker'})

    def join_wgrad(self, plan):
        """main stream waits for the filter-gradient side streams, not for the auxiliary one"""
        plan.ops.append(('join_wgrad', None, ()))
        plan.meta.append({'kernel': 'marker'})

    def adam(self, plan, lr, grad_scale=1.0, b1=0.9, b2=0.999, eps=1e-8, lo=0, hi=None):
        """TF-Adam over the arena slice [lo, hi)."""
        s = self.store
        hi = s.n if hi is None else hi
        if hi <= lo:
            return
        meta = {'kernel': 'adam_kernel'}
        o = lo * 4
        plan.add('adam[%d:%d]' % (lo, hi), self.lib.seg_adam, s.p.data_ptr() + o, s.g.data_ptr() + o, s.m.data_ptr() + o, s.v.data_ptr() + o,
                 hi - lo, lr, b1, b2, eps, grad_scale, s.step.data_ptr() + 8, **meta)

    def step_begin(self, plan, loss_buf, aux=True):
        """global_step += 1 and loss accumulator = 0, first thing of a training forward (auxiliary stream: off the
        critical path; nothing before the loss kernel / Adam reads either)."""
        meta = {'kernel': 'step_begin_kernel'}
        if aux:
            meta['side'] = 'aux'
        plan.add('step_begin', self.lib.seg_step_begin, self.store.step.data_ptr(), loss_buf.data_ptr(), **meta)
