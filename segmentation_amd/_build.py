"""Builds libseg_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library
is a plain C-ABI shared object (include/seg_hip.h) loaded through ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['conv_fwd.hip', 'conv_sweep.hip', 'conv_wgrad.hip', 'wgrad_sweep.hip', 'conv_first.hip', 'elementwise.hip', 'deconv_ops.hip', 'adv_ops.hip']
LIB = os.path.join(HERE, 'libseg_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result'] + os.environ.get('SEG_EXTRA_FLAGS', '').split()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'seg_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time (torch.distributed.run starts one process per GPU, all of which come through here): the others
    # wait for the lock and then find the library fresh
    import fcntl
    with open(os.path.join(objdir, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build_locked(objdir, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(objdir, verbose):

    def cc(src):
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        cmd = [HIPCC] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, r.stderr[-4000:]))
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    tmp = LIB + '.tmp.%d' % os.getpid()
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s' % r.stderr[-4000:])
    os.replace(tmp, LIB)                 # atomic: a concurrent loader sees the old or the new library, never half of one
    if verbose:
        print('built', LIB, file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
