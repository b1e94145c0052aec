"""Generates tests/golden/bilinear_golden.npz by importing the reference's only py3-importable
hot-path file, /root/reference/utils/upsampling.py (run in the build container only; the GPU box
has no /root/reference).  `xrange` is injected into the module namespace (the file is py2);
the file itself is untouched and none of its text is stored - only inputs and outputs.

    python tests/golden/make_bilinear_golden.py
"""
import hashlib
import importlib.util
import os
import numpy as np

REF = '/root/reference/utils/upsampling.py'
spec = importlib.util.spec_from_file_location('ref_upsampling', REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
mod.xrange = range

out = {}
for f in (1, 2, 3, 4, 8, 16, 32):
    out['ksize_%d' % f] = np.int64(mod.get_kernel_size(f))
for s in (1, 2, 3, 4, 5, 8, 16, 32, 64):
    out['filt_%d' % s] = np.asarray(mod.upsample_filt(s), np.float64)
sha = {}
for f, c in ((2, 2), (2, 4), (2, 21), (8, 2), (8, 4), (8, 21), (16, 21), (32, 2), (32, 21)):
    w = mod.bilinear_upsample_weights(f, c)
    assert w.dtype == np.float32
    sha['%d_%d' % (f, c)] = hashlib.sha256(np.ascontiguousarray(w).tobytes()).hexdigest()
    if w.nbytes <= 64 * 1024:
        out['weights_%d_%d' % (f, c)] = w
out['sha_keys'] = np.array(sorted(sha))
out['sha_vals'] = np.array([sha[k] for k in sorted(sha)])
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bilinear_golden.npz')
np.savez_compressed(dst, **out)
print('wrote', dst, {k: v[:16] for k, v in sha.items()})
