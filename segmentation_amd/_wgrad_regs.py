#!/usr/bin/env python
"""Register budget of every conv_wgrad_kernel instance, from the compiled ISA (no GPU needed; tests/test_build.py runs it).

The filter-gradient kernel issues its tile loads as inline asm and waits for them by hand one or two tiles later, so the
compiler does not know that the destination registers are not valid in between.  As long as every value has an
architectural VGPR of its own nothing moves; once an instance needs ~256 of them the register allocator parks values in
AGPRs or in scratch, and a staging register parked before the wait captures a load that has not landed (r02: garbage f32
gradients with a deeper LDS look-ahead, a memory fault in an unused r01 layout).  The check: no scratch, no spills, and
fewer than LIMIT architectural VGPRs in every instance.

    python -m segmentation_amd._wgrad_regs [--limit 250] [--list]      exit code 1 when an instance is over budget"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def instances(extra=()):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    src = os.path.join(ROOT, 'segmentation_amd', 'csrc', 'conv_wgrad.hip')
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'w.s')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only', '-S', '-I' + os.path.join(ROOT, 'include'),
               src, '-o', out] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        text = open(out).read()
    res = []
    for blk in text.split('  - .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        if 'conv_wgrad_kernel' not in name:
            continue
        ag = int(blk.split('\n')[0])
        vg = int(re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1))
        sp = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', blk).group(1))
        ps = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1))
        short = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+conv_wgrad_kernelI|EEvNS_3WgKE$', '', name).replace('ELi', ',').replace('Li', '')
        res.append({'instance': short, 'arch_vgprs': vg - ag if ag else vg, 'agprs': ag, 'spills': sp, 'scratch_bytes': ps})
    return res


def main():
    limit = 250
    if '--limit' in sys.argv:
        limit = int(sys.argv[sys.argv.index('--limit') + 1])
    inst = instances()
    bad = [i for i in inst if i['arch_vgprs'] >= limit or i['spills'] or i['scratch_bytes']]
    if '--list' in sys.argv:
        for i in sorted(inst, key=lambda r: -r['arch_vgprs']):
            print(i)
    for i in bad:
        print('OVER BUDGET', i)
    print('%d conv_wgrad_kernel instances, %d over budget (limit %d architectural VGPRs, no scratch)' % (len(inst), len(bad), limit))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
