#!/usr/bin/env python
"""Peak calibration on the box this runs on (BASELINE.md section 4 / SURVEY 8(d)): every roofline fraction in
bench.py is quoted against the guide's 2.5 PF dense bf16 / 8 TB/s HBM; this records what the box delivers to reference kernels:

  * hipBLASLt / rocBLAS bf16 GEMM through torch.matmul: 8192^3 and a convolution-like 50 176 x 128 x 1152 (conv3_2 of the 512^2
    shard as an explicit GEMM: pixels x cout x 9 cin), random operands;
  * bare MFMA loops on random register operands and a float4 copy (tools/micro/calib.hip).

Writes gpurun_out/calibration.json (committed copy: profiles/r04_calibration.json); bench.py reads the committed copy for
roofline.peak_calibrated.    python tools/calibrate.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gemm(torch, M, N, K, reps=20):
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(K, N, device='cuda') * 2 - 1).to(torch.bfloat16)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            c = a @ b
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return {'M': M, 'N': N, 'K': K, 'ms': round(best, 4), 'tflops': round(2.0 * M * N * K / best / 1e9, 1)}


def main():
    out = {}
    exe = '/tmp/seg_calib'
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-w', os.path.join(ROOT, 'tools', 'micro', 'calib.hip'), '-o', exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:]); sys.exit(1)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:]); sys.exit(1)
    out['micro'] = json.loads(r.stdout)
    import torch
    out['device'] = torch.cuda.get_device_name(0)
    out['gemm_bf16_torch_matmul'] = [gemm(torch, 8192, 8192, 8192), gemm(torch, 50176, 128, 1152, reps=50), gemm(torch, 50176, 256, 2304, reps=50)]
    n = 1 << 28                                   # 1 GiB of float32 each way through torch's own copy kernel
    a = torch.empty(n, dtype=torch.float32, device='cuda').normal_(); b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    out['copy_torch'] = {'gbs_read_plus_write': round(2.0 * n * 4 * 5 / e0.elapsed_time(e1) / 1e6)}
    m = out['micro']
    out['peak_calibrated'] = {
        'mfma_bf16_tflops': max(v['tflops'] for k, v in m.items() if k.startswith('mfma_')),
        'gemm_bf16_tflops': out['gemm_bf16_torch_matmul'][0]['tflops'],
        'hbm_gbs': max(m['copy_float4']['gbs_read_plus_write'], out['copy_torch']['gbs_read_plus_write']),
        'note': 'bare MFMA loop on random register operands / hipBLASLt 8192^3 bf16 on random data / float4 copy beyond the Infinity Cache',
    }
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'calibration.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
