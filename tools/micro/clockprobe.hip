// In-step clock sampler (diagnostic): ONE lane on a stream of its own records (s_memtime, s_memrealtime) pairs every `gap` ticks of the
// constant 100 MHz real-time counter while a train step runs on the other streams; shader clock = d(memtime) / d(realtime) * 100 MHz.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC clockprobe.hip -o libclockprobe.so      (tools/clock_probe.py builds and loads it)
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(long long* out, int n, int gap) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < n; ++i) {
    const long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    out[2 * i] = t; out[2 * i + 1] = r;
    while (__builtin_amdgcn_s_memrealtime() - r < gap) __builtin_amdgcn_s_sleep(32);     // (bounded: the real-time counter always advances)
  }
}
extern "C" int clock_probe(long long* out, int n, int gap, void* stream) {
  if (!out || n <= 0 || n > 100000 || gap <= 0 || gap > 100000) return -1;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out, n, gap);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
