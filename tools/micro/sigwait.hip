// Micro-benchmark: three ways for a side stream to start a kernel after main-stream kernel i has finished, and what each costs
// the MAIN stream (a chain of dependent ~20 us kernels, like the dgrad chain of a backward pass):
//   bare     main chain only
//   event    hipEventRecord(main) after kernel i + hipStreamWaitEvent(side) + side kernel            (what Plan.run does)
//   inkernel kernel i+1 stores i+1 to a flag when it STARTS (in-order queue: kernel i is complete and its end-of-kernel release
//            has happened); side: hipStreamWaitValue32(flag >= i+1) + side kernel.  No packet on the main queue.
// build: hipcc --offload-arch=gfx950 -O2 -o sigwait tools/micro/sigwait.hip ; run: ./sigwait [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void chain_kernel(float* buf, int n, unsigned* flag, unsigned val, int spin) {
  if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? buf[i] : 0.f;
  for (int k = 0; k < spin; ++k) v = v * 1.0000001f + 1e-7f;
  if (i < n) buf[i] = v;
}
__global__ void side_kernel(const float* src, float* dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * 2.f;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 40, n = 1 << 20, spin = argc > 2 ? atoi(argv[2]) : 3000;
  float *a, *b; unsigned* flag;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&flag, 64));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(flag, 0, 64));
  hipStream_t m, s; int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&m, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo));
  hipEvent_t e0, e1, ev[64];
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 64; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  unsigned base = 0;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, m));
      for (int i = 0; i < N; ++i) {
        ++base;
        hipLaunchKernelGGL(chain_kernel, dim3(n / 256), dim3(256), 0, m, a, n, mode == 2 ? flag : nullptr, base, spin);
        if (mode == 1 && i > 0) {            // side kernel i-1 needs main kernel i-1: record was placed after it (below)
        }
        if (mode == 1) { CK(hipEventRecord(ev[i % 64], m)); CK(hipStreamWaitEvent(s, ev[i % 64], 0)); hipLaunchKernelGGL(side_kernel, dim3(n / 256), dim3(256), 0, s, a, b, n); }
        if (mode == 2) { CK(hipStreamWaitValue32(s, flag, base + 1, hipStreamWaitValueGte, 0xffffffffu)); hipLaunchKernelGGL(side_kernel, dim3(n / 256), dim3(256), 0, s, a, b, n); }
      }
      if (mode == 2) { ++base; hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, m, a, 0, flag, base, 0); }   // releases the last waiter
      CK(hipEventRecord(e1, m));
      auto t1 = std::chrono::steady_clock::now();
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("%-9s main chain %7.2f us per kernel on the GPU timeline (host issue %6.2f us per iteration)\n",
                           mode == 0 ? "bare" : mode == 1 ? "event" : "inkernel", ms * 1e3 / N, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
  }
  return 0;
}
