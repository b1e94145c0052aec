// Micro-benchmark: what does the dependent-launch gap of a same-stream kernel chain cost, and can a software dependency
// (producer workgroups count themselves done; consumer workgroups poll that count after their own prologue) remove it?
//   plain      hipLaunchKernelGGL chain (barrier bit: kernel i+1 is dispatched after kernel i has drained)
//   plain+cnt  same, with the done-counter / fence code in the kernels (cost of the protocol itself)
//   anyorder   hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch): no barrier bit; the data dependency is the done-counter alone
// Every poll is bounded (s_memrealtime deadline): a consumer that gives up sets err[0] and the host reports it -- nothing
// can spin forever.  The chain's result is checked (buf[i] == N after N kernels), so a stale read across XCD L2s shows.
// build: hipcc --offload-arch=gfx950 -O2 -o pdl_probe tools/micro/pdl_probe.hip ; run: ./pdl_probe [N] [blocks] [spin]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void link_kernel(const float* in, float* out, int n, int spin, const unsigned* done_prev, unsigned need,
                                                    unsigned* done_me, int* err, int shift) {
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if (done_prev) {
    if (threadIdx.x == 0) {
      const long long t0 = __builtin_amdgcn_s_memrealtime();
      int good = 1;
      while (__hip_atomic_load(done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000LL) { good = 0; break; }      // 200 ms at 100 MHz
      }
      ok = good;
      if (!good) atomicAdd(err, 1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // read what ANOTHER workgroup (another XCD) of the previous kernel wrote
  const int j = (i + shift) % n;
  const float x = in[j];
  float v = x;
  for (int k = 0; k < spin; ++k) v = v * 1.0000001f + 1e-9f;
  if (ok) out[i] = x + 1.f + (v == 12345.678f ? 1.f : 0.f);
  if (done_me) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // write back this XCD's L2 before the count becomes visible
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done_me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 48, blocks = argc > 2 ? atoi(argv[2]) : 512, spin = argc > 3 ? atoi(argv[3]) : 400;
  const int n = blocks * 256;
  float *a, *b; unsigned* cnt; int* err;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&err, 64));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> h(n);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(a, 0, n * 4, st)); CK(hipMemsetAsync(cnt, 0, 4096 * 4, st)); CK(hipMemsetAsync(err, 0, 64, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      float *src = a, *dst = b;
      for (int i = 0; i < N; ++i) {
        const unsigned* dp = (mode >= 1 && i > 0) ? cnt + (i - 1) * 16 : nullptr;
        unsigned* dm = mode >= 1 ? cnt + i * 16 : nullptr;
        const int shift = 256 * 37 + 5;
        if (mode == 2)
          hipExtLaunchKernelGGL(link_kernel, dim3(blocks), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, (const float*)src, dst, n, spin, dp,
                                (unsigned)blocks, dm, err, shift);
        else
          hipLaunchKernelGGL(link_kernel, dim3(blocks), dim3(256), 0, st, (const float*)src, dst, n, spin, dp, (unsigned)blocks, dm, err, shift);
        float* t = src; src = dst; dst = t;
      }
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), src, n * 4, hipMemcpyDeviceToHost));
      int bad = 0; for (int i = 0; i < n; ++i) bad += h[i] != (float)N;
      int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      if (rep) printf("%-10s N %3d blocks %5d spin %5d : %8.1f us total %6.2f us/kernel  wrong %d  timeouts %d\n",
                      mode == 0 ? "plain" : mode == 1 ? "plain+cnt" : "anyorder", N, blocks, spin, ms * 1e3, ms * 1e3 / N, bad, herr);
    }
  }
  return 0;
}
