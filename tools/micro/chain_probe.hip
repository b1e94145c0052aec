// Micro-benchmark: per-kernel cost of a dependent same-stream chain of small kernels, launched eagerly against replayed as a
// hipGraph (the forward pass of the C2 step is 25 such launches).  build: hipcc --offload-arch=gfx950 -O2 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(float* a, int n, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? a[i] : 0.f;
  for (int s = 0; s < spin; ++s) v = v * 1.0000001f + 1e-9f;
  if (i < n) a[i] = v;
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 25, blocks = argc > 2 ? atoi(argv[2]) : 512, spin = argc > 3 ? atoi(argv[3]) : 0;
  const int n = blocks * 256;
  float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int j = 0; j < 20; ++j) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, st, a, n, spin);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("eager  N %3d blocks %5d spin %4d : %7.2f us per kernel\n", N, blocks, spin, ms * 1e3 / (20 * N));
  }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, st, a, n, spin);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int j = 0; j < 20; ++j) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("graph  N %3d blocks %5d spin %4d : %7.2f us per kernel\n", N, blocks, spin, ms * 1e3 / (20 * N));
  }
  return 0;
}
