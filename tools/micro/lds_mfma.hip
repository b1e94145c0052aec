// Micro-benchmark: MFMA rate of the tiled convolution's inner loop when it runs from LDS alone (no global traffic).
// A 4-wave workgroup, LDS filled once; per "tap" a wave reads FN filter fragments + FM pixel fragments (ds_read_b128 each,
// 1 KiB per wave) and issues FN*FM v_mfma_f32_16x16x32_bf16.  WGS workgroups per CU as in the real kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int FM, int FN, int PREFETCH>
__global__ __launch_bounds__(256) void k(float* out, int iters, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x * 16; i < lds_bytes; i += 256 * 16) *reinterpret_cast<f32x4*>(smem + i) = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int base_a = (lr * 64 + ((g ^ ((lr >> 1) & 2)) << 4));
  const int base_b = 36864 + ((wave * 32 + lr) * 64 + ((g ^ ((lr >> 1) & 2)) << 4));
  f32x4 acc[FN][FM];
  for (int a = 0; a < FN; ++a) for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  bf16x8 fa[2][FN], fb[2][FM];
  for (int a = 0; a < FN; ++a) fa[0][a] = *reinterpret_cast<bf16x8*>(smem + base_a + a * 1024);
  for (int b = 0; b < FM; ++b) fb[0][b] = *reinterpret_cast<bf16x8*>(smem + base_b + b * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (PREFETCH) {
#pragma unroll
        for (int a = 0; a < FN; ++a) fa[(tap + 1) & 1][a] = *reinterpret_cast<bf16x8*>(smem + base_a + a * 1024 + ((tap + 1) % 9) * 4096);
#pragma unroll
        for (int b = 0; b < FM; ++b) fb[(tap + 1) & 1][b] = *reinterpret_cast<bf16x8*>(smem + base_b + b * 1024 + ((tap + 1) % 3) * 64 + ((tap + 1) / 3 % 3) * 1152);
      }
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PREFETCH ? (tap & 1) : 0][a], fb[PREFETCH ? (tap & 1) : 0][b], acc[a][b], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < FN; ++a) for (int b = 0; b < FM; ++b) s += acc[a][b][0] + acc[a][b][3];
  if (s == 12345.f) out[0] = s;
}
template <int FM, int FN, int PF>
void run(int wgs_per_cu, const char* name) {
  float* out; hipMalloc(&out, 4);
  const int lds = 48 * 1024, iters = 400;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = k<FM, FN, PF>;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), lds, 0, out, iters, lds);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = 256.0 * wgs_per_cu * 4 * iters * 9 * FM * FN;
  const double flops = mfma * 16 * 16 * 32 * 2;
  const double ldsb = PF ? 256.0 * wgs_per_cu * 4 * iters * 9 * (FM + FN) * 1024 : 0;
  printf("%-28s FM %d FN %d WG/CU %d : %8.1f us  %7.1f TFLOP/s (%.0f %% of 2500)  LDS read %6.1f TB/s = %5.1f B/clk/CU @2.4GHz\n", name, FM, FN, wgs_per_cu, ms * 1e3,
         flops / ms / 1e9, flops / ms / 1e9 / 25.0, ldsb / ms / 1e9, ldsb / ms / 1e3 / 256 / 2.4e3);
  hipFree(out);
}
int main() {
  run<2, 4, 0>(3, "MFMA only (no LDS reads)");
  run<2, 4, 1>(3, "conv loop 32px x 64ch/wave");
  run<2, 4, 1>(2, "conv loop 32px x 64ch/wave");
  run<2, 4, 1>(1, "conv loop 32px x 64ch/wave");
  run<4, 4, 1>(2, "conv loop 64px x 64ch/wave");
  run<4, 4, 1>(1, "conv loop 64px x 64ch/wave");
  run<2, 2, 1>(3, "conv loop 32px x 32ch/wave");
  run<2, 2, 1>(4, "conv loop 32px x 32ch/wave");
  return 0;
}
