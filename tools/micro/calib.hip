// Peak calibration on the box (VERDICT r03 item 5 / BASELINE.md section 4): what the MFMA pipes and the HBM deliver to the simplest
// possible kernels, so that every roofline fraction can also be read against a MEASURED ceiling (bench.py: roofline.peak_calibrated).
//   mfma16 / mfma32 : bare v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 loops on RANDOM operands held in registers (zeros
//                     clock higher: MI355X_MICROARCH.md, DVFS give-back), W waves per SIMD, every CU busy; in-kernel clock from
//                     s_memtime / s_memrealtime
//   copy            : float4 copy of a buffer far beyond the 256 MiB Infinity Cache (read + write bytes / time)
// Prints one JSON object.   hipcc --offload-arch=gfx950 -O3 calib.hip -o calib && ./calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline bf16x8 rnd_frag(uint32_t seed) {
  bf16x8 v;
  for (int i = 0; i < 8; ++i) { seed = seed * 1664525u + 1013904223u; v[i] = (__bf16)(((int)(seed >> 9) % 2001 - 1000) * 1e-3f); }
  return v;
}

template <int SHAPE>     // 16: 16x16x32, 32: 32x32x16
__global__ __launch_bounds__(512) void mfma_loop(float* out, long long* stamps, int iters) {
  const uint32_t s0 = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd_frag(s0 + i); b[i] = rnd_frag(s0 * 7u + i); }
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * r], b[j + 2 * r], acc[i * 2 + j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && stamps) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  if (s == 12345.678f) out[0] = s;
}

__global__ void copy4(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) dst[i] = src[i];
}

template <int SHAPE>
void run_mfma(int waves_per_simd, const char* key, bool last) {
  float* out; hipMalloc(&out, 4);
  const int nb = 256, threads = 256 * waves_per_simd;
  long long* st; hipMalloc(&st, nb * 16);
  const int iters = SHAPE == 16 ? 40000 : 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {       // the first repetitions bring the chip to the clock it holds under this load
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(nb), dim3(threads), 0, 0, out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) best = std::min(best, ms);
  }
  std::vector<long long> h(nb * 2); hipMemcpy(h.data(), st, nb * 16, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int i = 0; i < nb; ++i) if (h[2 * i + 1] > 0) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  std::sort(clk.begin(), clk.end());
  const double mfmas = (double)nb * waves_per_simd * 4 * iters * (SHAPE == 16 ? 16 : 8);
  const double flops = mfmas * (SHAPE == 16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);
  const double cyc_per_mfma = clk.empty() ? 0 : (best * 1e-3 * clk[clk.size() / 2] * 1e6) / (iters * (SHAPE == 16 ? 16 : 8) * (double)waves_per_simd);
  printf("  \"%s\": {\"tflops\": %.1f, \"ms\": %.3f, \"clock_mhz_median\": %.0f, \"cycles_per_mfma_per_simd\": %.2f, \"waves_per_simd\": %d}%s\n", key, flops / best / 1e9, best,
         clk.empty() ? 0.0 : clk[clk.size() / 2], cyc_per_mfma, waves_per_simd, last ? "" : ",");
  hipFree(out); hipFree(st);
}

int main() {
  printf("{\n");
  run_mfma<16>(1, "mfma_16x16x32_bf16_1wave", false);
  run_mfma<16>(2, "mfma_16x16x32_bf16_2waves", false);
  run_mfma<32>(1, "mfma_32x32x16_bf16_1wave", false);
  run_mfma<32>(2, "mfma_32x32x16_bf16_2waves", false);
  const size_t bytes = (size_t)2 << 30;        // 2 GiB each way
  f32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(copy4, dim3(256 * 16), dim3(256), 0, 0, a, b, bytes / 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 1) best = std::min(best, ms);
  }
  printf("  \"copy_float4\": {\"gbs_read_plus_write\": %.0f, \"ms\": %.3f, \"bytes_each_way\": %zu}\n}\n", 2.0 * bytes / best / 1e6, best, bytes);
  return 0;
}
