// Micro-benchmark: global -> LDS fill rate per CU with global_load_lds_dwordx4 (and plain loads into VGPRs) for the
// access patterns of the conv patch ring.  One workgroup per CU, NL loader waves, each keeps DEPTH 1-KiB fills in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// pattern 0: contiguous 1 KiB per instruction; 1: 16 pixels x 64 B at 128-B stride (half lines); 2: 8 pixels x 128 B (full lines, 2 pixel rows apart every 18)
template <int DEPTH, int MODE>
__global__ __launch_bounds__(1024) void fill(const char* src, size_t bytes_per_wg, int iters, int pattern, unsigned* sink, int shared) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (shared ? 0 : (size_t)blockIdx.x * bytes_per_wg);   // shared: every workgroup reads the SAME window
  unsigned acc = 0;
  // each wave walks its own interleaved 1-KiB (or 2-KiB span) units
  for (int it = 0; it < iters; it += DEPTH) {
    u32x4 r[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      // 32-bit address arithmetic and a power-of-two window: the address math must stay far below the load issue rate
      const unsigned unit = (unsigned)(it + j) * nw + wave;
      unsigned off;
      if (pattern == 0) off = unit * 1024u + lane * 16u;
      else if (pattern == 1) off = unit * 2048u + (lane >> 2) * 128u + (lane & 3) * 16u;          // 64 B of every 128-B line
      else off = unit * 1024u + lane * 16u + (unit >> 1) * 13312u;                                // full lines, jumps to another image row every 2 KiB
      off &= (unsigned)bytes_per_wg - 1u;
      if (MODE == 0) glds16(base + off, lds + (wave * DEPTH + j) * 1024);
      else r[j] = *reinterpret_cast<const u32x4*>(base + off);
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) acc += r[j][0] ^ r[j][3];
    }
  }
  if (MODE == 0) acc = *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
  if (acc == 0x12345) sink[0] = acc;
}
int main() {
  const int ncu = 256; const size_t per = 1 << 20;        // 1 MiB window per CU: 256 MiB total, larger than the caches
  char* src; unsigned* sink;
  hipMalloc(&src, ncu * per); hipMalloc(&sink, 4); hipMemset(src, 1, ncu * per);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, int nl, int depth, int pattern, const char* name) {
    const int iters = 512;   // units per wave
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(ncu), dim3(64 * nl), nl * depth * 1024, 0, src, per, iters, pattern, sink, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ncu * nl * iters * 1024;
    printf("%-6s loaders %d depth %2d pattern %d : %7.1f us  %6.2f TB/s  %5.1f GB/s/CU\n", name, nl, depth, pattern, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / ncu);
  };
  // L2-resident working set (the filter rows every workgroup re-reads): 256 KiB window per CU
  {
    const size_t small = 256 << 10;
    auto run2 = [&](auto kern, int nl, int depth, const char* name) {
      const int iters = 2048;
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(64 * nl), nl * depth * 1024, 0, src, small, iters, 0, sink, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)ncu * nl * iters * 1024;
      printf("L2-resident %-5s waves %2d depth %2d : %7.1f us  %6.2f TB/s  %5.1f GB/s/CU\n", name, nl, depth, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / ncu);
    };
    for (int nl : {1, 4, 8, 12, 16}) { run2(fill<8, 0>, nl, 8, "glds"); run2(fill<8, 1>, nl, 8, "vgpr"); }
    // the SAME 64 KiB for every workgroup (one layer's 64x64 3x3 filters): L2 / L1 hits
    auto run3 = [&](auto kern, int nl, int depth, size_t win, const char* name) {
      const int iters = 2048;
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(64 * nl), nl * depth * 1024, 0, src, win, iters, 0, sink, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)ncu * nl * iters * 1024;
      printf("shared %4zu KiB %-5s waves %2d depth %2d : %7.1f us  %6.2f TB/s  %5.1f GB/s/CU\n", win >> 10, name, nl, depth, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / ncu);
    };
    for (size_t win : {(size_t)16 << 10, (size_t)64 << 10, (size_t)1024 << 10})
      for (int nl : {1, 4, 12}) { run3(fill<8, 0>, nl, 8, win, "glds"); run3(fill<8, 1>, nl, 8, win, "vgpr"); }
  }
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int nl : {1, 4}) {
      run(fill<4, 0>, nl, 4, pattern, "glds");
      run(fill<16, 0>, nl, 16, pattern, "glds");
      run(fill<32, 0>, nl, 32, pattern, "glds");
      run(fill<16, 1>, nl, 16, pattern, "vgpr");
      run(fill<32, 1>, nl, 32, pattern, "vgpr");
    }
  return 0;
}
