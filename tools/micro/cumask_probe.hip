// cumask_probe.hip -- which physical CUs does bit i of a hipExtStreamCreateWithCUMask mask enable on MI355X?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cumask_probe.hip -o tools/micro/cumask_probe && tools/micro/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
  }
  long long t0 = clock64();
  while (clock64() - t0 < spin) { }
}

int main() {
  const int NB = 2048;
  uint32_t* d; CK(hipMalloc(&d, NB * 8));
  std::vector<uint32_t> h(NB * 2);
  auto run = [&](hipStream_t st, const char* what) {
    hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, st, d, 20000);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus; int per_xcc[8] = {0};
    for (int i = 0; i < NB; ++i) {
      const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
      const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    for (uint32_t c : cus) per_xcc[(c >> 16) & 7]++;
    printf("%-28s distinct CUs %3zu  per XCC:", what, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2d", per_xcc[x]);
    printf("\n");
  };
  run(0, "default stream");
  struct M { const char* name; uint32_t w[8]; };
  M masks[] = {
    {"bits 0-31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
    {"bits 0-127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
    {"bits 128-255", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
    {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
    {"bits 0-7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
    {"bits 0,8,16,24 of word0", {0x01010101u, 0, 0, 0, 0, 0, 0, 0}},
    {"low 16 of every word", {0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu}},
    {"3 of 4 (0x77777777)", {0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u, 0x77777777u}},
  };
  for (auto& m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
    if (e != hipSuccess) { printf("%-28s create failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    run(st, m.name);
    hipStreamDestroy(st);
  }
  return 0;
}
