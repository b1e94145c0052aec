// conv_first.hip -- the first 3x3 conv of each model (Cin = input_channel <= 4, K = 9*Cin <= 36).
// /root/reference/models/unet.py:111-116 (conv1_1, VALID) and models/fcn.py:110-115 (conv1, SAME).
// With K = 27 an MFMA tile would be >50 % padding and the layer is HBM-bound on its 32-channel
// output anyway (25 FLOP/B), so it runs on the vector ALU: one thread per output pixel, 32 output
// channels in registers, filters broadcast from LDS, 64-byte-per-lane coalesced stores.
#include "common.h"

namespace {

constexpr int FT = 16;                     // 16x16 output pixels per workgroup
constexpr int FP = FT + 2;                 // patch edge

template <typename T>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* x, int B, int H, int W, int cin, const float* w,
                                                             const float* bias, int cout, int pad, seg_view dst, int Ho, int Wo,
                                                             int relu, int tiles_x, int tiles_y) {
  __shared__ float sx[FP * FP * 4];
  __shared__ __attribute__((aligned(16))) float sw[36 * 32];
  __shared__ float sb[32];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; const int b = t / tiles_y;
  const int co0 = blockIdx.y * 32;
  const int oy0 = ty * FT, ox0 = tx * FT;
  for (int i = tid; i < FP * FP * cin; i += 256) {
    const int c = i % cin, q = i / cin;
    const int iy = oy0 - pad + q / FP, ix = ox0 - pad + q % FP;
    sx[q * 4 + c] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((int64_t)b * H + iy) * W + ix) * cin + c] : 0.f;
  }
  for (int i = tid; i < 9 * cin * 32; i += 256) {
    const int co = i % 32, k = i / 32;            // k = tap*cin + c
    sw[k * 32 + co] = (co0 + co < cout) ? w[(int64_t)k * cout + co0 + co] : 0.f;
  }
  if (tid < 32) sb[tid] = (bias && co0 + tid < cout) ? bias[co0 + tid] : 0.f;
  __syncthreads();
  const int py = tid / FT, px = tid % FT;
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = sb[c];
  for (int u = 0; u < 3; ++u)
    for (int v = 0; v < 3; ++v)
      for (int c = 0; c < cin; ++c) {
        const float xv = sx[((py + u) * FP + px + v) * 4 + c];
        const f32x4* wr = reinterpret_cast<const f32x4*>(sw + ((u * 3 + v) * cin + c) * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 ww = wr[q];
          acc[q * 4 + 0] += xv * ww[0]; acc[q * 4 + 1] += xv * ww[1]; acc[q * 4 + 2] += xv * ww[2]; acc[q * 4 + 3] += xv * ww[3];
        }
      }
  const int oy = oy0 + py, ox = ox0 + px;
  if (oy < Ho && ox < Wo) {
    T* o = reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox) + co0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Vec8<T> ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) { float r = acc[q * 8 + e]; if (relu) r = fmaxf(r, 0.f); ov.set(e, r); }
      ov.store(o + q * 8);
    }
  }
}

// im2col of the raw input for the first layer's filter gradient: dst[b,y,x,k] = x[b, y+u-pad, x+v-pad, c] with
// k = (u*3+v)*cin + c (k < 9*cin, zero above), so that dW(HWIO) = the 1x1 "wgrad" of dst against dZ on the MFMA path.
template <typename T>
__global__ void im2col3x3_kernel(const float* x, int B, int H, int W, int cin, int pad, seg_view dst, int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo * 4;
  const int nk = 9 * cin;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int piece = i & 3; int64_t t = i >> 2;
    const int ox = t % Wo; t /= Wo;
    const int oy = t % Ho; const int b = t / Ho;
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = piece * 8 + e;
      float v = 0.f;
      if (k < nk) {
        const int tap = k / cin, c = k - tap * cin;
        const int iy = oy - pad + tap / 3, ix = ox - pad + tap % 3;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((int64_t)b * H + iy) * W + ix) * cin + c];
      }
      o.set(e, v);
    }
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox) + piece * 8);
  }
}

}  // namespace

extern "C" int seg_im2col3x3(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t pad, const seg_view* dst,
                             int32_t Ho, int32_t Wo, int32_t dtype, void* stream) {
  if (!x || !dst || !dst->ptr || cin < 1 || cin > 3 || B <= 0 || Ho != H + 2 * pad - 2 || Wo != W + 2 * pad - 2) { seg_set_error("im2col3x3: bad args (cin 1..3)"); return SEG_ERR_ARG; }
  if (dst->oy + Ho > dst->H || dst->ox + Wo > dst->W || dst->coff + 32 > dst->cs || dst->cs % 8 || dst->coff % 8) { seg_set_error("im2col3x3: destination window exceeds buffer"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * Ho * Wo * 4;
  int g = (int)((n + 255) / 256); if (g > 16384) g = 16384;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SEG_F32) SEG_LAUNCH(im2col3x3_kernel<float>, dim3(g), dim3(256), 0, st, x, B, H, W, cin, pad, *dst, Ho, Wo);
  else if (dtype == SEG_BF16) SEG_LAUNCH(im2col3x3_kernel<bf16_t>, dim3(g), dim3(256), 0, st, x, B, H, W, cin, pad, *dst, Ho, Wo);
  else { seg_set_error("im2col3x3: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("im2col3x3");
}

extern "C" int seg_conv_first_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const float* w_hwio, const float* bias,
                                  int32_t cout, int32_t pad, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t relu, int32_t dtype,
                                  void* stream) {
  if (!x || !w_hwio || !dst || !dst->ptr || cin < 1 || cin > 4 || cout < 1 || B <= 0) { seg_set_error("conv_first_fwd: bad args (cin must be 1..4)"); return SEG_ERR_ARG; }
  const int cp = cdiv(cout, 32) * 32;
  if (dst->oy + Ho > dst->H || dst->ox + Wo > dst->W || dst->coff + cp > dst->cs || dst->cs % 8 || dst->coff % 8) { seg_set_error("conv_first_fwd: destination window exceeds buffer"); return SEG_ERR_ARG; }
  if (Ho != H + 2 * pad - 2 || Wo != W + 2 * pad - 2) { seg_set_error("conv_first_fwd: inconsistent output extent"); return SEG_ERR_ARG; }
  const int tiles_x = cdiv(Wo, FT), tiles_y = cdiv(Ho, FT);
  dim3 grid(B * tiles_x * tiles_y, cp / 32);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SEG_F32) SEG_LAUNCH(conv_first_fwd_kernel<float>, grid, dim3(256), 0, st, x, B, H, W, cin, w_hwio, bias, cout, pad, *dst, Ho, Wo, relu, tiles_x, tiles_y);
  else if (dtype == SEG_BF16) SEG_LAUNCH(conv_first_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, x, B, H, W, cin, w_hwio, bias, cout, pad, *dst, Ho, Wo, relu, tiles_x, tiles_y);
  else { seg_set_error("conv_first_fwd: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("conv_first_fwd");
}
