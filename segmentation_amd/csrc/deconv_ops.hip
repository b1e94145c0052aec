// deconv_ops.hip -- the ops DeconvModel needs beyond the U-Net / FCN set (/root/reference/models/deconvolution.py:101-178):
// 5x5 stride-2 convolution and transposed convolution (direct correlation on the vector ALU: these layers carry < 15 % of
// the model's 1.5 GMAC per 512x512 image; its 3x3 / 2x2 layers run on the MFMA tiles of conv_fwd.hip / conv_wgrad.hip),
// k x k max-pool, batch norm fused with the ReLU-grad of the convolution in front of it (wavefront reductions, fixed
// summation order), bilinear resize.  NHWC, channels padded to 32, 16-byte vector access over channels.
#include "common.h"

namespace {

inline int grid_for(int64_t n, int per_block = 256, int cap = 16384) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

inline bool view_ok(const seg_view& v, int H, int W, int C) {
  return v.ptr && v.oy >= 0 && v.ox >= 0 && v.oy + H <= v.H && v.ox + W <= v.W && v.coff >= 0 && v.coff + C <= v.cs &&
         (v.cs % 8) == 0 && (v.coff % 8) == 0;
}

// 8 consecutive n-weights of row (u,v,k); entries at or beyond n_log read as 0 (the row is only n_log long)
SEG_DEV void wrow8(const float* wr, int n0, int n_log, float (&o)[8]) {
  if (n0 + 8 <= n_log && ((reinterpret_cast<uintptr_t>(wr + n0) & 15) == 0)) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(wr + n0), b = *reinterpret_cast<const f32x4*>(wr + n0 + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = n0 + e < n_log ? wr[n0 + e] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// direct correlation: forward form (one thread = one y pixel x 8 n-channels)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dconv_fwd_kernel(const seg_dconv_desc d) {
  const int N8 = d.y.c / 8, K8 = (d.xc + 7) / 8;
  const int64_t npix = (int64_t)d.B * d.Hy * d.Wy, total = npix * N8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n8 = (int)(i / npix);                       // slowest: a wave shares its weight rows
    int64_t t = i - (int64_t)n8 * npix;
    const int ox = t % d.Wy; t /= d.Wy;
    const int oy = t % d.Hy; const int b = t / d.Hy;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (d.bias != nullptr && n8 * 8 + e < d.bias_n) ? d.bias[n8 * 8 + e] : 0.f;
    if (n8 * 8 < d.yc) {
      for (int u = 0; u < d.KH; ++u) {
        const int iy = oy * d.stride + u - d.pad_t;
        if (iy < 0 || iy >= d.Hx) continue;
        for (int v = 0; v < d.KW; ++v) {
          const int ix = ox * d.stride + v - d.pad_l;
          if (ix < 0 || ix >= d.Wx) continue;
          const T* xp = reinterpret_cast<const T*>(d.x.ptr) + view_off(d.x, b, iy, ix);
          const float* wt = d.w + u * d.w_su + v * d.w_sv;
          for (int k8 = 0; k8 < K8; ++k8) {
            Vec8<T> xv; xv.load(xp + k8 * 8);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              const int k = k8 * 8 + kk;
              if (k >= d.xc) break;
              float wr[8];
              wrow8(wt + (int64_t)k * d.w_sk, n8 * 8, d.yc, wr);
              const float xs = xv.get(kk);
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[e] = fmaf(xs, wr[e], acc[e]);
            }
          }
        }
      }
    }
    Vec8<T> mk; mk.zero();
    if (d.mask.ptr) mk.load(reinterpret_cast<const T*>(d.mask.ptr) + view_off(d.mask, b, oy, ox) + n8 * 8);
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float r = n8 * 8 + e < d.yc ? acc[e] : 0.f;
      if (d.relu) r = fmaxf(r, 0.f);
      if (d.mask.ptr && !(mk.get(e) > 0.f)) r = 0.f;
      o.set(e, r);
    }
    o.store(reinterpret_cast<T*>(d.y.ptr) + view_off(d.y, b, oy, ox) + n8 * 8);
  }
}

// ------------------------------------------------------------------------------------------
// direct correlation: data-gradient / transposed-forward form (one thread = one x pixel x 8 k-channels)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dconv_bwd_data_kernel(const seg_dconv_desc d) {
  const int K8 = d.x.c / 8, N8 = (d.yc + 7) / 8;
  const int64_t npix = (int64_t)d.B * d.Hx * d.Wx, total = npix * K8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(i / npix);
    int64_t t = i - (int64_t)k8 * npix;
    const int ix = t % d.Wx; t /= d.Wx;
    const int iy = t % d.Hx; const int b = t / d.Hx;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (d.bias != nullptr && k8 * 8 + e < d.bias_n) ? d.bias[k8 * 8 + e] : 0.f;
    if (k8 * 8 < d.xc) {
      for (int u = 0; u < d.KH; ++u) {
        const int ty = iy + d.pad_t - u;
        if (ty < 0 || ty % d.stride) continue;
        const int oy = ty / d.stride;
        if (oy >= d.Hy) continue;
        for (int v = 0; v < d.KW; ++v) {
          const int tx = ix + d.pad_l - v;
          if (tx < 0 || tx % d.stride) continue;
          const int ox = tx / d.stride;
          if (ox >= d.Wy) continue;
          const T* yp = reinterpret_cast<const T*>(d.y.ptr) + view_off(d.y, b, oy, ox);
          const float* wt = d.w + u * d.w_su + v * d.w_sv;
          for (int n8 = 0; n8 < N8; ++n8) {
            Vec8<T> yv; yv.load(yp + n8 * 8);
            float ys[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ys[e] = yv.get(e);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              const int k = k8 * 8 + kk;
              if (k >= d.xc) break;
              float wr[8];
              wrow8(wt + (int64_t)k * d.w_sk, n8 * 8, d.yc, wr);
              float s = acc[kk];
#pragma unroll
              for (int e = 0; e < 8; ++e) s = fmaf(ys[e], wr[e], s);
              acc[kk] = s;
            }
          }
        }
      }
    }
    Vec8<T> mk; mk.zero();
    if (d.mask.ptr) mk.load(reinterpret_cast<const T*>(d.mask.ptr) + view_off(d.mask, b, iy, ix) + k8 * 8);
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float r = k8 * 8 + e < d.xc ? acc[e] : 0.f;
      if (d.relu) r = fmaxf(r, 0.f);
      if (d.mask.ptr && !(mk.get(e) > 0.f)) r = 0.f;
      o.set(e, r);
    }
    o.store(reinterpret_cast<T*>(d.x.ptr) + view_off(d.x, b, iy, ix) + k8 * 8);
  }
}

// ------------------------------------------------------------------------------------------
// filter gradient: one workgroup per (tap, 8 k, 8 n); threads stride over the y pixels with an 8x8 partial sum each, then
// a fixed-order reduction (butterfly inside a wave, the four waves through LDS in wave order) -> bitwise reproducible
// ------------------------------------------------------------------------------------------
SEG_DEV float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void dconv_wgrad_kernel(const seg_dconv_desc d, float* dw, float* db, int db_mode, int nps, float* ws) {
  __shared__ float red[4][72];
  // nps > 1: the pixels are dealt round-robin (in runs of 256) to nps workgroups per (tap, k, n) tile; each writes its partial
  // sums to ws[ps][tap][k][n] (+ bias partials) and dconv_wgrad_reduce_kernel adds the nps slabs in order
  const int tap = blockIdx.x / nps, ps = blockIdx.x % nps, u = tap / d.KW, v = tap % d.KW;
  const int k8 = blockIdx.y, n8 = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[a][e] = 0.f;
  float accb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) accb[e] = 0.f;
  const bool bias1 = db_mode == 1 && tap == 0 && k8 == 0, bias2 = db_mode == 2 && tap == 0 && n8 == 0;
  const int64_t npix = (int64_t)d.B * d.Hy * d.Wy;
  for (int64_t p = (int64_t)ps * 256 + tid; p < npix; p += (int64_t)256 * nps) {
    int64_t t = p;
    const int ox = t % d.Wy; t /= d.Wy;
    const int oy = t % d.Hy; const int b = t / d.Hy;
    Vec8<T> yv; yv.load(reinterpret_cast<const T*>(d.y.ptr) + view_off(d.y, b, oy, ox) + n8 * 8);
    if (bias1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) accb[e] += yv.get(e);
    }
    const int iy = oy * d.stride + u - d.pad_t, ix = ox * d.stride + v - d.pad_l;
    if (iy < 0 || iy >= d.Hx || ix < 0 || ix >= d.Wx) continue;
    Vec8<T> xv; xv.load(reinterpret_cast<const T*>(d.x.ptr) + view_off(d.x, b, iy, ix) + k8 * 8);
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const float xs = xv.get(a);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[a][e] = fmaf(xs, yv.get(e), acc[a][e]);
    }
  }
  if (bias2) {                                       // sum of x over its FULL extent (bias of a transposed convolution)
    const int64_t nx = (int64_t)d.B * d.Hx * d.Wx;
    for (int64_t p = (int64_t)ps * 256 + tid; p < nx; p += (int64_t)256 * nps) {
      int64_t t = p;
      const int ix = t % d.Wx; t /= d.Wx;
      const int iy = t % d.Hx; const int b = t / d.Hx;
      Vec8<T> xv; xv.load(reinterpret_cast<const T*>(d.x.ptr) + view_off(d.x, b, iy, ix) + k8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) accb[e] += xv.get(e);
    }
  }
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float s = wave_sum64(acc[a][e]);
      if (lane == 0) red[wave][a * 8 + e] = s;
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float s = wave_sum64(accb[e]);
    if (lane == 0) red[wave][64 + e] = s;
  }
  __syncthreads();
  if (tid < 72) {
    const float s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    if (nps > 1) {
      const int Kp = gridDim.y * 8, Np = gridDim.z * 8, NT = d.KH * d.KW;
      if (tid < 64) ws[(((int64_t)ps * NT + tap) * Kp + k8 * 8 + tid / 8) * Np + n8 * 8 + tid % 8] = s;
      else if (bias1) ws[(int64_t)nps * NT * Kp * Np + (int64_t)ps * (Kp > Np ? Kp : Np) + n8 * 8 + tid - 64] = s;
      else if (bias2) ws[(int64_t)nps * NT * Kp * Np + (int64_t)ps * (Kp > Np ? Kp : Np) + k8 * 8 + tid - 64] = s;
    } else if (tid < 64) {
      const int k = k8 * 8 + tid / 8, n = n8 * 8 + tid % 8;
      if (k < d.xc && n < d.yc) dw[u * d.w_su + v * d.w_sv + (int64_t)k * d.w_sk + n] = s;
    } else {
      const int e = tid - 64;
      if (bias1 && n8 * 8 + e < d.bias_n) db[n8 * 8 + e] = s;
      if (bias2 && k8 * 8 + e < d.bias_n) db[k8 * 8 + e] = s;
    }
  }
}

__global__ void dconv_wgrad_reduce_kernel(const seg_dconv_desc d, float* dw, float* db, int db_mode, int nps, int Kp, int Np, const float* ws) {
  const int NT = d.KH * d.KW;
  const int64_t total = (int64_t)NT * Kp * Np, slab = total, i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) {
    const int n = (int)(i % Np), k = (int)((i / Np) % Kp), tap = (int)(i / ((int64_t)Np * Kp));
    if (k < d.xc && n < d.yc) {
      float s = 0.f;
      for (int q = 0; q < nps; ++q) s += ws[q * slab + i];
      dw[(tap / d.KW) * d.w_su + (tap % d.KW) * d.w_sv + (int64_t)k * d.w_sk + n] = s;
    }
  }
  const int M = Kp > Np ? Kp : Np;
  if (db_mode && i < d.bias_n) {
    float s = 0.f;
    for (int q = 0; q < nps; ++q) s += ws[nps * slab + (int64_t)q * M + i];
    db[i] = s;
  }
}

// pixel splits of the filter gradient: enough workgroups to fill the chip when the (tap, k, n) tiles alone are few
inline int dconv_wgrad_nps(const seg_dconv_desc& d) {
  const int64_t tiles = (int64_t)d.KH * d.KW * ((d.xc + 7) / 8) * ((d.yc + 7) / 8);
  const int64_t npix = (int64_t)d.B * d.Hy * d.Wy;
  int64_t nps = (2048 + tiles - 1) / tiles;              // ~8 workgroups per CU
  const int64_t cap = npix / 1024;                        // >= 4 pixels per thread
  if (nps > cap) nps = cap;
  if (nps > 256) nps = 256;
  return nps < 1 ? 1 : (int)nps;
}

// ------------------------------------------------------------------------------------------
// k x k / stride k / VALID max-pool (no ReLU-mask fusion: the input is a batch-norm output)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_k_fwd_kernel(seg_view src, seg_view dst, int k, int B, int Ho, int Wo, int C8) {
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int c8 = t % C8; t /= C8;
    const int ox = t % Wo; t /= Wo;
    const int oy = t % Ho; const int b = t / Ho;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int u = 0; u < k; ++u)
      for (int v = 0; v < k; ++v) {
        Vec8<T> x; x.load(reinterpret_cast<const T*>(src.ptr) + view_off(src, b, oy * k + u, ox * k + v) + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], x.get(e));
      }
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, m[e]);
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox) + c8 * 8);
  }
}

// one thread per window position of the ceil grid: routes dpool to the FIRST maximum (row-major), zeroes the rest of the
// window and the trailing rows / columns no window covers
template <typename T>
__global__ void maxpool_k_bwd_kernel(seg_view src, seg_view dpool, seg_view dsrc, int k, int B, int H, int W, int C8) {
  const int Ho = H / k, Wo = W / k, Hw = (H + k - 1) / k, Ww = (W + k - 1) / k;
  const int64_t total = (int64_t)B * Hw * Ww * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int c8 = t % C8; t /= C8;
    const int wx = t % Ww; t /= Ww;
    const int wy = t % Hw; const int b = t / Hw;
    const bool full = wy < Ho && wx < Wo;
    float m[8]; int mi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; mi[e] = -1; }
    Vec8<T> dp; dp.zero();
    if (full) {
      dp.load(reinterpret_cast<const T*>(dpool.ptr) + view_off(dpool, b, wy, wx) + c8 * 8);
      for (int u = 0; u < k; ++u)
        for (int v = 0; v < k; ++v) {
          Vec8<T> x; x.load(reinterpret_cast<const T*>(src.ptr) + view_off(src, b, wy * k + u, wx * k + v) + c8 * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float xv = x.get(e); if (xv > m[e] || mi[e] < 0) { m[e] = xv; mi[e] = u * k + v; } }
        }
    }
    for (int u = 0; u < k; ++u)
      for (int v = 0; v < k; ++v) {
        const int yy = wy * k + u, xx = wx * k + v;
        if (yy >= H || xx >= W) continue;
        Vec8<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, (full && mi[e] == u * k + v) ? dp.get(e) : 0.f);
        o.store(reinterpret_cast<T*>(dsrc.ptr) + view_off(dsrc, b, yy, xx) + c8 * 8);
      }
  }
}

// ------------------------------------------------------------------------------------------
// batch norm (beta only) behind a ReLU: statistics in three fixed-order stages
//   partial  : NB workgroups, each a contiguous pixel range; thread (pixel lane, 8 channels) -> per-workgroup sums in ws
//   final    : one thread per channel adds the NB partials in order (double), derives mean / rstd (fwd) or the two means (bwd)
//   apply    : elementwise
// MODE 0: s1 = sum a, s2 = sum a^2.  MODE 1: s1 = sum dy, s2 = sum dy * xhat.
// ------------------------------------------------------------------------------------------
constexpr int BN_NB = 1024;      // partial-sum rows = workgroups of the statistics pass (256 left three quarters of the resident slots empty on the 512 x 512 maps)

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(seg_view a, seg_view dy, const float* stats, int B, int H, int W, int C, float* ws) {
  __shared__ float red[256][17];
  const int C8 = C / 8, PL = 256 / C8;                  // pixel lanes per workgroup
  const int tid = threadIdx.x, c8 = tid % C8, pl = tid / C8;
  const int64_t npix = (int64_t)B * H * W;
  const int64_t chunk = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = p0 + chunk < npix ? p0 + chunk : npix;
  float s1[8], s2[8], mean[8], rstd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = s2[e] = 0.f; mean[e] = 0.f; rstd[e] = 1.f; }
  if (MODE == 1 && pl < PL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { mean[e] = stats[c8 * 8 + e]; rstd[e] = stats[C + c8 * 8 + e]; }
  }
  if (pl < PL) {
    for (int64_t p = p0 + pl; p < p1; p += PL) {
      int64_t t = p;
      const int x = t % W; t /= W;
      const int y = t % H; const int b = t / H;
      Vec8<T> av; av.load(reinterpret_cast<const T*>(a.ptr) + view_off(a, b, y, x) + c8 * 8);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = av.get(e); s1[e] += v; s2[e] = fmaf(v, v, s2[e]); }
      } else {
        Vec8<T> gv; gv.load(reinterpret_cast<const T*>(dy.ptr) + view_off(dy, b, y, x) + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float g = gv.get(e); s1[e] += g; s2[e] = fmaf(g, (av.get(e) - mean[e]) * rstd[e], s2[e]); }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid][e] = s1[e]; red[tid][8 + e] = s2[e]; }
  __syncthreads();
  for (int idx = tid; idx < C8 * 16; idx += 256) {     // one (channel group, which of its 16 sums) each: pixel lanes added in order
    const int cc = idx / 16, j = idx % 16;
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * C8 + cc][j];
    ws[((int64_t)blockIdx.x * C + cc * 8 + (j & 7)) * 2 + (j >> 3)] = s;
  }
}

constexpr int BNF_SL = 128;            // row slices per workgroup of bn_final_kernel (x 8 channels = 1024 threads)
template <int MODE>
__global__ __launch_bounds__(1024) void bn_final_kernel(const float* ws, int nb, int C, int c_log, double inv_n, float eps, float decay, int training,
                                                        float* moving, float* stats, float* out2, float* dbeta, int dbeta_add) {
  // 8 channels x 128 slices per workgroup: a thread sums every 128th partial row (<= 8 loads), the slices meet in LDS in two fixed
  // stages.  (One thread per channel walking all nb rows was 60 us of pure load latency per batch norm; 32 channels x 8 slices 15 us;
  // 8 x 32 slices still 7.4 us of 32 dependent-latency loads -- 16 of these launches sit on the critical stream of a DeconvModel step.)
  __shared__ double r1[BNF_SL][8], r2[BNF_SL][8];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int b = sl; b < nb; b += BNF_SL) { s1 += (double)ws[((int64_t)b * C + c) * 2]; s2 += (double)ws[((int64_t)b * C + c) * 2 + 1]; }
  r1[sl][cl] = s1; r2[sl][cl] = s2;
  __syncthreads();
  if (sl < 16) {                          // stage 1: slice groups of 8, in order
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1 += r1[sl * 8 + q][cl]; s2 += r2[sl * 8 + q][cl]; }
  }
  __syncthreads();
  if (sl < 16) { r1[sl][cl] = s1; r2[sl][cl] = s2; }
  __syncthreads();
  if (sl != 0 || c >= C) return;
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { s1 += r1[q][cl]; s2 += r2[q][cl]; }
  if (MODE == 0) {
    float mean, var;
    if (training) {
      const double m = s1 * inv_n;
      double vv = s2 * inv_n - m * m; if (vv < 0.0) vv = 0.0;
      mean = (float)m; var = (float)vv;
      if (moving != nullptr && c < c_log) {
        moving[c] = decay * moving[c] + (1.f - decay) * mean;
        moving[C + c] = decay * moving[C + c] + (1.f - decay) * var;
      }
    } else {
      mean = moving[c]; var = moving[C + c];
    }
    if (c >= c_log) { mean = 0.f; var = 1.f; }          // pad channels: y = 0 * rstd + 0
    stats[c] = mean;
    stats[C + c] = 1.f / sqrtf(var + eps);
  } else {
    out2[c] = (float)(s1 * inv_n);
    out2[C + c] = (float)(s2 * inv_n);
    if (c < c_log) dbeta[c] = dbeta_add ? dbeta[c] + (float)s1 : (float)s1;
  }
}

template <typename T, int MODE>
__global__ void bn_apply_kernel(seg_view a, seg_view in2, seg_view out, const float* stats_g, const float* aux_g, int B, int H, int W, int C8, int C, int c_log) {
  // the per-channel constants go through LDS once per workgroup: read from global memory they were 16-24 scalar loads per
  // 16-byte element and the kernel ran at a quarter of the HBM rate
  extern __shared__ float sst[];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    sst[c] = stats_g[c]; sst[C + c] = stats_g[C + c];
    if (MODE == 0) { sst[2 * C + c] = c < c_log ? aux_g[c] : 0.f; sst[3 * C + c] = 0.f; }
    else { sst[2 * C + c] = aux_g[c]; sst[3 * C + c] = aux_g[C + c]; }
  }
  __syncthreads();
  const float* stats = sst;
  const float* aux = sst + 2 * C;
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8, x, y, b;
    if (i <= 0x7fffffff) {                               // (32-bit divisions: the 64-bit ones were most of this kernel's time)
      unsigned t = (unsigned)i;
      c8 = t % (unsigned)C8; t /= (unsigned)C8;
      x = t % (unsigned)W; t /= (unsigned)W;
      y = t % (unsigned)H; b = t / (unsigned)H;
    } else {
      int64_t t = i;
      c8 = t % C8; t /= C8;
      x = t % W; t /= W;
      y = t % H; b = (int)(t / H);
    }
    Vec8<T> av; av.load(reinterpret_cast<const T*>(a.ptr) + view_off(a, b, y, x) + c8 * 8);
    Vec8<T> o;
    if (MODE == 0) {                                     // aux = beta (c_log entries: pad channels stay 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const int c = c8 * 8 + e; o.set(e, c < c_log ? (av.get(e) - stats[c]) * stats[C + c] + aux[c] : 0.f); }
    } else {                                             // aux = {mean(dy), mean(dy * xhat)}
      Vec8<T> gv; gv.load(reinterpret_cast<const T*>(in2.ptr) + view_off(in2, b, y, x) + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        const float xh = (av.get(e) - stats[c]) * stats[C + c];
        const float da = stats[C + c] * (gv.get(e) - aux[c] - xh * aux[C + c]);
        o.set(e, av.get(e) > 0.f ? da : 0.f);
      }
    }
    o.store(reinterpret_cast<T*>(out.ptr) + view_off(out, b, y, x) + c8 * 8);
  }
}

// ------------------------------------------------------------------------------------------
// tf.image.resize_bilinear, align_corners=False
// ------------------------------------------------------------------------------------------
SEG_DEV void lerp_coord(int o, float scale, int n_in, int& i0, int& i1, float& l) {
  const float f = (float)o * scale;
  i0 = (int)floorf(f);
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  l = f - floorf(f);
}

// Batch norm + the k x k / stride-k max-pool behind it in one pass (DeconvModel bn1 -> pool 2x2, bn2 / bn3 -> pool 3x3,
// models/deconvolution.py:50-75): the normalisation is an increasing map per channel (rstd > 0) and so is the rounding to T, hence
// max over the window of round(bn(a)) == round(bn(max a)) bit for bit -- the normalised full-resolution tensor is never written or
// read (it had no other reader: the pool's backward routes to the first strict maximum of `a`, a maximum of the normalised tensor as
// well; where bf16 rounding makes a tie of distinct `a` values the unfused pair would pick the first tied pixel instead -- see the
// header).
template <typename T>
__global__ void bn_pool_apply_kernel(seg_view a, seg_view out, const float* stats_g, const float* beta_g, int k, int B, int Ho, int Wo, int C8, int C, int c_log) {
  extern __shared__ float sst[];
  for (int c = threadIdx.x; c < C; c += blockDim.x) { sst[c] = stats_g[c]; sst[C + c] = stats_g[C + c]; sst[2 * C + c] = c < c_log ? beta_g[c] : 0.f; }
  __syncthreads();
  const float* stats = sst;
  const float* aux = sst + 2 * C;
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8, x, y, b;
    if (i <= 0x7fffffff) {
      unsigned t = (unsigned)i;
      c8 = t % (unsigned)C8; t /= (unsigned)C8;
      x = t % (unsigned)Wo; t /= (unsigned)Wo;
      y = t % (unsigned)Ho; b = t / (unsigned)Ho;
    } else {
      int64_t t = i;
      c8 = t % C8; t /= C8;
      x = t % Wo; t /= Wo;
      y = t % Ho; b = (int)(t / Ho);
    }
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int u = 0; u < k; ++u)
      for (int v = 0; v < k; ++v) {
        Vec8<T> av; av.load(reinterpret_cast<const T*>(a.ptr) + view_off(a, b, y * k + u, x * k + v) + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], av.get(e));
      }
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int c = c8 * 8 + e; o.set(e, c < c_log ? (m[e] - stats[c]) * stats[C + c] + aux[c] : 0.f); }
    o.store(reinterpret_cast<T*>(out.ptr) + view_off(out, b, y, x) + c8 * 8);
  }
}

// Backward of that pair in two passes over `a` instead of three over `a` and two over the pool's full-resolution gradient: the
// gradient behind the batch norm is dpool at the first maximum of each window and zero elsewhere, so
//   sum dy = sum dpool,   sum dy * xhat = sum_windows dpool * xhat(max a)          (statistics: windows only)
//   dz = [a > 0] * rstd * (dy - mean(dy) - xhat * mean(dy * xhat))                  (every pixel, dy rebuilt per window)
// and that gradient tensor is never written (seg_maxpool_k_bwd + seg_bn_relu_bwd: 486 MB for the DeconvModel's bn1 at 16 x 512^2,
// here 235 MB).  The means are over ALL B*H*W pixels, as in seg_bn_relu_bwd.
template <typename T>
__global__ __launch_bounds__(256) void bn_pool_partial_bwd_kernel(seg_view a, seg_view dpool, const float* stats, int k, int B, int Hp, int Wp, int C, float* ws) {
  __shared__ float red[256][17];
  const int C8 = C / 8, PL = 256 / C8;
  const int tid = threadIdx.x, c8 = tid % C8, pl = tid / C8;
  const int64_t nwin = (int64_t)B * Hp * Wp;
  const int64_t chunk = (nwin + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * chunk, p1 = p0 + chunk < nwin ? p0 + chunk : nwin;
  float s1[8], s2[8], mean[8], rstd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = s2[e] = 0.f; mean[e] = 0.f; rstd[e] = 1.f; }
  if (pl < PL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { mean[e] = stats[c8 * 8 + e]; rstd[e] = stats[C + c8 * 8 + e]; }
    for (int64_t p = p0 + pl; p < p1; p += PL) {
      int64_t t = p;
      const int wx = t % Wp; t /= Wp;
      const int wy = t % Hp; const int b = t / Hp;
      float m[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
      for (int u = 0; u < k; ++u)
        for (int v = 0; v < k; ++v) {
          Vec8<T> av; av.load(reinterpret_cast<const T*>(a.ptr) + view_off(a, b, wy * k + u, wx * k + v) + c8 * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], av.get(e));
        }
      Vec8<T> gv; gv.load(reinterpret_cast<const T*>(dpool.ptr) + view_off(dpool, b, wy, wx) + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float g = gv.get(e); s1[e] += g; s2[e] = fmaf(g, (m[e] - mean[e]) * rstd[e], s2[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[tid][e] = s1[e]; red[tid][8 + e] = s2[e]; }
  __syncthreads();
  for (int idx = tid; idx < C8 * 16; idx += 256) {
    const int cc = idx / 16, j = idx % 16;
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * C8 + cc][j];
    ws[((int64_t)blockIdx.x * C + cc * 8 + (j & 7)) * 2 + (j >> 3)] = s;
  }
}

template <typename T>
__global__ void bn_pool_apply_bwd_kernel(seg_view a, seg_view dpool, seg_view dz, const float* stats_g, const float* means_g, int k, int B, int H, int W,
                                         int C8, int C) {
  extern __shared__ float sst[];
  for (int c = threadIdx.x; c < C; c += blockDim.x) { sst[c] = stats_g[c]; sst[C + c] = stats_g[C + c]; sst[2 * C + c] = means_g[c]; sst[3 * C + c] = means_g[C + c]; }
  __syncthreads();
  const float* stats = sst;
  const float* aux = sst + 2 * C;
  const int Ho = H / k, Wo = W / k, Hw = (H + k - 1) / k, Ww = (W + k - 1) / k;
  const int64_t total = (int64_t)B * Hw * Ww * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8, wx, wy, b;
    if (i <= 0x7fffffff) {
      unsigned t = (unsigned)i;
      c8 = t % (unsigned)C8; t /= (unsigned)C8;
      wx = t % (unsigned)Ww; t /= (unsigned)Ww;
      wy = t % (unsigned)Hw; b = t / (unsigned)Hw;
    } else {
      int64_t t = i;
      c8 = t % C8; t /= C8;
      wx = t % Ww; t /= Ww;
      wy = t % Hw; b = (int)(t / Hw);
    }
    const bool full = wy < Ho && wx < Wo;
    float m[8]; int mi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; mi[e] = -1; }
    Vec8<T> dp; dp.zero();
    if (full) {
      dp.load(reinterpret_cast<const T*>(dpool.ptr) + view_off(dpool, b, wy, wx) + c8 * 8);
      for (int u = 0; u < k; ++u)
        for (int v = 0; v < k; ++v) {
          Vec8<T> x; x.load(reinterpret_cast<const T*>(a.ptr) + view_off(a, b, wy * k + u, wx * k + v) + c8 * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float xv = x.get(e); if (xv > m[e] || mi[e] < 0) { m[e] = xv; mi[e] = u * k + v; } }
        }
    }
    for (int u = 0; u < k; ++u)
      for (int v = 0; v < k; ++v) {
        const int yy = wy * k + u, xx = wx * k + v;
        if (yy >= H || xx >= W) continue;
        Vec8<T> av; av.load(reinterpret_cast<const T*>(a.ptr) + view_off(a, b, yy, xx) + c8 * 8);      // (second touch: L1 / L2)
        Vec8<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = c8 * 8 + e;
          const float dy = (full && mi[e] == u * k + v) ? dp.get(e) : 0.f;
          const float xh = (av.get(e) - stats[c]) * stats[C + c];
          const float da = stats[C + c] * (dy - aux[c] - xh * aux[C + c]);
          o.set(e, av.get(e) > 0.f ? da : 0.f);
        }
        o.store(reinterpret_cast<T*>(dz.ptr) + view_off(dz, b, yy, xx) + c8 * 8);
      }
  }
}

template <typename T>
__global__ void resize_fwd_kernel(seg_view src, int Hs, int Ws, seg_view dst, int Hd, int Wd, int B, int C8) {
  const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
  const int64_t total = (int64_t)B * Hd * Wd * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int c8 = t % C8; t /= C8;
    const int x = t % Wd; t /= Wd;
    const int y = t % Hd; const int b = t / Hd;
    int y0, y1, x0, x1; float ly, lx;
    lerp_coord(y, sy, Hs, y0, y1, ly); lerp_coord(x, sx, Ws, x0, x1, lx);
    const T* sp = reinterpret_cast<const T*>(src.ptr);
    Vec8<T> tl, tr, bl, br;
    tl.load(sp + view_off(src, b, y0, x0) + c8 * 8); tr.load(sp + view_off(src, b, y0, x1) + c8 * 8);
    bl.load(sp + view_off(src, b, y1, x0) + c8 * 8); br.load(sp + view_off(src, b, y1, x1) + c8 * 8);
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float top = tl.get(e) + (tr.get(e) - tl.get(e)) * lx, bot = bl.get(e) + (br.get(e) - bl.get(e)) * lx;
      o.set(e, top + (bot - top) * ly);
    }
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, y, x) + c8 * 8);
  }
}

// adjoint in gather form: source pixel (iy, ix) collects every destination pixel whose 2x2 neighbourhood contains it
template <typename T>
__global__ void resize_bwd_kernel(seg_view dd, int Hd, int Wd, seg_view ds, int Hs, int Ws, int B, int C8) {
  const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
  const int64_t total = (int64_t)B * Hs * Ws * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int c8 = t % C8; t /= C8;
    const int ix = t % Ws; t /= Ws;
    const int iy = t % Hs; const int b = t / Hs;
    int ya = (int)floorf((float)(iy - 1) / sy) - 1, yb = (int)ceilf((float)(iy + 1) / sy) + 1;
    int xa = (int)floorf((float)(ix - 1) / sx) - 1, xb = (int)ceilf((float)(ix + 1) / sx) + 1;
    if (ya < 0) ya = 0; if (xa < 0) xa = 0; if (yb > Hd - 1) yb = Hd - 1; if (xb > Wd - 1) xb = Wd - 1;
    if (iy == Hs - 1) yb = Hd - 1;                       // clamped upper neighbours all land on the last row / column
    if (ix == Ws - 1) xb = Wd - 1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int y = ya; y <= yb; ++y) {
      int y0, y1; float ly;
      lerp_coord(y, sy, Hs, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int x = xa; x <= xb; ++x) {
        int x0, x1; float lx;
        lerp_coord(x, sx, Ws, x0, x1, lx);
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        Vec8<T> g; g.load(reinterpret_cast<const T*>(dd.ptr) + view_off(dd, b, y, x) + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(g.get(e), wy * wx, acc[e]);
      }
    }
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, acc[e]);
    o.store(reinterpret_cast<T*>(ds.ptr) + view_off(ds, b, iy, ix) + c8 * 8);
  }
}

// ------------------------------------------------------------------------------------------
// Dense layers (slim.fully_connected = a 1x1 "convolution" over 1x1 maps; the adversary's adv_fc1 is [B, 3528] x [3528, 1024],
// models/basemodel.py:255,260): weight-bandwidth-bound (14 MB of fp32 weights against 58 MFLOP), so the three forms below
// stream the weight matrix w[k][n] exactly once per 16 rows of the batch with coalesced reads along n and keep a fixed
// summation order.  The generic direct kernel took 1.9 ms for the forward alone (one thread walking all of k).
// ------------------------------------------------------------------------------------------
constexpr int FC_RB = 16;          // batch rows held in registers per pass

// y[b][n] = act(sum_k x[b][k] w[k][n] + bias[n]).  Block = 64 n-columns (a wave reads 256 contiguous bytes of a weight row)
// x 16 k-lanes.  The 16 batch rows of a 256-deep k chunk are staged in LDS as float [k][16 rows], so that a k step costs one
// weight load and four broadcast ds_read_b128 instead of sixteen uniform global loads (the first form of this kernel spent
// its time issuing those).  The k-lanes are reduced through LDS in lane order.
constexpr int FC_KC = 256;
// 16 columns x 64 k-lanes per workgroup (was 64 x 16: the adversary's [*, 3528] x [3528, 1024] layer was 16 workgroups on a
// 256-CU chip and took 45 us for 14 MB of weights)
constexpr int FC_NB = 16, FC_KL = 1024 / FC_NB;
template <typename T>
__global__ __launch_bounds__(1024) void fc_fwd_kernel(const seg_dconv_desc d) {
  extern __shared__ float fc_lds[];
  float* xs_ = fc_lds;                                  // [FC_KC][FC_RB]
  float* red = fc_lds + FC_KC * FC_RB;                  // [FC_KL k-lanes][FC_NB n][FC_RB + 1]
  const int tid = threadIdx.x, nl = tid % FC_NB, kl = tid / FC_NB;
  const int n = blockIdx.x * FC_NB + nl;
  const T* xp = reinterpret_cast<const T*>(d.x.ptr) + view_off(d.x, 0, 0, 0);
  T* yp = reinterpret_cast<T*>(d.y.ptr) + view_off(d.y, 0, 0, 0);
  const int64_t xs = (int64_t)d.x.H * d.x.W * d.x.cs, ys = (int64_t)d.y.H * d.y.W * d.y.cs;
  for (int b0 = 0; b0 < d.B; b0 += FC_RB) {
    float acc[FC_RB];
#pragma unroll
    for (int r = 0; r < FC_RB; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < d.xc; k0 += FC_KC) {
      __syncthreads();
      for (int i = tid; i < FC_KC * FC_RB; i += 1024) {     // coalesced along k within a row
        const int r = i / FC_KC, kk = i % FC_KC;
        xs_[kk * FC_RB + r] = (b0 + r < d.B && k0 + kk < d.xc) ? to_f32(xp[(b0 + r) * xs + k0 + kk]) : 0.f;
      }
      __syncthreads();
      if (n < d.yc) {
        const int kend = min(FC_KC, d.xc - k0);
#pragma unroll 4
        for (int kk = kl; kk < kend; kk += FC_KL) {
          const float wv = d.w[(int64_t)(k0 + kk) * d.w_sk + n];
          const f32x4* xr = reinterpret_cast<const f32x4*>(xs_ + kk * FC_RB);
#pragma unroll
          for (int q = 0; q < FC_RB / 4; ++q) {
            const f32x4 v = xr[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q * 4 + e] = fmaf(v[e], wv, acc[q * 4 + e]);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FC_RB; ++r) red[(kl * FC_NB + nl) * (FC_RB + 1) + r] = acc[r];
    __syncthreads();
    if (tid < FC_NB * FC_RB) {                          // thread (n2, r): one output element, k-lanes added in order
      const int n2l = tid % FC_NB, r = tid / FC_NB, n2 = blockIdx.x * FC_NB + n2l;
      float s = 0.f;
#pragma unroll 8
      for (int q = 0; q < FC_KL; ++q) s += red[(q * FC_NB + n2l) * (FC_RB + 1) + r];
      if (b0 + r < d.B && n2 < d.y.c) {
        if (n2 < d.yc) { if (d.bias != nullptr && n2 < d.bias_n) s += d.bias[n2]; if (d.relu) s = fmaxf(s, 0.f); } else s = 0.f;
        yp[(b0 + r) * ys + n2] = from_f32<T>(s);
      }
    }
  }
}

// dx[b][k] = sum_n dz[b][n] w[k][n]: one wave per k-row, lanes along n, wave reduction per batch row
template <typename T>
__global__ __launch_bounds__(256) void fc_bwd_data_kernel(const seg_dconv_desc d) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= d.x.c) return;
  const T* zp = reinterpret_cast<const T*>(d.y.ptr) + view_off(d.y, 0, 0, 0);
  T* xp = reinterpret_cast<T*>(d.x.ptr) + view_off(d.x, 0, 0, 0);
  const int64_t xs = (int64_t)d.x.H * d.x.W * d.x.cs, zs = (int64_t)d.y.H * d.y.W * d.y.cs;
  for (int b0 = 0; b0 < d.B; b0 += FC_RB) {
    float acc[FC_RB];
#pragma unroll
    for (int r = 0; r < FC_RB; ++r) acc[r] = 0.f;
    if (k < d.xc) {
      for (int n = lane; n < d.yc; n += 64) {
        const float wv = d.w[(int64_t)k * d.w_sk + n];
#pragma unroll
        for (int r = 0; r < FC_RB; ++r)
          if (b0 + r < d.B) acc[r] = fmaf(to_f32(zp[(b0 + r) * zs + n]), wv, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < FC_RB; ++r) {
      const float s = wave_sum64(acc[r]);
      if (lane == 0 && b0 + r < d.B) xp[(b0 + r) * xs + k] = from_f32<T>(s);
    }
  }
}

// dw[k][n] = sum_b x[b][k] dz[b][n] (batch rows added in order), db[n] = sum_b dz[b][n]; thread = one n, block = 8 k-rows
template <typename T>
__global__ __launch_bounds__(256) void fc_wgrad_kernel(const seg_dconv_desc d, float* dw, float* db) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * 8;
  if (n >= d.yc) return;
  const T* xp = reinterpret_cast<const T*>(d.x.ptr) + view_off(d.x, 0, 0, 0);
  const T* zp = reinterpret_cast<const T*>(d.y.ptr) + view_off(d.y, 0, 0, 0);
  const int64_t xs = (int64_t)d.x.H * d.x.W * d.x.cs, zs = (int64_t)d.y.H * d.y.W * d.y.cs;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  float sb = 0.f;
  for (int b = 0; b < d.B; ++b) {
    const float zv = to_f32(zp[b * zs + n]);
    sb += zv;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k0 + j < d.xc) s[j] = fmaf(to_f32(xp[b * xs + k0 + j]), zv, s[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (k0 + j < d.xc) dw[(int64_t)(k0 + j) * d.w_sk + n] = s[j];
  if (db != nullptr && blockIdx.y == 0 && n < d.bias_n) db[n] = sb;
}

inline bool is_dense(const seg_dconv_desc& d) {
  return d.KH == 1 && d.KW == 1 && d.stride == 1 && d.Hx == 1 && d.Wx == 1 && d.Hy == 1 && d.Wy == 1 && d.pad_t == 0 && d.pad_l == 0 &&
         d.mask.ptr == nullptr && d.w_sk >= d.yc;
}

int check_dconv(const seg_dconv_desc* dp, const char* what) {
  if (!dp) { seg_set_error("%s: null descriptor", what); return SEG_ERR_ARG; }
  const seg_dconv_desc& d = *dp;
  if (d.B <= 0 || d.Hx <= 0 || d.Wx <= 0 || d.Hy <= 0 || d.Wy <= 0 || d.KH <= 0 || d.KW <= 0 || d.stride <= 0) { seg_set_error("%s: empty extent", what); return SEG_ERR_ARG; }
  if (!view_ok(d.x, d.Hx, d.Wx, d.x.c) || !view_ok(d.y, d.Hy, d.Wy, d.y.c) || d.x.c <= 0 || d.x.c % 8 || d.y.c <= 0 || d.y.c % 8) { seg_set_error("%s: bad x / y view", what); return SEG_ERR_ARG; }
  if (d.xc <= 0 || d.xc > d.x.c || d.yc <= 0 || d.yc > d.y.c || !d.w) { seg_set_error("%s: bad channel counts / null weights", what); return SEG_ERR_ARG; }
  if (d.dtype != SEG_F32 && d.dtype != SEG_BF16) { seg_set_error("%s: bad dtype %d", what, d.dtype); return SEG_ERR_ARG; }
  // every y pixel must read only taps that the stride / padding arithmetic places (partly) inside x or in its zero padding
  if ((d.Hy - 1) * d.stride - d.pad_t >= d.Hx || (d.Wy - 1) * d.stride - d.pad_l >= d.Wx) { seg_set_error("%s: y extent %dx%d does not fit x %dx%d at stride %d", what, d.Hy, d.Wy, d.Hx, d.Wx, d.stride); return SEG_ERR_ARG; }
  return SEG_OK;
}

}  // namespace

extern "C" int seg_dconv_fwd(const seg_dconv_desc* dp, void* stream) {
  if (int rc = check_dconv(dp, "dconv_fwd")) return rc;
  const seg_dconv_desc& d = *dp;
  if (d.mask.ptr && !view_ok(d.mask, d.Hy, d.Wy, d.y.c)) { seg_set_error("dconv_fwd: bad mask view"); return SEG_ERR_ARG; }
  if (is_dense(d)) {
    const dim3 g((d.y.c + FC_NB - 1) / FC_NB);
    constexpr int lds = (FC_KC * FC_RB + FC_KL * FC_NB * (FC_RB + 1)) * 4;      // 84 KB
    static bool attr_done = false;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fc_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(fc_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        seg_set_error("fc_fwd: cannot raise dynamic LDS to %d", lds); return SEG_ERR_LAUNCH;
      }
      attr_done = true;
    }
    if (d.dtype == SEG_F32) SEG_LAUNCH(fc_fwd_kernel<float>, g, dim3(1024), lds, (hipStream_t)stream, d);
    else SEG_LAUNCH(fc_fwd_kernel<bf16_t>, g, dim3(1024), lds, (hipStream_t)stream, d);
    return seg_check_launch("fc_fwd");
  }
  const int g = grid_for((int64_t)d.B * d.Hy * d.Wy * (d.y.c / 8));
  if (d.dtype == SEG_F32) SEG_LAUNCH(dconv_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, d);
  else SEG_LAUNCH(dconv_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, d);
  return seg_check_launch("dconv_fwd");
}

extern "C" int seg_dconv_bwd_data(const seg_dconv_desc* dp, void* stream) {
  if (int rc = check_dconv(dp, "dconv_bwd_data")) return rc;
  const seg_dconv_desc& d = *dp;
  if (d.mask.ptr && !view_ok(d.mask, d.Hx, d.Wx, d.x.c)) { seg_set_error("dconv_bwd_data: bad mask view"); return SEG_ERR_ARG; }
  if (is_dense(d) && d.bias == nullptr && !d.relu) {
    const dim3 g((d.x.c + 3) / 4);
    if (d.dtype == SEG_F32) SEG_LAUNCH(fc_bwd_data_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, d);
    else SEG_LAUNCH(fc_bwd_data_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, d);
    return seg_check_launch("fc_bwd_data");
  }
  const int g = grid_for((int64_t)d.B * d.Hx * d.Wx * (d.x.c / 8));
  if (d.dtype == SEG_F32) SEG_LAUNCH(dconv_bwd_data_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, d);
  else SEG_LAUNCH(dconv_bwd_data_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, d);
  return seg_check_launch("dconv_bwd_data");
}

extern "C" int64_t seg_dconv_wgrad_ws_bytes(const seg_dconv_desc* dp) {
  if (!dp || is_dense(*dp)) return 0;
  const seg_dconv_desc& d = *dp;
  const int nps = dconv_wgrad_nps(d);
  if (nps <= 1) return 0;
  const int64_t Kp = (d.xc + 7) / 8 * 8, Np = (d.yc + 7) / 8 * 8;
  return (int64_t)nps * ((int64_t)d.KH * d.KW * Kp * Np + (Kp > Np ? Kp : Np)) * 4;
}

extern "C" int seg_dconv_wgrad(const seg_dconv_desc* dp, float* dw, float* db, int32_t db_mode, float* ws, int64_t ws_bytes, void* stream) {
  if (int rc = check_dconv(dp, "dconv_wgrad")) return rc;
  const seg_dconv_desc& d = *dp;
  if (!dw || db_mode < 0 || db_mode > 2 || (db_mode && (!db || d.bias_n <= 0))) { seg_set_error("dconv_wgrad: bad output / bias request"); return SEG_ERR_ARG; }
  if ((db_mode == 1 && d.bias_n > d.yc) || (db_mode == 2 && d.bias_n > d.xc)) { seg_set_error("dconv_wgrad: bias_n exceeds channels"); return SEG_ERR_ARG; }
  if (is_dense(d) && db_mode != 2) {
    const dim3 g((d.yc + 255) / 256, (d.xc + 7) / 8);
    if (d.dtype == SEG_F32) SEG_LAUNCH(fc_wgrad_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, d, dw, db_mode ? db : (float*)nullptr);
    else SEG_LAUNCH(fc_wgrad_kernel<bf16_t>, g, dim3(256), 0, (hipStream_t)stream, d, dw, db_mode ? db : (float*)nullptr);
    return seg_check_launch("fc_wgrad");
  }
  // (a caller without workspace gets the unsplit form: same result up to summation order)
  int nps = dconv_wgrad_nps(d);
  if (nps > 1 && (!ws || ws_bytes < seg_dconv_wgrad_ws_bytes(dp))) nps = 1;
  const int K8 = (d.xc + 7) / 8, N8 = (d.yc + 7) / 8;
  const dim3 grid(d.KH * d.KW * nps, K8, N8);
  if (d.dtype == SEG_F32) SEG_LAUNCH(dconv_wgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, d, dw, db, db_mode, nps, ws);
  else SEG_LAUNCH(dconv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, d, dw, db, db_mode, nps, ws);
  if (int rc = seg_check_launch("dconv_wgrad")) return rc;
  if (nps > 1) {
    const int64_t total = (int64_t)d.KH * d.KW * K8 * 8 * N8 * 8;
    const int64_t n = total > d.bias_n ? total : d.bias_n;
    SEG_LAUNCH(dconv_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, dw, db, db_mode, nps, K8 * 8, N8 * 8, (const float*)ws);
    return seg_check_launch("dconv_wgrad_reduce");
  }
  return SEG_OK;
}

extern "C" int seg_maxpool_k_fwd(const seg_view* src, const seg_view* dst, int32_t k, int32_t B, int32_t Ho, int32_t Wo, int32_t C,
                                 int32_t dtype, void* stream) {
  if (!src || !dst || k < 1 || k > 8 || C <= 0 || C % 8 || !view_ok(*src, Ho * k, Wo * k, C) || !view_ok(*dst, Ho, Wo, C)) { seg_set_error("maxpool_k_fwd: bad arguments"); return SEG_ERR_ARG; }
  const int g = grid_for((int64_t)B * Ho * Wo * (C / 8));
  if (dtype == SEG_F32) SEG_LAUNCH(maxpool_k_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, *src, *dst, k, B, Ho, Wo, C / 8);
  else if (dtype == SEG_BF16) SEG_LAUNCH(maxpool_k_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, *src, *dst, k, B, Ho, Wo, C / 8);
  else { seg_set_error("maxpool_k_fwd: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("maxpool_k_fwd");
}

extern "C" int seg_maxpool_k_bwd(const seg_view* src, const seg_view* dpool, const seg_view* dsrc, int32_t k, int32_t B, int32_t H,
                                 int32_t W, int32_t C, int32_t dtype, void* stream) {
  if (!src || !dpool || !dsrc || k < 1 || k > 8 || C <= 0 || C % 8 || H < k || W < k || !view_ok(*src, H, W, C) || !view_ok(*dsrc, H, W, C) ||
      !view_ok(*dpool, H / k, W / k, C)) { seg_set_error("maxpool_k_bwd: bad arguments"); return SEG_ERR_ARG; }
  const int g = grid_for((int64_t)B * ((H + k - 1) / k) * ((W + k - 1) / k) * (C / 8));
  if (dtype == SEG_F32) SEG_LAUNCH(maxpool_k_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, *src, *dpool, *dsrc, k, B, H, W, C / 8);
  else if (dtype == SEG_BF16) SEG_LAUNCH(maxpool_k_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, *src, *dpool, *dsrc, k, B, H, W, C / 8);
  else { seg_set_error("maxpool_k_bwd: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("maxpool_k_bwd");
}

extern "C" int64_t seg_bn_ws_bytes(int32_t C) { return ((int64_t)BN_NB * C * 2 + 2 * C) * 4; }

static int bn_nb(int64_t npix, int C8) {
  const int pl = 256 / C8;
  int64_t nb = (npix + (int64_t)pl * 16 - 1) / ((int64_t)pl * 16);      // >= 16 pixels per thread before another workgroup pays
  if (nb > BN_NB) nb = BN_NB;
  if (nb < 1) nb = 1;
  return (int)nb;
}

static int bn_fwd_launch(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, int32_t training,
                         float decay, float eps, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t rows, int32_t dtype,
                         void* stream, int32_t pool_k = 0);
extern "C" int seg_bn_fwd(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, int32_t training,
                          float decay, float eps, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t dtype,
                          void* stream) {
  return bn_fwd_launch(a, y, beta, moving, stats, training, decay, eps, B, H, W, C, c_log, ws, 0, dtype, stream);
}
/* seg_bn_fwd whose statistics pass has been done by the launch that produced `a`: ws holds `rows` rows [C][2] of per-channel (sum,
 * sum of squares) over disjoint pixel sets (seg_conv_first_gen_bn, seg_thin_up2x2_bn); training statistics only. */
extern "C" int seg_bn_fwd_rows(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, float decay, float eps,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t rows, int32_t dtype, void* stream) {
  if (rows < 1 || rows > BN_NB) { seg_set_error("bn_fwd_rows: 1..%d rows", BN_NB); return SEG_ERR_ARG; }
  return bn_fwd_launch(a, y, beta, moving, stats, 1, decay, eps, B, H, W, C, c_log, ws, rows, dtype, stream);
}
/* seg_bn_fwd with the k x k / stride-k max-pool (VALID) that consumes it: `pooled` [H/k, W/k] = max_pool(batch_norm(a)), bit for
 * bit what seg_bn_fwd + seg_maxpool_k_fwd write, without the normalised full-resolution tensor (seg_maxpool_k_bwd then takes `a` as
 * its source: the same first maximum).  rows: as seg_bn_fwd_rows (0: statistics pass over `a` here). */
extern "C" int seg_bn_pool_fwd(const seg_view* a, const seg_view* pooled, const float* beta, float* moving, float* stats, int32_t training,
                               float decay, float eps, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t rows,
                               int32_t k, int32_t dtype, void* stream) {
  if (k < 1 || k > 8 || H < k || W < k || rows < 0 || rows > BN_NB || (rows > 0 && !training)) { seg_set_error("bn_pool_fwd: pool 1..8, rows 0..%d (training only)", BN_NB); return SEG_ERR_ARG; }
  return bn_fwd_launch(a, pooled, beta, moving, stats, training, decay, eps, B, H, W, C, c_log, ws, rows, dtype, stream, k);
}
static int bn_fwd_launch(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, int32_t training,
                         float decay, float eps, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t rows, int32_t dtype,
                         void* stream, int32_t pool_k) {
  const int yH = pool_k ? H / pool_k : H, yW = pool_k ? W / pool_k : W;
  const bool stats_only = y == nullptr || y->ptr == nullptr;       // (the consumer normalises on load: seg_thin_conv3x3_bn)
  if (!a || !beta || !stats || !ws || C <= 0 || C % 8 || C > 2048 || c_log <= 0 || c_log > C || !view_ok(*a, H, W, C) || (!stats_only && !view_ok(*y, yH, yW, C)) ||
      (!training && !moving) || (dtype != SEG_F32 && dtype != SEG_BF16)) { seg_set_error("bn_fwd: bad arguments"); return SEG_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int64_t npix = (int64_t)B * H * W;
  const int nb = rows > 0 ? rows : bn_nb(npix, C / 8);
  seg_view none = *a;
  if (training && rows == 0) {
    if (dtype == SEG_F32) SEG_LAUNCH((bn_partial_kernel<float, 0>), dim3(nb), dim3(256), 0, st, *a, none, (const float*)nullptr, B, H, W, C, ws);
    else SEG_LAUNCH((bn_partial_kernel<bf16_t, 0>), dim3(nb), dim3(256), 0, st, *a, none, (const float*)nullptr, B, H, W, C, ws);
    if (int rc = seg_check_launch("bn_partial")) return rc;
  }
  SEG_LAUNCH(bn_final_kernel<0>, dim3((C + 7) / 8), dim3(8 * BNF_SL), 0, st, (const float*)ws, training ? nb : 0, C, c_log, 1.0 / (double)npix, eps, decay,
             training, moving, stats, (float*)nullptr, (float*)nullptr, 0);
  if (int rc = seg_check_launch("bn_final")) return rc;
  if (stats_only) return 0;
  if (pool_k) {
    const int gp = grid_for((int64_t)B * yH * yW * (C / 8));
    if (dtype == SEG_F32) SEG_LAUNCH(bn_pool_apply_kernel<float>, dim3(gp), dim3(256), (size_t)C * 12, st, *a, *y, (const float*)stats, beta, pool_k, B, yH, yW, C / 8, C, c_log);
    else SEG_LAUNCH(bn_pool_apply_kernel<bf16_t>, dim3(gp), dim3(256), (size_t)C * 12, st, *a, *y, (const float*)stats, beta, pool_k, B, yH, yW, C / 8, C, c_log);
    return seg_check_launch("bn_pool_apply");
  }
  const int g = grid_for(npix * (C / 8));
  if (dtype == SEG_F32) SEG_LAUNCH((bn_apply_kernel<float, 0>), dim3(g), dim3(256), (size_t)C * 16, st, *a, none, *y, (const float*)stats, beta, B, H, W, C / 8, C, c_log);
  else SEG_LAUNCH((bn_apply_kernel<bf16_t, 0>), dim3(g), dim3(256), (size_t)C * 16, st, *a, none, *y, (const float*)stats, beta, B, H, W, C / 8, C, c_log);
  return seg_check_launch("bn_apply");
}

extern "C" int seg_bn_relu_bwd(const seg_view* a, const seg_view* dy, const seg_view* dz, const float* stats, float* dbeta, int32_t dbeta_add,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t dtype, void* stream) {
  if (!a || !dy || !dz || !stats || !dbeta || !ws || C <= 0 || C % 8 || C > 2048 || c_log <= 0 || c_log > C || !view_ok(*a, H, W, C) ||
      !view_ok(*dy, H, W, C) || !view_ok(*dz, H, W, C) || (dtype != SEG_F32 && dtype != SEG_BF16)) { seg_set_error("bn_relu_bwd: bad arguments"); return SEG_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int64_t npix = (int64_t)B * H * W;
  const int nb = bn_nb(npix, C / 8);
  float* means = ws + (int64_t)BN_NB * C * 2;
  if (dtype == SEG_F32) SEG_LAUNCH((bn_partial_kernel<float, 1>), dim3(nb), dim3(256), 0, st, *a, *dy, stats, B, H, W, C, ws);
  else SEG_LAUNCH((bn_partial_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, st, *a, *dy, stats, B, H, W, C, ws);
  if (int rc = seg_check_launch("bn_partial_bwd")) return rc;
  SEG_LAUNCH(bn_final_kernel<1>, dim3((C + 7) / 8), dim3(8 * BNF_SL), 0, st, (const float*)ws, nb, C, c_log, 1.0 / (double)npix, 0.f, 0.f, 1,
             (float*)nullptr, (float*)nullptr, means, dbeta, dbeta_add);
  if (int rc = seg_check_launch("bn_final_bwd")) return rc;
  const int g = grid_for(npix * (C / 8));
  if (dtype == SEG_F32) SEG_LAUNCH((bn_apply_kernel<float, 1>), dim3(g), dim3(256), (size_t)C * 16, st, *a, *dy, *dz, stats, (const float*)means, B, H, W, C / 8, C, c_log);
  else SEG_LAUNCH((bn_apply_kernel<bf16_t, 1>), dim3(g), dim3(256), (size_t)C * 16, st, *a, *dy, *dz, stats, (const float*)means, B, H, W, C / 8, C, c_log);
  return seg_check_launch("bn_apply_bwd");
}

/* seg_maxpool_k_bwd (source `a`) + seg_bn_relu_bwd in two passes, without the pool's full-resolution gradient tensor: dz = masked
 * pre-activation gradient of the layer in front of the batch norm, dbeta as seg_bn_relu_bwd; dpool [H/k, W/k]. */
extern "C" int seg_bn_pool_relu_bwd(const seg_view* a, const seg_view* dpool, const seg_view* dz, const float* stats, float* dbeta, int32_t dbeta_add,
                                    int32_t k, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t dtype, void* stream) {
  if (!a || !dpool || !dz || !stats || !dbeta || !ws || k < 1 || k > 8 || H < k || W < k || C <= 0 || C % 8 || C > 2048 || c_log <= 0 || c_log > C ||
      !view_ok(*a, H, W, C) || !view_ok(*dz, H, W, C) || !view_ok(*dpool, H / k, W / k, C) || (dtype != SEG_F32 && dtype != SEG_BF16)) {
    seg_set_error("bn_pool_relu_bwd: bad arguments"); return SEG_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int Hp = H / k, Wp = W / k;
  const int64_t npix = (int64_t)B * H * W, nwin = (int64_t)B * Hp * Wp;
  const int nb = bn_nb(nwin * 4, C / 8);                     // (a window is k*k loads: workgroups pay earlier than per pixel)
  float* means = ws + (int64_t)BN_NB * C * 2;
  if (dtype == SEG_F32) SEG_LAUNCH(bn_pool_partial_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, *a, *dpool, stats, k, B, Hp, Wp, C, ws);
  else SEG_LAUNCH(bn_pool_partial_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, *a, *dpool, stats, k, B, Hp, Wp, C, ws);
  if (int rc = seg_check_launch("bn_pool_partial_bwd")) return rc;
  SEG_LAUNCH(bn_final_kernel<1>, dim3((C + 7) / 8), dim3(8 * BNF_SL), 0, st, (const float*)ws, nb, C, c_log, 1.0 / (double)npix, 0.f, 0.f, 1,
             (float*)nullptr, (float*)nullptr, means, dbeta, dbeta_add);
  if (int rc = seg_check_launch("bn_final_bwd")) return rc;
  const int g = grid_for((int64_t)B * ((H + k - 1) / k) * ((W + k - 1) / k) * (C / 8));
  if (dtype == SEG_F32) SEG_LAUNCH(bn_pool_apply_bwd_kernel<float>, dim3(g), dim3(256), (size_t)C * 16, st, *a, *dpool, *dz, stats, (const float*)means, k, B, H, W, C / 8, C);
  else SEG_LAUNCH(bn_pool_apply_bwd_kernel<bf16_t>, dim3(g), dim3(256), (size_t)C * 16, st, *a, *dpool, *dz, stats, (const float*)means, k, B, H, W, C / 8, C);
  return seg_check_launch("bn_pool_apply_bwd");
}

extern "C" int seg_resize_bilinear_fwd(const seg_view* src, int32_t Hs, int32_t Ws, const seg_view* dst, int32_t Hd, int32_t Wd, int32_t B,
                                       int32_t C, int32_t dtype, void* stream) {
  if (!src || !dst || C <= 0 || C % 8 || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || !view_ok(*src, Hs, Ws, C) || !view_ok(*dst, Hd, Wd, C)) { seg_set_error("resize_bilinear_fwd: bad arguments"); return SEG_ERR_ARG; }
  const int g = grid_for((int64_t)B * Hd * Wd * (C / 8));
  if (dtype == SEG_F32) SEG_LAUNCH(resize_fwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, *src, Hs, Ws, *dst, Hd, Wd, B, C / 8);
  else if (dtype == SEG_BF16) SEG_LAUNCH(resize_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, *src, Hs, Ws, *dst, Hd, Wd, B, C / 8);
  else { seg_set_error("resize_bilinear_fwd: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("resize_bilinear_fwd");
}

extern "C" int seg_resize_bilinear_bwd(const seg_view* ddst, int32_t Hd, int32_t Wd, const seg_view* dsrc, int32_t Hs, int32_t Ws, int32_t B,
                                       int32_t C, int32_t dtype, void* stream) {
  if (!ddst || !dsrc || C <= 0 || C % 8 || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || !view_ok(*dsrc, Hs, Ws, C) || !view_ok(*ddst, Hd, Wd, C)) { seg_set_error("resize_bilinear_bwd: bad arguments"); return SEG_ERR_ARG; }
  const int g = grid_for((int64_t)B * Hs * Ws * (C / 8));
  if (dtype == SEG_F32) SEG_LAUNCH(resize_bwd_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, *ddst, Hd, Wd, *dsrc, Hs, Ws, B, C / 8);
  else if (dtype == SEG_BF16) SEG_LAUNCH(resize_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, *ddst, Hd, Wd, *dsrc, Hs, Ws, B, C / 8);
  else { seg_set_error("resize_bilinear_bwd: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("resize_bilinear_bwd");
}

// ------------------------------------------------------------------------------------------------------------------
// The 5x5/s2 transposed convolutions of the DeconvModel on the MFMA kernels (models/deconvolution.py deconv1_0, deconv2_0,
// deconv2_1).  A transposed convolution T with filter [kh,kw,Cout,Cin] is the adjoint of the strided convolution C with
// the SAME memory read as HWIO [kh,kw,Cin_C = Cout,Cout_C = Cin]; C is a 1x1 convolution over the strided im2col of its
// input.  So:   T.forward(x)  = col2im( 1x1-dgrad_C(x) )        -> seg_col2im below (+ bias, ReLU)
//               T.dgrad(dz)   = 1x1-forward_C( im2col(dz) )     -> seg_im2col_act below
//               T.wgrad       = 1x1-wgrad_C( im2col(dz), x )    (the [1][kh*kw*Cout][Cin] result IS the TF filter layout)
// Both kernels here only move data; the arithmetic runs in conv_fwd.hip / conv_wgrad.hip.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_act_kernel(seg_view src, int B, int Hs, int Ws, int C_, int KH, int KW, int stride, int pad_t, int pad_l, seg_view dst,
                                  int Ho, int Wo, int pieces) {
  const int64_t total = (int64_t)B * Ho * Wo * pieces;
  const int nk = KH * KW * C_;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int piece = (int)(i % pieces);
    int64_t t = i / pieces;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho); const int b = (int)(t / Ho);
    Vec8<T> o;
    const int k0 = piece * 8;
    if ((C_ & 7) == 0 && k0 + 8 <= nk) {               // the eight k share one tap: one 16-byte load
      const int tap = k0 / C_, c = k0 - tap * C_;
      const int iy = oy * stride - pad_t + tap / KW, ix = ox * stride - pad_l + tap % KW;
      if (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) o.load(reinterpret_cast<const T*>(src.ptr) + view_off(src, b, iy, ix) + c);
      else o.zero();
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        float v = 0.f;
        if (k < nk) {
          const int tap = k / C_, c = k - tap * C_;
          const int iy = oy * stride - pad_t + tap / KW, ix = ox * stride - pad_l + tap % KW;
          if (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) v = (float)reinterpret_cast<const T*>(src.ptr)[view_off(src, b, iy, ix) + c];
        }
        o.set(e, v);
      }
    }
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox) + piece * 8);
  }
}

extern "C" int seg_im2col_act(const seg_view* src, int32_t B, int32_t Hs, int32_t Ws, int32_t C_, int32_t KH, int32_t KW, int32_t stride,
                              int32_t pad_t, int32_t pad_l, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t dtype, void* stream) {
  if (!src || !dst || !view_ok(*src, Hs, Ws, src->c) || C_ < 1 || C_ > src->c || KH < 1 || KW < 1 || stride < 1 || pad_t < 0 || pad_l < 0 || B <= 0) { seg_set_error("im2col_act: bad args"); return SEG_ERR_ARG; }
  if (!view_ok(*dst, Ho, Wo, dst->c) || dst->c % 32 || KH * KW * C_ > dst->c) { seg_set_error("im2col_act: destination must hold %d channels padded to a multiple of 32", KH * KW * C_); return SEG_ERR_ARG; }
  const int pieces = dst->c / 8;
  const int64_t n = (int64_t)B * Ho * Wo * pieces;
  if (dtype == SEG_F32) SEG_LAUNCH(im2col_act_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, *src, B, Hs, Ws, C_, KH, KW, stride, pad_t, pad_l, *dst, Ho, Wo, pieces);
  else if (dtype == SEG_BF16) SEG_LAUNCH(im2col_act_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, *src, B, Hs, Ws, C_, KH, KW, stride, pad_t, pad_l, *dst, Ho, Wo, pieces);
  else { seg_set_error("im2col_act: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("im2col_act");
}

// out[b,Y,X,c] = relu?( bias[c] + sum over taps (u,v) with (Y + pad_t - u) % s == 0, (X + pad_l - v) % s == 0 of
//                col[b, (Y + pad_t - u)/s, (X + pad_l - v)/s, (u*KW + v)*C + c] ),   c < C (channels beyond: zero); C % 8 == 0 takes 16-byte loads
template <typename T>
__global__ void col2im_kernel(seg_view col, int B, int Hi, int Wi, int C_, int KH, int KW, int stride, int pad_t, int pad_l, const float* bias,
                              int relu, seg_view dst, int Ho, int Wo) {
  const int C8 = dst.c / 8;
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    int64_t t = i / C8;
    const int X = (int)(t % Wo); t /= Wo;
    const int Y = (int)(t % Ho); const int b = (int)(t / Ho);
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    if (c8 * 8 < C_) {
      const int ne = C_ - c8 * 8 < 8 ? C_ - c8 * 8 : 8;           // live channels of this piece (the last piece of a ragged C)
      for (int u = (Y + pad_t) % stride; u < KH; u += stride) {
        const int iy = (Y + pad_t - u) / stride;
        if (Y + pad_t - u < 0 || iy >= Hi) continue;
        for (int v = (X + pad_l) % stride; v < KW; v += stride) {
          const int ix = (X + pad_l - v) / stride;
          if (X + pad_l - v < 0 || ix >= Wi) continue;
          const T* sp = reinterpret_cast<const T*>(col.ptr) + view_off(col, b, iy, ix) + (u * KW + v) * C_ + c8 * 8;
          if ((C_ & 7) == 0) {
            Vec8<T> s; s.load(sp);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += s.get(e);
          } else {                                                  // (a tap's channels are not 16-byte aligned: element loads)
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e < ne) a[e] += (float)sp[e];
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (bias && e < ne) a[e] += bias[c8 * 8 + e];
        if (relu) a[e] = fmaxf(a[e], 0.f);
      }
    }
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, a[e]);
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, Y, X) + c8 * 8);
  }
}

extern "C" int seg_col2im(const seg_view* col, int32_t B, int32_t Hi, int32_t Wi, int32_t C_, int32_t KH, int32_t KW, int32_t stride,
                          int32_t pad_t, int32_t pad_l, const float* bias, int32_t relu, const seg_view* dst, int32_t Ho, int32_t Wo,
                          int32_t dtype, void* stream) {
  if (!col || !dst || C_ < 1 || KH < 1 || KW < 1 || stride < 1 || pad_t < 0 || pad_l < 0 || B <= 0 || !view_ok(*col, Hi, Wi, col->c) || KH * KW * C_ > col->c ||
      !view_ok(*dst, Ho, Wo, dst->c) || C_ > dst->c) { seg_set_error("col2im: bad args"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * Ho * Wo * (dst->c / 8);
  if (dtype == SEG_F32) SEG_LAUNCH(col2im_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, *col, B, Hi, Wi, C_, KH, KW, stride, pad_t, pad_l, bias, relu, *dst, Ho, Wo);
  else if (dtype == SEG_BF16) SEG_LAUNCH(col2im_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, *col, B, Hi, Wi, C_, KH, KW, stride, pad_t, pad_l, bias, relu, *dst, Ho, Wo);
  else { seg_set_error("col2im: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("col2im");
}
