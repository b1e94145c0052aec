// conv_first.hip -- the first 3x3 conv of each model (Cin = input_channel <= 4, K = 9*Cin <= 36).
// /root/reference/models/unet.py:111-116 (conv1_1, VALID) and models/fcn.py:110-115 (conv1, SAME).
// With K = 27 an MFMA tile would be >50 % padding and the layer is HBM-bound on its 32-channel
// output anyway (25 FLOP/B), so it runs on the vector ALU: one thread per output pixel, 32 output
// channels in registers, filters broadcast from LDS, 64-byte-per-lane coalesced stores.
#include "common.h"
#include <stdlib.h>
#include <string.h>

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
    const int piece = i & 3;
    int ox, oy, b;
    if (i <= 0x7fffffff) { unsigned t = (unsigned)(i >> 2); ox = t % (unsigned)Wo; t /= (unsigned)Wo; oy = t % (unsigned)Ho; b = t / (unsigned)Ho; }   // (32-bit divisions)
    else { int64_t t = i >> 2; ox = t % Wo; t /= Wo; oy = t % Ho; b = (int)(t / Ho); }
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = piece * 8 + e;
      float v = 0.f;
      if (k < nk) {
        const int tap = cin == 3 ? k / 3 : k / cin, c = k - tap * cin;
        const int iy = oy - pad + tap / 3, ix = ox - pad + tap % 3;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((int64_t)b * H + iy) * W + ix) * cin + c];
      }
      o.set(e, v);
    }
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox) + piece * 8);
  }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 path: the same layer on the matrix cores, optionally fused with the 2x2/s2 max-pool that follows it in both
// models (models/unet.py pools conv1_1 itself, finding F11; models/fcn.py:116 pools conv1).
// K = 9*cin <= 27 is one 32-deep MFMA step: D[channel][pixel] = W[channel][k] * X[k][pixel].  A wave owns a block of
// 2 rows x 8 columns of output pixels per step (pixel p = lane&15 -> row p>>3, column p&7): lane (p, g = lane>>4)
// gathers k = 8g..8g+7 of its pixel straight from the float image (8 dword loads: a filter row's 3*cin values are
// contiguous in NHWC), converts to bf16 and that IS the B fragment -- no LDS, no im2col buffer.  Filter rows are
// permuted like the packed filters of conv_fwd.hip (MFMA m, row 4g'+r <-> channel 8g'+4m+r) so a lane ends up with 8
// consecutive channels of its pixel = one 16-byte store.  The 2x2 window of the pool is lanes p, p^1, p^8: two DPP
// max steps, then lanes with even column in the upper row store the pooled pixel.
// HBM-bound by design: 12 B read + 64 B (+16 B pooled) written per pixel.
// ---------------------------------------------------------------------------------------------------------
SEG_DEV float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }    // quad_perm [1,0,3,2]
SEG_DEV float dpp_ror8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); }   // row_ror:8

SEG_DEV unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// LDS-only barrier: __syncthreads() would also wait (vmcnt) for this wave's output stores of the previous tile
SEG_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FirstK {
  const float* x; const float* w; const float* bias;
  int B, H, W, cin, cout, pad, Ho, Wo, relu;
  seg_view dst;
  seg_view pool; int Hp, Wp;          // pool.ptr == nullptr: no fused pool
  int blocks_x, blocks_y;             // 2x8 pixel blocks per image
  int im2col;                         // 1: store the gathered fragment (the im2col row) instead of convolving
};

constexpr int FTH = 8, FTW = 32;                    // output pixels per workgroup tile (16 blocks of 2x8, 4 per wave)
constexpr int FPH = FTH + 2, FPW = FTW + 2;

template <int NG>      // NG = 32-channel groups of the output
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const FirstK P) {
  // input patch of the tile as floats, row stride RS words.  (Gathering the fragments straight from global memory
  // re-fetched ~10 cache lines per load instruction: 24 waves x ~1.3 KB of live lines thrash the 32 KB L1 and the
  // kernel ran at the L2->L1 rate, 70 us; through LDS every input line is fetched once per tile.)
  __shared__ float sx[FPH * (FPW * 3 + 1)];
  __shared__ float sbias[32 * NG];
  const int RS = FPW * P.cin + 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int nk = 9 * P.cin;
  int koff[8];                                       // LDS word offset of k = 8g+j relative to the pixel's patch origin; -1 = zero
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g + j;
    const int tap = k / P.cin, c = k - tap * P.cin;
    koff[j] = k < nk ? (tap / 3) * RS + (tap % 3) * P.cin + c : -1;
  }
  // filter fragments: MFMA m of group q, row i = 4g'+r (lane&15 = i) <-> channel 32q + 8g' + 4m + r
  Frag<bf16_t> fa[NG][2];
  float bv[NG][8];
  if (!P.im2col) {
#pragma unroll
    for (int q = 0; q < NG; ++q) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int ch = 32 * q + 8 * (p >> 2) + 4 * m + (p & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * g + j;
          fa[q][m].v[j] = (bf16_t)((k < nk && ch < P.cout) ? P.w[(int64_t)k * P.cout + ch] : 0.f);
        }
      }
    }
    // bias via LDS: registers filled by a global load before the loop make the compiler wait vmcnt(0) at their first
    // use in EVERY iteration (the counter also holds the previous block's stores); an LDS read is on lgkmcnt instead
    if (tid < 32 * NG) sbias[tid] = (P.bias && tid < P.cout) ? P.bias[tid] : 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[q][e] = sbias[32 * q + 8 * g + e];
  }
  const int tiles_x = P.blocks_x, tiles_y = P.blocks_y;          // here: FTH x FTW tiles per image
  const int per_img = tiles_x * tiles_y, total = P.B * per_img;
  const float lo = P.relu ? 0.f : -INFINITY;
  bf16_t* dstp = reinterpret_cast<bf16_t*>(P.dst.ptr);
  bf16_t* poolp = reinterpret_cast<bf16_t*>(P.pool.ptr);
  const int rowlen = FPW * P.cin;                                  // floats per patch row
  // vmcnt also counts stores and the compiler's wait insertion gives up across the loop back-edge (vmcnt(0)), so
  // (1) everything loaded before the loop is forced to have landed here (else every block's epilogue waited for the
  //     previous block's stores), and (2) the patch of tile t+1 is requested BEFORE the stores of tile t (its wait at
  //     the next loop top then counts the younger stores instead of draining them).
  constexpr int NPL = (FPH * FPW * 3 + 255) / 256;                 // patch floats per thread
  float pre[NPL];
  auto patch_load = [&](int t) {
    const int b = t / per_img; const int r = t - b * per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int oy0 = ty * FTH, ox0 = tx * FTW;
    const float* xb = P.x + (int64_t)b * P.H * P.W * P.cin;
#pragma unroll
    for (int n = 0; n < NPL; ++n) {
      const int i = tid + n * 256;
      const int py = i / rowlen, w = i - py * rowlen;
      const int iy = oy0 - P.pad + py, ixc = (ox0 - P.pad) * P.cin + w;     // row-contiguous: coalesced
      pre[n] = (i < FPH * rowlen && iy >= 0 && iy < P.H && ixc >= 0 && ixc < P.W * P.cin) ? xb[(int64_t)iy * P.W * P.cin + ixc] : 0.f;
    }
  };
  if ((int)blockIdx.x < total) patch_load(blockIdx.x);
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int b = t / per_img; const int r = t - b * per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int oy0 = ty * FTH, ox0 = tx * FTW;
    const int64_t dtile = view_off(P.dst, b, oy0, ox0);
    const int64_t ptile = poolp ? view_off(P.pool, b, oy0 >> 1, ox0 >> 1) : 0;
    lds_barrier();                                                 // previous tile's reads are done
#pragma unroll
    for (int n = 0; n < NPL; ++n) {
      const int i = tid + n * 256;
      const int py = i / rowlen, w = i - py * rowlen;
      if (i < FPH * rowlen) sx[py * RS + w] = pre[n];
    }
    lds_barrier();
    if (t + (int)gridDim.x < total) patch_load(t + gridDim.x);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int blk = wave * 4 + u;                                // 16 blocks: 4 block rows x 4 block columns
      const int ly = 2 * (blk >> 2) + (p >> 3), lx = 8 * (blk & 3) + (p & 7);
      const int oy = oy0 + ly, ox = ox0 + lx;
      const bool ok = oy < P.Ho && ox < P.Wo;
      const int base = ly * RS + lx * P.cin;
      Frag<bf16_t> fb;
#pragma unroll
      for (int j = 0; j < 8; ++j) fb.v[j] = (bf16_t)(koff[j] >= 0 ? sx[base + koff[j]] : 0.f);
      const int64_t doff = dtile + (ly * P.dst.W + lx) * P.dst.cs;
      if (P.im2col) {
        if (ok) *reinterpret_cast<bf16x8*>(dstp + doff + 8 * g) = fb.v;
        continue;
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        f32x4 a0 = f32x4{0, 0, 0, 0}, a1 = f32x4{0, 0, 0, 0};
        mma32(a0, fa[q][0], fb);
        mma32(a1, fa[q][1], fb);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaxf(a0[e] + bv[q][e], lo); v[4 + e] = fmaxf(a1[e] + bv[q][4 + e], lo); }
        Vec8<bf16_t> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, v[e]);
        if (ok) o.store(dstp + doff + 32 * q + 8 * g);
        if (poolp) {
          // pool the bf16-ROUNDED values (what the separate pool kernel would read).  After a ReLU they are >= 0, so
          // their bit patterns order like unsigned integers: the 2x2 max is two DPP steps of v_pk_max_u16 on the packed
          // pairs; pixels outside the image contribute 0 and windows that contain one are never stored.
          const bool st = ok && (p & 9) == 0 && (oy >> 1) < P.Hp && (ox >> 1) < P.Wp;
          bf16_t* pp = poolp + ptile + ((ly >> 1) * P.pool.W + (lx >> 1)) * P.pool.cs + 32 * q + 8 * g;
          if (P.relu) {
            u32x4 w = __builtin_bit_cast(u32x4, o.v);
            if (!ok) w = u32x4{0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              unsigned a = w[e];
              a = pk_max_u16(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true));
              a = pk_max_u16(a, (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xF, 0xF, true));
              w[e] = a;
            }
            if (st) *reinterpret_cast<u32x4*>(pp) = w;
          } else {
            Vec8<bf16_t> m;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float tv = ok ? o.get(e) : -INFINITY;
              tv = fmaxf(tv, dpp_xor1(tv));
              tv = fmaxf(tv, dpp_ror8(tv));
              m.set(e, tv);
            }
            if (st) m.store(pp);
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Round 3: the same layer with a quarter of the vector instructions (counters on the kernel above: 10.0 M VALU
// instructions per launch = 19 us of pure VALU issue at C2, the kernel was not HBM-bound but VALU-bound at 3 TB/s).
//   * The tile's input patch is staged ONCE as bf16 RGB0 pixels (8 bytes each): an MFMA B fragment is then two 8-byte
//     LDS reads instead of 8 float reads + 8 selects + 4 conversions.  K is laid out for that: k' = 32 s + 8 g + j with
//     filter row ty = 2 s + (g >> 1), filter column tx = 2 (g & 1) + (j >> 2), channel j & 3 -- two 32-deep MFMA steps,
//     the slots of the padding channel, of tx = 3 and of ty = 3 carry zero weights (the MFMAs are nowhere near a limit).
//   * The bias is the MFMA's C operand, ReLU is one packed signed-integer max per two channels on the converted bf16 bits.
//   * A wave owns 4 x 16 output pixels = 2 x 8 pool windows; MFMA block q holds window position q of all 16 windows, so a
//     lane sees the four pixels of ITS window one after the other: the pool is three packed maxima (no cross-lane traffic)
//     and the pooled row is stored by all 64 lanes (one 1 KB store per wave and tile; the DPP form stored with 16).
// Same arithmetic as before (bf16 operands, f32 accumulation, one rounding), so parity bounds are unchanged.
// ---------------------------------------------------------------------------------------------------------
constexpr int WPR = FTH + 3, WRS = 37;               // patch rows (+1: the ty = 3 slots read a row of zeros), row stride in pixels

SEG_DEV unsigned pk_relu_bf16(unsigned a) {         // both halves: negative (sign bit set) -> +0
  unsigned r;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(a));
  return r;
}

template <int NG, bool RELU, bool POOL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NG == 1 ? (RELU || !POOL ? 6 : 4) : 3, 8))) void conv_first_win_kernel(const FirstK P) {
  __shared__ __attribute__((aligned(16))) uint32_t sp[WPR * WRS * 2];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  for (int i = tid; i < WPR * WRS * 2; i += 256) sp[i] = 0u;      // (borders / the extra row stay zero for the whole launch)

  // filter fragments: MFMA m of channel group q, row i (= lane & 15) <-> channel 32 q + 8 (i >> 2) + 4 m + (i & 3)
  Frag<bf16_t> fa[NG][2][2];
  f32x4 bias4[NG][2];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int ch = 32 * q + 8 * (p >> 2) + 4 * m + (p & 3);
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ty = 2 * s_ + (g >> 1), tx = 2 * (g & 1) + (j >> 2), c = j & 3;
          const bool on = ty < 3 && tx < 3 && c < P.cin && ch < P.cout;
          fa[q][m][s_].v[j] = (bf16_t)(on ? P.w[(int64_t)((ty * 3 + tx) * P.cin + c) * P.cout + ch] : 0.f);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cb = 32 * q + 8 * g + 4 * m + r;                 // D row 4 g + r of MFMA m <-> this channel
        bias4[q][m][r] = (P.bias && cb < P.cout) ? P.bias[cb] : 0.f;
      }
    }
  // everything above is loaded before the loop: make it land now (see the vmcnt note in the kernel above)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int tiles_x = P.blocks_x, tiles_y = P.blocks_y;
  const int per_img = tiles_x * tiles_y, total = P.B * per_img;
  bf16_t* dstp = reinterpret_cast<bf16_t*>(P.dst.ptr);
  bf16_t* poolp = POOL ? reinterpret_cast<bf16_t*>(P.pool.ptr) : nullptr;
  // this lane's pool window inside the wave's 4 x 16 pixel region of the tile
  const int r0 = 4 * (wave >> 1) + 2 * (p >> 3), c0 = 16 * (wave & 1) + 2 * (p & 7);
  int baddr[2];                                                    // LDS byte address of the fragment of block 0, MFMA step s
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) baddr[s_] = ((r0 + 2 * s_ + (g >> 1)) * WRS + c0 + 2 * (g & 1)) * 8;
  const int dlane = (r0 * P.dst.W + c0) * P.dst.cs + 8 * g;
  const int plane = ((r0 >> 1) * P.pool.W + (c0 >> 1)) * P.pool.cs + 8 * g;

  constexpr int NPX = (FTH + 2) * (FTW + 2);                       // 340 staged pixels per tile
  float pre[2][3];
  auto patch_load = [&](int t) {
    const int b = t / per_img; const int r = t - b * per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const float* xb = P.x + (int64_t)b * P.H * P.W * P.cin;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int i = tid + n * 256;
      const int py = i / (FTW + 2), px = i - py * (FTW + 2);
      const int iy = ty * FTH - P.pad + py, ix = tx * FTW - P.pad + px;
      const bool in = i < NPX && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      const float* q = xb + ((int64_t)iy * P.W + ix) * P.cin;
#pragma unroll
      for (int c = 0; c < 3; ++c) pre[n][c] = (in && c < P.cin) ? q[c] : 0.f;
    }
  };
  if ((int)blockIdx.x < total) patch_load(blockIdx.x);
  __syncthreads();                                                 // the zero fill is complete
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int b = t / per_img; const int r = t - b * per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int oy0 = ty * FTH, ox0 = tx * FTW;
    lds_barrier();                                                 // previous tile's reads are done
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int i = tid + n * 256;
      if (i < NPX) {
        const int py = i / (FTW + 2), px = i - py * (FTW + 2);
        const bf16x4 v = bf16x4{(bf16_t)pre[n][0], (bf16_t)pre[n][1], (bf16_t)pre[n][2], (bf16_t)0.f};
        *reinterpret_cast<u32x2*>(&sp[(py * WRS + px) * 2]) = __builtin_bit_cast(u32x2, v);
      }
    }
    lds_barrier();
    if (t + (int)gridDim.x < total) patch_load(t + gridDim.x);     // in flight behind this tile's arithmetic and in front of its stores
    const int64_t dtile = view_off(P.dst, b, oy0, ox0);
    const int64_t ptile = POOL ? view_off(P.pool, b, oy0 >> 1, ox0 >> 1) : 0;
    const char* spb = reinterpret_cast<const char*>(sp);
    u32x4 pm[NG];                                                   // running packed maximum (ReLU layers) ...
    float pf[NG][8];                                                // ... or float maximum (no ReLU: bf16 bits do not order as integers)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int dy = q4 >> 1, dx = q4 & 1;
      const int oy = oy0 + r0 + dy, ox = ox0 + c0 + dx;
      const bool ok = oy < P.Ho && ox < P.Wo;
      Frag<bf16_t> fb[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const u32x2* src = reinterpret_cast<const u32x2*>(spb + baddr[s_] + (dy * WRS + dx) * 8);
        const u32x2 lo = src[0], hi = src[1];
        fb[s_].v = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        f32x4 a0 = bias4[q][0], a1 = bias4[q][1];
        mma32(a0, fa[q][0][0], fb[0]); mma32(a1, fa[q][1][0], fb[0]);
        mma32(a0, fa[q][0][1], fb[1]); mma32(a1, fa[q][1][1], fb[1]);
        const bf16x8 ob = bf16x8{(bf16_t)a0[0], (bf16_t)a0[1], (bf16_t)a0[2], (bf16_t)a0[3], (bf16_t)a1[0], (bf16_t)a1[1], (bf16_t)a1[2], (bf16_t)a1[3]};
        u32x4 o = __builtin_bit_cast(u32x4, ob);
        if (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pk_relu_bf16(o[e]);
        }
        if (ok) *reinterpret_cast<u32x4*>(dstp + dtile + dlane + (dy * P.dst.W + dx) * P.dst.cs + 32 * q) = o;
        if (POOL) {
          if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pm[q][e] = q4 == 0 ? o[e] : pk_max_u16(pm[q][e], o[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              pf[q][e] = q4 == 0 ? a0[e] : fmaxf(pf[q][e], a0[e]);
              pf[q][4 + e] = q4 == 0 ? a1[e] : fmaxf(pf[q][4 + e], a1[e]);
            }
          }
        }
      }
    }
    if (POOL) {
      const int py_ = (oy0 + r0) >> 1, px_ = (ox0 + c0) >> 1;
      if (py_ < P.Hp && px_ < P.Wp) {
#pragma unroll
        for (int q = 0; q < NG; ++q) {
          u32x4 o = pm[q];
          if (!RELU) {
            const bf16x8 ob = bf16x8{(bf16_t)pf[q][0], (bf16_t)pf[q][1], (bf16_t)pf[q][2], (bf16_t)pf[q][3], (bf16_t)pf[q][4], (bf16_t)pf[q][5], (bf16_t)pf[q][6], (bf16_t)pf[q][7]};
            o = __builtin_bit_cast(u32x4, ob);
          }
          *reinterpret_cast<u32x4*>(poolp + ptile + plane + 32 * q) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The window kernel for any small filter and stride on the raw image: K is laid out in PAIRS of horizontally adjacent taps
// (a pair = 2 RGB0 pixels = 16 contiguous LDS bytes = one lane's 8 k-slots): pair pi = 4 s + g of MFMA step s sits at filter row
// pi / PPR, columns 2 (pi % PPR) and + 1, PPR = ceil(KW / 2) pairs per filter row (3x3: the layout of the kernel above).  Slots
// past the filter (the odd column, the rows past KH) carry zero weights and read zero-filled / real pixels, never garbage.
// The DeconvModel's first layer (/root/reference/models/deconvolution.py:44-46, 5x5 / stride 2 SAME on RGB, 64 filters: 16 pairs =
// 4 MFMA steps) ran as im2col (250 MB) + a 1x1 convolution over it (335 MB): 131 + 53 us on the critical stream at 16 x 512^2; this
// reads the image once and writes the activation once (50 + 134 MB).
// ---------------------------------------------------------------------------------------------------------
struct GenK {
  const float* x; const float* w; const float* bias;
  int B, H, W, cin, cout, pad_t, pad_l, Ho, Wo;
  seg_view dst;
  int blocks_x, blocks_y;
  float* bn_ws; int bn_C;             // STATS: one row [bn_C][2] of per-channel (sum, sum of squares) of the stored values per workgroup
};

// STATS: the statistics pass of the batch norm that consumes this layer (seg_bn_fwd_rows) rides along -- every lane adds up the
// rounded values it stores, the workgroup leaves one row of partial sums in the batch norm's workspace (fixed order: lanes by
// butterfly, waves 0..3), and the 134 MB activation of the DeconvModel's conv1_0 is not read back for it.
template <int KH, int KW, int S, int NG, bool RELU, bool STATS>
__global__ __launch_bounds__(256) void conv_first_gen_kernel(const GenK P) {
  constexpr int PPR = (KW + 1) / 2, NP = KH * PPR, NS = (NP + 3) / 4;
  constexpr int PRL = (FTH - 1) * S + KH, PCL = (FTW - 1) * S + KW;               // patch rows / columns that are loaded
  constexpr int PR = (FTH - 1) * S + (4 * NS - 1) / PPR + 1, PC = (FTW - 1) * S + 2 * PPR;   // ... that the fragments may read
  constexpr int RS = PC | 1;                                                      // row stride in pixels (odd: rows start in different banks)
  constexpr int NPX = PRL * PCL, NLD = (NPX + 255) / 256;
  __shared__ __attribute__((aligned(16))) uint32_t sp[PR * RS * 2];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  for (int i = tid; i < PR * RS * 2; i += 256) sp[i] = 0u;        // (the slots no load covers stay zero for the whole launch)

  Frag<bf16_t> fa[NG][2][NS];
  f32x4 bias4[NG][2];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int ch = 32 * q + 8 * (p >> 2) + 4 * m + (p & 3);     // MFMA m of channel group q, A row p <-> this channel
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int pi = 4 * s_ + g;
          const int ty = pi / PPR, tx = 2 * (pi - ty * PPR) + (j >> 2), c = j & 3;
          const bool on = ty < KH && tx < KW && c < P.cin && ch < P.cout;
          fa[q][m][s_].v[j] = (bf16_t)(on ? P.w[(int64_t)((ty * KW + tx) * P.cin + c) * P.cout + ch] : 0.f);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cb = 32 * q + 8 * g + 4 * m + r;                 // D row 4 g + r of MFMA m <-> this channel
        bias4[q][m][r] = (P.bias && cb < P.cout) ? P.bias[cb] : 0.f;
      }
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int tiles_x = P.blocks_x, tiles_y = P.blocks_y;
  const int per_img = tiles_x * tiles_y, total = P.B * per_img;
  bf16_t* dstp = reinterpret_cast<bf16_t*>(P.dst.ptr);
  // this lane's 2 x 2 output pixels inside the wave's 4 x 16 region of the tile (as above)
  const int r0 = 4 * (wave >> 1) + 2 * (p >> 3), c0 = 16 * (wave & 1) + 2 * (p & 7);
  int baddr[NS];                                                   // LDS byte address of the pair of pixel (r0, c0), step s
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) {
    const int pi = 4 * s_ + g, ty = pi / PPR, tx = 2 * (pi - ty * PPR);
    baddr[s_] = ((r0 * S + ty) * RS + c0 * S + tx) * 8;
  }
  const int dlane = (r0 * P.dst.W + c0) * P.dst.cs + 8 * g;

  float st1[STATS ? NG : 1][8], st2[STATS ? NG : 1][8];
#pragma unroll
  for (int q = 0; q < (STATS ? NG : 1); ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) st1[q][e] = st2[q][e] = 0.f;
  float pre[NLD][3];
  auto patch_load = [&](int t) {
    const int b = t / per_img; const int r = t - b * per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const float* xb = P.x + (int64_t)b * P.H * P.W * P.cin;
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int i = tid + n * 256;
      const int py = i / PCL, px = i - py * PCL;
      const int iy = ty * FTH * S - P.pad_t + py, ix = tx * FTW * S - P.pad_l + px;
      const bool in = i < NPX && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      const float* q = xb + ((int64_t)iy * P.W + ix) * P.cin;
#pragma unroll
      for (int c = 0; c < 3; ++c) pre[n][c] = (in && c < P.cin) ? q[c] : 0.f;
    }
  };
  if ((int)blockIdx.x < total) patch_load(blockIdx.x);
  __syncthreads();                                                 // the zero fill is complete
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int b = t / per_img; const int r = t - b * per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int oy0 = ty * FTH, ox0 = tx * FTW;
    lds_barrier();                                                 // previous tile's reads are done
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int i = tid + n * 256;
      if (i < NPX) {
        const int py = i / PCL, px = i - py * PCL;
        const bf16x4 v = bf16x4{(bf16_t)pre[n][0], (bf16_t)pre[n][1], (bf16_t)pre[n][2], (bf16_t)0.f};
        *reinterpret_cast<u32x2*>(&sp[(py * RS + px) * 2]) = __builtin_bit_cast(u32x2, v);
      }
    }
    lds_barrier();
    if (t + (int)gridDim.x < total) patch_load(t + gridDim.x);     // in flight behind this tile's arithmetic and in front of its stores
    const int64_t dtile = view_off(P.dst, b, oy0, ox0);
    const char* spb = reinterpret_cast<const char*>(sp);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int dy = q4 >> 1, dx = q4 & 1;
      const int oy = oy0 + r0 + dy, ox = ox0 + c0 + dx;
      const bool ok = oy < P.Ho && ox < P.Wo;
      Frag<bf16_t> fb[NS];
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        const u32x2* src = reinterpret_cast<const u32x2*>(spb + baddr[s_] + (dy * S * RS + dx * S) * 8);
        const u32x2 lo = src[0], hi = src[1];
        fb[s_].v = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        f32x4 a0 = bias4[q][0], a1 = bias4[q][1];
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) { mma32(a0, fa[q][0][s_], fb[s_]); mma32(a1, fa[q][1][s_], fb[s_]); }
        const bf16x8 ob = bf16x8{(bf16_t)a0[0], (bf16_t)a0[1], (bf16_t)a0[2], (bf16_t)a0[3], (bf16_t)a1[0], (bf16_t)a1[1], (bf16_t)a1[2], (bf16_t)a1[3]};
        u32x4 o = __builtin_bit_cast(u32x4, ob);
        if (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = pk_relu_bf16(o[e]);
        }
        if (ok) *reinterpret_cast<u32x4*>(dstp + dtile + dlane + (dy * P.dst.W + dx) * P.dst.cs + 32 * q) = o;
        if (STATS && ok) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(o[e] << 16), hi = __uint_as_float(o[e] & 0xffff0000u);
            st1[q][2 * e] += lo; st2[q][2 * e] = fmaf(lo, lo, st2[q][2 * e]);
            st1[q][2 * e + 1] += hi; st2[q][2 * e + 1] = fmaf(hi, hi, st2[q][2 * e + 1]);
          }
        }
      }
    }
  }
  if constexpr (STATS) {
    __shared__ float red[4][NG * 32][2];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = st1[q][e], b2 = st2[q][e];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m); b2 += __shfl_xor(b2, m); }      // the 16 pixel lanes of channel group g
        if (p == 0) { red[wave][32 * q + 8 * g + e][0] = a; red[wave][32 * q + 8 * g + e][1] = b2; }
      }
    __syncthreads();
    for (int i = tid; i < NG * 32 * 2; i += 256) {
      const int c = i >> 1, j = i & 1;
      if (c < P.bn_C) P.bn_ws[((int64_t)blockIdx.x * P.bn_C + c) * 2 + j] = ((red[0][c][j] + red[1][c][j]) + red[2][c][j]) + red[3][c][j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Filter + bias gradient of that layer from the image itself (no im2col tensor): dW[kslot][o] = sum_px col[px][kslot] * dZ[px][o] as
// MFMAs whose reduction dimension is the PIXEL -- A = the tile's patch read "transposed" out of LDS (lane (kslot i, group g) picks
// its k-slot's value at 8 consecutive pixels of a tile row: eight 2-byte reads 16 bytes apart), B = the tile's dZ, transposed once
// into LDS as [o][pixel] so that a fragment is one 16-byte read.  The k-slot layout is the forward kernel's (pairs of taps, RGB0
// pixels); the spare channel of the pixel holds 1.0 inside the image, so the k-slot of tap (pad_t, pad_l) -- under it lies input pixel (S y, S x), always inside --
// accumulates the bias gradient.  Persistent workgroups keep the [NS*32][32 NG] sums in MFMA accumulators (wave w: k-slot groups
// 2w, 2w+1 of the eight), leave one row each in ws, and conv_first_gen_wgrad_final_kernel adds the rows in a fixed order.
// The DeconvModel's conv1_0 at 16 x 512^2: im2col (250 MB) + the 1x1 filter gradient over it (225 MB) = 112 + 97 us on a side
// stream; this reads 50 + 67 MB.
// ---------------------------------------------------------------------------------------------------------
template <int KH, int KW, int S, int NG>
__global__ __launch_bounds__(256) void conv_first_gen_wgrad_kernel(const GenK P, seg_view dzv, float* ws) {
  constexpr int PPR = (KW + 1) / 2, NP = KH * PPR, NS = (NP + 3) / 4;
  static_assert(NS == 4, "eight k-slot groups of 16 (two per wave)");
  constexpr int PRL = (FTH - 1) * S + KH, PCL = (FTW - 1) * S + KW;
  constexpr int PR = (FTH - 1) * S + (4 * NS - 1) / PPR + 1, PC = (FTW - 1) * S + 2 * PPR;
  constexpr int RS = PC | 1;
  constexpr int NPX = PRL * PCL, NLD = (NPX + 255) / 256;
  constexpr int NO = 32 * NG, ZRS = FTH * FTW + 8;              // dZ^T row stride in elements (16-byte aligned rows)
  constexpr int NZ = NO / 8;                                     // 16-byte dZ pieces per thread and tile
  __shared__ __attribute__((aligned(16))) uint32_t sp[PR * RS * 2];
  __shared__ __attribute__((aligned(16))) uint16_t zt[NO * ZRS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  for (int i = tid; i < PR * RS * 2; i += 256) sp[i] = 0u;

  f32x4 acc[2][2 * NG];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int o = 0; o < 2 * NG; ++o) acc[a][o] = f32x4{0.f, 0.f, 0.f, 0.f};
  // byte address (within sp) of this lane's k-slot at pixel (row 0, column 8 g) of the tile, for its two k-slot groups
  int abase[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int kslot = 16 * (2 * wave + a) + i16;
    const int pi = kslot >> 3, j = kslot & 7;
    const int ty = pi / PPR, tx = 2 * (pi - ty * PPR) + (j >> 2), ch = j & 3;
    abase[a] = ((ty * RS + (8 * g) * S + tx) * 4 + ch) * 2;
  }
  const int tiles_x = P.blocks_x, tiles_y = P.blocks_y;
  const int per_img = tiles_x * tiles_y, total = P.B * per_img;
  const bf16_t* dzp = reinterpret_cast<const bf16_t*>(dzv.ptr);

  float pre[NLD][3]; bool pin[NLD];
  u32x4 zpre[NZ];
  auto tile_load = [&](int t) {
    const int b = t / per_img; const int r = t - b * per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const float* xb = P.x + (int64_t)b * P.H * P.W * P.cin;
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int i = tid + n * 256;
      const int py = i / PCL, px = i - py * PCL;
      const int iy = ty * FTH * S - P.pad_t + py, ix = tx * FTW * S - P.pad_l + px;
      const bool in = i < NPX && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      const float* q = xb + ((int64_t)iy * P.W + ix) * P.cin;
      pin[n] = in;
#pragma unroll
      for (int c = 0; c < 3; ++c) pre[n][c] = (in && c < P.cin) ? q[c] : 0.f;
    }
#pragma unroll
    for (int n = 0; n < NZ; ++n) {
      const int pc = tid + n * 256;                              // piece: pixel pc / NZ of the tile, channels 8 (pc % NZ) ..
      const int px = pc / NZ, c8 = pc - px * NZ;
      const int oy = ty * FTH + (px >> 5), ox = tx * FTW + (px & 31);
      const bool ok = oy < P.Ho && ox < P.Wo;
      zpre[n] = ok ? *reinterpret_cast<const u32x4*>(dzp + view_off(dzv, b, oy, ox) + c8 * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  if ((int)blockIdx.x < total) tile_load(blockIdx.x);
  __syncthreads();
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    lds_barrier();                                                 // previous tile's reads are done
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
      const int i = tid + n * 256;
      if (i < NPX) {
        const int py = i / PCL, px = i - py * PCL;
        const bf16x4 v = bf16x4{(bf16_t)pre[n][0], (bf16_t)pre[n][1], (bf16_t)pre[n][2], (bf16_t)(pin[n] ? 1.f : 0.f)};
        *reinterpret_cast<u32x2*>(&sp[(py * RS + px) * 2]) = __builtin_bit_cast(u32x2, v);
      }
    }
#pragma unroll
    for (int n = 0; n < NZ; ++n) {
      const int pc = tid + n * 256;
      const int px = pc / NZ, c8 = pc - px * NZ;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        zt[(c8 * 8 + 2 * e) * ZRS + px] = (uint16_t)(zpre[n][e] & 0xffffu);
        zt[(c8 * 8 + 2 * e + 1) * ZRS + px] = (uint16_t)(zpre[n][e] >> 16);
      }
    }
    lds_barrier();
    if (t + (int)gridDim.x < total) tile_load(t + gridDim.x);      // in flight behind this tile's arithmetic
    const char* spb = reinterpret_cast<const char*>(sp);
#pragma unroll
    for (int r = 0; r < FTH; ++r) {                                // reduction step = one tile row of 32 pixels
      Frag<bf16_t> fb[2 * NG];
#pragma unroll
      for (int o = 0; o < 2 * NG; ++o)
        fb[o].v = *reinterpret_cast<const bf16x8*>(&zt[(16 * o + i16) * ZRS + 32 * r + 8 * g]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const char* ap = spb + abase[a] + r * (S * RS * 8);
        uint32_t w4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t lo = *reinterpret_cast<const uint16_t*>(ap + (2 * e) * (S * 8));
          const uint32_t hi = *reinterpret_cast<const uint16_t*>(ap + (2 * e + 1) * (S * 8));
          w4[e] = lo | (hi << 16);
        }
        Frag<bf16_t> fa;
        fa.v = __builtin_bit_cast(bf16x8, u32x4{w4[0], w4[1], w4[2], w4[3]});
#pragma unroll
        for (int o = 0; o < 2 * NG; ++o) mma32(acc[a][o], fa, fb[o]);
      }
    }
  }
  // one row [NS * 32][NO] of partial sums per workgroup: D row 4 g + r of k-slot group (2 wave + a), column i16 of output group o
  float* row = ws + (int64_t)blockIdx.x * (NS * 32 * NO);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int o = 0; o < 2 * NG; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) row[(16 * (2 * wave + a) + 4 * g + r) * NO + 16 * o + i16] = acc[a][o][r];
}

template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_first_gen_wgrad_final_kernel(const float* ws, int rows, int NO, int cin, int cout, int pad_t, int pad_l, float* dw, float* db) {
  constexpr int PPR = (KW + 1) / 2;
  __shared__ double part[32][8];
  const int el = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + el, RL = 128 * NO;
  double a = 0.0;
  if (i < RL)
    for (int r = sl; r < rows; r += 32) a += (double)ws[(int64_t)r * RL + i];
  part[sl][el] = a;
  __syncthreads();
  if (sl != 0 || i >= RL) return;
  a = 0.0;
#pragma unroll
  for (int q = 0; q < 32; ++q) a += part[q][el];
  const int kslot = i / NO, o = i - kslot * NO;
  const int pi = kslot >> 3, j = kslot & 7;
  const int ty = pi / PPR, tx = 2 * (pi - ty * PPR) + (j >> 2), ch = j & 3;
  if (o >= cout || ty >= KH || tx >= KW) return;
  if (ch < cin) dw[((int64_t)(ty * KW + tx) * cin + ch) * cout + o] = (float)a;
  else if (ch == 3 && ty == pad_t && tx == pad_l && db != nullptr) db[o] = (float)a;
}

}  // namespace

static int launch_first_mfma(const FirstK& P0, hipStream_t st) {
  FirstK P = P0;
  P.blocks_x = cdiv(P.Wo, FTW); P.blocks_y = cdiv(P.Ho, FTH);
  const int64_t total = (int64_t)P.B * P.blocks_x * P.blocks_y;
  int grid = (int)total; if (grid > 256 * 6) grid = 256 * 6;      // persistent: 6 workgroups per CU
  const int ng = P.im2col ? 1 : cdiv(P.cout, 32);
  // SEG_FIRST_IMPL = old / win forces either form (read once; seg_dbg_reload_env() re-reads it: the tests flip it between launches)
  const char* fi = seg_env("SEG_FIRST_IMPL");
  const bool old_form = fi && !strcmp(fi, "old");
  // The bf16-staged window form (round 3) for outputs that the 256 MB Infinity Cache absorbs: 23.0 against 31.1 us at C2 (16 x
  // 254^2 x 32: 83 MB written, 4.1 TB/s).  Its full-resolution stores are 64-byte halves of a line per instruction (a lane's
  // pixels are two apart) and once the writes really reach HBM -- 512^2 x 16: 332 MB -- it is the slower one (126-134 against
  // 117 us), so the big maps and the im2col mode keep the kernel below (SEG_FIRST_IMPL=old / win forces either).
  const bool win_form = fi && !strcmp(fi, "win");
  const double out_mb = (double)P.B * P.Ho * P.Wo * (ng * 32) * 2 * (P.pool.ptr ? 1.25 : 1.0) / 1e6;
  if (!P.im2col && !old_form && P.cin <= 3 && ng <= 2 && (win_form || out_mb <= 192.0)) {
    const int per_cu = ng == 1 ? 6 : 3;
    int g2 = (int)total; if (g2 > 256 * per_cu) g2 = 256 * per_cu;      // persistent: what the register budget keeps resident
    const int key = (ng - 1) * 4 + (P.relu ? 2 : 0) + (P.pool.ptr ? 1 : 0);
    switch (key) {
#define FW_CASE(K, NG_, R_, P_) case K: SEG_LAUNCH((conv_first_win_kernel<NG_, R_, P_>), dim3(g2), dim3(256), 0, st, P); break
      FW_CASE(0, 1, false, false); FW_CASE(1, 1, false, true); FW_CASE(2, 1, true, false); FW_CASE(3, 1, true, true);
      FW_CASE(4, 2, false, false); FW_CASE(5, 2, false, true); FW_CASE(6, 2, true, false); FW_CASE(7, 2, true, true);
#undef FW_CASE
    }
    return seg_check_launch("conv_first_win");
  }
  switch (ng) {
    case 1: SEG_LAUNCH(conv_first_mfma_kernel<1>, dim3(grid), dim3(256), 0, st, P); break;
    case 2: SEG_LAUNCH(conv_first_mfma_kernel<2>, dim3(grid), dim3(256), 0, st, P); break;
    default: seg_set_error("conv_first (bf16 MFMA path): cout %d > 64", P.cout); return SEG_ERR_UNSUPPORTED;
  }
  return seg_check_launch("conv_first_mfma");
}

extern "C" int seg_im2col3x3(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t pad, const seg_view* dst,
                             int32_t Ho, int32_t Wo, int32_t dtype, void* stream) {
  if (!x || !dst || !dst->ptr || cin < 1 || cin > 3 || B <= 0 || Ho != H + 2 * pad - 2 || Wo != W + 2 * pad - 2) { seg_set_error("im2col3x3: bad args (cin 1..3)"); return SEG_ERR_ARG; }
  if (dst->oy + Ho > dst->H || dst->ox + Wo > dst->W || dst->coff + 32 > dst->cs || dst->cs % 8 || dst->coff % 8) { seg_set_error("im2col3x3: destination window exceeds buffer"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * Ho * Wo * 4;
  int g = (int)((n + 255) / 256); if (g > 16384) g = 16384;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SEG_BF16) {            // same fragment gather as the forward kernel, stored instead of multiplied
    FirstK P = {};
    P.x = x; P.B = B; P.H = H; P.W = W; P.cin = cin; P.cout = 32; P.pad = pad; P.Ho = Ho; P.Wo = Wo; P.dst = *dst; P.im2col = 1;
    return launch_first_mfma(P, st);
  }
  if (dtype == SEG_F32) SEG_LAUNCH(im2col3x3_kernel<float>, dim3(g), dim3(256), 0, st, x, B, H, W, cin, pad, *dst, Ho, Wo);
  else if (dtype == SEG_BF16) SEG_LAUNCH(im2col3x3_kernel<bf16_t>, dim3(g), dim3(256), 0, st, x, B, H, W, cin, pad, *dst, Ho, Wo);
  else { seg_set_error("im2col3x3: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("im2col3x3");
}

// ---------------------------------------------------------------------------------------------------------
// General im2col of the raw float input: KH x KW window, any stride, explicit leading pads; channel k = (u*KW + v)*cin + c,
// zero-filled up to dst->c (a multiple of 32).  The DeconvModel's first layer (/root/reference/models/deconvolution.py: 5x5/s2
// SAME on 3 channels, K = 75 -> 96) becomes a 1x1 convolution over this tensor: forward and filter gradient on the MFMA
// kernels instead of the direct ones (1.6 + 2.6 ms -> ~0.2 ms at 512 x 512).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_kernel(const float* x, int B, int H, int W, int cin, int KH, int KW, int stride, int pad_t, int pad_l, seg_view dst,
                              int Ho, int Wo, int pieces) {
  const int64_t total = (int64_t)B * Ho * Wo * pieces;
  const int nk = KH * KW * cin;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int piece = (int)(i % pieces);
    int64_t t = i / pieces;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho); const int b = (int)(t / Ho);
    Vec8<T> o;
    // a filter ROW's KW * cin values are contiguous in the NHWC input: k = u * RW + j <-> x[(b, iy0 + u, ix0) * cin + j].  One
    // division per piece, none per element; windows that lie inside the image take no bounds checks (the per-element form with
    // two divisions and four compares ran the DeconvModel's first layer at 1.1 TB/s)
    const int RW = KW * cin;
    const int iy0 = oy * stride - pad_t, ix0 = ox * stride - pad_l;
    int u = piece * 8 / RW, j = piece * 8 - u * RW;
    const bool inside = iy0 >= 0 && iy0 + KH <= H && ix0 >= 0 && ix0 + KW <= W;
    const float* xr = x + (((int64_t)b * H + iy0) * W + ix0) * cin;      // (only dereferenced at positions inside the image)
    const int64_t rstride = (int64_t)W * cin;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = 0.f;
      if (u < KH) {
        if (inside) v = xr[u * rstride + j];
        else {
          const int iy = iy0 + u, ix = ix0 + j / cin;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xr[u * rstride + j];
        }
      }
      o.set(e, v);
      if (++j == RW) { j = 0; ++u; }
    }
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox) + piece * 8);
  }
}

// The same through LDS for small-cin inputs: a workgroup owns IT_TY x IT_TX output pixels, stages the float patch under them with
// coalesced row loads (one bounds test per float, no divisions) and every thread assembles its pixel's row from LDS.  The kernel
// above reads the patch with scalar global loads whose 64 lanes touch ~12 cache lines each: 114 us for the DeconvModel's
// 512^2 x 16 input (TA-bound at 2.2 TB/s of a 250 MB job).
constexpr int IT_TY = 4, IT_TX = 64;
template <typename T>
__global__ __launch_bounds__(256) void im2col_tile_kernel(const float* x, int B, int H, int W, int cin, int KH, int KW, int stride, int pad_t, int pad_l,
                                                          seg_view dst, int Ho, int Wo, int pieces, int tiles_x, int tiles_y) {
  extern __shared__ float sp_[];
  const int PH = (IT_TY - 1) * stride + KH, PWF = ((IT_TX - 1) * stride + KW) * cin;     // patch rows, floats per patch row
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; const int b = t / tiles_y;
  const int oy0 = ty * IT_TY, ox0 = tx * IT_TX;
  const int iy0 = oy0 * stride - pad_t, cx0 = (ox0 * stride - pad_l) * cin;              // first patch row / first float column of the patch in the image
  const int tid = threadIdx.x, rowf = W * cin;
  const float* xb = x + (int64_t)b * H * rowf;
  for (int r = 0; r < PH; ++r) {
    const int iy = iy0 + r;
    const bool rin = iy >= 0 && iy < H;
    const float* xr = xb + (int64_t)iy * rowf + cx0;
    for (int c = tid; c < PWF; c += 256) sp_[r * PWF + c] = (rin && cx0 + c >= 0 && cx0 + c < rowf) ? xr[c] : 0.f;
  }
  __syncthreads();
  const int ly = tid / IT_TX, lx = tid % IT_TX;
  const int oy = oy0 + ly, ox = ox0 + lx;
  if (oy >= Ho || ox >= Wo) return;
  const int RW = KW * cin;
  const float* base = sp_ + (ly * stride) * PWF + lx * stride * cin;
  T* o = reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox);
  int u = 0, j = 0;
  for (int p = 0; p < pieces; ++p) {
    Vec8<T> ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ov.set(e, u < KH ? base[u * PWF + j] : 0.f);
      if (++j == RW) { j = 0; ++u; }
    }
    ov.store(o + p * 8);
  }
}

extern "C" int seg_im2col(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t KH, int32_t KW, int32_t stride,
                          int32_t pad_t, int32_t pad_l, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t dtype, void* stream) {
  if (!x || !dst || !dst->ptr || cin < 1 || KH < 1 || KW < 1 || stride < 1 || pad_t < 0 || pad_l < 0 || B <= 0 || Ho <= 0 || Wo <= 0) { seg_set_error("im2col: bad args"); return SEG_ERR_ARG; }
  if (dst->c % 32 || KH * KW * cin > dst->c || dst->oy + Ho > dst->H || dst->ox + Wo > dst->W || dst->coff + dst->c > dst->cs || dst->cs % 8 || dst->coff % 8) {
    seg_set_error("im2col: destination window must hold %d channels padded to a multiple of 32", KH * KW * cin); return SEG_ERR_ARG;
  }
  if ((Ho - 1) * stride - pad_t >= H || (Wo - 1) * stride - pad_l >= W) { seg_set_error("im2col: output extent reaches past the input"); return SEG_ERR_ARG; }
  const int pieces = dst->c / 8;
  const int64_t n = (int64_t)B * Ho * Wo * pieces;
  int g = (int)((n + 255) / 256); if (g > 16384) g = 16384;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    const int64_t patch_floats = (int64_t)((IT_TY - 1) * stride + KH) * (((IT_TX - 1) * stride + KW) * cin);
    if (cin <= 4 && patch_floats * 4 <= 60 * 1024 && (dtype == SEG_F32 || dtype == SEG_BF16)) {
      const int tiles_x = cdiv(Wo, IT_TX), tiles_y = cdiv(Ho, IT_TY);
      const size_t lds = (size_t)patch_floats * 4;
      auto k32 = im2col_tile_kernel<float>; auto k16 = im2col_tile_kernel<bf16_t>;
      static bool attr_done = false;
      if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k32), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k16), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        attr_done = true;
      }
      const dim3 gt((unsigned)(B * tiles_y * tiles_x));
      if (dtype == SEG_F32) SEG_LAUNCH(k32, gt, dim3(256), lds, st, x, B, H, W, cin, KH, KW, stride, pad_t, pad_l, *dst, Ho, Wo, pieces, tiles_x, tiles_y);
      else SEG_LAUNCH(k16, gt, dim3(256), lds, st, x, B, H, W, cin, KH, KW, stride, pad_t, pad_l, *dst, Ho, Wo, pieces, tiles_x, tiles_y);
      return seg_check_launch("im2col_tile");
    }
  }
  if (dtype == SEG_F32) SEG_LAUNCH(im2col_kernel<float>, dim3(g), dim3(256), 0, st, x, B, H, W, cin, KH, KW, stride, pad_t, pad_l, *dst, Ho, Wo, pieces);
  else if (dtype == SEG_BF16) SEG_LAUNCH(im2col_kernel<bf16_t>, dim3(g), dim3(256), 0, st, x, B, H, W, cin, KH, KW, stride, pad_t, pad_l, *dst, Ho, Wo, pieces);
  else { seg_set_error("im2col: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("im2col");
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
  if (dtype == SEG_BF16 && cin <= 3 && cout <= 64) {
    FirstK P = {};
    P.x = x; P.w = w_hwio; P.bias = bias; P.B = B; P.H = H; P.W = W; P.cin = cin; P.cout = cout; P.pad = pad; P.Ho = Ho; P.Wo = Wo; P.relu = relu;
    P.dst = *dst;
    return launch_first_mfma(P, st);
  }
  if (dtype == SEG_F32) SEG_LAUNCH(conv_first_fwd_kernel<float>, grid, dim3(256), 0, st, x, B, H, W, cin, w_hwio, bias, cout, pad, *dst, Ho, Wo, relu, tiles_x, tiles_y);
  else if (dtype == SEG_BF16) SEG_LAUNCH(conv_first_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, x, B, H, W, cin, w_hwio, bias, cout, pad, *dst, Ho, Wo, relu, tiles_x, tiles_y);
  else { seg_set_error("conv_first_fwd: bad dtype"); return SEG_ERR_ARG; }
  return seg_check_launch("conv_first_fwd");
}

/* Small filter of any size / stride on the raw image as ONE pass (bf16, cin <= 3, cout <= 64): the DeconvModel's conv1_0
 * (/root/reference/models/deconvolution.py:44-46).  Instantiated: 5x5/s2, 3x3/s2, 3x3/s1, 7x7/s2. */
static int first_gen_grid(int64_t total, int ng, bool stats) {
  (void)stats;
  const int per_cu = ng == 1 ? 4 : 2;      // (what the filter fragments in registers leave resident)
  int g = total > 256 * per_cu ? 256 * per_cu : (int)total;
  return g < 1 ? 1 : g;
}

/* Workgroups = rows of batch-norm partial sums a seg_conv_first_gen_bn launch of this shape writes (<= 1024). */
extern "C" int32_t seg_conv_first_gen_rows(int32_t B, int32_t Ho, int32_t Wo, int32_t cout) {
  if (B < 1 || Ho < 1 || Wo < 1 || cout < 1 || cout > 64) return 0;
  return first_gen_grid((int64_t)B * cdiv(Wo, FTW) * cdiv(Ho, FTH), cdiv(cout, 32), true);
}

static int first_gen_launch(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const float* w_hwio, const float* bias,
                            int32_t cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_t, int32_t pad_l, const seg_view* dst,
                            int32_t Ho, int32_t Wo, int32_t relu, float* bn_ws, int32_t bn_C, int32_t dtype, void* stream) {
  if (!x || !w_hwio || !dst || !dst->ptr || cin < 1 || cin > 3 || cout < 1 || cout > 64 || B <= 0 || Ho <= 0 || Wo <= 0 || pad_t < 0 || pad_l < 0) {
    seg_set_error("conv_first_gen: bad args (cin 1..3, cout <= 64)"); return SEG_ERR_ARG; }
  if (dtype != SEG_BF16) { seg_set_error("conv_first_gen: bf16 only (f32: seg_im2col + the 1x1 convolution)"); return SEG_ERR_UNSUPPORTED; }
  const int cp = cdiv(cout, 32) * 32;
  if (dst->oy + Ho > dst->H || dst->ox + Wo > dst->W || dst->coff + cp > dst->cs || dst->cs % 8 || dst->coff % 8) { seg_set_error("conv_first_gen: destination window exceeds buffer"); return SEG_ERR_ARG; }
  if ((int64_t)(Ho - 1) * stride - pad_t >= H || (int64_t)(Wo - 1) * stride - pad_l >= W) { seg_set_error("conv_first_gen: output extent reaches past the input"); return SEG_ERR_ARG; }
  if ((int64_t)B * dst->H * dst->W * dst->cs >= ((int64_t)1 << 31) || (int64_t)B * H * W * cin >= ((int64_t)1 << 31)) { seg_set_error("conv_first_gen: tensor exceeds the 32-bit index range"); return SEG_ERR_UNSUPPORTED; }
  if (bn_ws && (bn_C < cout || bn_C > cp || bn_C % 8 || KH != 5 || KW != 5 || stride != 2)) { seg_set_error("conv_first_gen_bn: 5x5/s2 only, cout <= bn_C <= %d", cp); return SEG_ERR_UNSUPPORTED; }
  GenK P = {};
  P.x = x; P.w = w_hwio; P.bias = bias; P.B = B; P.H = H; P.W = W; P.cin = cin; P.cout = cout; P.pad_t = pad_t; P.pad_l = pad_l; P.Ho = Ho; P.Wo = Wo;
  P.dst = *dst; P.blocks_x = cdiv(Wo, FTW); P.blocks_y = cdiv(Ho, FTH); P.bn_ws = bn_ws; P.bn_C = bn_C;
  const int64_t total = (int64_t)B * P.blocks_x * P.blocks_y;
  const int ng = cdiv(cout, 32);
  const int g = first_gen_grid(total, ng, bn_ws != nullptr);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int key = KH * 1000 + KW * 100 + stride * 10 + ng;
  if (bn_ws) {
#define FG_BN(NG_) do { if (relu) SEG_LAUNCH((conv_first_gen_kernel<5, 5, 2, NG_, true, true>), dim3(g), dim3(256), 0, st, P); \
    else SEG_LAUNCH((conv_first_gen_kernel<5, 5, 2, NG_, false, true>), dim3(g), dim3(256), 0, st, P); } while (0)
    if (ng == 1) FG_BN(1); else FG_BN(2);
#undef FG_BN
    return seg_check_launch("conv_first_gen_bn");
  }
#define FG_CASE(KH_, KW_, S_, NG_) case KH_ * 1000 + KW_ * 100 + S_ * 10 + NG_: \
    if (relu) SEG_LAUNCH((conv_first_gen_kernel<KH_, KW_, S_, NG_, true, false>), dim3(g), dim3(256), 0, st, P); \
    else SEG_LAUNCH((conv_first_gen_kernel<KH_, KW_, S_, NG_, false, false>), dim3(g), dim3(256), 0, st, P); break
  switch (key) {
    FG_CASE(5, 5, 2, 1); FG_CASE(5, 5, 2, 2); FG_CASE(3, 3, 2, 1); FG_CASE(3, 3, 2, 2); FG_CASE(3, 3, 1, 1); FG_CASE(3, 3, 1, 2);
    FG_CASE(7, 7, 2, 1); FG_CASE(7, 7, 2, 2);
    default: seg_set_error("conv_first_gen: %dx%d / stride %d is not instantiated", KH, KW, stride); return SEG_ERR_UNSUPPORTED;
  }
#undef FG_CASE
  return seg_check_launch("conv_first_gen");
}

extern "C" int seg_conv_first_gen(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const float* w_hwio, const float* bias,
                                  int32_t cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_t, int32_t pad_l, const seg_view* dst,
                                  int32_t Ho, int32_t Wo, int32_t relu, int32_t dtype, void* stream) {
  return first_gen_launch(x, B, H, W, cin, w_hwio, bias, cout, KH, KW, stride, pad_t, pad_l, dst, Ho, Wo, relu, nullptr, 0, dtype, stream);
}

/* The same launch also leaving the statistics rows of the batch norm that consumes the layer (5x5 / stride 2): bn_ws is that batch
 * norm's workspace (seg_bn_ws_bytes(bn_C)), which seg_bn_fwd_rows(..., seg_conv_first_gen_rows(...)) then finishes. */
extern "C" int seg_conv_first_gen_bn(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const float* w_hwio, const float* bias,
                                     int32_t cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_t, int32_t pad_l, const seg_view* dst,
                                     int32_t Ho, int32_t Wo, int32_t relu, float* bn_ws, int32_t bn_C, int32_t dtype, void* stream) {
  if (!bn_ws) { seg_set_error("conv_first_gen_bn: no workspace"); return SEG_ERR_ARG; }
  return first_gen_launch(x, B, H, W, cin, w_hwio, bias, cout, KH, KW, stride, pad_t, pad_l, dst, Ho, Wo, relu, bn_ws, bn_C, dtype, stream);
}

/* Rows of partial sums (= persistent workgroups) and workspace bytes of seg_conv_first_gen_wgrad. */
static int first_gen_wgrad_rows(int64_t total) { return total > 512 ? 512 : (total < 1 ? 1 : (int)total); }
extern "C" int64_t seg_conv_first_gen_wgrad_ws_bytes(int32_t cout) {
  if (cout < 1 || cout > 64) return 0;
  return (int64_t)512 * 128 * (cdiv(cout, 32) * 32) * (int64_t)sizeof(float);
}

/* Filter + bias gradient of seg_conv_first_gen's layer straight from the image (5x5 / stride 2, pad <= 1, cin <= 3, cout <= 64, bf16):
 * dw_hwio float32 [KH][KW][cin][cout], db [cout] or NULL; two launches (partial rows in ws, then a fixed-order sum), deterministic. */
extern "C" int seg_conv_first_gen_wgrad(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const seg_view* dz, int32_t Ho, int32_t Wo,
                                        int32_t cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_t, int32_t pad_l, float* dw_hwio, float* db,
                                        void* ws, int64_t ws_bytes, int32_t dtype, void* stream) {
  if (!x || !dz || !dz->ptr || !dw_hwio || !ws || cin < 1 || cin > 3 || cout < 1 || cout > 64 || B <= 0 || Ho <= 0 || Wo <= 0 || pad_t < 0 || pad_l < 0) {
    seg_set_error("conv_first_gen_wgrad: bad args (cin 1..3, cout <= 64)"); return SEG_ERR_ARG; }
  if (dtype != SEG_BF16 || KH != 5 || KW != 5 || stride != 2 || pad_t >= KH || pad_l >= KW) { seg_set_error("conv_first_gen_wgrad: bf16, 5x5 / stride 2, pad < 5"); return SEG_ERR_UNSUPPORTED; }
  const int cp = cdiv(cout, 32) * 32;
  if (dz->oy + Ho > dz->H || dz->ox + Wo > dz->W || dz->coff + cp > dz->cs || dz->cs % 8 || dz->coff % 8 || dz->c < cp) { seg_set_error("conv_first_gen_wgrad: dZ window exceeds buffer"); return SEG_ERR_ARG; }
  if ((int64_t)(Ho - 1) * stride >= H || (int64_t)(Wo - 1) * stride >= W) { seg_set_error("conv_first_gen_wgrad: output extent reaches past the input"); return SEG_ERR_ARG; }
  if ((int64_t)B * dz->H * dz->W * dz->cs >= ((int64_t)1 << 31) || (int64_t)B * H * W * cin >= ((int64_t)1 << 31)) { seg_set_error("conv_first_gen_wgrad: tensor exceeds the 32-bit index range"); return SEG_ERR_UNSUPPORTED; }
  if (ws_bytes < seg_conv_first_gen_wgrad_ws_bytes(cout)) { seg_set_error("conv_first_gen_wgrad: workspace too small"); return SEG_ERR_ARG; }
  GenK P = {};
  P.x = x; P.B = B; P.H = H; P.W = W; P.cin = cin; P.cout = cout; P.pad_t = pad_t; P.pad_l = pad_l; P.Ho = Ho; P.Wo = Wo;
  P.blocks_x = cdiv(Wo, FTW); P.blocks_y = cdiv(Ho, FTH);
  const int64_t total = (int64_t)B * P.blocks_x * P.blocks_y;
  const int g = first_gen_wgrad_rows(total);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* wsf = reinterpret_cast<float*>(ws);
  if (cp == 32) SEG_LAUNCH((conv_first_gen_wgrad_kernel<5, 5, 2, 1>), dim3(g), dim3(256), 0, st, P, *dz, wsf);
  else SEG_LAUNCH((conv_first_gen_wgrad_kernel<5, 5, 2, 2>), dim3(g), dim3(256), 0, st, P, *dz, wsf);
  if (int rc = seg_check_launch("conv_first_gen_wgrad")) return rc;
  SEG_LAUNCH((conv_first_gen_wgrad_final_kernel<5, 5>), dim3(128 * cp / 8), dim3(256), 0, st, (const float*)wsf, g, cp, cin, cout, pad_t, pad_l, dw_hwio, db);
  return seg_check_launch("conv_first_gen_wgrad_final");
}

/* First layer + the 2x2/s2 max-pool (VALID) that consumes it, in one pass (bf16, cin <= 3, cout <= 64): writes both
 * the activation (dst) and its pooled map (pool, [Hp = Ho/2, Wp = Wo/2]). */
extern "C" int seg_conv_first_pool_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const float* w_hwio, const float* bias,
                                       int32_t cout, int32_t pad, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t relu,
                                       const seg_view* pool, int32_t Hp, int32_t Wp, int32_t dtype, void* stream) {
  if (!x || !w_hwio || !dst || !dst->ptr || !pool || !pool->ptr || cin < 1 || cin > 3 || cout < 1 || cout > 64 || B <= 0) { seg_set_error("conv_first_pool_fwd: bad args (cin 1..3, cout <= 64)"); return SEG_ERR_ARG; }
  if (dtype != SEG_BF16) { seg_set_error("conv_first_pool_fwd: bf16 only (f32: seg_conv_first_fwd + seg_maxpool2x2_fwd)"); return SEG_ERR_UNSUPPORTED; }
  const int cp = cdiv(cout, 32) * 32;
  if (Ho != H + 2 * pad - 2 || Wo != W + 2 * pad - 2 || Hp != Ho / 2 || Wp != Wo / 2) { seg_set_error("conv_first_pool_fwd: inconsistent extents"); return SEG_ERR_ARG; }
  if (dst->oy + Ho > dst->H || dst->ox + Wo > dst->W || dst->coff + cp > dst->cs || dst->cs % 8 || dst->coff % 8 ||
      pool->oy + Hp > pool->H || pool->ox + Wp > pool->W || pool->coff + cp > pool->cs || pool->cs % 8 || pool->coff % 8) {
    seg_set_error("conv_first_pool_fwd: destination window exceeds buffer"); return SEG_ERR_ARG;
  }
  FirstK P = {};
  P.x = x; P.w = w_hwio; P.bias = bias; P.B = B; P.H = H; P.W = W; P.cin = cin; P.cout = cout; P.pad = pad; P.Ho = Ho; P.Wo = Wo; P.relu = relu;
  P.dst = *dst; P.pool = *pool; P.Hp = Hp; P.Wp = Wp;
  return launch_first_mfma(P, reinterpret_cast<hipStream_t>(stream));
}
