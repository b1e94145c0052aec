// elementwise.hip -- the HBM-bound ops of the hot path: max-pool (+fused crop-grad add and
// ReLU-grad), per-pixel softmax cross-entropy, sigmoid+argmax, bias grad, fused Adam, weight
// packing, depthwise bilinear transposed conv, dropout, cast.  All NHWC, 16-byte vector access
// over the channel dimension, wave64 reductions.
#include "common.h"
#include <stdarg.h>
#include <string.h>

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void seg_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int seg_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { seg_set_error("%s: launch failed: %s", what, hipGetErrorString(e)); return SEG_ERR_LAUNCH; }
  return SEG_OK;
}
extern "C" const char* seg_last_error(void) { return g_err; }

// the kernel this host thread launched last, as spelled at the launch site (parentheses dropped); launch sites that go through a
// function-pointer variable ("kern": the templated convolution / filter-gradient dispatchers) leave "" -- their instance is
// reported by seg_conv2d_kernel_name / seg_conv2d_wgrad_kernel_name
// Environment switches of the library are read ONCE (seg_env caches the pointer per name); seg_dbg_reload_env() drops the cache --
// the tests flip SEG_FIRST_IMPL between launches of one process.
static int g_env_epoch = 1;
const char* seg_env(const char* name) {
  struct Ent { const char* name; const char* val; int epoch; };
  static Ent tab[16];
  for (int i = 0; i < 16; ++i) {
    if (tab[i].name && !strcmp(tab[i].name, name)) {
      if (tab[i].epoch != g_env_epoch) { tab[i].val = getenv(name); tab[i].epoch = g_env_epoch; }
      return tab[i].val;
    }
    if (!tab[i].name) { tab[i].name = name; tab[i].val = getenv(name); tab[i].epoch = g_env_epoch; return tab[i].val; }
  }
  return getenv(name);
}
extern "C" void seg_dbg_reload_env(void) { ++g_env_epoch; }

static thread_local char g_kname[160] = "", g_kname_out[160] = "";
void seg_note_kernel(const char* s) {
  if (g_kname[0]) return;                           // the FIRST kernel since the last query (a finishing pass does not rename the launch)
  if (!strstr(s, "_kernel")) return;                // (a launch through a function-pointer variable -- kern, k0, k32 ...: the dispatcher's own name query answers)
  int n = 0;
  for (const char* p = s; *p && n < 159; ++p) if (*p != '(' && *p != ')' && *p != ' ') g_kname[n++] = *p;
  g_kname[n] = 0;
}
extern "C" const char* seg_last_kernel_name(void) {
  memcpy(g_kname_out, g_kname, sizeof(g_kname));
  g_kname[0] = 0;
  return g_kname_out;
}
extern "C" int seg_version(void) { return 100; }

namespace {

SEG_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware workgroup order for stencil-like walks: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so
// with the identity mapping the rows above and below a workgroup's pixels are fetched by OTHER XCDs and every L2 pulls three rows
// per row of output.  This gives XCD x the x-th contiguous eighth of the virtual ids: vertical neighbours meet in one L2.
SEG_DEV unsigned xcd_block(unsigned b, unsigned n, int on = 1) {
  return (on && n % 8u == 0u) ? (b % 8u) * (n / 8u) + b / 8u : b;
}

// (b, y, x[, c8]) of a flat element index.  The element counts of this path fit 32 bits; 64-bit divisions by run-time extents
// are ~100 instructions each and were a third of the byte-moving kernels' issue time.
struct Idx3 { int x, y, b; };
SEG_DEV Idx3 split3(int64_t i, int W, int H) {
  Idx3 r;
  if (i <= 0x7fffffff) { unsigned t = (unsigned)i; r.x = t % (unsigned)W; t /= (unsigned)W; r.y = t % (unsigned)H; r.b = t / (unsigned)H; }
  else { int64_t t = i; r.x = t % W; t /= W; r.y = t % H; r.b = (int)(t / H); }
  return r;
}
struct Idx4 { int c8, x, y, b; };
SEG_DEV Idx4 split4(int64_t i, int C8, int W, int H) {
  Idx4 r;
  if (i <= 0x7fffffff) { unsigned t = (unsigned)i; r.c8 = t % (unsigned)C8; t /= (unsigned)C8; r.x = t % (unsigned)W; t /= (unsigned)W; r.y = t % (unsigned)H; r.b = t / (unsigned)H; }
  else { int64_t t = i; r.c8 = t % C8; t /= C8; r.x = t % W; t /= W; r.y = t % H; r.b = (int)(t / H); }
  return r;
}

inline int grid_for(int64_t n, int per_block = 256, int cap = 8192) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

inline bool view_ok(const seg_view* v, int H, int W, int C) {
  return v && v->ptr && v->oy >= 0 && v->ox >= 0 && v->oy + H <= v->H && v->ox + W <= v->W && v->coff >= 0 &&
         v->coff + C <= v->cs && (v->cs % 8) == 0 && (v->coff % 8) == 0;
}

// ------------------------------------------------------------------------------------------
// max-pool 2x2 / stride 2 / VALID
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_fwd_kernel(seg_view src, seg_view dst, uint8_t* idx, int B, int Ho, int Wo, int C8) {
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, Wo, Ho);
    const int c8 = q_.c8, ox = q_.x, oy = q_.y, b = q_.b;
    const T* sp = reinterpret_cast<const T*>(src.ptr);
    Vec8<T> v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k].load(sp + view_off(src, b, 2 * oy + (k >> 1), 2 * ox + (k & 1)) + c8 * 8);
    Vec8<T> o;
    uint8_t id[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float m = v[0].get(e); int mi = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) { const float x = v[k].get(e); if (x > m) { m = x; mi = k; } }
      o.set(e, m); id[e] = (uint8_t)mi;
    }
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, oy, ox) + c8 * 8);
    if (idx) {
      uint64_t pk = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) pk |= (uint64_t)id[e] << (8 * e);
      *reinterpret_cast<uint64_t*>(idx + (((int64_t)(b * Ho + oy) * Wo + ox) * C8 + c8) * 8) = pk;
    }
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(seg_view yact, seg_view dpool, seg_view add, int add_h, int add_w, int ay0, int ax0,
                                   seg_view dz, int B, int H, int W, int C8) {
  const int Hw = (H + 1) / 2, Ww = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Hw * Ww * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, Ww, Hw);
    const int c8 = q_.c8, wx = q_.x, wy = q_.y, b = q_.b;
    const bool full = wy < Ho && wx < Wo;
    Vec8<T> y[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = 2 * wy + (k >> 1), xx = 2 * wx + (k & 1);
      ok[k] = yy < H && xx < W;
      y[k].zero();
      if (ok[k]) y[k].load(reinterpret_cast<const T*>(yact.ptr) + view_off(yact, b, yy, xx) + c8 * 8);
    }
    Vec8<T> dp; dp.zero();
    const bool route = full && dpool.ptr != nullptr;
    if (route) dp.load(reinterpret_cast<const T*>(dpool.ptr) + view_off(dpool, b, wy, wx) + c8 * 8);
    Vec8<T> adv[4];                     // the skip gradient of the four pixels: requested with everything else (one round trip)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adv[k].zero();
      const int ay = 2 * wy + (k >> 1) - ay0, ax = 2 * wx + (k & 1) - ax0;
      if (ok[k] && add.ptr != nullptr && ay >= 0 && ay < add_h && ax >= 0 && ax < add_w)
        adv[k].load(reinterpret_cast<const T*>(add.ptr) + view_off(add, b, ay, ax) + c8 * 8);
    }
    int mi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float m = y[0].get(e); mi[e] = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) { const float x = y[k].get(e); if (x > m) { m = x; mi[e] = k; } }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!ok[k]) continue;
      const int yy = 2 * wy + (k >> 1), xx = 2 * wx + (k & 1);
      Vec8<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float gsum = adv[k].get(e);
        if (route && mi[e] == k) gsum += dp.get(e);
        o.set(e, y[k].get(e) > 0.f ? gsum : 0.f);
      }
      o.store(reinterpret_cast<T*>(dz.ptr) + view_off(dz, b, yy, xx) + c8 * 8);
    }
  }
}

// ------------------------------------------------------------------------------------------
// ReLU grad, cast+pad, dropout
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void relu_grad_kernel(seg_view dy, seg_view yact, seg_view dz, int B, int H, int W, int C8) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, W, H);
    const int c8 = q_.c8, x = q_.x, y = q_.y, b = q_.b;
    Vec8<T> g, a, o;
    g.load(reinterpret_cast<const T*>(dy.ptr) + view_off(dy, b, y, x) + c8 * 8);
    a.load(reinterpret_cast<const T*>(yact.ptr) + view_off(yact, b, y, x) + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, a.get(e) > 0.f ? g.get(e) : 0.f);
    o.store(reinterpret_cast<T*>(dz.ptr) + view_off(dz, b, y, x) + c8 * 8);
  }
}

// float32 window -> the bf16 value grid, still float32 (diagnostic: seg_round_bf16)
__global__ void round_bf16_kernel(seg_view src, seg_view dst, int B, int H, int W, int C8) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, W, H);
    Vec8<float> a, o;
    a.load(reinterpret_cast<const float*>(src.ptr) + view_off(src, q_.b, q_.y, q_.x) + q_.c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, (float)(bf16_t)a.get(e));
    o.store(reinterpret_cast<float*>(dst.ptr) + view_off(dst, q_.b, q_.y, q_.x) + q_.c8 * 8);
  }
}
__global__ void round_bf16_flat_kernel(const float* src, float* dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (float)(bf16_t)src[i];
}

template <typename T>
__global__ void cast_pad_kernel(const float* x, int64_t npix, int c, seg_view dst) {
  const int C8 = dst.cs / 8;
  const int64_t total = npix * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i / C8; const int c8 = i % C8;
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int ch = c8 * 8 + e; o.set(e, ch < c ? x[pix * c + ch] : 0.f); }
    o.store(reinterpret_cast<T*>(dst.ptr) + pix * dst.cs + c8 * 8);
  }
}

// Counter-based Bernoulli bits: one independent stream per seed.  key = splitmix64(seed) is XOR-ed into the 64-bit counter
// (offset + element index) and the result goes through the splitmix64 finaliser again; the draw is its high 32 bits.  The
// per-site seeds (seed+2, seed+5, ...) and the per-pass offsets (t << 40) of infer_mc therefore cannot alias: two streams
// meet only if key1 ^ key2 equals the XOR of two counters, and counters stay far below 2^48.  Restated bit for bit in
// oracle/np_ops.py (dropout_mask).
SEG_DEV uint64_t splitmix64(uint64_t k) {
  k += 0x9E3779B97F4A7C15ull; k = (k ^ (k >> 30)) * 0xBF58476D1CE4E5B9ull; k = (k ^ (k >> 27)) * 0x94D049BB133111EBull;
  return k ^ (k >> 31);
}
template <typename T>
__global__ void dropout_kernel(seg_view xin, seg_view yout, int B, int H, int W, int C8, float keep, uint64_t seed, uint64_t offset,
                               const int64_t* step_dev) {
  if (step_dev != nullptr) offset += (uint64_t)(*step_dev) << 40;     // (seg_dropout_step: a fresh mask per replayed training step)
  const int64_t total = (int64_t)B * H * W * C8;
  const float inv = 1.f / keep;
  const uint32_t thr = (uint32_t)(keep * 4294967295.0);
  const uint64_t key = splitmix64(seed);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, W, H);
    const int c8 = q_.c8, x = q_.x, y = q_.y, b = q_.b;
    Vec8<T> v, o;
    v.load(reinterpret_cast<const T*>(xin.ptr) + view_off(xin, b, y, x) + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t r = (uint32_t)(splitmix64(key ^ (offset + (uint64_t)i * 8 + e)) >> 32);
      o.set(e, r <= thr ? v.get(e) * inv : 0.f);
    }
    o.store(reinterpret_cast<T*>(yout.ptr) + view_off(yout, b, y, x) + c8 * 8);
  }
}

// ------------------------------------------------------------------------------------------
// softmax cross-entropy (+grad) and sigmoid/argmax
// ------------------------------------------------------------------------------------------
template <typename T, int NCP>     // NCP = classes rounded up to 4/8/16/32: bounds every unrolled loop
__global__ void softmax_xent_kernel(seg_view lg, const uint8_t* labels, int LH, int LW, int ly0, int lx0, int B, int H, int W,
                                    int nc, float inv_n, float gscale, float* loss_sum, seg_view dl, seg_view pr) {
  const int64_t total = (int64_t)B * H * W;
  float local = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx3 q_ = split3(i, W, H);
    const int x = q_.x, y = q_.y, b = q_.b;
    const float* z = reinterpret_cast<const float*>(lg.ptr) + view_off(lg, b, y, x);
    const int lab = labels[((int64_t)b * LH + y + ly0) * LW + x + lx0];
    float zv[NCP];
    float m = -INFINITY;
    // (the float logits of a pixel are read as 16-byte vectors: 32 scalar loads per thread made this kernel 3x its HBM time
    // at 21 classes; the view is 32-byte aligned and at least NCP floats wide: checked on the host)
#pragma unroll
    for (int c4 = 0; c4 < NCP / 4; ++c4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(z + c4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int c = c4 * 4 + e; zv[c] = c < nc ? v[e] : -INFINITY; m = fmaxf(m, zv[c]); }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCP; ++c) { zv[c] = c < nc ? expf(zv[c] - m) : 0.f; s += zv[c]; }
    const bool valid = lab < nc;
    const float logs = logf(s);
    const float zl = valid ? z[lab] : 0.f;
    if (valid) local += (logs - (zl - m));
    const float k = inv_n * gscale, rs = 1.f / s;
    T* o = reinterpret_cast<T*>(dl.ptr) + view_off(dl, b, y, x);
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      if (c8 * 8 >= dl.c) break;
      Vec8<T> ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        float gv = 0.f;
        if (c < NCP) gv = (c < nc && valid) ? (zv[c < NCP ? c : 0] * rs - (c == lab ? 1.f : 0.f)) * k : 0.f;
        ov.set(e, gv);
      }
      ov.store(o + c8 * 8);
    }
    if (pr.ptr != nullptr) {                 // (adversarial training: the class probabilities the adversary reads, from the same pass)
      T* po = reinterpret_cast<T*>(pr.ptr) + view_off(pr, b, y, x);
#pragma unroll
      for (int c8 = 0; c8 < 4; ++c8) {
        if (c8 * 8 >= pr.c) break;
        Vec8<T> ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = c8 * 8 + e;
          ov.set(e, c < NCP ? zv[c < NCP ? c : 0] * rs : 0.f);
        }
        ov.store(po + c8 * 8);
      }
    }
  }
  // one atomic per block (hundreds of waves adding to one address serialise at the memory side: ~20 us measured)
  __shared__ float wsum[4];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_n);
}

// ------------------------------------------------------------------------------------------
// Fused segmentation head of the U-Net train step: 1x1 'output' conv (logits) + softmax cross-entropy + the 1x1 conv's
// input gradient (with the ReLU-grad mask of its input) -- three launches of ~6 us each on the critical path become
// one.  CINP/8 lanes share a pixel: lane j holds input channels [8j, 8j+8), so a wave's loads and stores of the
// activation, its gradient and dlogits are contiguous 16-byte pieces; the class sums are completed with a butterfly
// over those lanes.  w is the fp32 HWIO filter [cin][nc] (models/unet.py:166,171-174, basemodel.py:59-70).
// ------------------------------------------------------------------------------------------
//
// DW: the 1x1 conv's filter and bias gradients are accumulated here as well -- lane j keeps the 8 x NCP products of its
// channels with the class gradients -- and leave as one row of partial sums per workgroup ([CINP*NCP] dW then [NCP] db,
// summed in a fixed order by head_dw_reduce_kernel).  dlogits is then not written at all: it had no other reader, and at
// 32 padded channels it was as large as the activation gradient.
template <typename T, int NCP, int CINP, bool DW>
__global__ __launch_bounds__(256) void head_xent_kernel(seg_view act, const float* w, const float* bias, int cin, const uint8_t* labels,
                                                        int LH, int LW, int ly0, int lx0, int B, int H, int W, int nc, float inv_n,
                                                        float* loss_sum, seg_view lg, seg_view dl, seg_view dact, float* dw_part) {
  constexpr int LP = CINP / 8;              // lanes per pixel
  constexpr bool WREG = false;              // true: the lane's 8 x NCP filter slice in registers (no faster, 50 VGPRs more)
  __shared__ float sw[CINP * NCP];
  __shared__ float sb[NCP];
  for (int i = threadIdx.x; i < CINP * NCP; i += 256) { const int k = i / NCP, c = i % NCP; sw[i] = (k < cin && c < nc) ? w[(int64_t)k * nc + c] : 0.f; }
  if (threadIdx.x < NCP) sb[threadIdx.x] = (bias && (int)threadIdx.x < nc) ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int j = threadIdx.x % LP;
  const float* swj = sw + j * 8 * NCP;
  float wr[WREG ? 8 * NCP : 1];
  if constexpr (WREG) {
#pragma unroll
    for (int i = 0; i < 8 * NCP; ++i) wr[i] = swj[i];
  }
  auto wgt = [&](int e, int c) -> float { if constexpr (WREG) return wr[e * NCP + c]; else return swj[e * NCP + c]; };
  const int total = B * H * W;              // < 2^31 pixels (checked by the host)
  constexpr int PPB = 256 / LP;
  const int stride = gridDim.x * PPB;
  float local = 0.f;
  // one 16-byte load per lane and pixel is too little in flight to cover the HBM latency: the next pixel's activation
  // and label are requested before this pixel's arithmetic starts
  float dwa[DW ? 8 * NCP : 1], dba[DW ? NCP : 1];
  if constexpr (DW) {
#pragma unroll
    for (int t = 0; t < 8 * NCP; ++t) dwa[t] = 0.f;
#pragma unroll
    for (int c = 0; c < NCP; ++c) dba[c] = 0.f;
  }
  Vec8<T> v_next; int lab_next = 0; Idx3 q_next = {0, 0, 0};
  auto request = [&](int i) {
    q_next = split3(i, W, H);
    v_next.load(reinterpret_cast<const T*>(act.ptr) + view_off(act, q_next.b, q_next.y, q_next.x) + j * 8);
    lab_next = labels[((int64_t)q_next.b * LH + q_next.y + ly0) * LW + q_next.x + lx0];
  };
  int i = blockIdx.x * PPB + threadIdx.x / LP;
  if (i < total) request(i);
  for (; i < total; i += stride) {
    if constexpr (!WREG) asm volatile("" ::: "memory");     // keep the filter in LDS: hoisted out of the loop it would not fit the registers
    const Vec8<T> v = v_next;
    const int lab = lab_next;
    const int x = q_next.x, y = q_next.y, b = q_next.b;
    if (i + stride < total) request(i + stride);
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = v.get(e);
    float zv[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) zv[c] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int c = 0; c < NCP; ++c) zv[c] = fmaf(a[e], wgt(e, c), zv[c]);
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
#pragma unroll
      for (int m_ = 1; m_ < LP; m_ <<= 1) zv[c] += __shfl_xor(zv[c], m_);
      zv[c] += sb[c];
    }
    float* zo = reinterpret_cast<float*>(lg.ptr) + view_off(lg, b, y, x);
#pragma unroll
    for (int c = 0; c < NCP; ++c) if (c % LP == j && c < nc) zo[c] = zv[c];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCP; ++c) { if (c >= nc) zv[c] = -INFINITY; m = fmaxf(m, zv[c]); }
    const bool valid = lab < nc;
    float zl = 0.f, s_ = 0.f;
#pragma unroll
    for (int c = 0; c < NCP; ++c) { if (c == lab) zl = zv[c]; zv[c] = c < nc ? expf(zv[c] - m) : 0.f; s_ += zv[c]; }
    if (valid && j == 0) local += (logf(s_) - (zl - m));
    const float rs = 1.f / s_;
    float g[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) g[c] = (c < nc && valid) ? (zv[c] * rs - (c == lab ? 1.f : 0.f)) * inv_n : 0.f;
    // lane j writes channel group j of dlogits (real classes first, zero padding behind them)
    if (dl.ptr != nullptr && j * 8 < dl.c) {
      T* o = reinterpret_cast<T*>(dl.ptr) + view_off(dl, b, y, x) + j * 8;
      Vec8<T> ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < NCP; ++c) if (c == j * 8 + e) v = g[c];
        ov.set(e, v);
      }
      ov.store(o);
    }
    // gradient of the 1x1 conv's input, masked by that input's ReLU.  dlogits is what the filter gradient reads, so the
    // same (dtype-rounded) values are used here as the separate dgrad launch would see.
#pragma unroll
    for (int c = 0; c < NCP; ++c) { Vec8<T> r; r.set(0, g[c]); g[c] = r.get(0); }
    if constexpr (DW) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int c = 0; c < NCP; ++c) dwa[e * NCP + c] = fmaf(a[e], g[c], dwa[e * NCP + c]);
#pragma unroll
      for (int c = 0; c < NCP; ++c) dba[c] += g[c];
    }
    T* da = reinterpret_cast<T*>(dact.ptr) + view_off(dact, b, y, x) + j * 8;
    Vec8<T> ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < NCP; ++c) acc = fmaf(g[c], wgt(e, c), acc);
      ov.set(e, a[e] > 0.f ? acc : 0.f);
    }
    ov.store(da);
  }
  __shared__ float wsum[4];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  if constexpr (DW) {
    // lanes with the same j across the wave, then the four waves through LDS, then one row of the workspace
    constexpr int ROW = CINP * NCP + NCP;
    __shared__ float part[4][ROW];
#pragma unroll
    for (int t = 0; t < 8 * NCP; ++t)
#pragma unroll
      for (int m_ = LP; m_ < 64; m_ <<= 1) dwa[t] += __shfl_xor(dwa[t], m_);
#pragma unroll
    for (int c = 0; c < NCP; ++c)
#pragma unroll
      for (int m_ = LP; m_ < 64; m_ <<= 1) dba[c] += __shfl_xor(dba[c], m_);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < LP) {
#pragma unroll
      for (int t = 0; t < 8 * NCP; ++t) part[wv][lane * 8 * NCP + t] = dwa[t];
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NCP; ++c) part[wv][CINP * NCP + c] = dba[c];
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ROW; t += 256)
      dw_part[(int64_t)blockIdx.x * ROW + t] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  } else {
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(loss_sum, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_n);
}

// Sums the workgroup rows of head_xent_kernel<.., DW> in row order (one workgroup per output element) into the fp32 HWIO
// filter gradient [cin][nc] and the bias gradient [nc].
__global__ __launch_bounds__(256) void head_dw_reduce_kernel(const float* __restrict__ part, int rows, int ncp, int cinp, int cin, int nc,
                                                             float* __restrict__ dw, float* __restrict__ db) {
  const int o = blockIdx.x, row = cinp * ncp + ncp;
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) s += part[(int64_t)r * row + o];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (o < cinp * ncp) { const int k = o / ncp, c = o % ncp; if (k < cin && c < nc) dw[k * nc + c] = red[0]; }
    else { const int c = o - cinp * ncp; if (c < nc && db) db[c] = red[0]; }
  }
}

__global__ void sigmoid_argmax_kernel(seg_view lg, int B, int H, int W, int nc, float* sig, float* out) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx3 q_ = split3(i, W, H);
    const int x = q_.x, y = q_.y, b = q_.b;
    const float* z = reinterpret_cast<const float*>(lg.ptr) + view_off(lg, b, y, x);
    float best = -1.f; int bi = 0;
    for (int c = 0; c < nc; ++c) {
      // sigmoid evaluated in double and rounded once to float32 (the oracle's definition); the argmax
      // is over the float32 values, first maximum wins (saturated ties -> lower index).
      const float s = (float)(1.0 / (1.0 + exp(-(double)z[c])));
      sig[i * nc + c] = s;
      if (s > best) { best = s; bi = c; }
    }
    out[i] = (float)bi;
  }
}

// ------------------------------------------------------------------------------------------
// bias grad: column sums of a [B*H*W, C] window
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void bias_grad_kernel(seg_view dz, int B, int H, int W, int n_log, float* db) {
  // one workgroup per 8-channel group: a fixed summation order (bitwise reproducible), db overwritten
  __shared__ float red[1024 * 8];
  const int tid = threadIdx.x, c8 = blockIdx.x;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int npix = B * H * W;
  for (int p = tid; p < npix; p += 1024) {
    const Idx3 q_ = split3(p, W, H);
    Vec8<T> v;
    v.load(reinterpret_cast<const T*>(dz.ptr) + view_off(dz, q_.b, q_.y, q_.x) + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += v.get(e);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[e * 1024 + tid] = s[e];
  __syncthreads();
  for (int h = 512; h > 0; h >>= 1) {
    if (tid < h) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[e * 1024 + tid] += red[e * 1024 + tid + h];
    }
    __syncthreads();
  }
  if (tid < 8 && c8 * 8 + tid < n_log) db[c8 * 8 + tid] = red[tid * 1024];
}

// ------------------------------------------------------------------------------------------
// Adam (TF variant) on a flat arena
// ------------------------------------------------------------------------------------------
// one element of the TF-Adam update: the ONLY place the arithmetic is written (flat and fused kernels must agree bit for bit)
SEG_DEV void adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps, float gs) {
#pragma clang fp contract(off)      // no mul+add fusion: which products get fused would depend on the surrounding kernel
  const float gr = g * gs;
  m = b1 * m + (1.f - b1) * gr;
  v = b2 * v + (1.f - b2) * gr * gr;
  p = p - lr_t * m / (sqrtf(v) + eps);
}
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                            float eps, float gs, const int64_t* step_dev) {
  __shared__ float s_lrt;
  if (threadIdx.x == 0) {
    const double t = (double)(*step_dev + 1);
    s_lrt = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = s_lrt;
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float p1 = pp[e], m1 = mm[e], v1 = vv[e];
      adam1(p1, gg[e], m1, v1, lr_t, b1, b2, eps, gs);
      pp[e] = p1; mm[e] = m1; vv[e] = v1;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    float p1 = p[i], m1 = m[i], v1 = v[i];
    adam1(p1, g[i], m1, v1, lr_t, b1, b2, eps, gs);
    p[i] = p1; m[i] = m1; v[i] = v1;
  }
}
__global__ void step_inc_kernel(int64_t* s) { if (threadIdx.x == 0 && blockIdx.x == 0) *s += 1; }
// start of a train step: step2[1] = completed steps so far (what seg_adam reads, t = step2[1] + 1), step2[0] = t; loss = 0
__global__ void step_begin_kernel(int64_t* s, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { const int64_t t0 = s[0]; s[1] = t0; s[0] = t0 + 1; if (loss) *loss = 0.f; }
}

// ------------------------------------------------------------------------------------------
// weight packing: fp32 TF layout -> [taps][K/32][n_total][32] (rows permuted inside each
// 32-block so that the two MFMA fragments of a block give every lane 8 consecutive channels)
// ------------------------------------------------------------------------------------------
SEG_DEV int seg_ci(const seg_pack_entry& e, int kpad) {   // padded concat channel -> logical (or -1)
  if (kpad < e.seg0_cp) return kpad < e.seg0_c ? kpad : -1;
  const int k1 = kpad - e.seg0_cp;
  return (k1 < e.seg1_c) ? e.seg0_c + k1 : -1;
}
// 32(k) x 32(n) tiles of one tap: reads run along the source's fastest axis, the tile is transposed through LDS when that
// axis is n, writes are one contiguous 32x32 block of the packed arena.  A 256-thread block owns PACK_TPB consecutive
// tiles and requests the source elements of ALL of them before it touches any: with one tile per block the launch was
// 15 000 blocks of three dependent round trips each (table -> entry -> elements), 46 us for 77 MB.
constexpr int PACK_TPB = 4;
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* arena, T* packed, const seg_pack_entry* tab, int n_entries, int64_t total_tiles) {
  __shared__ int s_e[PACK_TPB];
  __shared__ float tile[PACK_TPB][32][33];
  const int64_t tile0 = (int64_t)blockIdx.x * PACK_TPB;
  if (threadIdx.x < 64) {
    // entries are sorted by blk_start: the owner of a tile is (number of entries starting at or before it) - 1.
    // One wave tests 64 entries per pass in parallel (a serial scan by one lane cost ~4 us of latency per block).
    int cnt[PACK_TPB];
#pragma unroll
    for (int q = 0; q < PACK_TPB; ++q) cnt[q] = 0;
    for (int base = 0; base < n_entries; base += 64) {
      const int i = base + threadIdx.x;
      const int64_t bs = i < n_entries ? tab[i].blk_start : INT64_MAX;
#pragma unroll
      for (int q = 0; q < PACK_TPB; ++q) cnt[q] += __popcll(__ballot(tile0 + q >= bs));
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < PACK_TPB; ++q) s_e[q] = cnt[q] - 1;
    }
  }
  __syncthreads();
  float val[PACK_TPB][4];
  T* dst[PACK_TPB];
#pragma unroll
  for (int q = 0; q < PACK_TPB; ++q) {
    dst[q] = nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) val[q][i] = 0.f;
    if (tile0 + q >= total_tiles) continue;
    const seg_pack_entry e = tab[s_e[q]];
    // tile index -> (tap, chunk, n block)
    int64_t t = tile0 + q - e.blk_start;
    const int nblk = e.n_total / 32, nch = e.k_pad / 32;
    const int nb = t % nblk; t /= nblk;
    const int chunk = t % nch; const int tap = t / nch;
    const float* src = arena + e.src_off;
    const bool n_fast = (e.mode == SEG_PACK_CONV_FWD || e.mode == SEG_PACK_UP_DGRAD);   // source index runs fastest along n
    dst[q] = packed + e.dst_off + (((int64_t)tap * nch + chunk) * e.n_total + nb * 32) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int fast = idx & 31, slow = idx >> 5;
      const int kk = n_fast ? slow : fast, nn = n_fast ? fast : slow;     // element (k = chunk*32+kk, logical n = nb*32+nn)
      const int k = chunk * 32 + kk, n = nb * 32 + nn;
      if (e.mode == SEG_PACK_CONV_FWD) {
        const int ci = seg_ci(e, k);
        if (ci >= 0 && n < e.cout) val[q][i] = src[((int64_t)tap * e.cin + ci) * e.cout + n];
      } else if (e.mode == SEG_PACK_CONV_DGRAD) {
        const int ci = seg_ci(e, n);
        const int u = e.KH - 1 - tap / e.KW, v = e.KW - 1 - tap % e.KW;
        if (ci >= 0 && k < e.cout) val[q][i] = src[((int64_t)(u * e.KW + v) * e.cin + ci) * e.cout + k];
      } else if (e.mode == SEG_PACK_UP_FWD) {
        const int ci = seg_ci(e, k);
        const int tp = n / e.cout_pad, co = n % e.cout_pad;
        if (ci >= 0 && co < e.cout && tp < 4) val[q][i] = src[((int64_t)tp * e.cout + co) * e.cin + ci];
      } else {  // SEG_PACK_UP_DGRAD
        const int ci = seg_ci(e, n);
        if (ci >= 0 && k < e.cout) val[q][i] = src[((int64_t)tap * e.cout + k) * e.cin + ci];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PACK_TPB; ++q) {
    if (dst[q] == nullptr) continue;
    const bool n_fast = (tab[s_e[q]].mode == SEG_PACK_CONV_FWD || tab[s_e[q]].mode == SEG_PACK_UP_DGRAD);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int fast = idx & 31, slow = idx >> 5;
      tile[q][n_fast ? fast : slow][n_fast ? slow : fast] = val[q][i];
    }
  }
  __syncthreads();
  // 16-byte stores: 8 consecutive k of one packed row per thread (bf16: 128 threads cover a tile; f32: two passes)
  constexpr int EPT = 8;
#pragma unroll
  for (int q = 0; q < PACK_TPB; ++q) {
    if (dst[q] == nullptr) continue;
    for (int idx = threadIdx.x; idx < 1024 / EPT; idx += 256) {
      const int pos = idx >> 2, kk0 = (idx & 3) * EPT;                    // packed row position, first k
      const int nn = (((pos >> 2) & 3) << 3) | (((pos >> 4) & 1) << 2) | (pos & 3);   // -> logical row
      Vec8<T> o;
#pragma unroll
      for (int j = 0; j < EPT; ++j) o.set(j, tile[q][nn][kk0 + j]);
      o.store(dst[q] + pos * 32 + kk0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Re-pack with ONE read of the fp32 arena for both packed copies: the walk goes over the 32(k) x 32(n) tiles of the FORWARD
// packed layout (reads are whole 128-byte lines of the TF-layout arena); a workgroup writes, from LDS, the forward tile and,
// transposed, the one dgrad tile that holds the same weights (taps flipped, the roles of the channel blocks swapped).  (A form
// of this walk that also did the Adam update was bit-identical but slower at the serial tail of the step and was removed.)
// ------------------------------------------------------------------------------------------
constexpr int AP_TPB = 2;          // (4 / 8 tiles per workgroup: C2 step 0.966 / 0.972 against 0.961 ms)
template <typename T>
__global__ __launch_bounds__(256) void pack_dual_kernel(const float* p, T* packed, const seg_pack_entry* tab, const int64_t* dgrad_off,
                                                        int n_entries, int64_t total_tiles) {
  __shared__ int s_e[AP_TPB];
  __shared__ float tile[AP_TPB][32][33];
  const int64_t tile_blocks = (total_tiles + AP_TPB - 1) / AP_TPB;
  if ((int64_t)blockIdx.x >= tile_blocks) return;
  const int64_t tile0 = (int64_t)blockIdx.x * AP_TPB;
  if (threadIdx.x < 64) {
    int cnt[AP_TPB];
#pragma unroll
    for (int q = 0; q < AP_TPB; ++q) cnt[q] = 0;
    for (int base = 0; base < n_entries; base += 64) {
      const int i = base + threadIdx.x;
      const int64_t bs = i < n_entries ? tab[i].blk_start : INT64_MAX;
#pragma unroll
      for (int q = 0; q < AP_TPB; ++q) cnt[q] += __popcll(__ballot(tile0 + q >= bs));
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < AP_TPB; ++q) s_e[q] = cnt[q] - 1;
    }
  }
  __syncthreads();
  float pp[AP_TPB][4];
  int64_t sidx[AP_TPB][4];
  T* dstf[AP_TPB]; T* dstd[AP_TPB];
  bool nfast[AP_TPB];
#pragma unroll
  for (int q = 0; q < AP_TPB; ++q) {
    dstf[q] = nullptr; dstd[q] = nullptr; nfast[q] = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) { sidx[q][i] = -1; pp[q][i] = 0.f; }
    if (tile0 + q >= total_tiles) continue;
    const seg_pack_entry e = tab[s_e[q]];
    int64_t t = tile0 + q - e.blk_start;
    const int nblk = e.n_total / 32, nch = e.k_pad / 32;
    const int nb = t % nblk; t /= nblk;
    const int chunk = t % nch; const int tap = t / nch;
    const bool conv = e.mode == SEG_PACK_CONV_FWD;
    nfast[q] = conv;                             // source index runs fastest along n (conv) / along k (transposed conv)
    dstf[q] = packed + e.dst_off + (((int64_t)tap * nch + chunk) * e.n_total + nb * 32) * 32;
    const int64_t doff = dgrad_off[s_e[q]];
    const int ntaps = e.KH * e.KW;
    if (doff >= 0) {
      if (conv) dstd[q] = packed + doff + (((int64_t)(ntaps - 1 - tap) * nblk + nb) * e.k_pad + chunk * 32) * 32;
      else {
        const int tp = (nb * 32) / e.cout_pad, cob = ((nb * 32) % e.cout_pad) / 32;
        dstd[q] = packed + doff + (((int64_t)tp * (e.cout_pad / 32) + cob) * e.k_pad + chunk * 32) * 32;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int fast = idx & 31, slow = idx >> 5;
      const int kk = conv ? slow : fast, nn = conv ? fast : slow;
      const int k = chunk * 32 + kk, n = nb * 32 + nn;
      const int ci = seg_ci(e, k);
      if (conv) {
        if (ci >= 0 && n < e.cout) sidx[q][i] = e.src_off + ((int64_t)tap * e.cin + ci) * e.cout + n;
      } else {
        const int tp = n / e.cout_pad, co = n % e.cout_pad;
        if (ci >= 0 && co < e.cout && tp < 4) sidx[q][i] = e.src_off + ((int64_t)tp * e.cout + co) * e.cin + ci;
      }
      if (sidx[q][i] >= 0) pp[q][i] = p[sidx[q][i]];
    }
  }
#pragma unroll
  for (int q = 0; q < AP_TPB; ++q) {
    if (dstf[q] == nullptr) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * 256;
      const int fast = idx & 31, slow = idx >> 5;
      tile[q][nfast[q] ? fast : slow][nfast[q] ? slow : fast] = pp[q][i];       // [n][k]; 0 where the tile is padding
    }
  }
  __syncthreads();
  constexpr int EPT = 8;
#pragma unroll
  for (int q = 0; q < AP_TPB; ++q) {
    if (dstf[q] == nullptr) continue;
    for (int idx = threadIdx.x; idx < 1024 / EPT; idx += 256) {
      const int pos = idx >> 2, kk0 = (idx & 3) * EPT;                    // packed row position, first k
      const int nn = (((pos >> 2) & 3) << 3) | (((pos >> 4) & 1) << 2) | (pos & 3);   // -> logical row
      Vec8<T> o;
#pragma unroll
      for (int j = 0; j < EPT; ++j) o.set(j, tile[q][nn][kk0 + j]);
      o.store(dstf[q] + pos * 32 + kk0);
      if (dstd[q] != nullptr) {                                           // dgrad tile: rows = input channels, k = output channels
        Vec8<T> od;
#pragma unroll
        for (int j = 0; j < EPT; ++j) od.set(j, tile[q][kk0 + j][nn]);
        od.store(dstd[q] + pos * 32 + kk0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// depthwise bilinear transposed conv (+crop/pad, + skip add)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void bilinear_fwd_kernel(seg_view src, int Hs, int Ws, int f, const float* filt, seg_view add, seg_view dst,
                                    int Hd, int Wd, int cy, int cx, int B, int C8, int dst_f32) {
  const int k = 2 * f - f % 2, pb = (k - f) / 2;
  const int64_t total = (int64_t)B * Hd * Wd * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, Wd, Hd);
    const int c8 = q_.c8, x = q_.x, y = q_.y, b = q_.b;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int Y = y + cy, X = x + cx;
    if (Y >= 0 && Y < Hs * f && X >= 0 && X < Ws * f) {
      for (int u = (Y + pb) % f; u < k; u += f) {
        const int iy = (Y + pb - u) / f;
        if (iy < 0 || iy >= Hs) continue;
        for (int v = (X + pb) % f; v < k; v += f) {
          const int ix = (X + pb - v) / f;
          if (ix < 0 || ix >= Ws) continue;
          const float w = filt[u * k + v];
          Vec8<T> s; s.load(reinterpret_cast<const T*>(src.ptr) + view_off(src, b, iy, ix) + c8 * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] += w * s.get(e);
        }
      }
    }
    if (add.ptr != nullptr) {
      Vec8<T> s; s.load(reinterpret_cast<const T*>(add.ptr) + view_off(add, b, y, x) + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += s.get(e);
    }
    const int64_t off = view_off(dst, b, y, x) + c8 * 8;
    if (dst_f32) { Vec8<float> o; for (int e = 0; e < 8; ++e) o.set(e, a[e]); o.store(reinterpret_cast<float*>(dst.ptr) + off); }
    else { Vec8<T> o; for (int e = 0; e < 8; ++e) o.set(e, a[e]); o.store(reinterpret_cast<T*>(dst.ptr) + off); }
  }
}

// LP lanes share one (source pixel, 8 channels): lane r takes the filter rows r, r + LP, ... and the LP partial sums meet in a
// fixed butterfly.  (One thread per source pixel walked all k*k = 256 taps of the 8x layer alone: 131 k threads for a 134 MB
// tensor, 146 us per FCN-8s step.)
template <typename T, int LP>
__global__ void bilinear_bwd_kernel(seg_view mk, seg_view dzm, seg_view dd, int Hd, int Wd, int cy, int cx, int f, const float* filt, seg_view ds,
                                    int Hs, int Ws, int B, int C8, int dd_f32) {
  const int k = 2 * f - f % 2, pb = (k - f) / 2;
  const int64_t total = (int64_t)B * Hs * Ws * C8 * LP;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < total; i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + threadIdx.x;                    // (blockDim.x and the grid stride are multiples of LP: a group stays together)
    const bool on = i < total;
    int64_t t = on ? i / LP : 0;
    const int r = (int)(i % LP);
    const int c8 = t % C8; t /= C8;
    const int ix = t % Ws; t /= Ws;
    const int iy = t % Hs; const int b = t / Hs;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (on)
    for (int u = r; u < k; u += LP) {
      const int Y = iy * f + u - pb, y = Y - cy;          // Y: position in the SAME-cropped upsampled map
      if (Y < 0 || Y >= Hs * f || y < 0 || y >= Hd) continue;
      for (int v = 0; v < k; ++v) {
        const int X = ix * f + v - pb, x = X - cx;
        if (X < 0 || X >= Ws * f || x < 0 || x >= Wd) continue;
        const float w = filt[u * k + v];
        const int64_t off = view_off(dd, b, y, x) + c8 * 8;
        if (dd_f32) { Vec8<float> s; s.load(reinterpret_cast<const float*>(dd.ptr) + off); for (int e = 0; e < 8; ++e) a[e] += w * s.get(e); }
        else { Vec8<T> s; s.load(reinterpret_cast<const T*>(dd.ptr) + off); for (int e = 0; e < 8; ++e) a[e] += w * s.get(e); }
      }
    }
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += __shfl_xor(a[e], o, 64);
    if (on && r == 0) {
      Vec8<T> o; for (int e = 0; e < 8; ++e) o.set(e, a[e]);
      o.store(reinterpret_cast<T*>(ds.ptr) + view_off(ds, b, iy, ix) + c8 * 8);
      if (dzm.ptr != nullptr) {          // the same gradient behind the ReLU of `mk` (the score convolution's own activation)
        Vec8<T> m; m.load(reinterpret_cast<const T*>(mk.ptr) + view_off(mk, b, iy, ix) + c8 * 8);
        for (int e = 0; e < 8; ++e) o.set(e, m.get(e) > 0.f ? o.get(e) : 0.f);
        o.store(reinterpret_cast<T*>(dzm.ptr) + view_off(dzm, b, iy, ix) + c8 * 8);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// FCN head, fused: logits = crop(bilinear_up_f(score))  ->  softmax x-entropy  ->  dlogits, without the float logits
// tensor in between (models/fcn.py:199-218 + models/basemodel.py:59-70).  At 512^2 x 21 classes (padded 32) the separate
// launches wrote and re-read 268 MB of float logits per step (bilinear_fwd 89 us + softmax_xent 116 us of a 1.09 ms step);
// the score map is L2-resident, so the fused form moves only the labels and the dlogits.  LPX lanes share a pixel (4
// channels each): every access is one contiguous run per pixel, max / sum go through LPX-wide butterflies.
// logits_out (nullable): also store the float logits (tests, `y_hat`).
// ------------------------------------------------------------------------------------------
template <typename T, int LPX, int CPL, int F>   // LPX lanes share a pixel, CPL channels each; F: the factor at compile time (0 = run time)
__global__ void bilinear_xent_kernel(seg_view src, int Hs, int Ws, int f_rt, const float* filt, int cy, int cx, const uint8_t* labels, int LH, int LW,
                                     int ly0, int lx0, int B, int H, int W, int nc, float inv_n, float gscale, float* loss_sum, seg_view dl,
                                     seg_view lo) {
  const int f = F ? F : f_rt;
  const int k = 2 * f - f % 2, pb = (k - f) / 2;
  // blockIdx.x: a 256-thread run of one image row (x = column, r = channel group); blockIdx.y strides over the B*H rows: no
  // 64-bit divisions per pixel (they were most of this kernel's instructions)
  float local = 0.f;
  const int xi = blockIdx.x * blockDim.x + threadIdx.x;
  const int x = xi / LPX, r = xi % LPX;
  for (int row = blockIdx.y; row < B * H; row += gridDim.y) {
    const int b = row / H, y = row - b * H;
    const bool on = x < W;
    float zv[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) zv[e] = 0.f;
    const int Y = y + cy, X = x + cx;
    if (on && Y >= 0 && Y < Hs * f && X >= 0 && X < Ws * f) {
      for (int u = (Y + pb) % f; u < k; u += f) {
        const int iy = (Y + pb - u) / f;
        if (iy < 0 || iy >= Hs) continue;
        for (int v = (X + pb) % f; v < k; v += f) {
          const int ix = (X + pb - v) / f;
          if (ix < 0 || ix >= Ws) continue;
          const float w = filt[u * k + v];
          const T* sp = reinterpret_cast<const T*>(src.ptr) + view_off(src, b, iy, ix) + r * CPL;
#pragma unroll
          for (int q = 0; q < CPL / 4; ++q) {
            if constexpr (sizeof(T) == 2) {               // 4 channels per 8-byte load
              const bf16x4 v4 = *reinterpret_cast<const bf16x4*>(sp + q * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) zv[q * 4 + e] += w * (float)v4[e];
            } else {
              const f32x4 v4 = *reinterpret_cast<const f32x4*>(sp + q * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) zv[q * 4 + e] += w * v4[e];
            }
          }
        }
      }
    }
    if (on && lo.ptr != nullptr) {
      float* lp = reinterpret_cast<float*>(lo.ptr) + view_off(lo, b, y, x) + r * CPL;
#pragma unroll
      for (int q = 0; q < CPL / 4; ++q) *reinterpret_cast<f32x4*>(lp + q * 4) = f32x4{zv[q * 4], zv[q * 4 + 1], zv[q * 4 + 2], zv[q * 4 + 3]};
    }
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < CPL; ++e) { if (r * CPL + e >= nc) zv[e] = -INFINITY; m = fmaxf(m, zv[e]); }
#pragma unroll
    for (int o = LPX / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float ev[CPL], s = 0.f;
#pragma unroll
    for (int e = 0; e < CPL; ++e) { ev[e] = r * CPL + e < nc ? (sizeof(T) == 2 ? __expf(zv[e] - m) : expf(zv[e] - m)) : 0.f; s += ev[e]; }
#pragma unroll
    for (int o = LPX / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const int lab = on ? labels[((int64_t)b * LH + y + ly0) * LW + x + lx0] : 255;
    const bool valid = lab < nc;
    if (on && valid && lab / CPL == r) {
      float zl = 0.f;
#pragma unroll
      for (int e = 0; e < CPL; ++e) if (e == lab % CPL) zl = zv[e];
      local += (sizeof(T) == 2 ? __logf(s) : logf(s)) - (zl - m);
    }
    if (on) {
      const float kk = inv_n * gscale, rs = 1.f / s;
      T* op = reinterpret_cast<T*>(dl.ptr) + view_off(dl, b, y, x) + r * CPL;
#pragma unroll
      for (int q = 0; q < CPL / 4; ++q) {
        if (r * CPL + q * 4 >= dl.c) break;
        float gv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = r * CPL + q * 4 + e;
          gv[e] = (c < nc && valid) ? (ev[q * 4 + e] * rs - (c == lab ? 1.f : 0.f)) * kk : 0.f;
        }
        if constexpr (sizeof(T) == 2) *reinterpret_cast<bf16x4*>(op + q * 4) = bf16x4{(bf16_t)gv[0], (bf16_t)gv[1], (bf16_t)gv[2], (bf16_t)gv[3]};
        else *reinterpret_cast<f32x4*>(op + q * 4) = f32x4{gv[0], gv[1], gv[2], gv[3]};
      }
    }
  }
  __shared__ float wsum[4];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * inv_n);
}

// adjoint of the bilinear up-sampling, separable: the tent filter bank is an outer product (utils/upsampling.py:6-24), so the
// k x k gather becomes a horizontal pass (dd -> tmp[b][y][ix][c], float) and a vertical one (tmp -> ds): 2k taps instead of
// k*k per source pixel and every dd element read once instead of four times.  Row / column weights: filt[c0*k + j] / sqrt(filt[c0*k + c0]).
template <typename T>
__global__ void bilinear_bwd_h_kernel(seg_view dd, int Hd, int Wd, int cx, int f, const float* filt, float* tmp, int Ws, int B, int C8, int dd_f32) {
  const int k = 2 * f - f % 2, pb = (k - f) / 2, c0 = (k - 1) / 2;
  const float nrm = rsqrtf(filt[c0 * k + c0]);
  const int64_t total = (int64_t)B * Hd * Ws * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, Ws, Hd);
    const int c8 = q_.c8, ix = q_.x, y = q_.y, b = q_.b;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int v = 0; v < k; ++v) {
      const int X = ix * f + v - pb, x = X - cx;
      if (X < 0 || X >= Ws * f || x < 0 || x >= Wd) continue;
      const float w = filt[c0 * k + v] * nrm;
      const int64_t off = view_off(dd, b, y, x) + c8 * 8;
      if (dd_f32) { Vec8<float> s; s.load(reinterpret_cast<const float*>(dd.ptr) + off); for (int e = 0; e < 8; ++e) a[e] += w * s.get(e); }
      else { Vec8<T> s; s.load(reinterpret_cast<const T*>(dd.ptr) + off); for (int e = 0; e < 8; ++e) a[e] += w * s.get(e); }
    }
    Vec8<float> o; for (int e = 0; e < 8; ++e) o.set(e, a[e]);
    o.store(tmp + (((int64_t)b * Hd + y) * Ws + ix) * (C8 * 8) + c8 * 8);
  }
}

template <typename T>
__global__ void bilinear_bwd_v_kernel(seg_view mk, seg_view dzm, const float* tmp, int Hd, int cy, int f, const float* filt, seg_view ds, int Hs, int Ws, int B, int C8) {
  const int k = 2 * f - f % 2, pb = (k - f) / 2, c0 = (k - 1) / 2;
  const float nrm = rsqrtf(filt[c0 * k + c0]);
  const int64_t total = (int64_t)B * Hs * Ws * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx4 q_ = split4(i, C8, Ws, Hs);
    const int c8 = q_.c8, ix = q_.x, iy = q_.y, b = q_.b;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int u = 0; u < k; ++u) {
      const int Y = iy * f + u - pb, y = Y - cy;
      if (Y < 0 || Y >= Hs * f || y < 0 || y >= Hd) continue;
      const float w = filt[u * k + c0] * nrm;
      Vec8<float> s; s.load(tmp + (((int64_t)b * Hd + y) * Ws + ix) * (C8 * 8) + c8 * 8);
      for (int e = 0; e < 8; ++e) a[e] += w * s.get(e);
    }
    Vec8<T> o; for (int e = 0; e < 8; ++e) o.set(e, a[e]);
    o.store(reinterpret_cast<T*>(ds.ptr) + view_off(ds, b, iy, ix) + c8 * 8);
    if (dzm.ptr != nullptr) {
      Vec8<T> m; m.load(reinterpret_cast<const T*>(mk.ptr) + view_off(mk, b, iy, ix) + c8 * 8);
      for (int e = 0; e < 8; ++e) o.set(e, m.get(e) > 0.f ? o.get(e) : 0.f);
      o.store(reinterpret_cast<T*>(dzm.ptr) + view_off(dzm, b, iy, ix) + c8 * 8);
    }
  }
}

seg_view null_view() { seg_view v; memset(&v, 0, sizeof(v)); return v; }

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define DISPATCH(dtype, CALL_F32, CALL_BF16)                      \
  if ((dtype) == SEG_F32) { CALL_F32; }                           \
  else if ((dtype) == SEG_BF16) { CALL_BF16; }                    \
  else { seg_set_error("bad dtype %d", (int)(dtype)); return SEG_ERR_ARG; }

extern "C" int seg_maxpool2x2_fwd(const seg_view* src, const seg_view* dst, uint8_t* idx, int32_t B, int32_t Ho, int32_t Wo,
                                  int32_t C, int32_t dtype, void* stream) {
  if (!view_ok(src, 2 * Ho, 2 * Wo, C) || !view_ok(dst, Ho, Wo, C) || C % 8 || B <= 0) { seg_set_error("maxpool_fwd: bad views"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * Ho * Wo * (C / 8);
  DISPATCH(dtype,
           SEG_LAUNCH(maxpool_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *src, *dst, idx, B, Ho, Wo, C / 8),
           SEG_LAUNCH(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *src, *dst, idx, B, Ho, Wo, C / 8));
  return seg_check_launch("maxpool_fwd");
}

extern "C" int seg_maxpool2x2_bwd(const seg_view* y_act, const seg_view* dpool, const seg_view* add, int32_t add_h, int32_t add_w,
                                  int32_t add_y0, int32_t add_x0, const seg_view* dz, int32_t B, int32_t H, int32_t W, int32_t C,
                                  int32_t dtype, void* stream) {
  if (!view_ok(y_act, H, W, C) || !view_ok(dz, H, W, C) || C % 8 || B <= 0) { seg_set_error("maxpool_bwd: bad views"); return SEG_ERR_ARG; }
  if (dpool && dpool->ptr && !view_ok(dpool, H / 2, W / 2, C)) { seg_set_error("maxpool_bwd: bad dpool view"); return SEG_ERR_ARG; }
  if (add && add->ptr && (!view_ok(add, add_h, add_w, C) || add_y0 < 0 || add_x0 < 0)) { seg_set_error("maxpool_bwd: bad add view"); return SEG_ERR_ARG; }
  const seg_view dpv = (dpool && dpool->ptr) ? *dpool : null_view();
  const seg_view adv = (add && add->ptr) ? *add : null_view();
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
  DISPATCH(dtype,
           SEG_LAUNCH(maxpool_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *y_act, dpv, adv, add_h, add_w, add_y0, add_x0, *dz, B, H, W, C / 8),
           SEG_LAUNCH(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *y_act, dpv, adv, add_h, add_w, add_y0, add_x0, *dz, B, H, W, C / 8));
  return seg_check_launch("maxpool_bwd");
}

extern "C" int seg_relu_grad(const seg_view* dy, const seg_view* y_act, const seg_view* dz, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t dtype, void* stream) {
  if (!view_ok(dy, H, W, C) || !view_ok(y_act, H, W, C) || !view_ok(dz, H, W, C) || C % 8) { seg_set_error("relu_grad: bad views"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * H * W * (C / 8);
  DISPATCH(dtype,
           SEG_LAUNCH(relu_grad_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *dy, *y_act, *dz, B, H, W, C / 8),
           SEG_LAUNCH(relu_grad_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *dy, *y_act, *dz, B, H, W, C / 8));
  return seg_check_launch("relu_grad");
}

extern "C" int seg_round_bf16(const seg_view* src, const seg_view* dst, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!view_ok(src, H, W, C) || !view_ok(dst, H, W, C) || C % 8 || B < 1) { seg_set_error("round_bf16: bad views"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * H * W * (C / 8);
  SEG_LAUNCH(round_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), *src, *dst, B, H, W, C / 8);
  return seg_check_launch("round_bf16");
}
extern "C" int seg_round_bf16_flat(const float* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 1) { seg_set_error("round_bf16_flat: bad arguments"); return SEG_ERR_ARG; }
  SEG_LAUNCH(round_bf16_flat_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), src, dst, n);
  return seg_check_launch("round_bf16_flat");
}

extern "C" int seg_cast_pad(const float* x, int64_t npix, int32_t c, const seg_view* dst, int32_t dtype, void* stream) {
  if (!x || !dst || !dst->ptr || dst->cs % 8 || c > dst->cs || npix <= 0) { seg_set_error("cast_pad: bad args"); return SEG_ERR_ARG; }
  const int64_t n = npix * (dst->cs / 8);
  DISPATCH(dtype,
           SEG_LAUNCH(cast_pad_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, npix, c, *dst),
           SEG_LAUNCH(cast_pad_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, npix, c, *dst));
  return seg_check_launch("cast_pad");
}

static int dropout_launch(const seg_view* x, const seg_view* y, int32_t B, int32_t H, int32_t W, int32_t C, float keep, uint64_t seed,
                          uint64_t offset, const int64_t* step_dev, int32_t dtype, void* stream) {
  if (!view_ok(x, H, W, C) || !view_ok(y, H, W, C) || C % 8 || !(keep > 0.f && keep <= 1.f)) { seg_set_error("dropout: bad args"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * H * W * (C / 8);
  DISPATCH(dtype,
           SEG_LAUNCH(dropout_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *x, *y, B, H, W, C / 8, keep, seed, offset, step_dev),
           SEG_LAUNCH(dropout_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *x, *y, B, H, W, C / 8, keep, seed, offset, step_dev));
  return seg_check_launch("dropout");
}

extern "C" int seg_dropout(const seg_view* x, const seg_view* y, int32_t B, int32_t H, int32_t W, int32_t C, float keep,
                           uint64_t seed, uint64_t offset, int32_t dtype, void* stream) {
  return dropout_launch(x, y, B, H, W, C, keep, seed, offset, nullptr, dtype, stream);
}

extern "C" int seg_dropout_step(const seg_view* x, const seg_view* y, int32_t B, int32_t H, int32_t W, int32_t C, float keep,
                                uint64_t seed, uint64_t offset, const int64_t* step_dev, int32_t dtype, void* stream) {
  if (!step_dev) { seg_set_error("dropout_step: null step pointer"); return SEG_ERR_ARG; }
  return dropout_launch(x, y, B, H, W, C, keep, seed, offset, step_dev, dtype, stream);
}

extern "C" int seg_softmax_xent_probs(const seg_view* logits, const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0,
                                      int32_t B, int32_t H, int32_t W, int32_t n_classes, float inv_n, float grad_scale, float* loss_sum,
                                      const seg_view* dlogits, const seg_view* probs, int32_t dtype, void* stream);
extern "C" int seg_softmax_xent(const seg_view* logits, const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0,
                                int32_t B, int32_t H, int32_t W, int32_t n_classes, float inv_n, float grad_scale, float* loss_sum,
                                const seg_view* dlogits, int32_t dtype, void* stream) {
  return seg_softmax_xent_probs(logits, labels, LH, LW, ly0, lx0, B, H, W, n_classes, inv_n, grad_scale, loss_sum, dlogits, nullptr, dtype, stream);
}
extern "C" int seg_softmax_xent_probs(const seg_view* logits, const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0,
                                      int32_t B, int32_t H, int32_t W, int32_t n_classes, float inv_n, float grad_scale, float* loss_sum,
                                      const seg_view* dlogits, const seg_view* probs, int32_t dtype, void* stream) {
  if (probs && probs->ptr && (!view_ok(probs, H, W, probs->c) || probs->c % 8 || probs->c > 32 || n_classes > probs->c)) { seg_set_error("softmax_xent: bad probs view"); return SEG_ERR_ARG; }
  const seg_view prv = (probs && probs->ptr) ? *probs : seg_view{nullptr, 0, 0, 0, 0, 0, 0, 0};
  if (!logits || !logits->ptr || !labels || !loss_sum || !view_ok(dlogits, H, W, dlogits ? dlogits->c : 0)) { seg_set_error("softmax_xent: bad args"); return SEG_ERR_ARG; }
  if (n_classes < 1 || n_classes > 32 || n_classes > dlogits->c || dlogits->c % 8 || dlogits->c > 32) { seg_set_error("softmax_xent: n_classes %d unsupported (1..32)", n_classes); return SEG_ERR_UNSUPPORTED; }
  if (ly0 < 0 || lx0 < 0 || ly0 + H > LH || lx0 + W > LW) { seg_set_error("softmax_xent: label window out of range"); return SEG_ERR_ARG; }
  {
    const int ncp = n_classes <= 4 ? 4 : n_classes <= 8 ? 8 : n_classes <= 16 ? 16 : 32;     // floats read per pixel (16-byte vectors)
    if (logits->cs % 4 || logits->coff % 4 || logits->coff + ncp > logits->cs || (reinterpret_cast<uintptr_t>(logits->ptr) & 15)) {
      seg_set_error("softmax_xent: logits view must be 16-byte aligned with %d readable floats per pixel", ncp); return SEG_ERR_ARG;
    }
  }
  const int64_t n = (int64_t)B * H * W;
  const int g = grid_for(n, 256, 1024);
#define XENT_ARGS dim3(g), dim3(256), 0, ST(stream), *logits, labels, LH, LW, ly0, lx0, B, H, W, n_classes, inv_n, grad_scale, loss_sum, *dlogits, prv
#define XENT_NCP(TT) do { if (n_classes <= 4) SEG_LAUNCH((softmax_xent_kernel<TT, 4>), XENT_ARGS); \
    else if (n_classes <= 8) SEG_LAUNCH((softmax_xent_kernel<TT, 8>), XENT_ARGS); \
    else if (n_classes <= 16) SEG_LAUNCH((softmax_xent_kernel<TT, 16>), XENT_ARGS); \
    else SEG_LAUNCH((softmax_xent_kernel<TT, 32>), XENT_ARGS); } while (0)
  DISPATCH(dtype, XENT_NCP(float), XENT_NCP(bf16_t));
#undef XENT_NCP
#undef XENT_ARGS
  return seg_check_launch("softmax_xent");
}

// Workgroups of a seg_head_xent launch (its grid is a function of the shape only, so the partial-sum workspace can be sized
// and reduced without the launch being replayed).
static int head_xent_grid(int64_t B, int64_t H, int64_t W, int cpad) {
  // >= 4 pixels per lane group (a 68 x 68 map: 289 workgroups), at most 2048 workgroups (512 x 512: ~13 pixels each): the
  // per-workgroup partial row and the loss atomic stay cheap (16 k workgroups doubled the kernel time on the atomic alone)
  const int64_t px_per_wg = 256 / (cpad / 8) * 4;        // (2 / 1 pixels per lane group measured no better at C2: profiles/r03_step_structure_ab.txt)
  return grid_for(B * H * W, (int)px_per_wg, 2048);
}
static int head_ncp(int n_classes) { return n_classes <= 4 ? 4 : n_classes <= 8 ? 8 : n_classes <= 16 ? 16 : 32; }

extern "C" int64_t seg_head_xent_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t cin_pad, int32_t n_classes) {
  if (n_classes < 1 || n_classes > 8 || (cin_pad != 32 && cin_pad != 64) || B < 1 || H < 1 || W < 1) return 0;   // fused filter gradient: <= 8 classes
  const int ncp = head_ncp(n_classes);
  return (int64_t)head_xent_grid(B, H, W, cin_pad) * (cin_pad * ncp + ncp) * (int64_t)sizeof(float);
}

extern "C" int seg_head_xent(const seg_view* act, const float* w_hwio, const float* bias, int32_t cin, const uint8_t* labels, int32_t LH, int32_t LW,
                             int32_t ly0, int32_t lx0, int32_t B, int32_t H, int32_t W, int32_t n_classes, float inv_n, float* loss_sum,
                             const seg_view* logits, const seg_view* dlogits, const seg_view* dact, void* dw_ws, int64_t dw_ws_bytes,
                             int32_t dtype, void* stream) {
  if (!act || !act->ptr || !w_hwio || !labels || !loss_sum || !logits || !logits->ptr || !dlogits ||
      !view_ok(dact, H, W, dact ? dact->c : 0) || !view_ok(act, H, W, act->c)) { seg_set_error("head_xent: bad args"); return SEG_ERR_ARG; }
  const bool fused_dw = dw_ws != nullptr;
  if (!fused_dw && !view_ok(dlogits, H, W, dlogits->c)) { seg_set_error("head_xent: dlogits is required when the filter gradient is not fused"); return SEG_ERR_ARG; }
  if (dlogits->ptr && !view_ok(dlogits, H, W, dlogits->c)) { seg_set_error("head_xent: bad dlogits view"); return SEG_ERR_ARG; }
  if (n_classes < 1 || n_classes > 32 || n_classes > logits->cs || (dlogits->ptr && (n_classes > dlogits->c || dlogits->c % 8 || dlogits->c > 32))) {
    seg_set_error("head_xent: n_classes %d unsupported (1..32)", n_classes); return SEG_ERR_UNSUPPORTED; }
  if ((act->c != 32 && act->c != 64) || dact->c != act->c || cin < 1 || cin > act->c) { seg_set_error("head_xent: input channels %d unsupported (32 or 64 padded)", act->c); return SEG_ERR_UNSUPPORTED; }
  if (ly0 < 0 || lx0 < 0 || ly0 + H > LH || lx0 + W > LW) { seg_set_error("head_xent: label window out of range"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * H * W;
  if (n >= (int64_t)1 << 31) { seg_set_error("head_xent: %lld pixels exceed the 32-bit index range", (long long)n); return SEG_ERR_UNSUPPORTED; }
  if (fused_dw) {
    const int64_t need = seg_head_xent_ws_bytes(B, H, W, act->c, n_classes);
    if (need == 0) { seg_set_error("head_xent: the fused filter gradient covers 1..8 classes (got %d)", n_classes); return SEG_ERR_UNSUPPORTED; }
    if (dw_ws_bytes < need) { seg_set_error("head_xent: workspace of %lld bytes, %lld needed", (long long)dw_ws_bytes, (long long)need); return SEG_ERR_ARG; }
  }
  const int g = head_xent_grid(B, H, W, act->c);
  seg_view dlv = *dlogits;
  if (fused_dw) dlv.ptr = nullptr;
  float* part = reinterpret_cast<float*>(dw_ws);
#define HEAD_ARGS dim3(g), dim3(256), 0, ST(stream), *act, w_hwio, bias, cin, labels, LH, LW, ly0, lx0, B, H, W, n_classes, inv_n, loss_sum, *logits, dlv, *dact, part
#define HEAD_C(TT, NCP, DW) do { if (act->c == 32) SEG_LAUNCH((head_xent_kernel<TT, NCP, 32, DW>), HEAD_ARGS); else SEG_LAUNCH((head_xent_kernel<TT, NCP, 64, DW>), HEAD_ARGS); } while (0)
#define HEAD_NCP(TT) do { if (fused_dw) { if (n_classes <= 4) HEAD_C(TT, 4, true); else HEAD_C(TT, 8, true); } \
    else if (n_classes <= 4) HEAD_C(TT, 4, false); else if (n_classes <= 8) HEAD_C(TT, 8, false); else if (n_classes <= 16) HEAD_C(TT, 16, false); else HEAD_C(TT, 32, false); } while (0)
  DISPATCH(dtype, HEAD_NCP(float), HEAD_NCP(bf16_t));
#undef HEAD_NCP
#undef HEAD_C
#undef HEAD_ARGS
  return seg_check_launch("head_xent");
}

// The second half of the fused filter gradient: workspace rows -> dw [cin][n_classes] (fp32 HWIO) and db [n_classes].  Same
// shape arguments as the seg_head_xent launch that filled the workspace.
extern "C" int seg_head_dw_reduce(const void* dw_ws, int64_t dw_ws_bytes, int32_t B, int32_t H, int32_t W, int32_t cin_pad, int32_t cin,
                                  int32_t n_classes, float* dw, float* db, void* stream) {
  const int64_t need = seg_head_xent_ws_bytes(B, H, W, cin_pad, n_classes);
  if (!dw_ws || !dw || need == 0 || dw_ws_bytes < need || cin < 1 || cin > cin_pad) { seg_set_error("head_dw_reduce: bad args"); return SEG_ERR_ARG; }
  const int ncp = head_ncp(n_classes), rows = head_xent_grid(B, H, W, cin_pad);
  SEG_LAUNCH(head_dw_reduce_kernel, dim3(cin_pad * ncp + ncp), dim3(256), 0, ST(stream), reinterpret_cast<const float*>(dw_ws), rows, ncp, cin_pad, cin,
             n_classes, dw, db);
  return seg_check_launch("head_dw_reduce");
}

extern "C" int seg_sigmoid_argmax(const seg_view* logits, int32_t B, int32_t H, int32_t W, int32_t n_classes, float* sig, float* out,
                                  void* stream) {
  if (!logits || !logits->ptr || !sig || !out || n_classes < 1 || n_classes > logits->cs) { seg_set_error("sigmoid_argmax: bad args"); return SEG_ERR_ARG; }
  const int64_t n = (int64_t)B * H * W;
  SEG_LAUNCH(sigmoid_argmax_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), *logits, B, H, W, n_classes, sig, out);
  return seg_check_launch("sigmoid_argmax");
}

// ------------------------------------------------------------------------------------------
// 3x3 / stride-1 convolution of THIN tensors (<= 8 channels at a channel stride of 8; engine.Act(thin=True)) on the vector ALU:
// the DeconvModel's conv_out (n_classes -> n_classes at full resolution, models/deconvolution.py:170).  On the MFMA kernels the
// layer is 32 x 32 padded channels of matrix work for 2 x 2 real ones (146 us forward, 207 us data gradient at 512^2 x 16); here a
// thread owns an output pixel, reads its nine 16-byte neighbours and runs NC x NC x 9 FMAs -- HBM-bound (one read + one write of the
// map).  DGRAD: the same walk with the filter flipped and transposed (dX = dZ (*) rot180(W)^T), optional ReLU-grad mask.
// ------------------------------------------------------------------------------------------
template <typename T, int NC, bool OUT_F32>
__global__ __launch_bounds__(256) void thin_conv3x3_kernel(seg_view src, const float* w, const float* bias, int cin, int cout, int pad, int relu,
                                                           int dgrad, seg_view mask, seg_view dst, int B, int Ho, int Wo, int Hi, int Wi, int remap,
                                                           const float* bn_stats, const float* bn_beta, int bn_clog) {
  __shared__ float sw[9 * NC * NC];          // [tap][ci][co] as THIS launch consumes it (zero above the logical counts)
  __shared__ float sb[NC];
  // bn_stats != NULL: the source is the PRE-batch-norm activation and every value read is normalised on the fly, rounded as the
  // stored normalised tensor would have been (so the results are bit-identical to reading that tensor, which then need not exist)
  float bm[NC], br[NC], bb[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool on = bn_stats != nullptr && c < bn_clog;
    bm[c] = on ? bn_stats[c] : 0.f; br[c] = on ? bn_stats[8 + c] : (bn_stats != nullptr ? 0.f : 1.f); bb[c] = on ? bn_beta[c] : 0.f;
  }
  for (int i = threadIdx.x; i < 9 * NC * NC; i += 256) {
    const int tap = i / (NC * NC), ci = (i / NC) % NC, co = i % NC;
    float v = 0.f;
    if (!dgrad) { if (ci < cin && co < cout) v = w[((int64_t)tap * cin + ci) * cout + co]; }
    else { if (ci < cout && co < cin) v = w[((int64_t)(8 - tap) * cin + co) * cout + ci]; }      // input channel of this walk = the layer's output channel
    sw[i] = to_f32(from_f32<T>(v));          // (the compute dtype's weight grid, as the packed MFMA operands of every other layer)
  }
  if (threadIdx.x < NC) sb[threadIdx.x] = (bias != nullptr && !dgrad && (int)threadIdx.x < cout) ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int64_t total = (int64_t)B * Ho * Wo;
  const T* sp = reinterpret_cast<const T*>(src.ptr);
  const unsigned vb = xcd_block(blockIdx.x, gridDim.x, remap);
  for (int64_t i = vb * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx3 q_ = split3(i, Wo, Ho);
    float acc[NC];
#pragma unroll
    for (int co = 0; co < NC; ++co) acc[co] = sb[co];
    // all nine loads first, at clamped addresses, the out-of-image taps zeroed by a select: with a branch per tap the loads could
    // not be issued together and a pixel cost nine load latencies in a row
    if constexpr (NC <= 4) {
    Vec8<T> xv[9]; bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = q_.y - pad + t / 3, ix = q_.x - pad + t % 3;
      ok[t] = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
      const int cy = iy < 0 ? 0 : (iy >= Hi ? Hi - 1 : iy), cx = ix < 0 ? 0 : (ix >= Wi ? Wi - 1 : ix);
      xv[t].load(sp + view_off(src, q_.b, cy, cx));
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* wt = sw + t * NC * NC;
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        float xf = xv[t].get(ci);
        if (bn_stats != nullptr) xf = to_f32(from_f32<T>((xf - bm[ci]) * br[ci] + bb[ci]));
        xf = ok[t] ? xf : 0.f;
#pragma unroll
        for (int co = 0; co < NC; ++co) acc[co] = fmaf(xf, wt[ci * NC + co], acc[co]);
      }
    }
    } else {                                   // (8 x 8 channels: the hoisted form spills; one tap at a time)
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {
        const int iy = q_.y - pad + t / 3, ix = q_.x - pad + t % 3;
        if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
        Vec8<T> xv;
        xv.load(sp + view_off(src, q_.b, iy, ix));
        const float* wt = sw + t * NC * NC;
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
          float xf = xv.get(ci);
          if (bn_stats != nullptr) xf = to_f32(from_f32<T>((xf - bm[ci]) * br[ci] + bb[ci]));
#pragma unroll
          for (int co = 0; co < NC; ++co) acc[co] = fmaf(xf, wt[ci * NC + co], acc[co]);
        }
      }
    }
    if (relu) {
#pragma unroll
      for (int co = 0; co < NC; ++co) acc[co] = fmaxf(acc[co], 0.f);
    }
    if (mask.ptr != nullptr) {
      Vec8<T> mv;
      mv.load(reinterpret_cast<const T*>(mask.ptr) + view_off(mask, q_.b, q_.y, q_.x));
#pragma unroll
      for (int co = 0; co < NC; ++co) acc[co] = mv.get(co) > 0.f ? acc[co] : 0.f;
    }
    const int64_t doff = view_off(dst, q_.b, q_.y, q_.x);
    if (OUT_F32) {
      Vec8<float> o; o.zero();
#pragma unroll
      for (int co = 0; co < NC; ++co) o.set(co, acc[co]);
      o.store(reinterpret_cast<float*>(dst.ptr) + doff);
    } else {
      Vec8<T> o; o.zero();
#pragma unroll
      for (int co = 0; co < NC; ++co) o.set(co, acc[co]);
      o.store(reinterpret_cast<T*>(dst.ptr) + doff);
    }
  }
}

static int thin_conv_launch(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const float* w_hwio, const float* bias, int32_t cin,
                            int32_t cout, int32_t pad, int32_t relu, int32_t dgrad, const seg_view* mask, const seg_view* dst, int32_t Ho,
                            int32_t Wo, int32_t out_f32, const float* bn_stats, const float* bn_beta, int32_t dtype, void* stream);
extern "C" int seg_thin_conv3x3(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const float* w_hwio, const float* bias, int32_t cin,
                                int32_t cout, int32_t pad, int32_t relu, int32_t dgrad, const seg_view* mask, const seg_view* dst, int32_t Ho,
                                int32_t Wo, int32_t out_f32, int32_t dtype, void* stream) {
  return thin_conv_launch(src, B, Hi, Wi, w_hwio, bias, cin, cout, pad, relu, dgrad, mask, dst, Ho, Wo, out_f32, nullptr, nullptr, dtype, stream);
}
/* Forward of seg_thin_conv3x3 on the batch norm of `src` without that tensor: src is the PRE-batch-norm activation, bn_stats the batch
 * norm's [mean[8] | rstd[8]] (seg_bn_fwd's `stats` of an 8-channel layer) and bn_beta its beta[cin]; every value is normalised and
 * rounded on load exactly as seg_bn_fwd would have stored it (models/deconvolution.py:168-170: bn8 -> conv_out). */
extern "C" int seg_thin_conv3x3_bn(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const float* w_hwio, const float* bias, int32_t cin,
                                   int32_t cout, int32_t pad, int32_t relu, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t out_f32,
                                   const float* bn_stats, const float* bn_beta, int32_t dtype, void* stream) {
  if (!bn_stats || !bn_beta) { seg_set_error("thin_conv3x3_bn: statistics and beta are required"); return SEG_ERR_ARG; }
  return thin_conv_launch(src, B, Hi, Wi, w_hwio, bias, cin, cout, pad, relu, 0, nullptr, dst, Ho, Wo, out_f32, bn_stats, bn_beta, dtype, stream);
}
static int thin_conv_launch(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const float* w_hwio, const float* bias, int32_t cin,
                            int32_t cout, int32_t pad, int32_t relu, int32_t dgrad, const seg_view* mask, const seg_view* dst, int32_t Ho,
                            int32_t Wo, int32_t out_f32, const float* bn_stats, const float* bn_beta, int32_t dtype, void* stream) {
  const int ci_w = dgrad ? cout : cin, co_w = dgrad ? cin : cout;             // channels this walk reads / writes
  if (!src || !src->ptr || !dst || !dst->ptr || !w_hwio || cin < 1 || cin > 8 || cout < 1 || cout > 8 || pad < 0 || pad > 2 || B <= 0 ||
      src->cs != 8 || src->coff != 0 || dst->cs != 8 || dst->coff != 0 || !view_ok(dst, Ho, Wo, 8) ||
      src->oy + Hi > src->H || src->ox + Wi > src->W || Ho != Hi + 2 * pad - 2 || Wo != Wi + 2 * pad - 2 || Ho < 1 || Wo < 1 ||
      (mask && mask->ptr && (mask->cs != 8 || mask->coff != 0 || !view_ok(mask, Ho, Wo, 8))) || (out_f32 && dgrad)) {
    seg_set_error("thin_conv3x3: thin views (cs 8, coff 0), 1..8 channels, Ho = Hi + 2 pad - 2"); return SEG_ERR_ARG;
  }
  (void)ci_w; (void)co_w;
  const seg_view mk = (mask && mask->ptr) ? *mask : seg_view{nullptr, 0, 0, 0, 0, 0, 0, 0};
  const int m = cin > cout ? cin : cout;
  const int g = grid_for((int64_t)B * Ho * Wo, 256, 16384);
  const int remap = 1;                     // XCD-aware run order (r03: measured faster; the plain order is gone)
#define TC_ARGS dim3(g), dim3(256), 0, ST(stream), *src, w_hwio, bias, cin, cout, pad, relu, dgrad, mk, *dst, B, Ho, Wo, Hi, Wi, remap, bn_stats, bn_beta, cin
#define TC_NC(TT, F32) do { if (m <= 2) SEG_LAUNCH((thin_conv3x3_kernel<TT, 2, F32>), TC_ARGS); \
    else if (m <= 4) SEG_LAUNCH((thin_conv3x3_kernel<TT, 4, F32>), TC_ARGS); else SEG_LAUNCH((thin_conv3x3_kernel<TT, 8, F32>), TC_ARGS); } while (0)
  if (out_f32) { DISPATCH(dtype, TC_NC(float, true), TC_NC(bf16_t, true)); }
  else { DISPATCH(dtype, TC_NC(float, false), TC_NC(bf16_t, false)); }
#undef TC_NC
#undef TC_ARGS
  return seg_check_launch("thin_conv3x3");
}

// ------------------------------------------------------------------------------------------
// Filter + bias gradient of that layer (thin source, thin dZ) on the vector ALU, two stages in a fixed order:
//   dW[u][v][ci][co] = sum_{b,y,x} src[b, y - pad + u, x - pad + v, ci] * dz[b, y, x, co],   db[co] = sum dz[b, y, x, co]
// A thread walks output pixels and keeps TAPS x NC x NC (+ NC) sums; a workgroup leaves one row of partial sums, the second launch
// adds the rows.  NC = 8 splits the nine taps over blockIdx.y (64 sums per thread instead of 576).  On the MFMA walk the layer is
// 32 x 32 padded channels against 2 x 2 real ones: 317 us at 16 x 512^2 (the longest launch of the DeconvModel step) for two 67 MB
// tensors.
// ------------------------------------------------------------------------------------------
constexpr int TW_ROWS = 1024;                // partial-sum rows = workgroups per tap group
template <typename T, int NC, int TAPS>
__global__ __launch_bounds__(256) void thin_wgrad3x3_partial_kernel(seg_view src, seg_view dz, int pad, int B, int Ho, int Wo, int Hi, int Wi, float* ws, int remap,
                                  const float* bn_stats, const float* bn_beta, int bn_clog) {
  constexpr int NA = TAPS * NC * NC;
  float bm[NC], br[NC], bb[NC];                // (bn_stats != NULL: src is the pre-batch-norm activation, normalised on load as above)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const bool on = bn_stats != nullptr && c < bn_clog;
    bm[c] = on ? bn_stats[c] : 0.f; br[c] = on ? bn_stats[8 + c] : (bn_stats != nullptr ? 0.f : 1.f); bb[c] = on ? bn_beta[c] : 0.f;
  }
  const int tap0 = blockIdx.y * TAPS;
  float acc[NA], bsum[NC];
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) bsum[i] = 0.f;
  const int64_t total = (int64_t)B * Ho * Wo;
  const T* sp = reinterpret_cast<const T*>(src.ptr);
  const T* zp = reinterpret_cast<const T*>(dz.ptr);
  // a workgroup owns a CONTIGUOUS run of pixels (a few image rows) and an XCD a contiguous band of runs: the rows above and below
  // are this workgroup's own previous / next iterations (a grid-stride walk fetched every row three times: 187 us)
  const int64_t chunk = (total + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)xcd_block(blockIdx.x, gridDim.x, remap) * chunk, p1 = p0 + chunk < total ? p0 + chunk : total;
  for (int64_t i = p0 + threadIdx.x; i < p1; i += blockDim.x) {
    const Idx3 q_ = split3(i, Wo, Ho);
    Vec8<T> zv; zv.load(zp + view_off(dz, q_.b, q_.y, q_.x));
    float z[NC];
#pragma unroll
    for (int co = 0; co < NC; ++co) { z[co] = zv.get(co); bsum[co] += z[co]; }
    Vec8<T> xv[TAPS]; bool ok[TAPS];             // (all loads first, clamped addresses: see thin_conv3x3_kernel)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int tap = tap0 + t, u = tap / 3, v = tap - 3 * u;
      const int iy = q_.y - pad + u, ix = q_.x - pad + v;
      ok[t] = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
      const int cy = iy < 0 ? 0 : (iy >= Hi ? Hi - 1 : iy), cx = ix < 0 ? 0 : (ix >= Wi ? Wi - 1 : ix);
      xv[t].load(sp + view_off(src, q_.b, cy, cx));
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
      for (int ci = 0; ci < NC; ++ci) {
        float xf = xv[t].get(ci);
        if (bn_stats != nullptr) xf = to_f32(from_f32<T>((xf - bm[ci]) * br[ci] + bb[ci]));
        xf = ok[t] ? xf : 0.f;
#pragma unroll
        for (int co = 0; co < NC; ++co) acc[(t * NC + ci) * NC + co] = fmaf(xf, z[co], acc[(t * NC + ci) * NC + co]);
      }
    }
  }
  __shared__ float red[4][NA + NC];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NA; ++i) { const float a = wave_sum(acc[i]); if ((threadIdx.x & 63) == 0) red[wave][i] = a; }
#pragma unroll
  for (int i = 0; i < NC; ++i) { const float a = wave_sum(bsum[i]); if ((threadIdx.x & 63) == 0) red[wave][NA + i] = a; }
  __syncthreads();
  // row layout: [9 * NC * NC] filter sums (tap-major) then [NC] bias sums (written by tap group 0)
  float* row = ws + (int64_t)blockIdx.x * (9 * NC * NC + NC);
  for (int i = threadIdx.x; i < NA + NC; i += 256) {
    const float a = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    if (i < NA) row[tap0 * NC * NC + i] = a;
    else if (blockIdx.y == 0) row[9 * NC * NC + (i - NA)] = a;
  }
}

template <int NC>
__global__ __launch_bounds__(256) void thin_wgrad3x3_final_kernel(const float* ws, int rows, int cin, int cout, float* dw, float* db) {
  // 8 sums x 32 row slices per workgroup, the slices meet in LDS in a fixed order (a thread walking all rows alone is 512 dependent
  // load latencies)
  constexpr int RL = 9 * NC * NC + NC;
  __shared__ double part[32][8];
  const int el = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + el;
  double a = 0.0;
  if (i < RL)
    for (int r = sl; r < rows; r += 32) a += (double)ws[(int64_t)r * RL + i];
  part[sl][el] = a;
  __syncthreads();
  if (sl != 0 || i >= RL) return;
  a = 0.0;
#pragma unroll
  for (int q = 0; q < 32; ++q) a += part[q][el];
  if (i < 9 * NC * NC) {
    const int tap = i / (NC * NC), ci = (i / NC) % NC, co = i % NC;
    if (ci < cin && co < cout) dw[((int64_t)tap * cin + ci) * cout + co] = (float)a;
  } else if (db != nullptr && i - 9 * NC * NC < cout) db[i - 9 * NC * NC] = (float)a;
}

static int thin_wgrad_nc(int cin, int cout) { const int m = cin > cout ? cin : cout; return m <= 2 ? 2 : m <= 4 ? 4 : 8; }
extern "C" int64_t seg_thin_wgrad3x3_ws_bytes(int32_t cin, int32_t cout) {
  if (cin < 1 || cin > 8 || cout < 1 || cout > 8) return 0;
  const int nc = thin_wgrad_nc(cin, cout);
  return (int64_t)TW_ROWS * (9 * nc * nc + nc) * (int64_t)sizeof(float);
}

static int thin_wgrad_launch(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const seg_view* dz, int32_t Ho, int32_t Wo, int32_t cin,
                             int32_t cout, int32_t pad, float* dw_hwio, float* db, void* ws, int64_t ws_bytes, const float* bn_stats,
                             const float* bn_beta, int32_t dtype, void* stream);
extern "C" int seg_thin_wgrad3x3(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const seg_view* dz, int32_t Ho, int32_t Wo, int32_t cin,
                                 int32_t cout, int32_t pad, float* dw_hwio, float* db, void* ws, int64_t ws_bytes, int32_t dtype, void* stream) {
  return thin_wgrad_launch(src, B, Hi, Wi, dz, Ho, Wo, cin, cout, pad, dw_hwio, db, ws, ws_bytes, nullptr, nullptr, dtype, stream);
}
/* seg_thin_wgrad3x3 with the batch norm of `src` applied on load (src = the pre-batch-norm activation; see seg_thin_conv3x3_bn). */
extern "C" int seg_thin_wgrad3x3_bn(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const seg_view* dz, int32_t Ho, int32_t Wo, int32_t cin,
                                    int32_t cout, int32_t pad, float* dw_hwio, float* db, void* ws, int64_t ws_bytes, const float* bn_stats,
                                    const float* bn_beta, int32_t dtype, void* stream) {
  if (!bn_stats || !bn_beta) { seg_set_error("thin_wgrad3x3_bn: statistics and beta are required"); return SEG_ERR_ARG; }
  return thin_wgrad_launch(src, B, Hi, Wi, dz, Ho, Wo, cin, cout, pad, dw_hwio, db, ws, ws_bytes, bn_stats, bn_beta, dtype, stream);
}
static int thin_wgrad_launch(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const seg_view* dz, int32_t Ho, int32_t Wo, int32_t cin,
                             int32_t cout, int32_t pad, float* dw_hwio, float* db, void* ws, int64_t ws_bytes, const float* bn_stats,
                             const float* bn_beta, int32_t dtype, void* stream) {
  if (!src || !src->ptr || !dz || !dz->ptr || !dw_hwio || !ws || cin < 1 || cin > 8 || cout < 1 || cout > 8 || pad < 0 || pad > 2 || B <= 0 ||
      src->cs != 8 || src->coff != 0 || dz->cs != 8 || dz->coff != 0 || !view_ok(dz, Ho, Wo, 8) || src->oy + Hi > src->H || src->ox + Wi > src->W ||
      Ho != Hi + 2 * pad - 2 || Wo != Wi + 2 * pad - 2 || Ho < 1 || Wo < 1 || ws_bytes < seg_thin_wgrad3x3_ws_bytes(cin, cout)) {
    seg_set_error("thin_wgrad3x3: thin views (cs 8, coff 0), 1..8 channels, Ho = Hi + 2 pad - 2, workspace of seg_thin_wgrad3x3_ws_bytes"); return SEG_ERR_ARG;
  }
  const int nc = thin_wgrad_nc(cin, cout);
  const int g = grid_for((int64_t)B * Ho * Wo, 256, TW_ROWS);
  float* wsf = reinterpret_cast<float*>(ws);
  const int remap = 1;
#define TWG_ARGS(GY) dim3(g, GY), dim3(256), 0, ST(stream), *src, *dz, pad, B, Ho, Wo, Hi, Wi, wsf, remap, bn_stats, bn_beta, cin
#define TWG_NC(TT) do { if (nc == 2) SEG_LAUNCH((thin_wgrad3x3_partial_kernel<TT, 2, 9>), TWG_ARGS(1)); \
    else if (nc == 4) SEG_LAUNCH((thin_wgrad3x3_partial_kernel<TT, 4, 3>), TWG_ARGS(3)); else SEG_LAUNCH((thin_wgrad3x3_partial_kernel<TT, 8, 1>), TWG_ARGS(9)); } while (0)
  DISPATCH(dtype, TWG_NC(float), TWG_NC(bf16_t));
#undef TWG_NC
#undef TWG_ARGS
  if (int rc = seg_check_launch("thin_wgrad3x3_partial")) return rc;
  const int rl = 9 * nc * nc + nc;
  if (nc == 2) SEG_LAUNCH(thin_wgrad3x3_final_kernel<2>, dim3(cdiv(rl, 8)), dim3(256), 0, ST(stream), (const float*)wsf, g, cin, cout, dw_hwio, db);
  else if (nc == 4) SEG_LAUNCH(thin_wgrad3x3_final_kernel<4>, dim3(cdiv(rl, 8)), dim3(256), 0, ST(stream), (const float*)wsf, g, cin, cout, dw_hwio, db);
  else SEG_LAUNCH(thin_wgrad3x3_final_kernel<8>, dim3(cdiv(rl, 8)), dim3(256), 0, ST(stream), (const float*)wsf, g, cin, cout, dw_hwio, db);
  return seg_check_launch("thin_wgrad3x3_final");
}

// 2x2 / stride-2 transposed convolution INTO a thin tensor and its data gradient, on the vector ALU (the DeconvModel's deconv3_0:
// 32 -> n_classes channels, models/deconvolution.py:166): a thread owns one pixel of the SMALL map, i.e. a 2x2 block of the big one.
//   forward : big[b, 2y+a, 2x+c, co] = relu?(bias[co] + sum_ci small[b,y,x,ci] * w[a][c][co][ci])          (TF filter [2,2,Cout,Cin])
//   dgrad   : dsmall[b,y,x,ci] = (mask > 0) * sum_{a,c,co} dbig[b, 2y+a, 2x+c, co] * w[a][c][co][ci]
template <typename T, int NC, bool DGRAD>
__global__ __launch_bounds__(256) void thin_up2x2_kernel(seg_view small, seg_view big, const float* w, const float* bias, int cin, int cout,
                                                         int relu, seg_view mask, int B, int H, int W, float* bn_ws) {
  extern __shared__ float sw_up[];            // [tap][co < NC][ci < cp]  (zero above the logical counts), then bias[NC]
  const int cp = small.c;                     // padded input channels of the layer (a multiple of 8)
  for (int i = threadIdx.x; i < 4 * NC * cp; i += 256) {
    const int tap = i / (NC * cp), co = (i / cp) % NC, ci = i % cp;
    const float v = (co < cout && ci < cin) ? w[((int64_t)tap * cout + co) * cin + ci] : 0.f;
    sw_up[i] = to_f32(from_f32<T>(v));
  }
  float* sb = sw_up + 4 * NC * cp;
  if (threadIdx.x < NC) sb[threadIdx.x] = (bias != nullptr && (int)threadIdx.x < cout) ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int64_t total = (int64_t)B * H * W;
  const int G = cp / 8;
  float st1[NC], st2[NC];                     // bn_ws (forward): per-channel sum / sum of squares of the values this thread stores
#pragma unroll
  for (int co = 0; co < NC; ++co) st1[co] = st2[co] = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const Idx3 q_ = split3(i, W, H);
    const T* sp = reinterpret_cast<const T*>(small.ptr) + view_off(small, q_.b, q_.y, q_.x);
    T* bp = reinterpret_cast<T*>(big.ptr);
    if (!DGRAD) {
      float acc[4][NC];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int co = 0; co < NC; ++co) acc[t][co] = sb[co];
      for (int g = 0; g < G; ++g) {
        Vec8<T> xv; xv.load(sp + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xf = xv.get(e);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int co = 0; co < NC; ++co) acc[t][co] = fmaf(xf, sw_up[(t * NC + co) * cp + g * 8 + e], acc[t][co]);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Vec8<T> o; o.zero();
#pragma unroll
        for (int co = 0; co < NC; ++co) o.set(co, relu ? fmaxf(acc[t][co], 0.f) : acc[t][co]);
        o.store(bp + view_off(big, q_.b, 2 * q_.y + (t >> 1), 2 * q_.x + (t & 1)));
        if (bn_ws != nullptr) {
#pragma unroll
          for (int co = 0; co < NC; ++co) { const float v = o.get(co); st1[co] += v; st2[co] = fmaf(v, v, st2[co]); }
        }
      }
    } else {
      float z[4][NC];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        Vec8<T> zv; zv.load(bp + view_off(big, q_.b, 2 * q_.y + (t >> 1), 2 * q_.x + (t & 1)));
#pragma unroll
        for (int co = 0; co < NC; ++co) z[t][co] = zv.get(co);
      }
      T* dp = reinterpret_cast<T*>(small.ptr) + view_off(small, q_.b, q_.y, q_.x);
      const T* mp = mask.ptr ? reinterpret_cast<const T*>(mask.ptr) + view_off(mask, q_.b, q_.y, q_.x) : nullptr;
      for (int g = 0; g < G; ++g) {
        float a8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int co = 0; co < NC; ++co) a = fmaf(z[t][co], sw_up[(t * NC + co) * cp + g * 8 + e], a);
          a8[e] = a;
        }
        if (mp) {
          Vec8<T> mv; mv.load(mp + g * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) a8[e] = mv.get(e) > 0.f ? a8[e] : 0.f;
        }
        Vec8<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, a8[e]);
        o.store(dp + g * 8);
      }
    }
  }
  if (!DGRAD && bn_ws != nullptr) {           // one row [8][2] of partial sums per workgroup (the statistics pass of seg_bn_fwd_rows)
    __shared__ float red_up[4][NC][2];
    __syncthreads();
#pragma unroll
    for (int co = 0; co < NC; ++co) {
      const float a = wave_sum(st1[co]), b2 = wave_sum(st2[co]);
      if ((threadIdx.x & 63) == 0) { red_up[threadIdx.x >> 6][co][0] = a; red_up[threadIdx.x >> 6][co][1] = b2; }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
      const int c = threadIdx.x >> 1, j = threadIdx.x & 1;
      bn_ws[((int64_t)blockIdx.x * 8 + c) * 2 + j] = c < NC ? ((red_up[0][c < NC ? c : 0][j] + red_up[1][c < NC ? c : 0][j]) + red_up[2][c < NC ? c : 0][j]) + red_up[3][c < NC ? c : 0][j] : 0.f;
    }
  }
}

static int thin_up_launch(const seg_view* small, const seg_view* big, int32_t B, int32_t H, int32_t W, const float* w_tf, const float* bias,
                          int32_t cin, int32_t cout, int32_t relu, int32_t dgrad, const seg_view* mask, float* bn_ws, int32_t dtype, void* stream);
extern "C" int seg_thin_up2x2(const seg_view* small, const seg_view* big, int32_t B, int32_t H, int32_t W, const float* w_tf, const float* bias,
                              int32_t cin, int32_t cout, int32_t relu, int32_t dgrad, const seg_view* mask, int32_t dtype, void* stream) {
  return thin_up_launch(small, big, B, H, W, w_tf, bias, cin, cout, relu, dgrad, mask, nullptr, dtype, stream);
}
/* Rows of batch-norm partial sums a seg_thin_up2x2_bn launch writes (= its workgroups, <= 1024). */
extern "C" int32_t seg_thin_up2x2_rows(int32_t B, int32_t H, int32_t W) { return (B < 1 || H < 1 || W < 1) ? 0 : grid_for((int64_t)B * H * W, 256, 1024); }
/* Forward of seg_thin_up2x2 that also leaves the statistics rows ([rows][8][2] floats) of the batch norm consuming `big` in bn_ws
 * (that batch norm's workspace; seg_bn_fwd_rows finishes it): models/deconvolution.py:166-168 deconv3_0 -> bn8. */
extern "C" int seg_thin_up2x2_bn(const seg_view* small, const seg_view* big, int32_t B, int32_t H, int32_t W, const float* w_tf, const float* bias,
                                 int32_t cin, int32_t cout, int32_t relu, float* bn_ws, int32_t dtype, void* stream) {
  if (!bn_ws) { seg_set_error("thin_up2x2_bn: no workspace"); return SEG_ERR_ARG; }
  return thin_up_launch(small, big, B, H, W, w_tf, bias, cin, cout, relu, 0, nullptr, bn_ws, dtype, stream);
}
static int thin_up_launch(const seg_view* small, const seg_view* big, int32_t B, int32_t H, int32_t W, const float* w_tf, const float* bias,
                          int32_t cin, int32_t cout, int32_t relu, int32_t dgrad, const seg_view* mask, float* bn_ws, int32_t dtype, void* stream) {
  if (!small || !small->ptr || !big || !big->ptr || !w_tf || cin < 1 || cout < 1 || cout > 8 || B <= 0 || small->c % 8 || cin > small->c || small->c > 512 ||
      !view_ok(small, H, W, small->c) || big->cs != 8 || big->coff != 0 || !view_ok(big, 2 * H, 2 * W, 8) ||
      (mask && mask->ptr && (!dgrad || !view_ok(mask, H, W, small->c)))) {
    seg_set_error("thin_up2x2: small [H,W,cin padded to 8..512], big thin [2H,2W,<= 8]; mask only with dgrad"); return SEG_ERR_ARG;
  }
  const seg_view mk = (mask && mask->ptr) ? *mask : seg_view{nullptr, 0, 0, 0, 0, 0, 0, 0};
  const int g = bn_ws ? seg_thin_up2x2_rows(B, H, W) : grid_for((int64_t)B * H * W, 256, 16384);
  const int nc = cout <= 2 ? 2 : cout <= 4 ? 4 : 8;
  const size_t lds = (size_t)(4 * nc * small->c + nc) * sizeof(float);
#define TU_ARGS dim3(g), dim3(256), lds, ST(stream), *small, *big, w_tf, dgrad ? (const float*)nullptr : bias, cin, cout, relu, mk, B, H, W, bn_ws
#define TU_NC(TT, DG) do { if (nc == 2) SEG_LAUNCH((thin_up2x2_kernel<TT, 2, DG>), TU_ARGS); else if (nc == 4) SEG_LAUNCH((thin_up2x2_kernel<TT, 4, DG>), TU_ARGS); \
    else SEG_LAUNCH((thin_up2x2_kernel<TT, 8, DG>), TU_ARGS); } while (0)
  if (dgrad) { DISPATCH(dtype, TU_NC(float, true), TU_NC(bf16_t, true)); }
  else { DISPATCH(dtype, TU_NC(float, false), TU_NC(bf16_t, false)); }
#undef TU_NC
#undef TU_ARGS
  return seg_check_launch("thin_up2x2");
}

// Two-stage form for the big maps (the one-workgroup-per-8-channels kernel above is 4 workgroups for a 32-channel tensor: 51 us
// for 67 MB at 256^2 x 16): BG_NB workgroups sum whole pixels (all channel groups, consecutive lanes = consecutive 16-byte
// pieces) into one row of partial sums each, a second launch adds the rows in a fixed order.
constexpr int BG_NB = 512;
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(seg_view dz, int B, int H, int W, int G8, float* ws) {
  __shared__ float red[8 * 256];
  const int tid = threadIdx.x, g = tid % G8, pl = tid / G8, PL = 256 / G8;
  const int64_t npix = (int64_t)B * H * W;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < npix; p += (int64_t)gridDim.x * PL) {
    const Idx3 q_ = split3(p, W, H);
    Vec8<T> v;
    v.load(reinterpret_cast<const T*>(dz.ptr) + view_off(dz, q_.b, q_.y, q_.x) + g * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += v.get(e);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[e * 256 + tid] = s[e];
  __syncthreads();
  for (int h = PL / 2; h > 0; h >>= 1) {                  // pixel lanes pl and pl + h of the same channel group: tid and tid + h * G8
    if (pl < h) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[e * 256 + tid] += red[e * 256 + tid + h * G8];
    }
    __syncthreads();
  }
  if (pl == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ws[(int64_t)blockIdx.x * (G8 * 8) + g * 8 + e] = red[e * 256 + tid];
  }
}
__global__ __launch_bounds__(256) void bias_grad_final_kernel(const float* ws, int nb, int C, int n_log, float* db) {
  __shared__ float r[32][8];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
  float s = 0.f;
  if (c < C) for (int b = sl; b < nb; b += 32) s += ws[(int64_t)b * C + c];
  r[sl][cl] = s;
  __syncthreads();
  if (sl != 0 || c >= n_log) return;
  s = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) s += r[q][cl];
  db[c] = s;
}

extern "C" int64_t seg_bias_grad_ws_bytes(int32_t C) { return (C > 0 && C % 8 == 0 && 256 % (C / 8) == 0) ? (int64_t)BG_NB * C * 4 : 0; }

extern "C" int seg_bias_grad_ws(const seg_view* dz, int32_t B, int32_t H, int32_t W, int32_t n_log, float* db, float* ws, int64_t ws_bytes,
                                int32_t dtype, void* stream) {
  if (!dz || !dz->ptr || !db || !ws || dz->c % 8 || n_log > dz->c || n_log < 1 || !view_ok(dz, H, W, dz->c) || (int64_t)B * H * W >= ((int64_t)1 << 31)) { seg_set_error("bias_grad_ws: bad args"); return SEG_ERR_ARG; }
  const int64_t need = seg_bias_grad_ws_bytes(dz->c);
  if (need == 0 || ws_bytes < need) { seg_set_error("bias_grad_ws: %d channels / workspace of %lld bytes (needs %lld)", dz->c, (long long)ws_bytes, (long long)need); return SEG_ERR_ARG; }
  const int G8 = dz->c / 8;
  DISPATCH(dtype,
           SEG_LAUNCH(bias_grad_partial_kernel<float>, dim3(BG_NB), dim3(256), 0, ST(stream), *dz, B, H, W, G8, ws),
           SEG_LAUNCH(bias_grad_partial_kernel<bf16_t>, dim3(BG_NB), dim3(256), 0, ST(stream), *dz, B, H, W, G8, ws));
  if (int rc = seg_check_launch("bias_grad_partial")) return rc;
  SEG_LAUNCH(bias_grad_final_kernel, dim3((dz->c + 7) / 8), dim3(256), 0, ST(stream), (const float*)ws, BG_NB, dz->c, n_log, db);
  return seg_check_launch("bias_grad_final");
}

extern "C" int seg_bias_grad(const seg_view* dz, int32_t B, int32_t H, int32_t W, int32_t n_log, float* db, int32_t dtype, void* stream) {
  if (!dz || !dz->ptr || !db || dz->c % 8 || n_log > dz->c || n_log < 1 || !view_ok(dz, H, W, dz->c) || (int64_t)B * H * W >= ((int64_t)1 << 31)) { seg_set_error("bias_grad: bad args"); return SEG_ERR_ARG; }
  const int g = (n_log + 7) / 8;
  DISPATCH(dtype,
           SEG_LAUNCH(bias_grad_kernel<float>, dim3(g), dim3(1024), 0, ST(stream), *dz, B, H, W, n_log, db),
           SEG_LAUNCH(bias_grad_kernel<bf16_t>, dim3(g), dim3(1024), 0, ST(stream), *dz, B, H, W, n_log, db));
  return seg_check_launch("bias_grad");
}

extern "C" int seg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                        float grad_scale, const int64_t* step_dev, void* stream) {
  if (!p || !g || !m || !v || !step_dev || n <= 0) { seg_set_error("adam: bad args"); return SEG_ERR_ARG; }
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) { seg_set_error("adam: arenas must be 16-byte aligned"); return SEG_ERR_ARG; }
  SEG_LAUNCH(adam_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, ST(stream), p, g, m, v, n, lr, b1, b2, eps, grad_scale, step_dev);
  return seg_check_launch("adam");
}

extern "C" int seg_pack_weights_dual(const float* arena, void* packed, const seg_pack_entry* fwd_table_dev, const int64_t* dgrad_dst_off_dev,
                                     int32_t n_entries, int64_t total_tiles, int32_t dtype, void* stream) {
  if (!arena || !packed || !fwd_table_dev || !dgrad_dst_off_dev || n_entries <= 0 || total_tiles <= 0) { seg_set_error("pack_dual: bad args"); return SEG_ERR_ARG; }
  const int64_t tb = (total_tiles + AP_TPB - 1) / AP_TPB;
  if (tb > 0x7fffffff) { seg_set_error("pack_dual: too many blocks"); return SEG_ERR_ARG; }
  DISPATCH(dtype,
           SEG_LAUNCH(pack_dual_kernel<float>, dim3((unsigned)tb), dim3(256), 0, ST(stream), arena, reinterpret_cast<float*>(packed), fwd_table_dev, dgrad_dst_off_dev, n_entries, total_tiles),
           SEG_LAUNCH(pack_dual_kernel<bf16_t>, dim3((unsigned)tb), dim3(256), 0, ST(stream), arena, reinterpret_cast<bf16_t*>(packed), fwd_table_dev, dgrad_dst_off_dev, n_entries, total_tiles));
  return seg_check_launch("pack_weights_dual");
}

extern "C" int seg_step_begin(int64_t* step2_dev, float* loss_sum, void* stream) {
  if (!step2_dev) { seg_set_error("step_begin: null"); return SEG_ERR_ARG; }
  SEG_LAUNCH(step_begin_kernel, dim3(1), dim3(64), 0, ST(stream), step2_dev, loss_sum);
  return seg_check_launch("step_begin");
}

extern "C" int seg_step_increment(int64_t* step_dev, void* stream) {
  if (!step_dev) { seg_set_error("step_increment: null"); return SEG_ERR_ARG; }
  SEG_LAUNCH(step_inc_kernel, dim3(1), dim3(64), 0, ST(stream), step_dev);
  return seg_check_launch("step_increment");
}

extern "C" int seg_pack_weights(const float* arena, void* packed, const seg_pack_entry* table_dev, int32_t n_entries,
                                int64_t total_blocks, int32_t dtype, void* stream) {
  if (!arena || !packed || !table_dev || n_entries <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffff) { seg_set_error("pack: bad args"); return SEG_ERR_ARG; }
  DISPATCH(dtype,
           SEG_LAUNCH(pack_kernel<float>, dim3((unsigned)((total_blocks + PACK_TPB - 1) / PACK_TPB)), dim3(256), 0, ST(stream), arena, reinterpret_cast<float*>(packed), table_dev, n_entries, total_blocks),
           SEG_LAUNCH(pack_kernel<bf16_t>, dim3((unsigned)((total_blocks + PACK_TPB - 1) / PACK_TPB)), dim3(256), 0, ST(stream), arena, reinterpret_cast<bf16_t*>(packed), table_dev, n_entries, total_blocks));
  return seg_check_launch("pack_weights");
}

extern "C" int seg_bilinear_up_fwd(const seg_view* src, int32_t Hs, int32_t Ws, int32_t factor, const float* filt, const seg_view* add,
                                   const seg_view* dst, int32_t Hd, int32_t Wd, int32_t cy, int32_t cx, int32_t B, int32_t C,
                                   int32_t dst_f32, int32_t dtype, void* stream) {
  if (!view_ok(src, Hs, Ws, C) || !view_ok(dst, Hd, Wd, C) || !filt || factor < 1 || C % 8) { seg_set_error("bilinear_fwd: bad args"); return SEG_ERR_ARG; }
  if (add && add->ptr && !view_ok(add, Hd, Wd, C)) { seg_set_error("bilinear_fwd: bad add view"); return SEG_ERR_ARG; }
  const seg_view adv = (add && add->ptr) ? *add : null_view();
  const int64_t n = (int64_t)B * Hd * Wd * (C / 8);
  DISPATCH(dtype,
           SEG_LAUNCH(bilinear_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *src, Hs, Ws, factor, filt, adv, *dst, Hd, Wd, cy, cx, B, C / 8, dst_f32),
           SEG_LAUNCH(bilinear_fwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, ST(stream), *src, Hs, Ws, factor, filt, adv, *dst, Hd, Wd, cy, cx, B, C / 8, dst_f32));
  return seg_check_launch("bilinear_fwd");
}

extern "C" int seg_bilinear_up_bwd(const seg_view* ddst, int32_t Hd, int32_t Wd, int32_t cy, int32_t cx, int32_t factor, const float* filt,
                                   const seg_view* dsrc, int32_t Hs, int32_t Ws, int32_t B, int32_t C, int32_t ddst_f32,
                                   const seg_view* mask_act, const seg_view* dz_masked, int32_t dtype, void* stream) {
  if (!view_ok(ddst, Hd, Wd, C) || !view_ok(dsrc, Hs, Ws, C) || !filt || factor < 1 || C % 8) { seg_set_error("bilinear_bwd: bad args"); return SEG_ERR_ARG; }
  if ((mask_act != nullptr) != (dz_masked != nullptr) || (mask_act && (!view_ok(mask_act, Hs, Ws, C) || !view_ok(dz_masked, Hs, Ws, C)))) { seg_set_error("bilinear_bwd: bad masked output"); return SEG_ERR_ARG; }
  const seg_view mkv = mask_act ? *mask_act : null_view(), dzv = dz_masked ? *dz_masked : null_view();
  const int64_t n = (int64_t)B * Hs * Ws * (C / 8);
#define BIL_BWD(TT, LP) SEG_LAUNCH((bilinear_bwd_kernel<TT, LP>), dim3(grid_for(n * LP, 256, 16384)), dim3(256), 0, ST(stream), mkv, dzv, *ddst, Hd, Wd, cy, cx, factor, filt, *dsrc, Hs, Ws, B, C / 8, ddst_f32)
#define BIL_BWD_F(TT) do { if (factor >= 8) BIL_BWD(TT, 16); else if (factor >= 2) BIL_BWD(TT, 4); else BIL_BWD(TT, 1); } while (0)
  DISPATCH(dtype, BIL_BWD_F(float), BIL_BWD_F(bf16_t));
#undef BIL_BWD_F
#undef BIL_BWD
  return seg_check_launch("bilinear_bwd");
}

extern "C" int seg_bilinear_xent(const seg_view* src, int32_t Hs, int32_t Ws, int32_t factor, const float* filt, int32_t cy, int32_t cx,
                                 const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0, int32_t B, int32_t H, int32_t W,
                                 int32_t n_classes, float inv_n, float grad_scale, float* loss_sum, const seg_view* dlogits,
                                 const seg_view* logits_out, int32_t dtype, void* stream) {
  if (!src || !labels || !loss_sum || !filt || factor < 1 || !view_ok(dlogits, H, W, dlogits ? dlogits->c : 0) || !view_ok(src, Hs, Ws, src->c)) { seg_set_error("bilinear_xent: bad args"); return SEG_ERR_ARG; }
  if (n_classes < 1 || n_classes > 32 || n_classes > dlogits->c || dlogits->c % 8 || dlogits->c > 32 || src->c < n_classes) { seg_set_error("bilinear_xent: n_classes %d unsupported (1..32)", n_classes); return SEG_ERR_UNSUPPORTED; }
  if (ly0 < 0 || lx0 < 0 || ly0 + H > LH || lx0 + W > LW) { seg_set_error("bilinear_xent: label window out of range"); return SEG_ERR_ARG; }
  const int ncp = n_classes <= 4 ? 4 : n_classes <= 8 ? 8 : n_classes <= 16 ? 16 : 32;
  if (src->c < ncp && src->c % 8) { seg_set_error("bilinear_xent: score map narrower than %d channels", ncp); return SEG_ERR_ARG; }
  seg_view lov = null_view();
  if (logits_out && logits_out->ptr) {
    if (!view_ok(logits_out, H, W, logits_out->c) || logits_out->c < ncp || (reinterpret_cast<uintptr_t>(logits_out->ptr) & 15) || logits_out->cs % 4 || logits_out->coff % 4) { seg_set_error("bilinear_xent: bad logits_out view"); return SEG_ERR_ARG; }
    lov = *logits_out;
  }
  // lanes per pixel: measured at 512^2 x 21 classes (32 padded): 1 lane 92 us, 2 lanes 81, 4 lanes 94, 8 lanes 129 -- one lane per
  // pixel repeats no index arithmetic but scatters its stores, eight lanes repeat all of it eight times
  const int lpx = ncp >= 16 ? 2 : 1;
  const int gx = (W * lpx + 255) / 256;
  int gy = 4096 / gx; if (gy > B * H) gy = B * H; if (gy < 1) gy = 1;      // ~4096 workgroups (one atomic each), rows strided
#define BX_ARGS dim3(gx, gy), dim3(256), 0, ST(stream), *src, Hs, Ws, factor, filt, cy, cx, labels, LH, LW, ly0, lx0, B, H, W, n_classes, inv_n, grad_scale, loss_sum, *dlogits, lov
#define BX_F(TT, LL, CC) do { if (factor == 8) SEG_LAUNCH((bilinear_xent_kernel<TT, LL, CC, 8>), BX_ARGS); else if (factor == 16) SEG_LAUNCH((bilinear_xent_kernel<TT, LL, CC, 16>), BX_ARGS); \
    else if (factor == 32) SEG_LAUNCH((bilinear_xent_kernel<TT, LL, CC, 32>), BX_ARGS); else SEG_LAUNCH((bilinear_xent_kernel<TT, LL, CC, 0>), BX_ARGS); } while (0)
#define BX_L(TT) do { if (ncp == 4) BX_F(TT, 1, 4); else if (ncp == 8) BX_F(TT, 1, 8); else if (ncp == 16) BX_F(TT, 2, 8); else BX_F(TT, 2, 16); } while (0)
  DISPATCH(dtype, BX_L(float), BX_L(bf16_t));
#undef BX_L
#undef BX_F
#undef BX_ARGS
  return seg_check_launch("bilinear_xent");
}

extern "C" int64_t seg_bilinear_up_bwd_ws_bytes(int32_t B, int32_t Hd, int32_t Ws, int32_t C) { return (int64_t)B * Hd * Ws * C * 4; }

extern "C" int seg_bilinear_up_bwd_sep(const seg_view* ddst, int32_t Hd, int32_t Wd, int32_t cy, int32_t cx, int32_t factor, const float* filt,
                                       const seg_view* dsrc, int32_t Hs, int32_t Ws, int32_t B, int32_t C, int32_t ddst_f32, float* ws,
                                       int64_t ws_bytes, const seg_view* mask_act, const seg_view* dz_masked, int32_t dtype, void* stream) {
  if ((mask_act != nullptr) != (dz_masked != nullptr) || (mask_act && (!view_ok(mask_act, Hs, Ws, C) || !view_ok(dz_masked, Hs, Ws, C)))) { seg_set_error("bilinear_bwd_sep: bad masked output"); return SEG_ERR_ARG; }
  const seg_view mkv = mask_act ? *mask_act : null_view(), dzv = dz_masked ? *dz_masked : null_view();
  if (!view_ok(ddst, Hd, Wd, C) || !view_ok(dsrc, Hs, Ws, C) || !filt || factor < 1 || C % 8 || !ws || ws_bytes < seg_bilinear_up_bwd_ws_bytes(B, Hd, Ws, C)) { seg_set_error("bilinear_bwd_sep: bad args / workspace"); return SEG_ERR_ARG; }
  const int64_t n1 = (int64_t)B * Hd * Ws * (C / 8), n2 = (int64_t)B * Hs * Ws * (C / 8);
  DISPATCH(dtype,
           SEG_LAUNCH(bilinear_bwd_h_kernel<float>, dim3(grid_for(n1, 256, 16384)), dim3(256), 0, ST(stream), *ddst, Hd, Wd, cx, factor, filt, ws, Ws, B, C / 8, ddst_f32),
           SEG_LAUNCH(bilinear_bwd_h_kernel<bf16_t>, dim3(grid_for(n1, 256, 16384)), dim3(256), 0, ST(stream), *ddst, Hd, Wd, cx, factor, filt, ws, Ws, B, C / 8, ddst_f32));
  if (int rc = seg_check_launch("bilinear_bwd_h")) return rc;
  DISPATCH(dtype,
           SEG_LAUNCH(bilinear_bwd_v_kernel<float>, dim3(grid_for(n2, 256, 16384)), dim3(256), 0, ST(stream), mkv, dzv, (const float*)ws, Hd, cy, factor, filt, *dsrc, Hs, Ws, B, C / 8),
           SEG_LAUNCH(bilinear_bwd_v_kernel<bf16_t>, dim3(grid_for(n2, 256, 16384)), dim3(256), 0, ST(stream), mkv, dzv, (const float*)ws, Hd, cy, factor, filt, *dsrc, Hs, Ws, B, C / 8));
  return seg_check_launch("bilinear_bwd_v");
}
