// conv_wgrad.hip -- filter gradient (Conv2DBackpropFilter) as an MFMA GEMM whose reduction
// dimension is the pixel index:  dW[tap][ci][co] = sum_{b,y,x} X[b, y*S+u-pt, x*S+v-pl, ci] * dZ[b,y,x,co].
// Serves the 3x3 / 1x1 convs and, with X := dZ_big (stride 2, 2x2 taps) and dZ := x_small, the
// 2x2/s2 transposed conv (TF autodiff of /root/reference/models/unet.py:111-166, models/fcn.py:110-128).
//
// Both MFMA operands need K (= pixel) contiguous per lane while NHWC keeps channels contiguous, so the
// bf16 path reads its fragments with ds_read_b64_tr_b16 (4 pixels x 16 channels, delivered
// column-major); the f32 path reads single elements.  A workgroup owns one 32-channel chunk of X and
// BN channels of dZ for ALL taps, keeps the KH*KW*32*BN partial sums in registers while it walks its
// share of the pixel tiles, and flushes them ONCE, with plain coalesced stores, into its own slab of a
// workspace; a second small kernel sums the `ksplit` slabs in a fixed order into the TF-layout gradient
// (deterministic, no atomics: 768 workgroups hammering one 36 KB filter with f32 atomics measured 10-30x
// slower than the MFMA work).  The bias gradient rides along as one extra MFMA against an all-ones fragment.
#include "common.h"

namespace {

struct WgK {
  seg_wgrad_desc d;
  int tiles_x, tiles_y, ntiles, ksplit;
  int nchunks0, nchunks, nblk;   // K chunks of src0 / total; BN blocks
  int k_pad, n_pad;              // slab = [taps][k_pad][n_pad] floats followed by bias[max(k_pad,n_pad)]
  int64_t slab;                  // floats per split
};

template <typename T> SEG_DEV Frag<T> ones_frag();
template <> SEG_DEV Frag<bf16_t> ones_frag<bf16_t>() {
  Frag<bf16_t> f;
  for (int i = 0; i < 8; ++i) f.v[i] = (bf16_t)1.0f;
  return f;
}
template <> SEG_DEV Frag<float> ones_frag<float>() {
  Frag<float> f; f.lo = f32x4{1, 1, 1, 1}; f.hi = f32x4{1, 1, 1, 1}; return f;
}

template <typename T> struct TrRead;
template <> struct TrRead<bf16_t> {
  // lane (G = lane>>4, qr = (lane&15)>>2, p = lane&3) passes the address of (pixel 8G+4h+qr, ch0+4p);
  // receives channel ch0+(lane&15), pixels 8G+4h..+3.
  static SEG_DEV Frag<bf16_t> read(const char* lo, const char* hi) {
    typedef bf16x4 __attribute__((address_space(3))) * lp;
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(lo));
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(hi));
    Frag<bf16_t> f;
    f.v = bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
  }
};

template <typename T, int TH, int TW, int KH, int KW, int S, int WCI, int WCO, int FCI, int FCO>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgK P) {
  constexpr int BM = TH * TW;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, NPIX = PH * PW;
  constexpr int NT = KH * KW;
  constexpr int ES = sizeof(T);
  constexpr int BN = 16 * FCO * WCO;
  constexpr int RSP = 32 * ES + 16;            // patch row stride (bytes)
  constexpr int RSZ = BN * ES + 16;            // dZ tile row stride
  constexpr int PPIECES = 32 * ES / 16, ZPIECES = BN * ES / 16, EPP = 16 / ES;
  constexpr int PATCH_BYTES = ((NPIX * RSP + 15) / 16) * 16;
  constexpr int NPP = (NPIX * PPIECES + 255) / 256;
  constexpr int NZP = (BM * ZPIECES + 255) / 256;
  constexpr int KS = BM / 32;
  static_assert(WCI * WCO == 4 && 16 * FCI * WCI == 32, "wave layout");
  static_assert(BM % 32 == 0, "tile pixels");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;
  char* sZ = smem + PATCH_BYTES;

  const seg_wgrad_desc& d = P.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wci = wave / WCO, wco = wave % WCO;
  const int lr = lane & 15, G = lane >> 4;

  const int chunk = blockIdx.x / P.nblk, nb = blockIdx.x % P.nblk;
  const bool first = chunk < P.nchunks0;
  const seg_view& sv = first ? d.src0 : d.src1;
  const int cbase = first ? chunk * 32 : (chunk - P.nchunks0) * 32;   // padded channel base inside its source
  const T* srcp = reinterpret_cast<const T*>(sv.ptr);
  const T* dzp = reinterpret_cast<const T*>(d.dz.ptr);
  const int n0 = nb * BN;

  f32x4 acc[NT][FCI][FCO];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < FCI; ++a)
#pragma unroll
      for (int c = 0; c < FCO; ++c) acc[t][a][c] = f32x4{0, 0, 0, 0};

  // bias gradient: mode 1 = column sums of dz (conv; only chunk-0 workgroups), mode 2 = sums of src over
  // pixels and taps (transposed conv, where src is the big dZ map; only nb-0 workgroups)
  const bool bias1 = d.bias_mode == 1 && chunk == 0 && wci == 0;
  const bool bias2 = d.bias_mode == 2 && nb == 0 && wco == 0;
  f32x4 accb1[FCO], accb2[FCI];
#pragma unroll
  for (int c = 0; c < FCO; ++c) accb1[c] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < FCI; ++a) accb2[a] = f32x4{0, 0, 0, 0};
  const Frag<T> ones = ones_frag<T>();

  u32x4 rp[NPP], rz[NZP];
  auto prefetch = [&](int tile) {
    int t = tile;
    const int tx = t % P.tiles_x; t /= P.tiles_x;
    const int ty = t % P.tiles_y; const int b = t / P.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const T* sb = srcp + (int64_t)b * sv.H * sv.W * sv.cs + sv.coff + cbase;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int idx = tid + i * 256;
      rp[i] = u32x4{0, 0, 0, 0};
      if (idx < NPIX * PPIECES) {
        const int q = idx / PPIECES, h = idx % PPIECES;
        const int iy = oy0 * S - d.pad_t + q / PW, ix = ox0 * S - d.pad_l + q % PW;
        if (iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi)
          rp[i] = *reinterpret_cast<const u32x4*>(sb + ((int64_t)(iy + sv.oy) * sv.W + ix + sv.ox) * sv.cs + h * EPP);
      }
    }
    const T* zb = dzp + (int64_t)b * d.dz.H * d.dz.W * d.dz.cs + d.dz.coff + n0;
#pragma unroll
    for (int i = 0; i < NZP; ++i) {
      const int idx = tid + i * 256;
      rz[i] = u32x4{0, 0, 0, 0};
      if (BM * ZPIECES % 256 == 0 || idx < BM * ZPIECES) {
        const int m = idx / ZPIECES, h = idx % ZPIECES;
        const int oy = oy0 + m / TW, ox = ox0 + m % TW;
        if (oy < d.Ho && ox < d.Wo)
          rz[i] = *reinterpret_cast<const u32x4*>(zb + ((int64_t)(oy + d.dz.oy) * d.dz.W + ox + d.dz.ox) * d.dz.cs + h * EPP);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int idx = tid + i * 256;
      if (idx < NPIX * PPIECES) *reinterpret_cast<u32x4*>(sP + (idx / PPIECES) * RSP + (idx % PPIECES) * 16) = rp[i];
    }
#pragma unroll
    for (int i = 0; i < NZP; ++i) {
      const int idx = tid + i * 256;
      if (BM * ZPIECES % 256 == 0 || idx < BM * ZPIECES)
        *reinterpret_cast<u32x4*>(sZ + (idx / ZPIECES) * RSZ + (idx % ZPIECES) * 16) = rz[i];
    }
  };

  // ---- per-lane fragment addresses ----
  // bf16: transposed reads; lane supplies (pixel 8G+4h+qr, channels +4p).  f32: element reads of pixel 8G+s.
  constexpr bool BF = (sizeof(T) == 2);
  constexpr int NA = BF ? 2 : 8;
  int pa[KS][NA], za[NA];
  {
    const int qr = lr >> 2, p = lr & 3;
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      const int mloc = BF ? (8 * G + 4 * e + qr) : (8 * G + e);       // pixel within a 32-pixel K step
      za[e] = mloc * RSZ + (BF ? 8 * p : 4 * lr);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int m = ks * 32 + mloc;
        pa[ks][e] = (((m / TW) * S) * PW + (m % TW) * S) * RSP + (BF ? 8 * p : 4 * lr);
      }
    }
  }
  const int a_ch = (wci * FCI) * 16 * ES;          // byte offset of this wave's first X channel fragment
  const int z_ch = (wco * FCO) * 16 * ES;

  auto read_frag = [&](const char* base, const int* addr, int off) -> Frag<T> {
    if constexpr (BF) {
      return TrRead<bf16_t>::read(base + addr[0] + off, base + addr[1] + off);
    } else {
      Frag<float> f;
      f.lo = f32x4{*reinterpret_cast<const float*>(base + addr[0] + off), *reinterpret_cast<const float*>(base + addr[1] + off),
                   *reinterpret_cast<const float*>(base + addr[2] + off), *reinterpret_cast<const float*>(base + addr[3] + off)};
      f.hi = f32x4{*reinterpret_cast<const float*>(base + addr[4] + off), *reinterpret_cast<const float*>(base + addr[5] + off),
                   *reinterpret_cast<const float*>(base + addr[6] + off), *reinterpret_cast<const float*>(base + addr[7] + off)};
      return f;
    }
  };

  int tile = blockIdx.y;
  if (tile < P.ntiles) prefetch(tile);
  for (; tile < P.ntiles; tile += P.ksplit) {
    __syncthreads();
    commit();
    __syncthreads();
    if (tile + P.ksplit < P.ntiles) prefetch(tile + P.ksplit);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Frag<T> fz[FCO];
#pragma unroll
      for (int c = 0; c < FCO; ++c) fz[c] = read_frag(sZ, za, ks * 32 * RSZ + z_ch + c * 16 * ES);
      if (bias1) {
#pragma unroll
        for (int c = 0; c < FCO; ++c) mma32(accb1[c], ones, fz[c]);
      }
#pragma unroll
      for (int u = 0; u < KH; ++u)
#pragma unroll
        for (int v = 0; v < KW; ++v)
#pragma unroll
          for (int a = 0; a < FCI; ++a) {
            Frag<T> fx = read_frag(sP, pa[ks], (u * PW + v) * RSP + a_ch + a * 16 * ES);
#pragma unroll
            for (int c = 0; c < FCO; ++c) mma32(acc[u * KW + v][a][c], fx, fz[c]);
            if (bias2) mma32(accb2[a], fx, ones);
          }
    }
  }

  // ---- flush into this split's slab: D[row = ci][col = co]; lane: co = lr, ci = 4G + r ----
  float* slab = d.ws + (int64_t)blockIdx.y * P.slab;
  const int kbase = chunk * 32;                                    // padded concat channel base
#pragma unroll
  for (int a = 0; a < FCI; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kbase + (wci * FCI + a) * 16 + 4 * G + r;
#pragma unroll
      for (int c = 0; c < FCO; ++c) {
        const int co = n0 + (wco * FCO + c) * 16 + lr;
#pragma unroll
        for (int t = 0; t < NT; ++t) slab[((int64_t)t * P.k_pad + k) * P.n_pad + co] = acc[t][a][c][r];
      }
    }
  float* bslab = slab + (int64_t)NT * P.k_pad * P.n_pad;
  if (bias1 && G == 0) {
#pragma unroll
    for (int c = 0; c < FCO; ++c) bslab[n0 + (wco * FCO + c) * 16 + lr] = accb1[c][0];      // row 0 of D
  }
  if (bias2 && lr == 0) {
#pragma unroll
    for (int a = 0; a < FCI; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) bslab[kbase + (wci * FCI + a) * 16 + 4 * G + r] = accb2[a][r];   // column 0 of D
  }
}

// Sums the ksplit slabs (fixed order => bitwise reproducible) and scatters to the logical TF layout.
// SP = 1: one thread per output (few splits).  SP = 16: a 256-thread block owns 16 consecutive outputs,
// thread (jj = t/16, oo = t%16) sums splits s = jj, jj+16, ... (4 independent chains), then the 16 partial
// sums meet in LDS and are added in jj order.
SEG_DEV void reduce_addr(int64_t i, int64_t nw, int taps, int k_pad, int n_pad, int seg0_c, int seg0_cp, int k_log, int n_log,
                         float* dw, float* db, int64_t& src, float*& dst) {
  if (i < nw) {
    const int n = i % n_log; int64_t t = i / n_log;
    const int k = t % k_log; const int tap = t / k_log;
    const int kp = k < seg0_c ? k : seg0_cp + (k - seg0_c);
    src = ((int64_t)tap * k_pad + kp) * n_pad + n;
    dst = dw + i;
  } else {
    src = (int64_t)taps * k_pad * n_pad + (i - nw);
    dst = db + (i - nw);
  }
}

template <int SP>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, int ksplit, int64_t slab, int taps, int k_pad, int n_pad,
                                                           int seg0_c, int seg0_cp, int seg1_c, int n_log, float* dw, int bias_mode,
                                                           int bias_n, float* db) {
  const int k_log = seg0_c + seg1_c;
  const int64_t nw = (int64_t)taps * k_log * n_log;
  const int64_t total = nw + (bias_mode ? bias_n : 0);
  if (SP == 1) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      int64_t src; float* dst;
      reduce_addr(i, nw, taps, k_pad, n_pad, seg0_c, seg0_cp, k_log, n_log, dw, db, src, dst);
      float a = 0.f;
      for (int s = 0; s < ksplit; ++s) a += ws[(int64_t)s * slab + src];
      *dst = a;
    }
  } else {
    __shared__ float red[16][17];
    const int oo = threadIdx.x & 15, jj = threadIdx.x >> 4;
    for (int64_t i0 = (int64_t)blockIdx.x * 16; i0 < total; i0 += (int64_t)gridDim.x * 16) {
      const int64_t i = i0 + oo;
      int64_t src = 0; float* dst = nullptr;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      if (i < total) {
        reduce_addr(i, nw, taps, k_pad, n_pad, seg0_c, seg0_cp, k_log, n_log, dw, db, src, dst);
        int s = jj;
        for (; s + 48 < ksplit; s += 64) {
          a0 += ws[(int64_t)s * slab + src]; a1 += ws[(int64_t)(s + 16) * slab + src];
          a2 += ws[(int64_t)(s + 32) * slab + src]; a3 += ws[(int64_t)(s + 48) * slab + src];
        }
        for (; s < ksplit; s += 16) a0 += ws[(int64_t)s * slab + src];
      }
      __syncthreads();
      red[jj][oo] = (a0 + a1) + (a2 + a3);
      __syncthreads();
      if (jj == 0 && i < total) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += red[q][oo];
        *dst = a;
      }
    }
  }
}

thread_local char* g_wname_out = nullptr;      // name-query mode: report the kernel instance, launch nothing
thread_local int g_wname_cap = 0;
thread_local int32_t* g_plan_ks = nullptr;     // planning mode: report ksplit / workspace bytes, launch nothing
thread_local int64_t* g_plan_bytes = nullptr;

template <typename T, int TH, int TW, int KH, int KW, int S, int WCI, int WCO, int FCI, int FCO>
int launch_cfg(const WgK& P0, hipStream_t st) {
  constexpr int BN = 16 * FCO * WCO, ES = sizeof(T);
  if (g_wname_out) {
    snprintf(g_wname_out, g_wname_cap, "conv_wgrad_kernel<%s,%d,%d,%d,%d,%d,%d,%d,%d,%d>", ES == 2 ? "bf16" : "f32", TH, TW, KH, KW, S, WCI, WCO, FCI, FCO);
    return SEG_OK;
  }
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
  constexpr int PATCH_BYTES = ((PH * PW * (32 * ES + 16) + 15) / 16) * 16;
  constexpr int LDS = PATCH_BYTES + TH * TW * (BN * ES + 16);
  WgK P = P0;
  P.tiles_x = cdiv(P.d.Wo, TW); P.tiles_y = cdiv(P.d.Ho, TH);
  P.ntiles = P.d.B * P.tiles_x * P.tiles_y;
  P.nblk = cdiv(P.d.dz.c, BN);
  if (P.d.dz.c % BN) { seg_set_error("wgrad: dz channels %d not a multiple of BN %d", P.d.dz.c, BN); return SEG_ERR_ARG; }
  const int base = P.nchunks * P.nblk;
  // K split: enough workgroups to fill the chip once (every split costs one slab write + read of this
  // workgroup tile: ~9*32*BN*4 bytes), never more splits than pixel tiles.
  int ks = P.d.ksplit > 0 ? P.d.ksplit : cdiv(256, base);
  if (ks > P.ntiles) ks = P.ntiles;
  if (ks < 1) ks = 1;
  P.ksplit = ks;
  P.k_pad = P.nchunks * 32; P.n_pad = P.d.dz.c;
  const int64_t bias_len = P.k_pad > P.n_pad ? P.k_pad : P.n_pad;
  P.slab = (int64_t)KH * KW * P.k_pad * P.n_pad + bias_len;
  if (g_plan_ks) { *g_plan_ks = ks; *g_plan_bytes = P.slab * ks * 4; return SEG_OK; }
  if (!P.d.ws || P.d.ws_bytes < P.slab * ks * 4) { seg_set_error("wgrad: workspace too small (%lld < %lld bytes)", (long long)P.d.ws_bytes, (long long)(P.slab * ks * 4)); return SEG_ERR_ARG; }
  auto kern = conv_wgrad_kernel<T, TH, TW, KH, KW, S, WCI, WCO, FCI, FCO>;
  static bool attr_done = false;
  if (!attr_done && LDS > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      seg_set_error("wgrad: cannot raise dynamic LDS to %d", LDS); return SEG_ERR_LAUNCH;
    }
    attr_done = true;
  }
  SEG_LAUNCH(kern, dim3(base, ks), dim3(256), LDS, st, P);
  int rc = seg_check_launch("conv_wgrad");
  if (rc) return rc;
  const int64_t total = (int64_t)KH * KW * (P.d.src0_clog + P.d.src1_clog) * P.d.n_log + (P.d.bias_mode ? P.d.bias_n : 0);
  if (ks <= 4) {
    int rg = (int)((total + 255) / 256); if (rg > 4096) rg = 4096;
    SEG_LAUNCH(wgrad_reduce_kernel<1>, dim3(rg), dim3(256), 0, st, P.d.ws, ks, P.slab, KH * KW, P.k_pad, P.n_pad, P.d.src0_clog, P.d.src0.c,
               P.d.src1_clog, P.d.n_log, P.d.dw, P.d.bias_mode, P.d.bias_n, P.d.db);
  } else {
    int rg = (int)((total + 15) / 16); if (rg > 8192) rg = 8192;
    SEG_LAUNCH(wgrad_reduce_kernel<16>, dim3(rg), dim3(256), 0, st, P.d.ws, ks, P.slab, KH * KW, P.k_pad, P.n_pad, P.d.src0_clog, P.d.src0.c,
               P.d.src1_clog, P.d.n_log, P.d.dw, P.d.bias_mode, P.d.bias_n, P.d.db);
  }
  return seg_check_launch("wgrad_reduce");
}

template <typename T, int KH, int KW, int S>
int launch_k(const WgK& P, hipStream_t st) {
  const seg_wgrad_desc& d = P.d;
  int cfg = d.cfg;
  const bool small = (long)cdiv(d.Ho, 8) * 8 * cdiv(d.Wo, 8) * 8 < (long)cdiv(d.Ho, 8) * 8 * cdiv(d.Wo, 16) * 16;
  if (cfg == 0) {
    const int bn = (d.dz.c % 128 == 0) ? 128 : (d.dz.c % 64 == 0 ? 64 : 32);
    cfg = (bn == 128 ? 1 : bn == 64 ? 2 : 3) + (small ? 3 : 0);
  }
  switch (cfg) {
    case 1: return launch_cfg<T, 8, 16, KH, KW, S, 1, 4, 2, 2>(P, st);   // 128 px, 32 ci x 128 co
    case 2: return launch_cfg<T, 8, 16, KH, KW, S, 1, 4, 2, 1>(P, st);   // 128 px, 32 ci x 64 co
    case 3: return launch_cfg<T, 8, 16, KH, KW, S, 2, 2, 1, 1>(P, st);   // 128 px, 32 ci x 32 co
    case 4: return launch_cfg<T, 8, 8, KH, KW, S, 1, 4, 2, 2>(P, st);    //  64 px
    case 5: return launch_cfg<T, 8, 8, KH, KW, S, 1, 4, 2, 1>(P, st);
    case 6: return launch_cfg<T, 8, 8, KH, KW, S, 2, 2, 1, 1>(P, st);
    default: seg_set_error("wgrad: unknown cfg %d", cfg); return SEG_ERR_ARG;
  }
}

template <typename T>
int launch_t(const WgK& P, hipStream_t st) {
  const seg_wgrad_desc& d = P.d;
  if (d.KH == 3 && d.KW == 3 && d.stride == 1) return launch_k<T, 3, 3, 1>(P, st);
  if (d.KH == 1 && d.KW == 1 && d.stride == 1) return launch_k<T, 1, 1, 1>(P, st);
  if (d.KH == 2 && d.KW == 2 && d.stride == 2) return launch_k<T, 2, 2, 2>(P, st);
  seg_set_error("wgrad: unsupported kernel %dx%d stride %d", d.KH, d.KW, d.stride);
  return SEG_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int seg_conv2d_wgrad(const seg_wgrad_desc* dp, void* stream);
extern "C" int seg_conv2d_wgrad_kernel_name(const seg_wgrad_desc* dp, char* buf, int32_t cap) {
  if (!buf || cap <= 0) { seg_set_error("kernel_name: bad buffer"); return SEG_ERR_ARG; }
  buf[0] = 0;
  g_wname_out = buf; g_wname_cap = cap;
  const int rc = seg_conv2d_wgrad(dp, nullptr);
  g_wname_out = nullptr;
  return rc;
}

extern "C" int seg_conv2d_wgrad_plan(const seg_wgrad_desc* dp, int32_t* ksplit, int64_t* ws_bytes) {
  if (!ksplit || !ws_bytes) { seg_set_error("wgrad_plan: null output"); return SEG_ERR_ARG; }
  g_plan_ks = ksplit; g_plan_bytes = ws_bytes;
  const int rc = seg_conv2d_wgrad(dp, nullptr);
  g_plan_ks = nullptr; g_plan_bytes = nullptr;
  return rc;
}

extern "C" int seg_conv2d_wgrad(const seg_wgrad_desc* dp, void* stream) {
  if (!dp) { seg_set_error("wgrad: null descriptor"); return SEG_ERR_ARG; }
  const seg_wgrad_desc& d = *dp;
  if (!d.src0.ptr || !d.dz.ptr || !d.dw) { seg_set_error("wgrad: null pointer"); return SEG_ERR_ARG; }
  if (d.bias_mode < 0 || d.bias_mode > 2 || (d.bias_mode && (!d.db || d.bias_n <= 0))) { seg_set_error("wgrad: bad bias request"); return SEG_ERR_ARG; }
  if ((d.bias_mode == 1 && d.bias_n > d.dz.c) || (d.bias_mode == 2 && d.bias_n > d.src0.c)) { seg_set_error("wgrad: bias_n exceeds channels"); return SEG_ERR_ARG; }
  if (d.src0.c <= 0 || d.src0.c % 32 || (d.src1.ptr && (d.src1.c <= 0 || d.src1.c % 32)) || d.dz.c <= 0 || d.dz.c % 32) {
    seg_set_error("wgrad: channel counts must be positive multiples of 32"); return SEG_ERR_ARG;
  }
  if (d.src0_clog > d.src0.c || (d.src1.ptr && d.src1_clog > d.src1.c) || d.n_log > d.dz.c) {
    seg_set_error("wgrad: logical channels exceed padded"); return SEG_ERR_ARG;
  }
  if (d.B <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.Hi <= 0 || d.Wi <= 0) { seg_set_error("wgrad: empty extent"); return SEG_ERR_ARG; }
  if (d.src0.oy + d.Hi > d.src0.H || d.src0.ox + d.Wi > d.src0.W || d.src0.coff + d.src0.c > d.src0.cs ||
      (d.src1.ptr && (d.src1.oy + d.Hi > d.src1.H || d.src1.ox + d.Wi > d.src1.W || d.src1.coff + d.src1.c > d.src1.cs)) ||
      d.dz.oy + d.Ho > d.dz.H || d.dz.ox + d.Wo > d.dz.W || d.dz.coff + d.dz.c > d.dz.cs) {
    seg_set_error("wgrad: window exceeds its buffer"); return SEG_ERR_ARG;
  }
  WgK P;
  P.d = d;
  if (!d.src1.ptr) { P.d.src1 = d.src0; P.d.src1.c = 0; P.d.src1_clog = 0; }
  P.nchunks0 = d.src0.c / 32;
  P.nchunks = P.nchunks0 + (d.src1.ptr ? d.src1.c / 32 : 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d.dtype == SEG_F32) return launch_t<float>(P, st);
  if (d.dtype == SEG_BF16) return launch_t<bf16_t>(P, st);
  seg_set_error("wgrad: bad dtype %d", d.dtype);
  return SEG_ERR_ARG;
}
