// wgrad_sweep.hip -- filter gradient (Conv2DBackpropFilter + BiasAddGrad) of the 3x3 / stride-1 convolutions, bf16, as a
// wave-specialised MFMA walk:  dW[tap][ci][co] = sum_{b,y,x} X[b, y+u-pt, x+v-pl, ci] * dZ[b,y,x,co]
// (TF autodiff of /root/reference/models/unet.py:111-166, models/fcn.py:110-128).  Round 3; replaces the register-staged tile walk
// of conv_wgrad.hip for these layers (that file keeps f32 / 1x1 / 2x2-stride-2 / the first layer).
//
// What is different, and why (profiles/r02_sq_counters.json: 44 % of a lone workgroup's cycles in the
// MFMAs, 31 % issuing staging / commit / address instructions, 18 % parked at two barriers per tile):
//   * 8 waves = 4 MFMA waves + 4 loader waves.  The loaders move every tile global -> registers -> LDS (full 64-byte pixel
//     rows per lane group, so every load instruction covers whole cache lines; the first form of this kernel filled the
//     32-byte-per-pixel planes with global_load_lds and spent 3.5 us per 76 KB tile touching 32 lines per instruction) through
//     2 LDS stages + 1 register stage = two tiles of look-ahead; ONE s_barrier per tile.  The loads are plain C++ loads whose
//     results the compiler tracks, so no hand-written waitcnt can be broken by a register-allocation change;
//     tools/check_wgrad_regs.py still lints the register budget of both walks at build time (segmentation_amd/_build.py).
//   * The K index of the GEMM is a LINEARISED window of output pixels (TB images x TR rows x TC columns, padded to 32): a
//     10 x 10 map is 4 K steps (78 % useful) instead of two 8 x 16 tiles (39 %), an 8 x 8 map shares a window with the next
//     images.  ds_read_b64_tr_b16 takes one address per pixel row, so the window shape only lives in a per-lane offset table.
//   * LDS image = one plane per 16 channels, 32 bytes per pixel: a 32-lane half of a transposed read covers 8 consecutive
//     pixels = 256 contiguous bytes = every bank once, with no swizzle (planes are 64 bytes apart modulo the bank period).
//   * Deep layers split the TAPS (9 / 3 / 1 per workgroup) instead of K: every workgroup sweeps all pixels and stores its part of
//     dW directly -- no partial-sum slabs, no reduction launch.  Shallow layers still split K into slabs (wgrad_reduce_kernel).
//   * The bias gradient is one extra MFMA per K step in one wave per dZ fragment (ones x dZ), not one per fragment in every wave.
#include "common.h"
#include "wgrad_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_sw[4] = {0, 0, 0, 0};

struct SwK {
  seg_wgrad_desc d;
  int TR, TC, TB, PR, PC;      // window: TB images x TR x TC output pixels; patch rows / columns per image
  int NPX, KS, NPATCH, XS;     // window pixels, K steps of 32, patch pixels (TB*PR*PC), 32-pixel groups per X plane
  uint32_t m_prpc, m_pc, m_trtc, m_tc;   // reciprocals: q / d == umulhi(q, 2^32 / d + 1) for q < 2^16, d >= 2
  int wy_n, wx_n, wb_n, nwin, ksplit;
  int nchunks0, nchunks, nblk;
  int k_pad, n_pad; long long slab; int direct;
};

// lane (G = lane>>4, qr = (lane&15)>>2, p = lane&3) passes the address of (pixel row, channels 4p..4p+3) for the two halves of
// its K slice; receives channel (lane&15) of 4 + 4 pixels (ds_read_b64_tr_b16: cdna_hip_programming.md T10)
SEG_DEV Frag<bf16_t> tr_read(const char* lo, const char* hi) {
  typedef bf16x4 __attribute__((address_space(3))) * lp;
  bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(lo));
  bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(hi));
  Frag<bf16_t> f;
  f.v = bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return f;
}

#ifdef SEG_STAMPS
// debug builds only (tools/stamp_sweep.py): s_memrealtime stamps (100 MHz, one clock for the whole chip) of wave 0 (MFMA) and
// wave 4 (loader) of every workgroup: [wg][0] row 0: entry, set-up done, walk done, flush done, then (barrier passed, tile
// computed) per tile; row 1: loader set-up done, then the time each of its tile issues was complete
__device__ long long* g_swstamps = nullptr;
__device__ int g_swfilter[3] = {0, 0, 0};      // (Ho, k_pad, n_pad) of the one layer that stamps inside a whole step; 0 = every launch
#define SWSTAMP(row, idx) do { if (swst && (idx) < 32) swst[(row) * 32 + (idx)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SWSTAMP(row, idx) do { } while (0)
#endif

constexpr int sweep_xplane(int xpxmax) { return xpxmax * 32 + 64; }
constexpr int sweep_zplane(int ksmax) { return ksmax * 1024 + 64; }

// CIW x COW = 4 MFMA waves; a wave owns 16 ci x (16 FCO) co for NU x NV taps.  KSMAX / XPXMAX: capacity of a stage (K steps,
// patch pixels per plane); the actual window comes with the launch (SwK).
template <int CIW, int COW, int FCO, int NU, int NV, int KSMAX, int XPXMAX>
__global__ __launch_bounds__(512) void wgrad_sweep_kernel(const SwK P) {
  using T = bf16_t;
  constexpr int NT = NU * NV;
  constexpr int CP = CIW, ZP = FCO * COW;
  constexpr int CIT = 16 * CIW, BN = 16 * ZP;
  // plane strides: + 64 bytes so that the four planes a pixel's pieces go to start 16 banks apart (ds_write_b128 of the loaders)
  constexpr int XPLANE = sweep_xplane(XPXMAX), ZPLANE = sweep_zplane(KSMAX);
  constexpr int STAGE = CP * XPLANE + ZP * ZPLANE;
  constexpr int NSTAGE = 2;
  static_assert(CIW * COW == 4 && FCO <= CIW, "wave layout (one bias fragment per wave at most)");
  static_assert(NSTAGE * STAGE <= 160 * 1024 && XPXMAX % 32 == 0, "LDS budget");
  static_assert((NU == 3 && NV == 3) || (NU == 1 && NV == 3) || (NU == 1 && NV == 1), "tap groups: all 9, one filter row, one tap");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const seg_wgrad_desc& d = P.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x / P.nblk, nb = blockIdx.x % P.nblk;
  const int split = blockIdx.y, tg = blockIdx.z;
  const int u0 = NU == 3 ? 0 : (NV == 3 ? tg : tg / 3);       // first tap of this workgroup's group
  const int v0 = NV == 3 ? 0 : tg % 3;
  const bool first = chunk < P.nchunks0;
  const seg_view& sv = first ? d.src0 : d.src1;
  const int cbase = (first ? chunk : chunk - P.nchunks0) * CIT;
  const int n0 = nb * BN;
  const int ntile = split < P.nwin ? (P.nwin - split + P.ksplit - 1) / P.ksplit : 0;
  auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
#ifdef SEG_STAMPS
  const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  long long* swst = (g_swstamps && wg_lin < 1024 && lane == 0 && (wave == 0 || wave == 4) &&
                     (g_swfilter[0] == 0 || (P.d.Ho == g_swfilter[0] && P.k_pad == g_swfilter[1] && P.n_pad == g_swfilter[2])))
                        ? g_swstamps + (int64_t)wg_lin * 64 : nullptr;
  if (wave == 0) SWSTAMP(0, 0);
#endif

  if (wave < 4) {
    // =========================================== MFMA waves ===========================================
    const int wci = wave / COW, wco = wave % COW;
    const int lr = lane & 15, G = lane >> 4, qr = lr >> 2, p4 = lr & 3;
    // K slot (8G + 4e + qr) of a 32-pixel step <- window pixel mloc(e): any bijection is valid as long as both operands use
    // it; this one keeps a 32-lane half on 8 consecutive pixels (256 contiguous bytes of a plane: conflict-free)
    int xoff[KSMAX][2], zoff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int mloc = 16 * (G >> 1) + 8 * e + 4 * (G & 1) + qr;
      zoff[e] = CP * XPLANE + (wco * FCO) * ZPLANE + mloc * 32 + 8 * p4;
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks) {
        const uint32_t p = ks * 32 + mloc;
        uint32_t pp = 0;                       // padded K slots read patch pixel 0 (finite data); their dZ rows are zero
        if ((int)p < P.NPX) {
          const uint32_t tb = __umulhi(p, P.m_trtc), rem = p - tb * (uint32_t)(P.TR * P.TC);
          const uint32_t r = __umulhi(rem, P.m_tc), c = rem - r * (uint32_t)P.TC;
          pp = tb * (uint32_t)(P.PR * P.PC) + r * (uint32_t)P.PC + c;
        }
        xoff[ks][e] = wci * XPLANE + (int)pp * 32 + 8 * p4;
      }
    }
    int toff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) toff[t] = __builtin_amdgcn_readfirstlane(((NU == 3 ? t / 3 : 0) * P.PC + (NV == 3 ? t % 3 : 0)) * 32);

    f32x4 acc[NT][FCO];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < FCO; ++c) acc[t][c] = f32x4{0, 0, 0, 0};
    f32x4 accb = f32x4{0, 0, 0, 0};
    // bias gradient = column sums of dZ: wave wci adds fragment c == wci (one extra MFMA per K step, ones x dZ); only the
    // workgroups of the first channel chunk and tap group own it
    const bool bias_on = d.bias_mode == 1 && chunk == 0 && tg == 0 && wci < FCO;
    Frag<T> ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones.v[i] = (bf16_t)1.0f;
    const int KS = P.KS;

    auto compute = [&](const char* sx) {
      // flattened (K step, tap) sequence, order pinned: the X fragment of step st+LA and the dZ fragments of the next K step are
      // read before the MFMAs of step st.  D[row = co][col = ci] (dZ is the A operand): a lane ends up with 4 consecutive co of
      // one ci = 16 contiguous bytes of dW[tap][ci][co] -- one dwordx4 store per fragment in the flush instead of four dwords
      constexpr int NSTEP = KSMAX * NT;
      constexpr int LA = 3;
      Frag<T> fx[LA + 1], fz[2][FCO];
      auto issue_x = [&](int st1) {
        const int ks1 = st1 / NT, tap1 = st1 % NT;
        const char* b = sx + toff[tap1];
        fx[st1 % (LA + 1)] = tr_read(b + xoff[ks1][0], b + xoff[ks1][1]);
      };
      auto issue_z = [&](int ks1) {
#pragma unroll
        for (int c = 0; c < FCO; ++c) fz[ks1 & 1][c] = tr_read(sx + zoff[0] + ks1 * 1024 + c * ZPLANE, sx + zoff[1] + ks1 * 1024 + c * ZPLANE);
      };
      __builtin_amdgcn_sched_barrier(0);
      issue_z(0);
#pragma unroll
      for (int i = 0; i < LA; ++i) if (i < NSTEP) issue_x(i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KSMAX; ++ks) {
        if (ks >= KS) break;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
          const int st = ks * NT + tap;
          if (st + LA < NSTEP) issue_x(st + LA);
          if (tap == 0 && ks + 1 < KSMAX) issue_z(ks + 1);
          if (tap == 0 && bias_on) {
#pragma unroll
            for (int c = 0; c < FCO; ++c) if (c == wci) mma32(accb, fz[ks & 1][c], ones);
          }
#pragma unroll
          for (int c = 0; c < FCO; ++c) mma32(acc[tap][c], fz[ks & 1][c], fx[st % (LA + 1)]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    int slot = 0;
    SWSTAMP(0, 1);
    for (int t = 0; t < ntile; ++t) {
      barrier();                                   // tile t has landed (loaders), everybody has finished tile t-1
      SWSTAMP(0, 4 + 2 * t);
      compute(smem + slot * STAGE);
      SWSTAMP(0, 5 + 2 * t);
      slot = slot + 1 == NSTAGE ? 0 : slot + 1;
    }
    SWSTAMP(0, 2);

    // ---- flush: D[row = co][col = ci]; lane: ci = lr, co = 4G .. 4G + 3 ----
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // (dW rows of the arena are only dword-aligned)
    const int kbase = chunk * CIT;
    const int k_log_n = d.src0_clog + d.src1_clog;
    float* slab = d.ws + (int64_t)split * P.slab;
    const int k = kbase + wci * 16 + lr;
    const int cil = k - (first ? 0 : d.src0.c);                    // padded channel inside its source
    const int kl = first ? (cil < d.src0_clog ? cil : -1) : (cil < d.src1_clog ? d.src0_clog + cil : -1);
#pragma unroll
    for (int c = 0; c < FCO; ++c) {
      const int co = n0 + (wco * FCO + c) * 16 + 4 * G;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tap = (u0 + (NU == 3 ? t / 3 : 0)) * 3 + v0 + (NV == 3 ? t % 3 : 0);
        if (P.direct) {
          if (kl >= 0) {
            float* o = d.dw + ((int64_t)tap * k_log_n + kl) * d.n_log + co;
            if (co + 3 < d.n_log) *reinterpret_cast<f32x4u*>(o) = acc[t][c];
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r) if (co + r < d.n_log) o[r] = acc[t][c][r];
            }
          }
        } else {
          *reinterpret_cast<f32x4*>(slab + ((int64_t)tap * P.k_pad + k) * P.n_pad + co) = acc[t][c];
        }
      }
    }
    if (bias_on && lr == 0) {                                      // column 0 of D = sums of dz over the pixels
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = n0 + (wco * FCO + wci) * 16 + 4 * G + r;
        if (P.direct) { if (co < d.bias_n) d.db[co] = accb[r]; }
        else slab[(int64_t)9 * P.k_pad * P.n_pad + co] = accb[r];
      }
    }
#ifdef SEG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SWSTAMP(0, 3);
    return;
  }

  // ============================================= loader waves =============================================
  // Register-staged: every lane moves 16-byte pieces, 8 (4) consecutive lanes = one pixel's 128 (64) bytes = whole cache lines
  // (a first version filled the planes with global_load_lds: one 1-KiB instruction = 32 pixels x 32 bytes = 32 different
  // lines, and the loaders, not the MFMAs, set the tile time: ~3.5 us per 76-KiB tile on 109 workgroups).  Plain loads, so the
  // compiler counts them; the tile after next is IN REGISTERS while the next one sits in the second LDS stage: two tiles of
  // look-ahead from two stages.
  const int ll = (wave - 4) * 64 + lane;                 // 0..255
  constexpr int XPPP = CIT / 8, ZPPP = BN / 8;           // 16-byte pieces per pixel
  constexpr int NXJ = (XPXMAX * XPPP + 255) / 256, NZJ = KSMAX * 32 * ZPPP / 256;
  static_assert(KSMAX * 32 * ZPPP % 256 == 0, "whole piece rounds");
  const T* srcp = reinterpret_cast<const T*>(sv.ptr);
  const T* dzp = reinterpret_cast<const T*>(d.dz.ptr);
  // per piece slot (static index j): byte offset of this lane's 16 bytes relative to the window's first pixel, its patch /
  // window coordinates (edge windows test them): y | x << 10 | image << 20 | inside << 31, and its LDS byte address in a stage
  uint32_t xrel[NXJ], xpk[NXJ], zrel[NZJ], zpk[NZJ];
  // (the LDS address of slot j is affine in j: 256 pieces further = 256 / XPPP pixels further in the same plane)
  const int xlds0 = (ll % XPPP >> 1) * XPLANE + (ll / XPPP) * 32 + (ll % XPPP & 1) * 16;
  const int zlds0 = CP * XPLANE + (ll % ZPPP >> 1) * ZPLANE + (ll / ZPPP) * 32 + (ll % ZPPP & 1) * 16;
#pragma unroll
  for (int j = 0; j < NXJ; ++j) {
    const uint32_t i = j * 256 + ll, q = i / XPPP, pc = i % XPPP;
    const bool in = (int)q < P.NPATCH;
    const uint32_t qq = in ? q : 0;
    const uint32_t tb = __umulhi(qq, P.m_prpc), rem = qq - tb * (uint32_t)(P.PR * P.PC);
    const uint32_t py = __umulhi(rem, P.m_pc), px = rem - py * (uint32_t)P.PC;
    xrel[j] = (((tb * (uint32_t)sv.H + py) * (uint32_t)sv.W + px) * (uint32_t)sv.cs + pc * 8) * 2u;
    xpk[j] = py | (px << 10) | (tb << 20) | (in ? 0x80000000u : 0u);
  }
#pragma unroll
  for (int j = 0; j < NZJ; ++j) {
    const uint32_t i = j * 256 + ll, p = i / ZPPP, pc = i % ZPPP;
    const bool in = (int)p < P.NPX;
    const uint32_t qq = in ? p : 0;
    const uint32_t tb = __umulhi(qq, P.m_trtc), rem = qq - tb * (uint32_t)(P.TR * P.TC);
    const uint32_t r = __umulhi(rem, P.m_tc), c = rem - r * (uint32_t)P.TC;
    zrel[j] = (((tb * (uint32_t)d.dz.H + r) * (uint32_t)d.dz.W + c) * (uint32_t)d.dz.cs + pc * 8) * 2u;
    zpk[j] = r | (c << 10) | (tb << 20) | (in ? 0x80000000u : 0u);
  }
  const char* const zero = reinterpret_cast<const char*>(g_zero16_sw);

  // window coordinates: decoded once, then advanced by ksplit windows per tile (no divisions in the walk)
  struct WinIt { int b, y, x; };
  auto decode = [&](int w) { WinIt it; it.x = w % P.wx_n; w /= P.wx_n; it.y = w % P.wy_n; it.b = w / P.wy_n; return it; };
  auto advance = [&](WinIt& it, const WinIt& st) {
    it.x += st.x; if (it.x >= P.wx_n) { it.x -= P.wx_n; ++it.y; }
    it.y += st.y; if (it.y >= P.wy_n) { it.y -= P.wy_n; ++it.b; }
    it.b += st.b;
  };
  WinIt it = decode(split);
  const WinIt stp = decode(P.ksplit);
  const int64_t s_img = (int64_t)sv.H * sv.W * sv.cs, z_img = (int64_t)d.dz.H * d.dz.W * d.dz.cs;
  u32x4 rx[NXJ], rz[NZJ];

#ifdef SEG_STAMPS
  int nissued = 0;
  SWSTAMP(1, 0);
#endif
  const int Hi_ = d.Hi, Wi_ = d.Wi, Ho_ = d.Ho, Wo_ = d.Wo, Bn_ = d.B, pt_ = d.pad_t, pl_ = d.pad_l;
  const int TB_ = P.TB, TR_ = P.TR, TC_ = P.TC, PR_ = P.PR, PC_ = P.PC;
  const T* const xb0 = srcp + sv.coff + cbase + ((int64_t)sv.oy * sv.W + sv.ox) * sv.cs;
  const T* const zb0 = dzp + d.dz.coff + n0 + ((int64_t)d.dz.oy * d.dz.W + d.dz.ox) * d.dz.cs;
  const int s_row = sv.W * sv.cs, s_col = sv.cs, z_row = d.dz.W * d.dz.cs, z_col = d.dz.cs;
  auto fetch = [&]() {                           // the window `it` points at -> registers
    const int bq = __builtin_amdgcn_readfirstlane(it.b), wy = __builtin_amdgcn_readfirstlane(it.y), wx = __builtin_amdgcn_readfirstlane(it.x);
    const int b0 = bq * TB_, oy0 = wy * TR_, ox0 = wx * TC_;
    const int iy0 = oy0 - pt_ + u0, ix0 = ox0 - pl_ + v0;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + PR_ <= Hi_ && ix0 + PC_ <= Wi_ && oy0 + TR_ <= Ho_ && ox0 + TC_ <= Wo_ && b0 + TB_ <= Bn_;
    const char* xb = reinterpret_cast<const char*>(xb0 + b0 * s_img + ((int64_t)iy0 * s_row + (int64_t)ix0 * s_col));
    const char* zb = reinterpret_cast<const char*>(zb0 + b0 * z_img + ((int64_t)oy0 * z_row + (int64_t)ox0 * z_col));
    // Every slot is loaded, whatever the window size (slots behind the patch re-read pixel 0 / the zero word and are not stored):
    // a load under a condition makes its destination a phi, and the compiler then shuffles the whole register set per slot.
    if (interior) {                              // every pixel of the window exists: no tests
#pragma unroll
      for (int j = 0; j < NXJ; ++j) rx[j] = *reinterpret_cast<const u32x4*>(xb + xrel[j]);
#pragma unroll
      for (int j = 0; j < NZJ; ++j) rz[j] = *reinterpret_cast<const u32x4*>((int)zpk[j] < 0 ? zb + zrel[j] : zero);
    } else {                                     // edge window: all tests evaluated without branches (no short-circuit)
#pragma unroll
      for (int j = 0; j < NXJ; ++j) {
        uint32_t pk = xpk[j];
        asm volatile("" : "+v"(pk));             // (keeps the unpacking HERE: hoisted out of the walk it triples the table's registers)
        const int py = pk & 1023, px = (pk >> 10) & 1023, tb = (pk >> 20) & 2047;
        const bool ok = ((int)pk < 0) & ((unsigned)(iy0 + py) < (unsigned)Hi_) & ((unsigned)(ix0 + px) < (unsigned)Wi_) & (b0 + tb < Bn_);
        rx[j] = *reinterpret_cast<const u32x4*>(ok ? xb + xrel[j] : zero);
      }
#pragma unroll
      for (int j = 0; j < NZJ; ++j) {
        uint32_t pk = zpk[j];
        asm volatile("" : "+v"(pk));
        const int r = pk & 1023, c = (pk >> 10) & 1023, tb = (pk >> 20) & 2047;
        const bool ok = ((int)pk < 0) & (oy0 + r < Ho_) & (ox0 + c < Wo_) & (b0 + tb < Bn_);      // K slots behind the window: ZERO rows of dZ
        rz[j] = *reinterpret_cast<const u32x4*>(ok ? zb + zrel[j] : zero);
      }
    }
    advance(it, stp);
#ifdef SEG_STAMPS
    if (swst) { SWSTAMP(1, 1 + nissued); ++nissued; }
#endif
  };
  auto commit = [&](char* sbase) {               // registers -> stage
#pragma unroll
    for (int j = 0; j < NXJ; ++j) if ((int)xpk[j] < 0) *reinterpret_cast<u32x4*>(sbase + xlds0 + j * (256 / XPPP) * 32) = rx[j];   // (pieces behind the patch are not stored)
#pragma unroll
    for (int j = 0; j < NZJ; ++j) *reinterpret_cast<u32x4*>(sbase + zlds0 + j * (256 / ZPPP) * 32) = rz[j];
  };

  // tile 0 -> stage 0 before the first barrier; from then on, right behind barrier t (everybody has finished tile t-1: its
  // stage is free) tile t+1 goes from the registers into that stage and the loads of tile t+2 are issued
  if (ntile > 0) { fetch(); commit(smem); }
  if (ntile > 1) fetch();
  for (int t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // my LDS stores of tile t are done
    barrier();
    if (t + 1 < ntile) commit(smem + ((t + 1) & 1) * STAGE);
    if (t + 2 < ntile) fetch();
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------
struct Geom { int TR, TC, TB, PR, PC, NPX, KS, NPATCH, XS, wy_n, wx_n, wb_n, nwin; double cyc_win; /* us per window */ };

// window for (Ho, Wo, B): whole images when a map fits a stage (several images per window when it is tiny), else the TR x TC
// rectangle with the least (padded K steps + per-window overhead)
bool choose_window(int B, int Ho, int Wo, int NU, int NV, int KSMAX, int XPXMAX, int mfma_per_step, int loader_rounds, Geom* out) {
  const int KMAX = KSMAX * 32;
  double best = 1e30; bool found = false;
  auto consider = [&](int TR, int TC, int TB) {
    const int PR = TR + NU - 1, PC = TC + NV - 1;
    const int NPX = TB * TR * TC, NPATCH = TB * PR * PC;
    if (NPX > KMAX || NPATCH > XPXMAX || TR < 1 || TC < 1 || TB < 1 || TR > 1023 || TC > 1023 || PR > 1023 || PC > 1023) return;
    if (TR * TC < 2 || PR * PC < 2 || TC < 2 || PC < 2) return;            // (the reciprocals need divisors >= 2)
    Geom g;
    g.TR = TR; g.TC = TC; g.TB = TB; g.PR = PR; g.PC = PC; g.NPX = NPX; g.NPATCH = NPATCH;
    g.KS = (NPX + 31) / 32; g.XS = (NPATCH + 31) / 32;
    g.wy_n = cdiv(Ho, TR); g.wx_n = cdiv(Wo, TC); g.wb_n = cdiv(B, TB);
    g.nwin = g.wy_n * g.wx_n * g.wb_n;
    // time of one window [us] (r03 stamps, tools/stamp_sweep.py): the MFMA waves run KS * mfma_per_step MFMAs of 16 cycles at the
    // ~1.55 GHz the chip holds in an LDS + MFMA loop; the loader waves need ~0.1 us per round of 256 16-byte pieces (ds_write_b128
    // bandwidth) and every round of the instance is executed whatever the window size; whichever is longer sets the pace
    const double t_mfma = g.KS * mfma_per_step * 16.0 / 1550.0 + 0.1, t_load = 0.1 * loader_rounds + 0.15;
    g.cyc_win = t_mfma > t_load ? t_mfma : t_load;
    const double cost = g.nwin * g.cyc_win;
    if (cost < best) { best = cost; *out = g; found = true; }
  };
  if (Ho * Wo <= KMAX) {
    for (int tb = 1; tb <= B && tb * Ho * Wo <= KMAX; ++tb) consider(Ho, Wo, tb);
  }
  static const int tcs[] = {8, 12, 16, 20, 24, 32, 40, 48, 64, 96, 128};
  for (int tc0 : tcs) {
    const int tc = tc0 < Wo ? tc0 : Wo;
    for (int tr = 1; tr <= Ho && tr * tc <= KMAX; ++tr) consider(tr, tc, 1);
  }
  if (Wo <= KMAX) for (int tr = 1; tr <= Ho && tr * Wo <= KMAX; ++tr) consider(tr, Wo, 1);
  return found;
}

struct Choice { int layout, tgs, cls, ks; Geom g; double cost; };

template <int CIW, int COW, int FCO, int NU, int NV, int KSMAX, int XPXMAX>
int launch_inst(const SwK& P, int tgs, hipStream_t st, const WgQuery& q) {
  constexpr int STAGE = CIW * sweep_xplane(XPXMAX) + FCO * COW * sweep_zplane(KSMAX);
  constexpr int LDS = 2 * STAGE;
  if (q.name_out) {
    snprintf(q.name_out, q.name_cap, "wgrad_sweep_kernel<%d,%d,%d,%d,%d,%d,%d>", CIW, COW, FCO, NU, NV, KSMAX, XPXMAX);
    return SEG_OK;
  }
  auto kern = wgrad_sweep_kernel<CIW, COW, FCO, NU, NV, KSMAX, XPXMAX>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      seg_set_error("wgrad_sweep: cannot raise dynamic LDS to %d", LDS); return SEG_ERR_LAUNCH;
    }
    attr_done = true;
  }
  SEG_LAUNCH(kern, dim3(P.nchunks * P.nblk, P.ksplit, tgs), dim3(512), LDS, st, P);
  return seg_check_launch("wgrad_sweep");
}

// instance table: layout (0 = 64 ci x 64 co, 1 = 32 x 64, 2 = 64 x 32, 3 = 32 x 32), tap groups (1, 3, 9), class (0 = 4 K steps /
// 192 patch pixels / 3 stages, 1 = 8 K steps / 352 patch pixels / 2 stages; the 32-channel layouts: 3 stages either way)
bool inst_exists(int layout, int tgs, int cls) {
  if (layout == 0) return true;
  return tgs == 1 && cls == 1;
}
void inst_caps(int layout, int cls, int* ksmax, int* xpx) {
  (void)layout;
  *ksmax = cls == 0 ? 4 : 8; *xpx = cls == 0 ? 192 : 352;
}

int launch_choice(const SwK& P, const Choice& c, hipStream_t st, const WgQuery& q) {
#define SW_CASE(L, TG, CL, ...) if (c.layout == L && c.tgs == TG && c.cls == CL) return launch_inst<__VA_ARGS__>(P, TG, st, q)
  SW_CASE(0, 1, 0, 4, 1, 4, 3, 3, 4, 192);
  SW_CASE(0, 1, 1, 4, 1, 4, 3, 3, 8, 352);
  SW_CASE(0, 3, 0, 4, 1, 4, 1, 3, 4, 192);
  SW_CASE(0, 3, 1, 4, 1, 4, 1, 3, 8, 352);
  SW_CASE(0, 9, 0, 4, 1, 4, 1, 1, 4, 192);
  SW_CASE(0, 9, 1, 4, 1, 4, 1, 1, 8, 352);
  SW_CASE(1, 1, 1, 2, 2, 2, 3, 3, 8, 352);
  SW_CASE(2, 1, 1, 4, 1, 2, 3, 3, 8, 352);
  SW_CASE(3, 1, 1, 2, 2, 1, 3, 3, 8, 352);
#undef SW_CASE
  seg_set_error("wgrad_sweep: no instance for layout %d, %d tap groups, class %d", c.layout, c.tgs, c.cls);
  return SEG_ERR_UNSUPPORTED;
}

inline int sweep_target_wgs() {
  static const int v = getenv("SEG_WGRAD_WGS") ? atoi(getenv("SEG_WGRAD_WGS")) : 128;
  return v > 0 ? v : 128;
}

// Decomposition of one layer: tap groups x K splits x window class, by a cost model [us] (eval below)
bool choose(const seg_wgrad_desc& d, int layout, Choice* best) {
  static const int CITs[] = {64, 32, 64, 32}, BNs[] = {64, 64, 32, 32}, FCOs[] = {4, 2, 2, 1};
  const int CIT = CITs[layout], BN = BNs[layout], FCO = FCOs[layout];
  const int kin = d.src0.c + (d.src1.ptr ? d.src1.c : 0);
  const int base = (kin / CIT) * (d.dz.c / BN);
  const int target = d.target_wgs > 0 ? d.target_wgs : sweep_target_wgs();
  // experiments: SEG_SWEEP_FORCE="tgs,ks,cls" (0 = automatic) -- tools/wgrad_micro.py
  int f_tgs = 0, f_ks = 0, f_cls = -1;
  if (const char* e = getenv("SEG_SWEEP_FORCE")) sscanf(e, "%d,%d,%d", &f_tgs, &f_ks, &f_cls);
  if (d.cfg >= 100) { const int t = d.cfg - 100; if (t == 1 || t == 3 || t == 9) f_tgs = t; if (d.cfg >= 200) { f_tgs = (d.cfg - 200) / 10; f_cls = (d.cfg - 200) % 10; } }
  if (d.ksplit > 0) f_ks = d.ksplit;
  bool found = false;
  best->cost = 1e30;
  const double reduce_us = 5.5;
  static const int tg_opts[] = {1, 3, 9};
  for (int tgs : tg_opts) {
    if (f_tgs && tgs != f_tgs) continue;
    const int NU = tgs == 1 ? 3 : 1, NV = tgs == 9 ? 1 : 3, NT = NU * NV;
    for (int cls = 0; cls < 2; ++cls) {
      if (f_cls >= 0 && cls != f_cls) continue;
      if (!inst_exists(layout, tgs, cls)) continue;
      int ksmax, xpx; inst_caps(layout, cls, &ksmax, &xpx);
      Geom g;
      const int rounds_ld = (xpx * (CIT / 8) + 255) / 256 + ksmax * 32 * (BN / 8) / 256;
      if (!choose_window(d.B, d.Ho, d.Wo, NU, NV, ksmax, xpx, NT * FCO + 1, rounds_ld, &g)) continue;
      const double slab_tile = (double)NT * CIT * BN * 4.0;
      auto eval = [&](int ks) {
        // [us] launch 3 + rounds * (4.5 until the first window has landed + windows + flush at ~28 ns per KB of accumulators (slab;
        // the direct form stores a third of that per byte... measured 1.6 against 4.1 us)) + the reduction launch
        const long wgs = (long)base * tgs * ks;
        const double rounds = (double)((wgs + target - 1) / target);
        const double tiles = (double)((g.nwin + ks - 1) / ks);
        const double t_wg = 4.5 + tiles * g.cyc_win + slab_tile / 1024.0 * (ks > 1 ? 0.028 : 0.011);
        const double t_s = ks > 1 ? reduce_us + wgs * slab_tile / 6.0e6 : 0.0;
        const double cost = 3.0 + rounds * t_wg + t_s;
        if (cost < best->cost) { best->cost = cost; best->layout = layout; best->tgs = tgs; best->cls = cls; best->ks = ks; best->g = g; found = true; }
      };
      if (f_ks) { eval(f_ks < g.nwin ? f_ks : g.nwin); continue; }
      for (int ks = 1; ks <= g.nwin; ks = ks < 8 ? ks + 1 : ks + (ks + 3) / 4) {
        if ((long)base * tgs * ks > 4L * target && ks > 1) break;
        eval(ks);
      }
    }
  }
  return found;
}

}  // namespace

#ifdef SEG_STAMPS
extern "C" int seg_dbg_set_swfilter(int ho, int k_pad, int n_pad) {
  const int v[3] = {ho, k_pad, n_pad};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_swfilter), v, sizeof(v)) == hipSuccess ? 0 : -1;
}
extern "C" int seg_dbg_set_swstamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_swstamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

int seg_wgrad_sweep(const seg_wgrad_desc& d, const WgQuery& q, hipStream_t st, int* rc) {
  static const bool off = getenv("SEG_WGRAD_IMPL") && !strcmp(getenv("SEG_WGRAD_IMPL"), "old");
  if (off || d.dtype != SEG_BF16 || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.im2col_x || d.pool_y.ptr || d.bias_mode == 2) return 0;
  if (d.cfg > 0 && d.cfg < 100) return 0;                    // an explicit layout of the register-staged kernel
  if (d.B <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.Hi != d.Ho + 2 - 2 * d.pad_t || d.Wi != d.Wo + 2 - 2 * d.pad_l) return 0;
  const bool two = d.src1.ptr != nullptr;
  const bool ci64 = d.src0.c % 64 == 0 && (!two || d.src1.c % 64 == 0), co64 = d.dz.c % 64 == 0;
  const int layout = ci64 ? (co64 ? 0 : 2) : (co64 ? 1 : 3);
  Choice c;
  if (!choose(d, layout, &c)) return 0;
  const Geom& g = c.g;
  // 32-bit in-window byte offsets: the TB images of a window stay below 4 GB in every tensor
  if ((int64_t)g.TB * d.src0.H * d.src0.W * d.src0.cs * 2 >= ((int64_t)1 << 32) || (two && (int64_t)g.TB * d.src1.H * d.src1.W * d.src1.cs * 2 >= ((int64_t)1 << 32)) ||
      (int64_t)g.TB * d.dz.H * d.dz.W * d.dz.cs * 2 >= ((int64_t)1 << 32)) return 0;
  static const int CITs[] = {64, 32, 64, 32}, BNs[] = {64, 64, 32, 32};
  const int CIT = CITs[layout], BN = BNs[layout];
  SwK P;
  P.d = d;
  if (!two) { P.d.src1 = d.src0; P.d.src1.c = 0; P.d.src1_clog = 0; }
  P.TR = g.TR; P.TC = g.TC; P.TB = g.TB; P.PR = g.PR; P.PC = g.PC; P.NPX = g.NPX; P.KS = g.KS; P.NPATCH = g.NPATCH; P.XS = g.XS;
  auto magic = [](int dv) { return (uint32_t)((((uint64_t)1) << 32) / (uint64_t)dv + 1); };
  P.m_prpc = magic(g.PR * g.PC); P.m_pc = magic(g.PC); P.m_trtc = magic(g.TR * g.TC); P.m_tc = magic(g.TC);
  P.wy_n = g.wy_n; P.wx_n = g.wx_n; P.wb_n = g.wb_n; P.nwin = g.nwin; P.ksplit = c.ks;
  P.nchunks0 = d.src0.c / CIT; P.nchunks = P.nchunks0 + (two ? d.src1.c / CIT : 0); P.nblk = d.dz.c / BN;
  P.k_pad = P.nchunks * CIT; P.n_pad = d.dz.c;
  const int64_t bias_len = P.k_pad > P.n_pad ? P.k_pad : P.n_pad;
  P.slab = (int64_t)9 * P.k_pad * P.n_pad + bias_len;
  P.direct = c.ks == 1;
  *rc = SEG_OK;
  if (q.name_out) { *rc = launch_choice(P, c, st, q); return 1; }
  if (q.plan_ks) { *q.plan_ks = c.ks; *q.plan_bytes = P.direct ? 0 : P.slab * c.ks * 4; return 1; }
  if (q.job_out && P.direct) { q.job_out->nblocks = 0; return 1; }
  if (!P.direct && (!d.ws || d.ws_bytes < P.slab * c.ks * 4)) {
    seg_set_error("wgrad: workspace too small (%lld < %lld bytes)", (long long)d.ws_bytes, (long long)(P.slab * c.ks * 4)); *rc = SEG_ERR_ARG; return 1;
  }
  if (d.phase != 2 && !q.job_out) {
    *rc = launch_choice(P, c, st, q);
    if (*rc) return 1;
  }
  if (P.direct || (d.phase == 1 && !q.job_out)) return 1;
  RedArgs RA;
  RA.ws = d.ws; RA.dw = d.dw; RA.db = d.db; RA.slab = P.slab; RA.ksplit = c.ks; RA.taps = 9; RA.k_pad = P.k_pad; RA.n_pad = P.n_pad;
  RA.seg0_c = d.src0_clog; RA.seg0_cp = d.src0.c; RA.seg1_c = two ? d.src1_clog : 0; RA.n_log = d.n_log; RA.bias_mode = d.bias_mode; RA.bias_n = d.bias_n;
  *rc = seg_wgrad_reduce_launch(RA, c.ks, q, st);
  return 1;
}
