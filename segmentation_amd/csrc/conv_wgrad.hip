// conv_wgrad.hip -- filter gradient (Conv2DBackpropFilter) as an MFMA GEMM whose reduction
// dimension is the pixel index:  dW[tap][ci][co] = sum_{b,y,x} X[b, y*S+u-pt, x*S+v-pl, ci] * dZ[b,y,x,co].
// Serves the 3x3 / 1x1 convs and, with X := dZ_big (stride 2, 2x2 taps) and dZ := x_small, the
// 2x2/s2 transposed conv (TF autodiff of /root/reference/models/unet.py:111-166, models/fcn.py:110-128).
//
// Both MFMA operands need K (= pixel) contiguous per lane while NHWC keeps channels contiguous, so the
// bf16 path reads its fragments with ds_read_b64_tr_b16 (4 pixels x 16 channels, delivered
// column-major); the f32 path reads single elements.  A workgroup owns one CIT-channel chunk of X (32 or 64) and
// BN channels of dZ for ALL taps, keeps the KH*KW*CIT*BN partial sums in registers while it walks its
// share of the pixel tiles, and flushes them ONCE, with plain coalesced stores, into its own slab of a
// workspace; a second small kernel sums the `ksplit` slabs in a fixed order into the TF-layout gradient
// (deterministic, no atomics: 768 workgroups hammering one 36 KB filter with f32 atomics measured 10-30x
// slower than the MFMA work).  The bias gradient rides along as one extra MFMA against an all-ones fragment.
#include "common.h"
#include "wgrad_common.h"
#include <stdlib.h>

namespace {

struct WgK {
  seg_wgrad_desc d;
  int tiles_x, tiles_y, ntiles, ksplit;
  int nchunks0, nchunks, nblk;   // K chunks of src0 / total; BN blocks
  int k_pad, n_pad;              // slab = [taps][k_pad][n_pad] floats followed by bias[max(k_pad,n_pad)]
  int64_t slab;                  // floats per split
  int direct;                    // ksplit == 1: store straight into dw/db, no reduce pass
};

// Debug builds only (SEG_EXTRA_FLAGS=-DSEG_WABL=bits, tools/wgrad_ablate.sh): parts of the tile walk compiled out one at a
// time to see which of them the run time follows.  1 MFMAs, 2 LDS fragment reads, 4 LDS commits, 8 global loads.
// Results are garbage with any bit set.
#ifndef SEG_WABL
#define SEG_WABL 0
#endif

SEG_DEV void frag_keep(const Frag<bf16_t>& f) { asm volatile("" :: "v"(f.v)); }
SEG_DEV void frag_keep(const Frag<float>& f) { asm volatile("" :: "v"(f.lo), "v"(f.hi)); }
SEG_DEV void frag_undef(Frag<bf16_t>& f) { asm volatile("" : "=v"(f.v)); }
SEG_DEV void frag_undef(Frag<float>& f) { asm volatile("" : "=v"(f.lo), "=v"(f.hi)); }

#ifdef SEG_STAMPS
__device__ long long* g_wstamps = nullptr;     // debug builds only: [workgroup][4 rows][32] s_memtime stamps of wave 0
#define WSTAMP(row, idx) do { if (wstp && (idx) < 32) wstp[(row) * 32 + (idx)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(row, idx) do { } while (0)
#endif

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

// On-the-fly im2col of the first layer (IMC = input channels 1..3): raw-image loads of N dwords.
typedef __attribute__((ext_vector_type(3))) uint32_t u32x3;
template <int N> struct XVsel { typedef uint32_t type; };
template <> struct XVsel<2> { typedef u32x2 type; };
template <> struct XVsel<3> { typedef u32x3 type; };
template <int N> SEG_DEV uint32_t xv_get(const typename XVsel<N>::type& r, int i) { if constexpr (N == 1) return r; else return r[i]; }
template <int N, int IMM> SEG_DEV void xload_s(typename XVsel<N>::type& r, unsigned voff, uint64_t sbase) {
  if constexpr (N == 3) asm volatile("global_load_dwordx3 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  else if constexpr (N == 2) asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  else asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
template <int N> SEG_DEV void xload_p(typename XVsel<N>::type& r, const float* ptr) {
  if constexpr (N == 3) asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  else if constexpr (N == 2) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  else asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
}

// RS = 1: "row split" -- blockIdx.z selects one filter row u, the workgroup keeps only KW taps in registers.
// Used for the deep layers (few pixel tiles, big filters) where splitting K cannot create enough workgroups:
// it multiplies the workgroup count by KH without any partial-sum traffic.
//
// IMC > 0 (first layer, models/unet.py:111 conv1_1 / models/fcn.py:110 conv1): src0 is VIRTUAL -- the 3x3 im2col of the dense
// float32 input image (d.im2col_*), channel tap*IMC + ci of output pixel (y,x) = image[b, y+u-pad, x+v-pad, ci] (zero outside).
// The 1x1 walk over 256-pixel tiles stays as it is; only the staging differs: thread q of the workgroup gathers the 9*IMC
// values of tile pixel q straight from the image (nine loads of IMC dwords: neighbours overlap in cache, the image is read
// from HBM once), rounds them to T and writes the pixel's 32-channel row of the LDS patch.  The [B,Ho,Wo,32] im2col tensor
// (as large as the layer's output) is never written or read back.
template <int DT, int TH, int TW, int KH, int KW, int S, int WCI, int WCO, int FCI, int FCO, int RS, int IMC = 0>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgK P) {
  using T = typename DtSel<DT>::type;
  constexpr bool IM = IMC > 0;
  constexpr int ICH = IMC & 3;                   // input channels of the virtual im2col source
  constexpr bool PDZ = (IMC & 4) != 0;           // dZ is virtual too: rebuilt from the 2x2 max-pool that consumes this layer (below)
  using XV = typename XVsel<IM ? ICH : 1>::type;
  constexpr int BM = TH * TW;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, NPIX = PH * PW;
  constexpr int NU = RS ? 1 : KH;
  constexpr int NT = NU * KW;                 // taps held by this workgroup
  constexpr int NTF = KH * KW;                // taps of the filter
  constexpr int ES = sizeof(T);
  constexpr int BN = 16 * FCO * WCO;
  constexpr int CIT = 16 * FCI * WCI;         // X channels per workgroup: 32, or 64 (bf16 register-reuse layouts below)
  // Wave layout matters more than the tile: every tap needs its own (shifted) X fragment, the dZ fragment is shared by all
  // taps.  A wave that owns FCI x FCO fragments reads 9*FCI + FCO fragments from LDS per 9*FCI*FCO MFMAs: the 32x32 tile
  // with one fragment pair per wave (r01 default) reads 1.1 fragments per MFMA = 284 B/clk per CU, MORE than the LDS delivers
  // (256 B/clk) -- it was LDS-read-bound at 0.06-0.10 of the MFMA peak; 16 ci x 64 co per wave (FCI 1, FCO 4) reads 0.36.
  // LDS row strides.  bf16: +32 B of padding and the pixel<->k permutation below make every ds_read_b64_tr_b16
  // conflict-free (each 32-lane half then reads 8 consecutive pixel rows whose 32-byte chunks fall in 8 distinct
  // bank groups: 96 B and BN*2+32 B strides are 3 resp. odd multiples of 32 B); f32 (parity mode): +16 B.
  constexpr int RSP = CIT * ES + (ES == 2 ? 32 : 16);  // patch row stride (bytes)
  constexpr int RSZ = BN * ES + (ES == 2 ? 32 : 16);   // dZ tile row stride
  constexpr int PPIECES = CIT * ES / 16, ZPIECES = BN * ES / 16, EPP = 16 / ES;
  constexpr int PATCH_BYTES = ((NPIX * RSP + 15) / 16) * 16;
  constexpr int NPP = (NPIX * PPIECES + 255) / 256;
  constexpr int NZP = (BM * ZPIECES + 255) / 256;
  constexpr int KS = BM / 32;
  static_assert(WCI * WCO == 4 && (CIT == 32 || CIT == 64), "wave layout");
  static_assert(BM % 32 == 0, "tile pixels");
  static_assert(!IM || (KH == 1 && KW == 1 && S == 1 && BM == 256 && CIT == 32 && ICH >= 1), "im2col staging: one tile pixel per thread");
  static_assert(!PDZ || (sizeof(T) == 2 && BN == 32 && TH == 16 && TW == 16), "pooled-dZ staging: one (window, 8-channel group) per thread");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;
  char* sZ = smem + PATCH_BYTES;

  const seg_wgrad_desc& d = P.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wci = wave / WCO, wco = wave % WCO;
  const int lr = lane & 15, G = lane >> 4;

  const int chunk = blockIdx.x / P.nblk, nb = blockIdx.x % P.nblk;
  const int u0 = RS ? blockIdx.z : 0;
  const bool first = chunk < P.nchunks0;
  const seg_view& sv = first ? d.src0 : d.src1;
  const int cbase = first ? chunk * CIT : (chunk - P.nchunks0) * CIT; // padded channel base inside its source
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
  // bias sums are always accumulated (one extra MFMA per K step / per tap for the stride-2 form) and only written
  // by the workgroups that own them: keeps the MFMA loop free of branches
  constexpr bool B2 = (S == 2);                 // stride-2 instantiations serve the transposed conv (bias = sums of src)
  const bool bias1 = !B2 && d.bias_mode == 1 && chunk == 0 && wci == 0 && u0 == 0;
  const bool bias2 = B2 && d.bias_mode == 2 && nb == 0 && wco == 0;
  f32x4 accb1[FCO], accb2[FCI];
#pragma unroll
  for (int c = 0; c < FCO; ++c) accb1[c] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < FCI; ++a) accb2[a] = f32x4{0, 0, 0, 0};
  const Frag<T> ones = ones_frag<T>();

  // ---- staging: everything that does not depend on the tile is computed once per thread ----
  // piece i of this thread: patch pixel (py,px) / dZ pixel (ty,tx), 16-byte piece h; element offset relative to the
  // tile origin and the LDS byte address are tile-invariant.  Interior tiles (the vast majority) take the
  // unconditional path; edge tiles re-derive the pixel coordinates and zero-fill.
  int sp_off[NPP], sp_lds[NPP], sz_off[NZP], sz_lds[NZP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int idx = tid + i * 256;
    const int q = idx / PPIECES, h = idx % PPIECES;
    sp_off[i] = ((q / PW) * sv.W + (q % PW)) * sv.cs + h * EPP;
    sp_lds[i] = idx < NPIX * PPIECES ? q * RSP + h * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < NZP; ++i) {
    const int idx = tid + i * 256;
    const int m = idx / ZPIECES, h = idx % ZPIECES;
    sz_off[i] = ((m / TW) * d.dz.W + (m % TW)) * d.dz.cs + h * EPP;
    sz_lds[i] = (BM * ZPIECES % 256 == 0 || idx < BM * ZPIECES) ? m * RSZ + h * 16 : -1;
  }
  // Two staging sets = two tiles of global loads in flight.  The loads are inline asm and the waits are explicit
  // (s_waitcnt vmcnt(N) with the set's registers tied to it): the compiler's own wait insertion cannot count across
  // the loop back-edge and falls back to vmcnt(0), which would drain the other set's prefetch at every commit.
  // Every load is issued unconditionally (clamped address; a bit mask remembers which pieces to zero), so each
  // prefetch is exactly NLD wave-instructions and "the other set is younger" is vmcnt(NLD).
  constexpr int NXL = IM ? 9 : NPP;               // loads per thread for the X side of a tile
  constexpr int NZL = PDZ ? 9 : NZP;              // ... and for the dZ side (pooled-dZ staging: 4 activations, 1 pooled gradient, 4 skip gradients)
  constexpr int NLD = NXL + NZL;
  static_assert(NLD <= 60, "vmcnt range / mask bits");
  constexpr bool DUAL = sizeof(T) == 2 && !PDZ;      // f32 (parity mode) keeps one set: twice the registers per piece; so does the
                                                     // pooled-dZ staging (18 loads of 16 bytes per thread and tile: two sets would not fit the register budget)
  u32x4 rpA[NPP], rzA[NZP], rpB[DUAL ? NPP : 1], rzB[DUAL ? NZP : 1];
  XV xA[IM ? 9 : 1], xB[IM && DUAL ? 9 : 1];
  u32x4 qA[PDZ ? 9 : 1], qB[PDZ && DUAL ? 9 : 1];   // pooled-dZ staging: [0..3] activation window, [4] pooled gradient, [5..8] skip gradient
  uint64_t okA = 0, okB = 0;
  auto gload = [&](u32x4& r, const T* ptr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory"); };
  // interior tiles: wave-uniform 64-bit base in SGPRs + the thread's constant 32-bit byte offset (no per-load 64-bit VALU)
  auto gload_s = [&](u32x4& r, unsigned voff, uint64_t sbase) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory"); };
  auto uniform64 = [&](const void* p) -> uint64_t {      // the pointer IS wave-uniform; make it so for the register allocator
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    // gfx9 hazard: a VALU write of an SGPR (v_readfirstlane) needs 5 wait states before a VMEM instruction reads that SGPR
    // as its base.  hipcc pads nothing for inline asm (the loads below ARE inline asm), and with no pad the first load of a
    // tile can go out with the PREVIOUS contents of the register pair: a stale base (silently the previous tile's data) or
    // garbage (memory access fault: f32 8x16 stride-2 tile at 512x512, where the pair had just held kernel arguments).
    // The nops are tied to the pair as in/out operands, so every use of the base is ordered behind them.
    asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
    return ((uint64_t)hi << 32) | lo;
  };
  unsigned sp_boff[NPP], sz_boff[NZP];
  uint64_t full = 0;                                    // pieces this thread owns at all (interior tiles: all valid)
#pragma unroll
  for (int i = 0; i < NPP; ++i) { sp_boff[i] = sp_lds[i] >= 0 ? (unsigned)sp_off[i] * ES : 0u; full |= (uint64_t)(sp_lds[i] >= 0 ? 1 : 0) << i; }
#pragma unroll
  for (int i = 0; i < NZP; ++i) { sz_boff[i] = sz_lds[i] >= 0 ? (unsigned)sz_off[i] * ES : 0u; full |= (uint64_t)(sz_lds[i] >= 0 ? 1 : 0) << (NXL + i); }
  if constexpr (IM) full |= 0x1ffull;
  // tile coordinates advance incrementally (no divisions in the walk): a stride of `step` tiles = (sb_, sy_, sx_)
  struct TileIt { int b, ty, tx; };
  auto tile_decode = [&](int t) { TileIt it; it.tx = t % P.tiles_x; t /= P.tiles_x; it.ty = t % P.tiles_y; it.b = t / P.tiles_y; return it; };
  auto tile_advance = [&](TileIt& it, const TileIt& st) {
    it.tx += st.tx; if (it.tx >= P.tiles_x) { it.tx -= P.tiles_x; ++it.ty; }
    it.ty += st.ty; if (it.ty >= P.tiles_y) { it.ty -= P.tiles_y; ++it.b; }
    it.b += st.b;
  };
  // tile (b, ty, tx) -> first element of its patch / dZ window: base + b * image + ty * tile-row step + tx * tile-column step.
  // The steps are hoisted (an image plane is < 2^31 elements, checked by the host): per tile two 32-bit products and one
  // 64 x 32 one per pointer instead of a chain of 64-bit multiplies -- the scalar address arithmetic was a third of the
  // non-MFMA instructions of a tile (r02 counters: 44 % of a lone workgroup's cycles in the MFMAs, 31 % issuing other work).
  const int s_ty = TH * S * sv.W * sv.cs, s_tx = TW * S * sv.cs, z_ty = TH * d.dz.W * d.dz.cs, z_tx = TW * d.dz.cs;
  const int64_t s_img = (int64_t)sv.H * sv.W * sv.cs, z_img = (int64_t)d.dz.H * d.dz.W * d.dz.cs;
  const T* const s_base = srcp + sv.coff + cbase + ((int64_t)(sv.oy - d.pad_t) * sv.W + (sv.ox - d.pad_l)) * sv.cs;
  const T* const z_base = dzp + d.dz.coff + n0 + ((int64_t)d.dz.oy * d.dz.W + d.dz.ox) * d.dz.cs;
  auto prefetch = [&](const TileIt& it, u32x4 (&rp)[NPP], u32x4 (&rz)[NZP], uint64_t& okm, bool& inter) {
    const int b = __builtin_amdgcn_readfirstlane(it.b), ty = __builtin_amdgcn_readfirstlane(it.ty), tx = __builtin_amdgcn_readfirstlane(it.tx);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - d.pad_t, ix0 = ox0 * S - d.pad_l;
    const T* sb = s_base + b * s_img + (ty * s_ty + tx * s_tx);
    const T* zb = z_base + b * z_img + (ty * z_ty + tx * z_tx);
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + PH <= d.Hi && ix0 + PW <= d.Wi && oy0 + TH <= d.Ho && ox0 + TW <= d.Wo;
    inter = interior;
    if (interior) {
      const uint64_t sbu = uniform64(sb), zbu = uniform64(zb);
#pragma unroll
      for (int i = 0; i < NPP; ++i) gload_s(rp[i], sp_boff[i], sbu);
#pragma unroll
      for (int i = 0; i < NZP; ++i) gload_s(rz[i], sz_boff[i], zbu);
      okm = full;
      return;
    }
    uint64_t m = 0;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int q = (tid + i * 256) / PPIECES;
      const int iy = iy0 + q / PW, ix = ix0 + q % PW;
      const bool ok = sp_lds[i] >= 0 && iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi;
      m |= (uint64_t)(ok ? 1 : 0) << i;
      gload(rp[i], ok ? sb + sp_off[i] : srcp);
    }
#pragma unroll
    for (int i = 0; i < NZP; ++i) {
      const int mm = (tid + i * 256) / ZPIECES;
      const bool ok = sz_lds[i] >= 0 && oy0 + mm / TW < d.Ho && ox0 + mm % TW < d.Wo;
      m |= (uint64_t)(ok ? 1 : 0) << (NXL + i);
      gload(rz[i], ok ? zb + sz_off[i] : dzp);
    }
    okm = m;
  };
  // im2col staging: tile pixel q = tid; image offsets of its three filter rows relative to the tile's first input pixel
  const float* imx = d.im2col_x;
  const int imH = d.im2col_h, imW = d.im2col_w, impad = d.im2col_pad;
  unsigned x_boff[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) x_boff[u] = (unsigned)(((tid / TW + u) * imW + tid % TW) * (IM ? ICH : 1) * 4);
  // ---- pooled-dZ staging (PDZ) ----
  // dZ of the first layer = (its activation y > 0) * (the pooled gradient routed to the window's first maximum + the gradient of
  // its other consumer) -- exactly seg_maxpool2x2_bwd (models/unet.py pool1 over conv1_1 with conv1_2 as second consumer, finding
  // F11; models/fcn.py:116 pool1) -- rebuilt per tile from y, dpool and the skip gradient instead of being written by a launch of
  // its own on the critical stream and read back.  Thread t owns window t / 4 of the tile's 8 x 8 windows and channel group t % 4:
  // nine 16-byte loads give the four dZ pieces of that window.  Mask bits: 9..12 pixel k inside the map, 13 routing on,
  // 14..17 skip gradient of pixel k present.
  const seg_view& pyv = d.pool_y; const seg_view& pdv = d.pool_dp; const seg_view& pav = d.pool_add;
  const int pwin = tid >> 2, ph = tid & 3, pwy = pwin >> 3, pwx = pwin & 7;
  const T* const py_base = reinterpret_cast<const T*>(pyv.ptr) + pyv.coff + n0 + ((int64_t)pyv.oy * pyv.W + pyv.ox) * pyv.cs + ph * 8;
  const T* const pd_base = reinterpret_cast<const T*>(pdv.ptr) + pdv.coff + n0 + ((int64_t)pdv.oy * pdv.W + pdv.ox) * pdv.cs + ph * 8;
  const T* const pa_base = reinterpret_cast<const T*>(pav.ptr) + pav.coff + n0 + ((int64_t)(pav.oy - d.pool_add_y0) * pav.W + (pav.ox - d.pool_add_x0)) * pav.cs + ph * 8;
  const int64_t py_img = (int64_t)pyv.H * pyv.W * pyv.cs, pd_img = (int64_t)pdv.H * pdv.W * pdv.cs, pa_img = (int64_t)pav.H * pav.W * pav.cs;
  auto prefetch_pdz = [&](int b, int oy0, int ox0, bool tile_in, u32x4 (&q)[PDZ ? 9 : 1], uint64_t& m) {
    if constexpr (PDZ) {
      const int Hp = d.Ho >> 1, Wp = d.Wo >> 1;
      const bool has_dp = pdv.ptr != nullptr, has_add = pav.ptr != nullptr;
      const T* yb = py_base + b * py_img + ((int64_t)oy0 * pyv.W + ox0) * pyv.cs;
      const T* db = pd_base + b * pd_img + ((int64_t)(oy0 >> 1) * pdv.W + (ox0 >> 1)) * pdv.cs;
      const T* ab = pa_base + b * pa_img + ((int64_t)oy0 * pav.W + ox0) * pav.cs;
      const int wy = (oy0 >> 1) + pwy, wx = (ox0 >> 1) + pwx;
      const bool route = has_dp && wy < Hp && wx < Wp;
      // NOTE every load below is ONE unconditional asm statement with a selected address.  An asm load under a condition makes
      // its destination a phi for the compiler, which may then implement it as "load into a temporary, copy" -- a copy made
      // before the load has landed (tried: skipping the skip-gradient loads of tiles outside the window gave NaN gradients as
      // soon as a workgroup walked more than one tile).
      m |= (uint64_t)(route ? 1 : 0) << 13;
      gload(q[4], route ? db + ((int64_t)pwy * pdv.W + pwx) * pdv.cs : reinterpret_cast<const T*>(pyv.ptr));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ly = 2 * pwy + (k >> 1), lx = 2 * pwx + (k & 1);
        const int y = oy0 + ly, x = ox0 + lx;
        const bool ok = tile_in || (y < d.Ho && x < d.Wo);
        const int ay = y - d.pool_add_y0, ax = x - d.pool_add_x0;
        const bool oka = ok && has_add && ay >= 0 && ay < d.pool_add_h && ax >= 0 && ax < d.pool_add_w;
        m |= (uint64_t)(ok ? 1 : 0) << (9 + k);
        m |= (uint64_t)(oka ? 1 : 0) << (14 + k);
        gload(q[k], ok ? yb + ((int64_t)ly * pyv.W + lx) * pyv.cs : reinterpret_cast<const T*>(pyv.ptr));
        gload(q[5 + k], oka ? ab + ((int64_t)ly * pav.W + lx) * pav.cs : reinterpret_cast<const T*>(pyv.ptr));
      }
    }
  };
  auto prefetch_im = [&](const TileIt& it, XV (&xr)[IM ? 9 : 1], u32x4 (&rz)[NZP], u32x4 (&q)[PDZ ? 9 : 1], uint64_t& okm, bool& inter) {
    if constexpr (IM) {
      const int b = __builtin_amdgcn_readfirstlane(it.b), ty = __builtin_amdgcn_readfirstlane(it.ty), tx = __builtin_amdgcn_readfirstlane(it.tx);
      const int oy0 = ty * TH, ox0 = tx * TW;
      const int iy0 = oy0 - impad, ix0 = ox0 - impad;
      const float* xb = imx + (((int64_t)b * imH + iy0) * imW + ix0) * ICH;
      const T* zb = z_base + b * z_img + (ty * z_ty + tx * z_tx);
      const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + TH + 2 <= imH && ix0 + TW + 2 <= imW && oy0 + TH <= d.Ho && ox0 + TW <= d.Wo;
      inter = interior;
      if (interior) {
        const uint64_t xbu = uniform64(xb), zbu = PDZ ? 0 : uniform64(zb);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          xload_s<ICH, 0>(xr[u * 3 + 0], x_boff[u], xbu);
          xload_s<ICH, ICH * 4>(xr[u * 3 + 1], x_boff[u], xbu);
          xload_s<ICH, ICH * 8>(xr[u * 3 + 2], x_boff[u], xbu);
        }
        if constexpr (PDZ) {
          uint64_t mq = 0x1ffull;
          prefetch_pdz(b, oy0, ox0, true, q, mq);
          okm = mq;
        } else {
#pragma unroll
          for (int i = 0; i < NZP; ++i) gload_s(rz[i], sz_boff[i], zbu);
          okm = full;
        }
        return;
      }
      uint64_t m = 0;
      const int oy = oy0 + tid / TW, ox = ox0 + tid % TW;
      const bool pix = oy < d.Ho && ox < d.Wo;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy - impad + t / 3, ix = ox - impad + t % 3;
        const bool ok = pix && iy >= 0 && iy < imH && ix >= 0 && ix < imW;
        m |= (uint64_t)(ok ? 1 : 0) << t;
        xload_p<ICH>(xr[t], ok ? imx + (((int64_t)b * imH + iy) * imW + ix) * ICH : imx);
      }
      if constexpr (PDZ) {
        prefetch_pdz(b, oy0, ox0, false, q, m);
      } else {
#pragma unroll
        for (int i = 0; i < NZP; ++i) {
          const int mm = (tid + i * 256) / ZPIECES;
          const bool ok = sz_lds[i] >= 0 && oy0 + mm / TW < d.Ho && ox0 + mm % TW < d.Wo;
          m |= (uint64_t)(ok ? 1 : 0) << (NXL + i);
          gload(rz[i], ok ? zb + sz_off[i] : dzp);
        }
      }
      okm = m;
    }
  };
  // waits until this set's loads have landed; `younger` = the other set's prefetch was issued after them
  auto wait_set = [&](u32x4 (&rp)[NPP], u32x4 (&rz)[NZP], bool younger) {
    if (younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NPP; ++i) asm volatile("" : "+v"(rp[i]));      // the registers are defined "here" for the compiler
#pragma unroll
    for (int i = 0; i < NZP; ++i) asm volatile("" : "+v"(rz[i]));
  };
  // inter (wave-uniform): the tile was interior, every piece this thread owns is valid -- plain stores, no selects
  auto commit_z = [&](const u32x4 (&rz)[NZP], uint64_t okm, bool inter) {
    if (inter) {
#pragma unroll
      for (int i = 0; i < NZP; ++i)
        if (sz_lds[i] >= 0) *reinterpret_cast<u32x4*>(sZ + sz_lds[i]) = rz[i];
      return;
    }
#pragma unroll
    for (int i = 0; i < NZP; ++i)
      if (sz_lds[i] >= 0) *reinterpret_cast<u32x4*>(sZ + sz_lds[i]) = ((okm >> (NXL + i)) & 1) ? rz[i] : u32x4{0, 0, 0, 0};
  };
  auto commit = [&](const u32x4 (&rp)[NPP], const u32x4 (&rz)[NZP], uint64_t okm, bool inter) {
    if (inter) {
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        if (sp_lds[i] >= 0) *reinterpret_cast<u32x4*>(sP + sp_lds[i]) = rp[i];
    } else {
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        if (sp_lds[i] >= 0) *reinterpret_cast<u32x4*>(sP + sp_lds[i]) = ((okm >> i) & 1) ? rp[i] : u32x4{0, 0, 0, 0};
    }
    commit_z(rz, okm, inter);
  };
  auto wait_set_im = [&](XV (&xr)[IM ? 9 : 1], u32x4 (&rz)[NZP], u32x4 (&q)[PDZ ? 9 : 1], bool younger) {
    if (younger) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < (IM ? 9 : 1); ++i) asm volatile("" : "+v"(xr[i]));
    if constexpr (PDZ) {
#pragma unroll
      for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(q[i]));
    } else {
#pragma unroll
      for (int i = 0; i < NZP; ++i) asm volatile("" : "+v"(rz[i]));
    }
  };
  // the four dZ pieces of this thread's window (same arithmetic and rounding as maxpool_bwd_kernel: float sum, one rounding to T)
  auto commit_pdz = [&](const u32x4 (&q)[PDZ ? 9 : 1], uint64_t okm) {
    if constexpr (PDZ) {
      const bool route = (okm >> 13) & 1;
      Vec8<T> yv[4], av[4], dp, o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { yv[k].v = __builtin_bit_cast(bf16x8, q[k]); av[k].v = __builtin_bit_cast(bf16x8, q[5 + k]); }
      dp.v = __builtin_bit_cast(bf16x8, q[4]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float yk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) yk[k] = ((okm >> (9 + k)) & 1) ? yv[k].get(e) : 0.f;
        float m = yk[0]; int mi = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) if (yk[k] > m) { m = yk[k]; mi = k; }
        const float dpe = route ? dp.get(e) : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float g = ((okm >> (14 + k)) & 1) ? av[k].get(e) : 0.f;
          if (route && mi == k) g += dpe;
          o[k].set(e, yk[k] > 0.f ? g : 0.f);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k].store(sZ + ((2 * pwy + (k >> 1)) * TW + 2 * pwx + (k & 1)) * RSZ + ph * 16);
    }
  };
  auto commit_im = [&](const XV (&xr)[IM ? 9 : 1], const u32x4 (&rz)[NZP], const u32x4 (&q)[PDZ ? 9 : 1], uint64_t okm, bool inter) {
    if constexpr (IM) {
      // the pixel's 32-channel row: k = tap * ICH + ci (the HWIO order of the filter gradient), zeros behind 9 * ICH
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        Vec8<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = h * 8 + e;
          float v = 0.f;
          if (k < 9 * ICH) { const int t = k / ICH, ci = k % ICH; v = (inter || ((okm >> t) & 1)) ? __builtin_bit_cast(float, xv_get<ICH>(xr[t], ci)) : 0.f; }
          o.set(e, v);
        }
        o.store(sP + tid * RSP + h * 8 * ES);
      }
      if constexpr (PDZ) commit_pdz(q, okm); else commit_z(rz, okm, inter);
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
      // pixel within a 32-pixel K step that feeds MFMA k-slot (8G + 4e + qr) [bf16] / (8G + e) [f32]; both operands use
      // the same map, so any bijection is valid -- this one keeps a 32-lane half on 8 consecutive pixels
      const int mloc = BF ? (16 * (G >> 1) + 8 * e + 4 * (G & 1) + qr) : (8 * G + e);
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
    if constexpr ((SEG_WABL & 2) != 0) { Frag<T> f; frag_undef(f); return f; }
    else if constexpr (BF) {
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

  // The tile walk is a serial chain (load -> LDS -> MFMA) and a workgroup is often alone on its CU (ksplit targets ~256
  // workgroups), so TWO tiles of global loads are kept in flight in two register sets; the barriers are LDS-only
  // (lgkmcnt + s_barrier): __syncthreads() would also drain vmcnt, i.e. wait for the prefetch just issued.
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto compute = [&]() {
    // Software-pipelined over the flattened (K step, tap) sequence with the ORDER PINNED (sched_barrier per step):
    // the transposed reads of step s+LA are issued before the MFMA(s) of step s.  Left to itself the scheduler
    // batches reads far ahead (204 VGPRs -> 2 waves/SIMD) and still waits on LDS before most MFMAs.
    constexpr int NSTEP = KS * NT;
    // X fragments in flight (steps ahead).  Deeper look-ahead (sized in cycles: 5 steps for the 64 x 64 layout, 12 for 32 x 32)
    // was measured in r02: no change of the tile time on the s_memtime stamps -- the LDS latency is not what the loop waits
    // for -- and in the f32 256-pixel-tile kernel the extra fragments pushed the register allocator past 256 VGPRs, where it
    // parks values in AGPRs: a copy of a staging register made BEFORE the manual vmcnt wait captures a load that has not
    // landed (conv3_2's f32 gradient came out as garbage).  The staging registers must stay where the asm loads put them,
    // so the register budget of these kernels is part of their correctness: keep it small.
    constexpr int LA = 3;
    constexpr int ZLA = NT >= LA ? 1 : LA;                 // dZ fragments in flight (K steps ahead)
    Frag<T> fx[LA + 1][FCI], fz[ZLA + 1][FCO];
    const int rs_off = u0 * PW * RSP;
    auto issue_x = [&](int st1) {
      const int ks1 = st1 / NT, tap1 = st1 % NT;
#pragma unroll
      for (int a = 0; a < FCI; ++a)
        fx[st1 % (LA + 1)][a] = read_frag(sP, pa[ks1], rs_off + ((tap1 / KW) * PW + tap1 % KW) * RSP + a_ch + a * 16 * ES);
    };
    auto issue_z = [&](int ks1) {
#pragma unroll
      for (int c = 0; c < FCO; ++c) fz[ks1 % (ZLA + 1)][c] = read_frag(sZ, za, ks1 * 32 * RSZ + z_ch + c * 16 * ES);
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < ZLA; ++i) if (i < KS) issue_z(i);
#pragma unroll
    for (int i = 0; i < LA; ++i) if (i < NSTEP) issue_x(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int ks = st / NT, tap = st % NT;
      if (st + LA < NSTEP) issue_x(st + LA);
      if (tap == 0 && ks + ZLA < KS) issue_z(ks + ZLA);
      if (!B2 && tap == 0) {
#pragma unroll
        for (int c = 0; c < FCO; ++c) if (!(SEG_WABL & 1)) mma32(accb1[c], ones, fz[ks % (ZLA + 1)][c]);
      }
#pragma unroll
      for (int a = 0; a < FCI; ++a) {
#pragma unroll
        for (int c = 0; c < FCO; ++c) {
          if (!(SEG_WABL & 1)) mma32(acc[tap][a][c], fx[st % (LA + 1)][a], fz[ks % (ZLA + 1)][c]);
          else { frag_keep(fx[st % (LA + 1)][a]); frag_keep(fz[ks % (ZLA + 1)][c]); }       // (the reads stay live)
        }
        if (B2 && !(SEG_WABL & 1)) mma32(accb2[a], fx[st % (LA + 1)][a], ones);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  bool intA = false, intB = false;
  auto prefetchA = [&](const TileIt& it) { if constexpr ((SEG_WABL & 8) != 0) { okA = full; } else if constexpr (IM) prefetch_im(it, xA, rzA, qA, okA, intA); else prefetch(it, rpA, rzA, okA, intA); };
  auto prefetchB = [&](const TileIt& it) { if constexpr ((SEG_WABL & 8) != 0) { okB = full; } else if constexpr (DUAL) { if constexpr (IM) prefetch_im(it, xB, rzB, qB, okB, intB); else prefetch(it, rpB, rzB, okB, intB); } };
  auto waitA = [&](bool younger) { if constexpr (IM) wait_set_im(xA, rzA, qA, younger); else wait_set(rpA, rzA, younger); };
  auto waitB = [&](bool younger) { if constexpr (DUAL) { if constexpr (IM) wait_set_im(xB, rzB, qB, younger); else wait_set(rpB, rzB, younger); } };
  auto commitA = [&]() { if constexpr ((SEG_WABL & 4) != 0) { asm volatile("" :: "v"(rpA[0]), "v"(rzA[0])); } else if constexpr (IM) commit_im(xA, rzA, qA, okA, intA); else commit(rpA, rzA, okA, intA); };
  auto commitB = [&]() { if constexpr ((SEG_WABL & 4) != 0) { asm volatile("" :: "v"(rpB[0]), "v"(rzB[0])); } else if constexpr (DUAL) { if constexpr (IM) commit_im(xB, rzB, qB, okB, intB); else commit(rpB, rzB, okB, intB); } };
  int tile = blockIdx.y;
  const int ks_ = P.ksplit;
#ifdef SEG_STAMPS
  long long* wstp = (g_wstamps && blockIdx.x == 0 && blockIdx.z == 0 && blockIdx.y < 64 && tid == 0) ? g_wstamps + (int64_t)blockIdx.y * 128 : nullptr;
  int it_ = 0;
#endif
  WSTAMP(3, 0);
  TileIt itA = tile_decode(tile < P.ntiles ? tile : 0);
  if (tile < P.ntiles) prefetchA(itA);
  if constexpr (DUAL) {
    const TileIt st2 = tile_decode(2 * ks_);            // each set strides by two splits
    TileIt itB = tile_decode(tile + ks_ < P.ntiles ? tile + ks_ : 0);
    bool youngerB = tile + ks_ < P.ntiles;              // set B was issued after the pending set A
    if (youngerB) prefetchB(itB);
    for (; tile < P.ntiles; tile += 2 * ks_) {
      waitA(youngerB);
      WSTAMP(0, it_);
      lds_barrier();                                     // previous tile's LDS reads are done
      commitA();
      lds_barrier();
      WSTAMP(1, it_);
      const bool moreA = tile + 2 * ks_ < P.ntiles;
      if (moreA) { tile_advance(itA, st2); prefetchA(itA); }
      compute();
      WSTAMP(2, it_);
#ifdef SEG_STAMPS
      ++it_;
#endif
      if (tile + ks_ >= P.ntiles) break;
      waitB(moreA);
      WSTAMP(0, it_);
      lds_barrier();
      commitB();
      lds_barrier();
      WSTAMP(1, it_);
      youngerB = tile + 3 * ks_ < P.ntiles;
      if (youngerB) { tile_advance(itB, st2); prefetchB(itB); }
      compute();
      WSTAMP(2, it_);
#ifdef SEG_STAMPS
      ++it_;
#endif
    }
  } else {
    const TileIt st1 = tile_decode(ks_);
    for (; tile < P.ntiles; tile += ks_) {
      waitA(false);
      lds_barrier();
      commitA();
      lds_barrier();
      if (tile + ks_ < P.ntiles) { tile_advance(itA, st1); prefetchA(itA); }
      compute();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WSTAMP(3, 1);

  // ---- flush: D[row = ci][col = co]; lane: co = lr, ci = 4G + r ----
  // ksplit == 1: straight into the TF-layout gradient; else into this split's slab (plain stores).
  const int kbase = chunk * CIT;                                   // padded concat channel base
  const int k_log_n = d.src0_clog + d.src1_clog;
  float* slab = d.ws + (int64_t)blockIdx.y * P.slab;
#pragma unroll
  for (int a = 0; a < FCI; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kbase + (wci * FCI + a) * 16 + 4 * G + r;
      const int cil = k - (first ? 0 : d.src0.c);                  // padded channel inside its source
      const int kl = first ? (cil < d.src0_clog ? cil : -1) : (cil < d.src1_clog ? d.src0_clog + cil : -1);
#pragma unroll
      for (int c = 0; c < FCO; ++c) {
        const int co = n0 + (wco * FCO + c) * 16 + lr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int tg = u0 * KW + t;
          if (P.direct) {
            if (kl >= 0 && co < d.n_log) d.dw[((int64_t)tg * k_log_n + kl) * d.n_log + co] = acc[t][a][c][r];
          } else {
            slab[((int64_t)tg * P.k_pad + k) * P.n_pad + co] = acc[t][a][c][r];
          }
        }
      }
    }
  float* bslab = slab + (int64_t)NTF * P.k_pad * P.n_pad;
  if (bias1 && G == 0) {
#pragma unroll
    for (int c = 0; c < FCO; ++c) {                                // row 0 of D = column sums of dz
      const int co = n0 + (wco * FCO + c) * 16 + lr;
      if (P.direct) { if (co < d.bias_n) d.db[co] = accb1[c][0]; }
      else bslab[co] = accb1[c][0];
    }
  }
  if (bias2 && lr == 0) {
#pragma unroll
    for (int a = 0; a < FCI; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {                                // column 0 of D = row sums of src
        const int k = kbase + (wci * FCI + a) * 16 + 4 * G + r;
        if (P.direct) { if (k < d.bias_n) d.db[k] = accb2[a][r]; }
        else bslab[k] = accb2[a][r];
      }
  }
}

// Sums the ksplit slabs (fixed association => bitwise reproducible) and scatters to the logical TF layout (RedArgs / RedJob:
// wgrad_common.h).  One thread owns V consecutive n (V = 4: 16-byte coalesced loads, needs n_log % 4 == 0).  SG = 16: a block =
// 16 outputs x 16 split groups, group g sums splits g, g+16, ... (8 loads in flight), partials meet in LDS and
// are added in group order; SG = 1 (few splits): one thread per output.
template <int V>
SEG_DEV void reduce_body(const RedArgs& A, const int sg_log, const int blk, const int nblk, float* red) {
  const float* ws = A.ws; float* dw = A.dw; float* db = A.db;
  const int ksplit = A.ksplit, taps = A.taps, k_pad = A.k_pad, n_pad = A.n_pad, seg0_c = A.seg0_c, seg0_cp = A.seg0_cp, seg1_c = A.seg1_c,
            n_log = A.n_log, bias_mode = A.bias_mode, bias_n = A.bias_n;
  const int64_t slab = A.slab;
  const int k_log = seg0_c + seg1_c;
  const int nq = n_log / V;
  const int64_t nw = (int64_t)taps * k_log * nq;
  const int64_t nb = bias_mode ? (bias_n + V - 1) / V : 0;          // bias handled in units of V as well
  const int64_t total = nw + nb;
  const int SG = 1 << sg_log, OPB = 256 >> sg_log;                  // split groups; outputs (units of V) per block
  const int ol = threadIdx.x & (OPB - 1), g = threadIdx.x >> (8 - sg_log);
  const int rstride = OPB * V + 1;
  for (int64_t i0 = (int64_t)blk * OPB; i0 < total; i0 += (int64_t)nblk * OPB) {
    const int64_t i = i0 + ol;
    const float* src = ws;
    float* dst = nullptr;
    int valid = 0;                                                  // elements of this unit that exist
    if (i < nw) {
      const int q = i % nq; int64_t t = i / nq;
      const int k = t % k_log; const int tap = t / k_log;
      const int kp = k < seg0_c ? k : seg0_cp + (k - seg0_c);
      src = ws + ((int64_t)tap * k_pad + kp) * n_pad + q * V;
      dst = dw + ((int64_t)tap * k_log + k) * n_log + q * V;
      valid = V;
    } else if (i < total) {
      const int j0 = (int)(i - nw) * V;
      src = ws + (int64_t)taps * k_pad * n_pad + j0;
      dst = db + j0;
      valid = bias_n - j0 < V ? bias_n - j0 : V;
    }
    float a[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = 0.f;
    // group g sums splits g, g+SG, ...: 8 predicated loads are issued before any is consumed, so a thread pays one
    // memory latency per 8 splits (SG is chosen so that this is a single batch whenever ksplit <= 256)
    for (int s = g; s < ksplit && valid > 0; s += 8 * SG) {
      float v[8][V];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int sj = s + j * SG;
        const bool on = sj < ksplit;
        const float* p = src + (int64_t)(on ? sj : s) * slab;
        if (V == 4 && valid == 4) {
          f32x4 x = *reinterpret_cast<const f32x4*>(p);
          if (!on) x = f32x4{0.f, 0.f, 0.f, 0.f};
          v[j][0] = x[0]; v[j][1 % V] = x[1]; v[j][2 % V] = x[2]; v[j][3 % V] = x[3];
        } else {
#pragma unroll
          for (int e = 0; e < V; ++e) v[j][e] = (on && e < valid) ? p[e] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += v[j][e];
    }
    if (sg_log > 0) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < V; ++e) red[g * rstride + ol * V + e] = a[e];
      __syncthreads();
      if (g == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          float r = 0.f;
          for (int q = 0; q < SG; ++q) r += red[q * rstride + ol * V + e];
          a[e] = r;
        }
      }
    }
    if (g == 0 && valid > 0) {
      if (V == 4 && valid == 4 && (((uintptr_t)dst) & 15) == 0) *reinterpret_cast<f32x4*>(dst) = f32x4{a[0], a[1 % V], a[2 % V], a[3 % V]};
      else {
#pragma unroll
        for (int e = 0; e < V; ++e) if (e < valid) dst[e] = a[e];
      }
    }
  }
}

constexpr int RED_LDS = 256 * 4 + 64;       // floats: SG rows of (OPB*V + 1)

template <int V>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const RedArgs A, const int sg_log) {
  __shared__ float red[RED_LDS];
  reduce_body<V>(A, sg_log, blockIdx.x, gridDim.x, red);
}

// Every pending slab reduction of a backward segment in one launch.  block -> job: the first_block column is
// staged in LDS with one load per thread (njobs <= 256 per launch), then scanned.
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const RedJob* __restrict__ jobs, const int njobs) {
  __shared__ float red[RED_LDS];
  __shared__ int first[256];
  if ((int)threadIdx.x < njobs) first[threadIdx.x] = jobs[threadIdx.x].first_block;
  __syncthreads();
  int lo = 0;
  for (int j = 1; j < njobs; ++j) if (first[j] <= (int)blockIdx.x) lo = j;
  lo = __builtin_amdgcn_readfirstlane(lo);
  const RedJob& J = jobs[lo];
  const int blk = blockIdx.x - J.first_block, nblk = J.nblocks;
  if (J.flags & 1) reduce_body<4>(J.a, J.flags >> 4, blk, nblk, red);
  else reduce_body<1>(J.a, J.flags >> 4, blk, nblk, red);
}

thread_local char* g_wname_out = nullptr;      // name-query mode: report the kernel instance, launch nothing
thread_local int g_wname_cap = 0;
thread_local int32_t* g_plan_ks = nullptr;     // planning mode: report ksplit / workspace bytes, launch nothing
thread_local int64_t* g_plan_bytes = nullptr;
thread_local RedJob* g_job_out = nullptr;      // job-capture mode: describe the slab reduction, launch nothing

// Workgroups a filter-gradient launch aims for (K splits x channel tiles).  128, half the CUs: in the train step these launches
// share the chip with the data gradients of the critical stream, and every K split costs a partial-sum slab that the reduction
// reads back; stand-alone a layer is fastest near 256, but the step is 4 % (256^2) / 3 % (512^2) faster at 128 and slower
// again at 96 (profiles/r02_wgs_sweep.txt).  SEG_WGRAD_WGS overrides.
inline int wgrad_target_wgs() {
  static const int v = getenv("SEG_WGRAD_WGS") ? atoi(getenv("SEG_WGRAD_WGS")) : 128;
  return v > 0 ? v : 128;
}

template <typename T, int TH, int TW, int KH, int KW, int S, int WCI, int WCO, int FCI, int FCO, int IMC = 0>
int launch_cfg(const WgK& P0, hipStream_t st) {
  constexpr int BN = 16 * FCO * WCO, ES = sizeof(T), CIT = 16 * FCI * WCI;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
  constexpr int PADB = ES == 2 ? 32 : 16;
  constexpr int PATCH_BYTES = ((PH * PW * (CIT * ES + PADB) + 15) / 16) * 16;
  constexpr int LDS = PATCH_BYTES + TH * TW * (BN * ES + PADB);
  WgK P = P0;
  if (P.d.src0.c % CIT || (P.d.src1.c && P.d.src1.c % CIT)) { seg_set_error("wgrad: source channels %d,%d not multiples of the %d-channel tile", P.d.src0.c, P.d.src1.c, CIT); return SEG_ERR_ARG; }
  P.nchunks0 = P.d.src0.c / CIT;
  P.nchunks = P.nchunks0 + P.d.src1.c / CIT;
  P.tiles_x = cdiv(P.d.Wo, TW); P.tiles_y = cdiv(P.d.Ho, TH);
  P.ntiles = P.d.B * P.tiles_x * P.tiles_y;
  P.nblk = cdiv(P.d.dz.c, BN);
  if (P.d.dz.c % BN) { seg_set_error("wgrad: dz channels %d not a multiple of BN %d", P.d.dz.c, BN); return SEG_ERR_ARG; }
  const int base = P.nchunks * P.nblk;
  // Parallelism: (1) deep layers (few pixel tiles, big filters): split the KH filter rows over blockIdx.z -- no
  // partial sums; (2) split K (pixel tiles) until ~256 workgroups exist; every split costs one slab write + read
  // of the workgroup tile, so never more than needed and never more than there are tiles.
  const bool rs = KH > 1 && P.d.bias_mode != 2 && P.ntiles <= 64 && base < 192;
  const int wg = base * (rs ? KH : 1);
  // (the first layer's filter gradient is the LAST kernel of the backward pass: it has the chip to itself and is bound by the
  // latency of its image / activation gathers, so it fills every resident slot instead of sharing with the critical stream)
  const int target_wgs = IMC ? 256 : (P.d.target_wgs > 0 ? P.d.target_wgs : wgrad_target_wgs());
  auto k0 = conv_wgrad_kernel<Tr<T>::DT, TH, TW, KH, KW, S, WCI, WCO, FCI, FCO, 0, IMC>;
  auto k1 = conv_wgrad_kernel<Tr<T>::DT, TH, TW, KH, KW, S, WCI, WCO, FCI, FCO, (KH > 1 ? 1 : 0), IMC>;
  static int occ = 0;                      // resident workgroups per CU of this instance
  if (occ == 0) {
    if (LDS > 48 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k0), 256, LDS) != hipSuccess || nb < 1) nb = 1;
    occ = nb > 4 ? 4 : nb;
  }
  // small filters (<= 64 KB of partial sums per workgroup) are cheap to split: fill every resident slot of the chip
  const int64_t wg_tile_bytes = (int64_t)KH * KW * CIT * BN * 4;
  int ks = P.d.ksplit > 0 ? P.d.ksplit : cdiv(target_wgs, wg);
  if (P.d.ksplit <= 0 && wg_tile_bytes <= 64 * 1024 && P.ntiles / ks > 8) {
    // long serial tile walks (first layer: 31 tiles per workgroup): use every resident slot, keep >= 8 tiles each
    int ks2 = cdiv(target_wgs * occ, wg);
    if (ks2 > P.ntiles / 8) ks2 = P.ntiles / 8;
    if (ks2 > ks) ks = ks2;
  }
  if (ks > P.ntiles) ks = P.ntiles;
  if (ks < 1) ks = 1;
  P.ksplit = ks;
  P.direct = ks == 1;
  P.k_pad = P.nchunks * CIT; P.n_pad = P.d.dz.c;
  const int64_t bias_len = P.k_pad > P.n_pad ? P.k_pad : P.n_pad;
  P.slab = (int64_t)KH * KW * P.k_pad * P.n_pad + bias_len;
  if (g_wname_out) {
    snprintf(g_wname_out, g_wname_cap, "conv_wgrad_kernel<%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d>", ES == 2 ? "bf16" : "f32", TH, TW, KH, KW, S, WCI, WCO, FCI, FCO, rs ? 1 : 0, IMC);
    return SEG_OK;
  }
  if (g_plan_ks) { *g_plan_ks = ks; *g_plan_bytes = P.direct ? 0 : P.slab * ks * 4; return SEG_OK; }
  if (g_job_out && P.direct) { g_job_out->nblocks = 0; return SEG_OK; }
  if (!P.direct && (!P.d.ws || P.d.ws_bytes < P.slab * ks * 4)) { seg_set_error("wgrad: workspace too small (%lld < %lld bytes)", (long long)P.d.ws_bytes, (long long)(P.slab * ks * 4)); return SEG_ERR_ARG; }
  int rc = SEG_OK;
  if (P.d.phase != 2 && !g_job_out) {
    if (rs) SEG_LAUNCH(k1, dim3(base, ks, KH), dim3(256), LDS, st, P);
    else SEG_LAUNCH(k0, dim3(base, ks, 1), dim3(256), LDS, st, P);
    rc = seg_check_launch("conv_wgrad");
  }
  if (rc || P.direct || (P.d.phase == 1 && !g_job_out)) return rc;
  RedArgs RA;
  RA.ws = P.d.ws; RA.dw = P.d.dw; RA.db = P.d.db; RA.slab = P.slab; RA.ksplit = ks; RA.taps = KH * KW; RA.k_pad = P.k_pad; RA.n_pad = P.n_pad;
  RA.seg0_c = P.d.src0_clog; RA.seg0_cp = P.d.src0.c; RA.seg1_c = P.d.src1_clog; RA.n_log = P.d.n_log; RA.bias_mode = P.d.bias_mode; RA.bias_n = P.d.bias_n;
  WgQuery q = WgQuery(); q.job_out = g_job_out;
  return seg_wgrad_reduce_launch(RA, ks, q, st);
}

template <typename T, int KH, int KW, int S>
int launch_k(const WgK& P, hipStream_t st) {
  const seg_wgrad_desc& d = P.d;
  if constexpr (KH == 1 && KW == 1 && S == 1) {
    if (d.im2col_x) {                       // first layer: virtual im2col source, 256-pixel tiles, 32 x 32 channels
      if constexpr (sizeof(T) == 2) {
        if (d.pool_y.ptr) {                 // ... and dZ rebuilt from the max-pool that consumes the layer (IMC bit 2)
          switch (d.im2col_cin) {
            case 1: return launch_cfg<T, 16, 16, 1, 1, 1, 2, 2, 1, 1, 5>(P, st);
            case 2: return launch_cfg<T, 16, 16, 1, 1, 1, 2, 2, 1, 1, 6>(P, st);
            default: return launch_cfg<T, 16, 16, 1, 1, 1, 2, 2, 1, 1, 7>(P, st);
          }
        }
      }
      switch (d.im2col_cin) {
        case 1: return launch_cfg<T, 16, 16, 1, 1, 1, 2, 2, 1, 1, 1>(P, st);
        case 2: return launch_cfg<T, 16, 16, 1, 1, 1, 2, 2, 1, 1, 2>(P, st);
        default: return launch_cfg<T, 16, 16, 1, 1, 1, 2, 2, 1, 1, 3>(P, st);
      }
    }
  }
  int cfg = d.cfg;
  const bool small = (long)cdiv(d.Ho, 8) * 8 * cdiv(d.Wo, 8) * 8 < (long)cdiv(d.Ho, 8) * 8 * cdiv(d.Wo, 16) * 16;
  if (cfg == 0 && sizeof(T) == 2) {
    // bf16: layout AND pixel tile by a cost model calibrated on stand-alone runs of every U-Net layer at 256 and 512
    // (tools/wgrad_micro.py, profiles/r02_wgrad_micro_*.txt).  Layouts: A = 64 ci x 64 co (16 ci x 64 co per wave: 0.36 LDS
    // fragment reads per MFMA, ~750 TF/s when the chip is full), B = 32 ci x 64 co (0.61, ~620), C = 32 x 32 (1.1: LDS-read
    // bound, ~480; ~540 with 256-pixel tiles).  Bigger accumulators mean bigger partial-sum slabs per K split (A: 147 KB per
    // workgroup), so A wins on the big maps (512 x 512: 540-717 TF/s for kernel + reduction) and B / C on the small ones.
    //   cost [us] = 8 + rounds * ceil(tiles / ksplit) * t_tile + (ksplit > 1 ? 3 + ksplit * slab_bytes / 4.5 TB/s : 0)
    const int layout = 2;          // every layout is a candidate (0 was the r01 choice, C only; 1 = B and C)
    // (the layout choice keeps the stand-alone calibration at 256 workgroups: choosing layouts for 128 was measured slower)
    const int cm_wgs = 256;
    const bool ci64 = d.src0.c % 64 == 0 && (!d.src1.ptr || d.src1.c % 64 == 0), co64 = d.dz.c % 64 == 0;
    struct Cand { int cfg, th, tw, cit, bn, occ; double rate; };
    static const Cand cands[] = {{11, 8, 16, 64, 64, 1, 750.}, {14, 8, 8, 64, 64, 1, 750.}, {12, 8, 16, 32, 64, 2, 620.}, {15, 8, 8, 32, 64, 2, 620.},
                                 {3, 8, 16, 32, 32, 2, 480.}, {6, 8, 8, 32, 32, 3, 480.}, {9, 16, 16, 32, 32, 1, 540.}};
    double best = 1e30;
    const int kin = d.src0.c + (d.src1.ptr ? d.src1.c : 0);
    for (const Cand& c : cands) {
      if (c.cit == 64 && (!ci64 || !co64 || S != 1 || layout < 2)) continue;
      if (c.bn == 64 && c.cit == 32 && (!co64 || layout < 1)) continue;
      if (c.cfg == 9 && S != 1) continue;
      const long ntiles = (long)d.B * cdiv(d.Ho, c.th) * cdiv(d.Wo, c.tw);
      const int base = (kin / c.cit) * (d.dz.c / c.bn);
      const bool rs = KH > 1 && d.bias_mode != 2 && ntiles <= 64 && base < 192;
      const int wg = base * (rs ? KH : 1);
      long ks = cdiv(cm_wgs, wg); if (ks > ntiles) ks = ntiles; if (ks < 1) ks = 1;
      const double t_tile = 2.0 * c.th * c.tw * KH * KW * c.cit * c.bn / (rs ? KH : 1) / (c.rate * 1e6 / 256.0);   // us on one CU
      const double rounds = (double)cdiv((int)(wg * ks), 256 * c.occ);
      const double slab = (double)KH * KW * c.cit * c.bn * 4.0 * base;           // bytes per K split (all output tiles)
      const double cost = 8.0 + rounds * (double)((ntiles + ks - 1) / ks) * t_tile + (ks > 1 ? 3.0 + ks * slab / 4.5e6 : 0.0);
      if (cost < best) { best = cost; cfg = c.cfg; }
    }
  } else if (cfg == 0) {
    cfg = (small || S == 2) ? 6 : 3;                                                     // f32 (parity mode): 32 ci x 32 co, 128 / 64 pixels
  }
  // The tile loads of this kernel are inline asm waited for by hand one or two tiles later: a staging register that the
  // register allocator parks somewhere else in between (an AGPR copy, a scratch spill) captures a load that has not landed,
  // and the register itself may be reused and then overwritten when the load does land (garbage gradients / a memory fault;
  // both seen in r02 on instances that needed ~256 architectural VGPRs).  Only instances that stay well below 256 are
  // offered -- tests/test_build.py checks the compiled ISA of every one -- which leaves out the r01 layouts with two X
  // fragments per wave (cfg 1, 2, 4, 5, 8) and, for f32, the 256-pixel tile.
  if (S != 1 && cfg == 9) { seg_set_error("wgrad: 256-pixel tiles are stride-1 only (cfg %d)", cfg); return SEG_ERR_UNSUPPORTED; }
  switch (cfg) {
    case 3: if constexpr (sizeof(T) == 2 || S == 1) return launch_cfg<T, 8, 16, KH, KW, S, 2, 2, 1, 1>(P, st); else break;   // 128 px, 32 ci x 32 co
    case 6: return launch_cfg<T, 8, 8, KH, KW, S, 2, 2, 1, 1>(P, st);    //  64 px
    default: break;
  }
  if constexpr (sizeof(T) == 2) {
    switch (cfg) {      // bf16 only: the 256-pixel tile and the register-reuse layouts (a wave owns 16 ci x (16 FCO) co for all taps)
      case 9: if constexpr (S == 1) return launch_cfg<T, 16, 16, KH, KW, S, 2, 2, 1, 1>(P, st); else break;  // 256 px, 32 ci x 32 co
      case 11: if constexpr (S == 1) return launch_cfg<T, 8, 16, KH, KW, S, 4, 1, 1, 4>(P, st); else break;  // 128 px, 64 ci x 64 co
      case 12: return launch_cfg<T, 8, 16, KH, KW, S, 2, 2, 1, 2>(P, st);  // 128 px, 32 ci x 64 co
      case 14: if constexpr (S == 1) return launch_cfg<T, 8, 8, KH, KW, S, 4, 1, 1, 4>(P, st); else break;   //  64 px, 64 ci x 64 co
      case 15: return launch_cfg<T, 8, 8, KH, KW, S, 2, 2, 1, 2>(P, st);   //  64 px, 32 ci x 64 co
      default: break;
    }
  }
  seg_set_error("wgrad: layout cfg %d is not offered for this kernel / dtype (3, 6; bf16: 9, 11, 12, 14, 15)", cfg); return SEG_ERR_UNSUPPORTED;
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

int seg_wgrad_reduce_launch(const RedArgs& RA, int ks, const WgQuery& q, hipStream_t st) {
  const bool v4 = (RA.n_log % 4 == 0) && (RA.n_pad % 4 == 0);
  const int V = v4 ? 4 : 1;
  const int k_log = RA.seg0_c + RA.seg1_c;
  const int64_t units = (int64_t)RA.taps * k_log * (RA.n_log / V) + (RA.bias_mode ? (RA.bias_n + V - 1) / V : 0);
  // split groups: one 8-deep load batch per thread whenever ksplit <= 256 (sg = ceil(ks/8) rounded up to a power of two)
  int sg_log = 0;
  while ((8 << sg_log) < ks && sg_log < 5) ++sg_log;
  const int opb = 256 >> sg_log;
  int rg = (int)((units + opb - 1) / opb); if (rg > 16384) rg = 16384;
  if (q.job_out) { q.job_out->a = RA; q.job_out->nblocks = rg; q.job_out->flags = (v4 ? 1 : 0) | (sg_log << 4); q.job_out->first_block = 0; q.job_out->pad_ = 0; return SEG_OK; }
  if (v4) SEG_LAUNCH((wgrad_reduce_kernel<4>), dim3(rg), dim3(256), 0, st, RA, sg_log);
  else SEG_LAUNCH((wgrad_reduce_kernel<1>), dim3(rg), dim3(256), 0, st, RA, sg_log);
  return seg_check_launch("wgrad_reduce");
}

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
  if (d.im2col_x) {
    if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad_t || d.pad_l || d.src1.ptr || d.src0.c != 32 || d.im2col_cin < 1 || d.im2col_cin > 3 ||
        d.src0_clog != 9 * d.im2col_cin || d.im2col_pad < 0 || d.im2col_pad > 1 || d.Hi != d.Ho || d.Wi != d.Wo ||
        d.Ho != d.im2col_h + 2 * d.im2col_pad - 2 || d.Wo != d.im2col_w + 2 * d.im2col_pad - 2) {
      seg_set_error("wgrad: inconsistent im2col source (1x1 walk over [B,Ho,Wo,32], 9*cin logical channels, cin 1..3)"); return SEG_ERR_ARG;
    }
  }
  if (d.pool_y.ptr) {
    const seg_view& y = d.pool_y; const seg_view& p = d.pool_dp; const seg_view& a = d.pool_add;
    auto inside = [](const seg_view& v, int H, int W, int C) { return v.oy >= 0 && v.ox >= 0 && v.oy + H <= v.H && v.ox + W <= v.W && v.coff >= 0 && v.coff + C <= v.cs && v.cs % 8 == 0 && v.coff % 8 == 0; };
    if (!d.im2col_x || d.dtype != SEG_BF16 || y.c != d.dz.c || !inside(y, d.Ho, d.Wo, y.c) ||
        (p.ptr && (p.c != y.c || !inside(p, d.Ho / 2, d.Wo / 2, y.c))) ||
        (a.ptr && (a.c != y.c || d.pool_add_h <= 0 || d.pool_add_w <= 0 || d.pool_add_y0 < 0 || d.pool_add_x0 < 0 || !inside(a, d.pool_add_h, d.pool_add_w, y.c))) ||
        (int64_t)y.H * y.W * y.cs >= ((int64_t)1 << 31)) {
      seg_set_error("wgrad: inconsistent pooled-dZ source (first layer, bf16; y [Ho,Wo,n], dpool [Ho/2,Wo/2,n], optional skip gradient window)"); return SEG_ERR_ARG;
    }
  }
  if (d.bias_mode < 0 || d.bias_mode > 2 || (d.bias_mode && (!d.db || d.bias_n <= 0))) { seg_set_error("wgrad: bad bias request"); return SEG_ERR_ARG; }
  if ((d.bias_mode == 1 && d.bias_n > d.dz.c) || (d.bias_mode == 2 && d.bias_n > d.src0.c)) { seg_set_error("wgrad: bias_n exceeds channels"); return SEG_ERR_ARG; }
  if (d.src0.c <= 0 || d.src0.c % 32 || (d.src1.ptr && (d.src1.c <= 0 || d.src1.c % 32)) || d.dz.c <= 0 || d.dz.c % 32) {
    seg_set_error("wgrad: channel counts must be positive multiples of 32"); return SEG_ERR_ARG;
  }
  if (d.src0_clog > d.src0.c || (d.src1.ptr && d.src1_clog > d.src1.c) || d.n_log > d.dz.c) {
    seg_set_error("wgrad: logical channels exceed padded"); return SEG_ERR_ARG;
  }
  if (d.B <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.Hi <= 0 || d.Wi <= 0) { seg_set_error("wgrad: empty extent"); return SEG_ERR_ARG; }
  if ((int64_t)d.src0.H * d.src0.W * d.src0.cs >= ((int64_t)1 << 31) || (int64_t)d.dz.H * d.dz.W * d.dz.cs >= ((int64_t)1 << 31) ||
      (d.src1.ptr && (int64_t)d.src1.H * d.src1.W * d.src1.cs >= ((int64_t)1 << 31))) {
    seg_set_error("wgrad: an image plane of 2^31 elements or more is not supported"); return SEG_ERR_UNSUPPORTED;
  }
  // thin operands (seg_wgrad_desc.thin): 8-channel records read as 32 channels -- what lies beyond channel 8 only reaches
  // filter-gradient entries that are never stored
  auto thin_ok = [](const seg_view& v, int clog) { return v.cs == 8 && v.coff == 0 && v.c == 32 && clog <= 8; };
  const bool ts = (d.thin & 1) != 0, tz = (d.thin & 2) != 0;
  if ((ts && (!thin_ok(d.src0, d.src0_clog) || (d.src1.ptr && !thin_ok(d.src1, d.src1_clog)) || d.im2col_x || d.bias_mode == 2 && d.bias_n > 8)) ||
      (tz && (!thin_ok(d.dz, d.n_log) || d.pool_y.ptr || (d.bias_mode == 1 && d.bias_n > 8)))) {
    seg_set_error("wgrad: a thin operand has cs 8, coff 0, c 32 and <= 8 logical channels"); return SEG_ERR_ARG;
  }
  if (d.src0.oy + d.Hi > d.src0.H || d.src0.ox + d.Wi > d.src0.W || (!ts && d.src0.coff + d.src0.c > d.src0.cs) ||
      (d.src1.ptr && (d.src1.oy + d.Hi > d.src1.H || d.src1.ox + d.Wi > d.src1.W || (!ts && d.src1.coff + d.src1.c > d.src1.cs))) ||
      d.dz.oy + d.Ho > d.dz.H || d.dz.ox + d.Wo > d.dz.W || (!tz && d.dz.coff + d.dz.c > d.dz.cs)) {
    seg_set_error("wgrad: window exceeds its buffer"); return SEG_ERR_ARG;
  }
  // 3x3 layers whose INPUT comes in 32-channel chunks (the U-Net's conv1_2, conv2_1, conv9_1 [32 | 32], conv9_2): the register-staged
  // walk of this file with the 32 ci x 32 co layout -- 16 x 16 pixel tiles on big maps, 8 x 16 on small ones.  The wave-specialised
  // walk owns 16 ci x 16 co per wave there and is bound by its loader waves (one window of x and dZ per 36 MFMAs): measured
  // stand-alone at 512^2 (profiles/r04_wgrad32_micro.txt) conv9_1 205 -> 114 us, conv9_2 / conv1_2 101 -> 62, conv2_1 93 -> 80; at
  // 256^2 with the in-step 64-workgroup target conv2_1 52 -> 38, conv9_1 28 -> 21, the 32 -> 32 layers level.
  seg_wgrad_desc d32;
  const seg_wgrad_desc* dq = dp;
  if (d.dtype == SEG_BF16 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.cfg == 0 && !d.im2col_x && !d.pool_y.ptr && !d.thin &&
      (d.src0.c % 64 != 0 || (d.src1.ptr && d.src1.c % 64 != 0))) {
    d32 = d; d32.cfg = (int64_t)d.Ho * d.Wo >= 160 * 160 ? 9 : 3; dq = &d32;
  }
  const seg_wgrad_desc& de = *dq;
  {
    // bf16 3x3 / stride 1: the wave-specialised walk (wgrad_sweep.hip) unless the caller asks for a layout of this file (cfg 1..99)
    WgQuery q; q.name_out = g_wname_out; q.name_cap = g_wname_cap; q.plan_ks = g_plan_ks; q.plan_bytes = g_plan_bytes; q.job_out = g_job_out;
    int rc = SEG_OK;
    if (seg_wgrad_sweep(de, q, reinterpret_cast<hipStream_t>(stream), &rc)) return rc;
  }
  WgK P;
  P.d = de;
  if (!d.src1.ptr) { P.d.src1 = d.src0; P.d.src1.c = 0; P.d.src1_clog = 0; }
  P.nchunks0 = P.nchunks = 0;                     // set by launch_cfg from its channel tile
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d.dtype == SEG_F32) return launch_t<float>(P, st);
  if (d.dtype == SEG_BF16) return launch_t<bf16_t>(P, st);
  seg_set_error("wgrad: bad dtype %d", d.dtype);
  return SEG_ERR_ARG;
}

/* Batched slab reduction: describe the reductions of n wgrad descriptors (already planned: ws / ksplit set) as
 * 96-byte job records in host memory; the caller copies them to the device once and replays the launch. */
extern "C" int seg_wgrad_reduce_batch_plan(const seg_wgrad_desc* const* descs, int32_t n, void* jobs_host, int64_t cap_bytes,
                                           int32_t* njobs, int32_t* total_blocks) {
  if (!descs || !jobs_host || !njobs || !total_blocks || n < 0) { seg_set_error("reduce_batch_plan: bad arguments"); return SEG_ERR_ARG; }
  RedJob* out = reinterpret_cast<RedJob*>(jobs_host);
  int nj = 0, blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (nj >= 256) { seg_set_error("reduce_batch_plan: more than 256 jobs in one batch"); return SEG_ERR_ARG; }
    if ((int64_t)(nj + 1) * (int64_t)sizeof(RedJob) > cap_bytes) { seg_set_error("reduce_batch_plan: job buffer too small"); return SEG_ERR_ARG; }
    RedJob j = RedJob();
    g_job_out = &j;
    const int rc = seg_conv2d_wgrad(descs[i], nullptr);
    g_job_out = nullptr;
    if (rc) return rc;
    if (j.nblocks <= 0) continue;                 // ksplit == 1: the wgrad stored its result directly
    j.first_block = blocks; blocks += j.nblocks;
    out[nj++] = j;
  }
  *njobs = nj; *total_blocks = blocks;
  return SEG_OK;
}

extern "C" int seg_wgrad_reduce_batch(const void* jobs_dev, int32_t njobs, int32_t total_blocks, void* stream) {
  if (njobs <= 0 || total_blocks <= 0) return SEG_OK;
  if (!jobs_dev) { seg_set_error("reduce_batch: null job table"); return SEG_ERR_ARG; }
  SEG_LAUNCH(wgrad_reduce_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const RedJob*>(jobs_dev), njobs);
  return seg_check_launch("wgrad_reduce_batch");
}

#ifdef SEG_STAMPS
extern "C" int seg_dbg_set_wstamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
