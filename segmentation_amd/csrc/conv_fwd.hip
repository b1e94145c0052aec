// conv_fwd.hip -- NHWC implicit-GEMM convolution on MFMA, LDS-staged input patch + filter rows.
//
// One kernel template serves: 3x3 / 1x1 conv forward (VALID or SAME), its dgrad (a "full"
// correlation of dZ with flipped, transposed filters), the 2x2/s2 transposed conv forward
// (4 independent 1x1 GEMMs with a scatter epilogue) and its dgrad (a 2x2/s2 conv).
// Replaces the TF kernels behind slim.convolution2d / slim.convolution2d_transpose at
// /root/reference/models/unet.py:111-166 and models/fcn.py:110-128,192,195.
//
// GEMM orientation: D[row = out channel][col = pixel] = W[row][k] * X[k][col], so that after the
// MFMA every lane owns 8 consecutive output channels of one pixel (two 16x16 fragments whose rows
// are interleaved by the packed-weight row permutation) and stores them with one 16-byte write.
//
// Workgroup = 256 threads = 4 waves; output tile = TH x TW pixels of one image x BN channels.
// Per 32-channel K chunk the WG stages (a) the (TH-1)S+KH x (TW-1)S+KW input patch, (b) all
// KH*KW filter taps for its BN rows, then runs KH*KW taps x fragments of MFMA out of LDS; the
// next chunk's global loads are in flight (registers) while the current one computes.
#include "common.h"
#include <stdlib.h>

int seg_conv_ring(const seg_conv_desc& d, char* name_out, int name_cap, hipStream_t st, int* rc);    // conv_ring.hip

namespace {

struct ConvK {          // kernel-side copy of the descriptor (trivially copyable)
  seg_conv_desc d;
  int tiles_x, tiles_y; // spatial tiles per image
  int nchunks0, nchunks; // K chunks from src0 / total
  int lin_tr, lin_tc;   // linearised tile (conv_fwd_kernel<..., LIN>): rows x columns of the window, lin_tr * lin_tc <= 128 pixel slots
  int ntw;              // conv_fwd_mt_kernel: consecutive tiles per workgroup
};

// ---- epilogue: bias, ReLU, ReLU-grad mask, store 8 channels per lane ----
// Stamps (s_memtime) showed the naive epilogue at ~5000 of a wave's ~23000 cycles: the bias and mask loads sat on the
// critical path one after the other and every store recomputed 64-bit view offsets.  So: (1) all bias loads are issued
// together (EpiCtx; holding them across the K loop cost 26 VGPRs = one wave per SIMD of occupancy, measured slower),
// (2) all mask / accumulate loads are issued back to back before the first use, (3) addresses are one 64-bit image base +
// 32-bit in-image offsets.
template <int NJ>
struct EpiCtx {
  int co[NJ], ua[NJ], uc[NJ];      // first of the lane's 8 channels (dst-relative), transposed-conv tap
  bool on[NJ];
  float bv[NJ][8];
};

// Bias of the workgroup's BN output channels, fetched once at kernel start into LDS (bias_to_lds) so that the epilogue
// does not begin with a global round trip (every workgroup is a serial chain of such round trips, tools/ablate_conv.py).
template <int BN>
SEG_DEV float bias_fetch(const seg_conv_desc& d, int n0, int tid) {
  float v = 0.f;
  if (tid < BN) {
    const int np = d.n_off + n0 + tid;
    const int bi = d.up2 ? np % d.up_cout : np;
    if (d.bias != nullptr && bi < d.bias_n) v = d.bias[bi];
  }
  return v;
}
template <int BN>
SEG_DEV void bias_to_lds(float v, int tid, float* sb) {          // call AFTER the first chunk's loads have been issued
  if (tid < BN) sb[tid] = v;
}

template <int BN, int WN, int NJ>
SEG_DEV void epi_setup(const seg_conv_desc& d, int n0, int wn, int g, EpiCtx<NJ>& E, const float* sb = nullptr) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int nl = n0 + wn * (BN / WN) + j * 32 + 8 * g;       // launch-local channel
    const int np = d.n_off + nl;                                 // packed row index
    E.on[j] = nl < d.n_count;
    E.co[j] = nl; E.ua[j] = 0; E.uc[j] = 0;
    if (d.up2) { const int tp = np / d.up_cout; E.co[j] = np - tp * d.up_cout; E.ua[j] = tp >> 1; E.uc[j] = tp & 1; }
    if (d.n_store > 0 && E.co[j] >= d.n_store) E.on[j] = false;        // thin destination: the pixel's record is n_store channels wide
    if (sb != nullptr) continue;                               // bias comes from LDS: epi_bias, after the last MFMAs
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int bi = d.up2 ? E.co[j] + e : np + e;
      E.bv[j][e] = (d.bias != nullptr && bi < d.bias_n) ? d.bias[bi] : 0.f;
    }
  }
}

template <int BN, int WN, int NJ>
SEG_DEV void epi_bias(const float* sb, int wn, int g, EpiCtx<NJ>& E) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb + wn * (BN / WN) + j * 32 + 8 * g);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(sb + wn * (BN / WN) + j * 32 + 8 * g + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { E.bv[j][e] = b0[e]; E.bv[j][4 + e] = b1[e]; }
  }
}

// The epilogue in two parts: epi_issue computes the store offsets and ISSUES the ReLU-mask / accumulate loads (a dgrad
// reads the forward activation it masks with); the tiled kernel calls it before the MFMAs of the LAST K chunk, when the
// staging registers of the prefetch are free, so that this round trip overlaps them; conv_epilogue consumes them.
template <typename T, int NJ, int FM>
struct EpiPre {
  int poff_d[FM];                      // in-window pixel offsets (elements) for tap (0,0); -1 = outside the output
  int64_t dbase;
  Vec8<T> mk[NJ][FM], old[NJ][FM];
};

// JSEL >= 0: only channel-fragment pair JSEL is handled (the other one goes to another destination: conv_fwd_kernel<.., SPLIT>)
template <typename T, int TH, int TW, int BN, int WM, int FM, int FN, int JSEL = -1>
SEG_DEV void epi_issue(const seg_conv_desc& d, const EpiCtx<FN / 2>& E, int b, int oy0, int ox0, int wm, int lr, EpiPre<T, FN / 2, FM>& R, int lin_tc = 0, int lin_n = 0) {
  constexpr int BM = TH * TW, NJ = FN / 2;
  const int sc = d.up2 ? 2 : 1;
  const T* mbase = reinterpret_cast<const T*>(d.mask.ptr) + ((int64_t)(b * d.mask.H + d.mask.oy) * d.mask.W + d.mask.ox) * d.mask.cs + d.mask.coff;
  R.dbase = ((int64_t)(b * d.dst.H + d.dst.oy) * d.dst.W + d.dst.ox) * d.dst.cs + d.dst.coff;
  int poff_m[FM];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int m = wm * (BM / WM) + fm * 16 + lr;
    // (lin_tc > 0: the linearised tile -- slot m is pixel (m / lin_tc, m % lin_tc) of the window, slots >= lin_n are padding)
    const int oy = oy0 + (lin_tc > 0 ? m / lin_tc : m / TW), ox = ox0 + (lin_tc > 0 ? m % lin_tc : m % TW);
    const bool ok = oy < d.Ho && ox < d.Wo && (lin_tc == 0 || m < lin_n);
    R.poff_d[fm] = ok ? (sc * oy * d.dst.W + sc * ox) * d.dst.cs : -1;
    poff_m[fm] = (sc * oy * d.mask.W + sc * ox) * d.mask.cs;
  }
  if (d.mask.ptr != nullptr) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
        if ((JSEL < 0 || j == JSEL) && E.on[j] && R.poff_d[fm] >= 0) R.mk[j][fm].load(mbase + poff_m[fm] + (E.ua[j] * d.mask.W + E.uc[j]) * d.mask.cs + E.co[j]);
  }
  if (d.accum != 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
        if ((JSEL < 0 || j == JSEL) && E.on[j] && R.poff_d[fm] >= 0)
          R.old[j][fm].load(reinterpret_cast<const T*>(d.dst.ptr) + R.dbase + R.poff_d[fm] + (E.ua[j] * d.dst.W + E.uc[j]) * d.dst.cs + E.co[j]);
  }
}

template <typename T, int TH, int TW, int BN, int WM, int WN, int FM, int FN, bool POOL = false, int JSEL = -1>
SEG_DEV void conv_epilogue(const seg_conv_desc& d, f32x4 (&acc)[FN][FM], const EpiCtx<FN / 2>& E, const EpiPre<T, FN / 2, FM>& R, int b, int oy0, int ox0, int wm, int lr) {
  constexpr int NJ = FN / 2;
  const int64_t dbase = R.dbase;
  const int (&poff_d)[FM] = R.poff_d;
  const Vec8<T> (&mk)[NJ][FM] = R.mk;
  const Vec8<T> (&old)[NJ][FM] = R.old;
  const bool has_mask = d.mask.ptr != nullptr, has_acc = d.accum != 0;
  const float lo = d.relu ? 0.f : -INFINITY;
  if constexpr (POOL) {
    // Fused 2x2/s2 VALID max-pool of this layer's output (the slim.max_pool2d that consumes it): with the 8x16 tile a wave
    // owns two full tile rows, fragment fm = row 2*wm + fm, lane lr = column -- the 2x2 window is (fm 0, fm 1) x (lane,
    // lane^1).  Pooling the bf16-ROUNDED values gives the bits the separate pool kernel would produce.
    static_assert(TW == 16 && WM == 4 && FM == 2, "pool fusion needs the 8x16 tile");
    T* pbase = reinterpret_cast<T*>(d.pool.ptr) + ((int64_t)(b * d.pool.H + d.pool.oy) * d.pool.W + d.pool.ox) * d.pool.cs + d.pool.coff;
    const int py = (oy0 >> 1) + wm, px = (ox0 + lr) >> 1;
    const bool pst = (lr & 1) == 0 && py < d.pool_h && px < d.pool_w;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!E.on[j]) continue;
      Vec8<T> o[2];
#pragma unroll
      for (int fm = 0; fm < 2; ++fm) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[fm].set(e, fmaxf(acc[2 * j][fm][e] + E.bv[j][e], lo));
          o[fm].set(4 + e, fmaxf(acc[2 * j + 1][fm][e] + E.bv[j][4 + e], lo));
        }
        if (poff_d[fm] >= 0) o[fm].store(reinterpret_cast<T*>(d.dst.ptr) + dbase + poff_d[fm] + E.co[j]);
      }
      Vec8<T> m;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = fmaxf(o[0].get(e), o[1].get(e));
        t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true)));   // lane ^ 1
        m.set(e, t);
      }
      if (pst) m.store(pbase + ((int64_t)py * d.pool.W + px) * d.pool.cs + E.co[j]);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if ((JSEL >= 0 && j != JSEL) || !E.on[j]) continue;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      if (poff_d[fm] < 0) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = fmaxf(acc[2 * j][fm][e] + E.bv[j][e], lo);
        v[4 + e] = fmaxf(acc[2 * j + 1][fm][e] + E.bv[j][4 + e], lo);
      }
      if (has_acc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += old[j][fm].get(e);
      }
      if (has_mask) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mk[j][fm].get(e) > 0.f ? v[e] : 0.f;
      }
      const int64_t doff = dbase + poff_d[fm] + (E.ua[j] * d.dst.W + E.uc[j]) * d.dst.cs + E.co[j];
      if (d.out_f32) {
        Vec8<float> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, v[e]);
        o.store(reinterpret_cast<float*>(d.dst.ptr) + doff);
      } else {
        Vec8<T> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, v[e]);
        o.store(reinterpret_cast<T*>(d.dst.ptr) + doff);
      }
    }
  }
}

template <typename T, int TH, int TW, int BN, int WM, int WN, int FM, int FN, bool POOL = false>
SEG_DEV void conv_epilogue(const seg_conv_desc& d, f32x4 (&acc)[FN][FM], const EpiCtx<FN / 2>& E, int b, int oy0, int ox0, int wm, int lr) {
  EpiPre<T, FN / 2, FM> R;
  epi_issue<T, TH, TW, BN, WM, FM, FN>(d, E, b, oy0, ox0, wm, lr, R);
  conv_epilogue<T, TH, TW, BN, WM, WN, FM, FN, POOL>(d, acc, E, R, b, oy0, ox0, wm, lr);
}

// dgrad of a channel-concat input: one launch, two destinations; a workgroup's BN block lies entirely in one of them
SEG_DEV void select_dst(seg_conv_desc& d, int n0) {
  if (d.n_split > 0 && n0 >= d.n_split) {
    d.dst = d.dst1; d.dst.coff -= d.n_split;
    d.mask = d.mask1; d.mask.coff -= d.n_split;
  }
}

// seg_conv_desc.signal: announce "this launch has started" (hence: everything before it on its stream has finished)
SEG_DEV void conv_signal(const seg_conv_desc& d) {
  if (d.signal != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    __hip_atomic_store(d.signal, d.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifdef SEG_ABLATE
// Debug builds only (SEG_EXTRA_FLAGS=-DSEG_ABLATE, tools/ablate_conv.py): parts of the tiled kernel switched off one at a
// time to see which of them its run time follows.  1 patch loads, 2 filter loads, 4 LDS reads + MFMAs, 8 epilogue,
// 16 LDS commits.  Results are garbage with any bit set.
__device__ int g_ablate = 0;
extern "C" int seg_dbg_set_ablate(int bits) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &bits, sizeof(int)) == hipSuccess ? 0 : -1; }
#define ABL(bit) (abl & (bit))
#else
#define ABL(bit) false
#endif

// ---- split K: `ksplit` workgroups (blockIdx.z) share the K chunks of one output tile -------------------------------------
// Every workgroup stores its partial accumulators (f32, lane-major: one 16-byte store per fragment and lane, 1 KB per wave
// instruction), makes them visible to the other XCDs (agent-scope release: this part has one L2 per XCD) and takes a ticket; the
// LAST arriver re-reads all partials -- its own included -- and adds them in split order, so the sum has ONE association whoever
// does it (bitwise reproducible), then runs the ordinary epilogue.  Returns false for the workgroups that are done.
template <int FN, int FM>
SEG_DEV bool splitk_combine(const seg_conv_desc& d, f32x4 (&acc2)[FN][FM], int tile_lin, int kz, int tid, int nthreads, int* s_flag) {
  constexpr int NA = FN * FM;
#define acc(i) acc2[(i) / FM][(i) % FM]
  typedef unsigned long long u64;
  const int ks = d.ksplit;
  // The partials travel as agent-scope relaxed atomics (sc1: written through / read past the XCD-local L2), so no cache-wide
  // release (buffer_wbl2) or acquire (buffer_inv) is needed: those act on the WHOLE L2 of the XCD and, with 500-1000 workgroups
  // each issuing one beside the filter gradients, made the train step 50 % slower (1.46 against 0.97 ms).
  u64* wsb = reinterpret_cast<u64*>(d.splitk_ws) + (((int64_t)tile_lin * ks + kz) * (int64_t)(NA * nthreads)) * 2;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const f32x4 v = acc(i);
    u64 lo, hi;
    __builtin_memcpy(&lo, reinterpret_cast<const char*>(&v), 8); __builtin_memcpy(&hi, reinterpret_cast<const char*>(&v) + 8, 8);
    u64* q = wsb + ((int64_t)i * nthreads + tid) * 2;
    __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's partials have been performed at agent scope
  __syncthreads();
  if (tid == 0) *s_flag = __hip_atomic_fetch_add(d.splitk_tickets + tile_lin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*s_flag != ks - 1) return false;
  const u64* w0 = reinterpret_cast<const u64*>(d.splitk_ws) + ((int64_t)tile_lin * ks * (int64_t)(NA * nthreads)) * 2;
  for (int z = 0; z < ks; ++z) {
    const u64* wz = w0 + (int64_t)z * (NA * nthreads) * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const u64* q = wz + ((int64_t)i * nthreads + tid) * 2;
      const u64 lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f32x4 v;
      __builtin_memcpy(reinterpret_cast<char*>(&v), &lo, 8); __builtin_memcpy(reinterpret_cast<char*>(&v) + 8, &hi, 8);
      if (z == 0) acc(i) = v; else acc(i) += v;
    }
  }
  if (tid == 0) __hip_atomic_store(d.splitk_tickets + tile_lin, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
#undef acc
  return true;
}

// Occupancy the register allocator must keep (waves per SIMD = workgroups per CU): what the LDS footprint allows for the
// 128-pixel bf16 tiles (3 workgroups of 48 KB with 64 output channels, 4 of 30 KB with 32); the f32 parity mode is left alone.
constexpr int conv_min_waves(int dt, int bm, int bn) { return dt != 1 || bm != 128 ? 1 : (bn == 64 ? 3 : 4); }

// LIN (3x3 / stride 1 only): the 128 pixel slots of the tile are the pixels of a lin_tr x lin_tc window in row-major order, the
// window shape chosen per layer at launch (lin_pick) -- a 10 x 10 map is ONE tile (78 % useful slots) instead of two 8 x 16 tiles
// (39 %), a 26 x 26 map 6 tiles of 9 x 13 instead of 8.  The patch area of LDS is sized for LIN_CAP pixels.
constexpr int LIN_CAP = 224;

// SPLIT (two-destination data gradient of a [32 | 32]-channel concat, n_split = 32 = BN / 2): ONE 64-channel block computes both
// halves from one patch -- channel-fragment pair 0 goes to dst / mask, pair 1 to dst1 / mask1 -- instead of two 32-channel blocks
// that each stage the same patch (conv9_1's data gradient: one K chunk, bandwidth-bound).
template <int DT, int TH, int TW, int BN, int WM, int WN, int KH, int KW, int S, bool POOL = false, bool LIN = false, bool SPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_min_waves(DT, TH * TW, BN))))
void conv_fwd_kernel(const ConvK P) {
#ifdef SEG_ABLATE
  const int abl = __builtin_amdgcn_readfirstlane(g_ablate);
  if (abl & 32) return;                            // launch + dispatch only
#endif
  using T = typename DtSel<DT>::type;
  using TT = Tr<T>;
  constexpr int BM = TH * TW;
  static_assert(!LIN || (S == 1 && KH == 3 && KW == 3 && !POOL && TH * TW == 128), "linearised tiles: 3x3 / s1, 128 slots");
  constexpr int PH = (TH - 1) * S + KH, PW_C = (TW - 1) * S + KW, NPIX = LIN ? LIN_CAP : PH * PW_C;
  constexpr int NT = KH * KW;
  constexpr int PIECES = TT::PIECES, EPP = TT::EPP, RSTR = TT::RSTR;
  constexpr int PATCH_BYTES = ((NPIX * RSTR + 15) / 16) * 16;
  constexpr int NPP = (NPIX * PIECES + 255) / 256;       // patch pieces per thread
  const int PW = LIN ? P.lin_tc + 2 : PW_C;              // patch row length in pixels
  const int npix = LIN ? (P.lin_tr + 2) * PW : NPIX;     // patch pixels in use
  const int lin_tc = LIN ? P.lin_tc : 0, lin_n = LIN ? P.lin_tr * P.lin_tc : 0;
  constexpr int NWP = (NT * BN * PIECES + 255) / 256;    // weight pieces per thread
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(FN % 2 == 0, "channel fragments come in pairs");
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 32) == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;
  char* sW = smem + PATCH_BYTES;
  float* sB = reinterpret_cast<float*>(smem + PATCH_BYTES + NT * BN * RSTR);

  seg_conv_desc d = P.d;
  conv_signal(d);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, g = lane >> 4;

  int t = blockIdx.x;
  const int tx = t % P.tiles_x; t /= P.tiles_x;
  const int ty = t % P.tiles_y; const int b = t / P.tiles_y;
  const int oy0 = ty * (LIN ? P.lin_tr : TH), ox0 = tx * (LIN ? P.lin_tc : TW);
  const int n0 = blockIdx.y * BN;                 // within this launch's n range
  static_assert(!SPLIT || (BN == 64 && WN == 1 && !POOL && !LIN), "split destinations: the 64-channel block with both fragment pairs in every wave");
  if constexpr (!SPLIT) select_dst(d, n0);
  const float bias_r = bias_fetch<BN>(d, n0, tid);

  // ---- per-thread staging descriptors (constant over the K loop) ----
  int p_lds[NPP];
  int p_off0[NPP], p_off1[NPP];                   // element offsets inside image b of each source; -1 = zero fill
  const int64_t img0 = (int64_t)b * d.src0.H * d.src0.W * d.src0.cs;
  const int64_t img1 = (int64_t)b * d.src1.H * d.src1.W * d.src1.cs;
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int idx = tid + i * 256;
    p_lds[i] = -1; p_off0[i] = -1; p_off1[i] = -1;
    if (idx < npix * PIECES) {
      const int q = idx / PIECES, h = idx % PIECES;
      const int py = q / PW, px = q % PW;
      const int iy = oy0 * S - d.pad_t + py, ix = ox0 * S - d.pad_l + px;
      p_lds[i] = TT::lds_off(q, h);
      if (iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi) {
        p_off0[i] = ((iy + d.src0.oy) * d.src0.W + ix + d.src0.ox) * d.src0.cs + d.src0.coff + h * EPP;
        p_off1[i] = ((iy + d.src1.oy) * d.src1.W + ix + d.src1.ox) * d.src1.cs + d.src1.coff + h * EPP;
      }
    }
  }
  const T* src0 = reinterpret_cast<const T*>(d.src0.ptr) + img0;
  const T* src1 = reinterpret_cast<const T*>(d.src1.ptr) + img1;
  const T* wp = reinterpret_cast<const T*>(d.w_packed);

  u32x4 rp[NPP], rw[NWP];

  auto prefetch = [&](int c) {
    const bool first = c < P.nchunks0;
    const T* sb = first ? src0 + c * 32 : src1 + (c - P.nchunks0) * 32;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int off = p_off1[i] ^ ((p_off0[i] ^ p_off1[i]) & -(int)first);     // (a select of two registers; hipcc turned `?:` into a scratch table)
      rp[i] = u32x4{0, 0, 0, 0};
      if (off >= 0 && !ABL(1)) rp[i] = *reinterpret_cast<const u32x4*>(sb + off);
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int idx = tid + i * 256;
      if (NT * BN * PIECES % 256 == 0 || idx < NT * BN * PIECES) {
        const int tap = idx / (BN * PIECES), row = (idx / PIECES) % BN, h = idx % PIECES;
        const int64_t off = ((int64_t)(tap * P.nchunks + c) * d.n_total + d.n_off + n0 + row) * 32 + h * EPP;
        if (ABL(2)) { rw[i] = u32x4{1, 2, 3, 4}; continue; }
        rw[i] = *reinterpret_cast<const u32x4*>(wp + off);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < NPP; ++i)
      if (p_lds[i] >= 0) *reinterpret_cast<u32x4*>(sP + p_lds[i]) = rp[i];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int idx = tid + i * 256;
      if (NT * BN * PIECES % 256 == 0 || idx < NT * BN * PIECES) {
        const int tap = idx / (BN * PIECES), row = (idx / PIECES) % BN, h = idx % PIECES;
        *reinterpret_cast<u32x4*>(sW + TT::lds_off(tap * BN + row, h)) = rw[i];
      }
    }
  };

  // split K (d.ksplit > 1): this workgroup walks chunks [c0, c1) of the tile; see splitk_combine
  const int ksn = d.ksplit > 1 ? d.ksplit : 1, kz = d.ksplit > 1 ? (int)blockIdx.z : 0;
  const int c0 = (int)((long)P.nchunks * kz / ksn), c1 = (int)((long)P.nchunks * (kz + 1) / ksn);
  prefetch(c0);                                   // first: everything below runs in the shadow of this round trip
  bias_to_lds<BN>(bias_r, tid, sB);               // visible after the K loop's barriers

  // ---- per-lane operand addresses ----
  int a_addr[FN];          // weight rows: packed order => fragment fn reads rows fn*16 + lr
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) a_addr[fn] = frag_addr<T>(wn * (BN / WN) + fn * 16 + lr, g);
  int b_addr[NT][FM];      // pixel rows per tap
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int m = wm * (BM / WM) + fm * 16 + lr;
    int py = (m / TW) * S, px = (m % TW) * S;
    if constexpr (LIN) { const int mm = m < lin_n ? m : 0; py = mm / lin_tc; px = mm % lin_tc; }     // (padding slots read pixel 0: finite, never stored)
#pragma unroll
    for (int u = 0; u < KH; ++u)
#pragma unroll
      for (int v = 0; v < KW; ++v) b_addr[u * KW + v][fm] = frag_addr<T>((py + u) * PW + px + v, g);
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0, 0, 0, 0};

#ifdef SEG_ABLATE
  if (abl & 64) {                                  // prologue only (the address tables are kept alive by a never-taken store)
    int sum = 0;
#pragma unroll
    for (int i = 0; i < NPP; ++i) sum += p_lds[i] + p_off0[i] + p_off1[i];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) sum += a_addr[fn];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int tp = 0; tp < NT; ++tp) sum += b_addr[tp][fm];
    if (sum == 0x7fffffff) *reinterpret_cast<int*>(smem) = sum;
    return;
  }
#endif
  auto compute = [&]() {
    // fragments of tap t+1 are read from LDS before the MFMAs of tap t (explicit software pipeline)
    Frag<T> fa[2][FN], fb[2][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) fa[0][fn] = lds_read_frag_at<T>(sW + a_addr[fn]);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) fb[0][fm] = lds_read_frag_at<T>(sP + b_addr[0][fm]);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      if (tap + 1 < NT) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) fa[(tap + 1) & 1][fn] = lds_read_frag_at<T>(sW + a_addr[fn] + (tap + 1) * BN * RSTR);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) fb[(tap + 1) & 1][fm] = lds_read_frag_at<T>(sP + b_addr[tap + 1 < NT ? tap + 1 : 0][fm]);
      }
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) mma32(acc[fn][fm], fa[tap & 1][fn], fb[tap & 1][fm]);
    }
  };

  for (int c = c0; c + 1 < c1; ++c) {
    __syncthreads();
    if (!ABL(16)) commit();
    __syncthreads();
    prefetch(c + 1);
    if (!ABL(4)) compute();
  }
  // last chunk: nothing left to prefetch, so the epilogue's mask / accumulate loads fly during its commit and its MFMAs instead.
  // (Issued BEFORE the commit waits for the chunk's data: a layer with a single K chunk -- the 32-channel data gradients of the
  // full-resolution maps, bandwidth-bound -- then has its mask round trip in flight together with the patch, not behind it.)
  EpiCtx<FN / 2> epi;
  epi_setup<BN, WN, FN / 2>(d, n0, wn, g, epi, sB);
  EpiPre<T, FN / 2, FM> pre;
  if constexpr (SPLIT) {
    seg_conv_desc d1 = d;                            // the second destination, addressed like select_dst does for whole blocks
    d1.dst = d.dst1; d1.dst.coff -= d.n_split;
    d1.mask = d.mask1; d1.mask.coff -= d.n_split;
    EpiPre<T, FN / 2, FM> pre1;
    epi_issue<T, TH, TW, BN, WM, FM, FN, 0>(d, epi, b, oy0, ox0, wm, lr, pre);
    epi_issue<T, TH, TW, BN, WM, FM, FN, 1>(d1, epi, b, oy0, ox0, wm, lr, pre1);
    __syncthreads();
    commit();
    __syncthreads();
    compute();
    if (ksn > 1) {
      int* s_flag = reinterpret_cast<int*>(sP);
      if (!splitk_combine<FN, FM>(d, acc, (int)(blockIdx.x + gridDim.x * blockIdx.y), kz, tid, 256, s_flag)) return;
    }
    epi_bias<BN, WN, FN / 2>(sB, wn, g, epi);
    conv_epilogue<T, TH, TW, BN, WM, WN, FM, FN, false, 0>(d, acc, epi, pre, b, oy0, ox0, wm, lr);
    conv_epilogue<T, TH, TW, BN, WM, WN, FM, FN, false, 1>(d1, acc, epi, pre1, b, oy0, ox0, wm, lr);
    return;
  }
  constexpr bool EARLY = !LIN && sizeof(T) == 2;    // (the linearised and the f32 instances have no registers to spare across the commit: 264 VGPRs)
  if constexpr (EARLY) epi_issue<T, TH, TW, BN, WM, FM, FN>(d, epi, b, oy0, ox0, wm, lr, pre, lin_tc, lin_n);
  __syncthreads();
  if (!ABL(16)) commit();
  __syncthreads();
  if constexpr (!EARLY) epi_issue<T, TH, TW, BN, WM, FM, FN>(d, epi, b, oy0, ox0, wm, lr, pre, lin_tc, lin_n);
  if (!ABL(4)) compute();
  if (ksn > 1) {
    int* s_flag = reinterpret_cast<int*>(sP);       // (the patch is dead: every wave is behind its last LDS read after the barrier inside)
    if (!splitk_combine<FN, FM>(d, acc, (int)(blockIdx.x + gridDim.x * blockIdx.y), kz, tid, 256, s_flag)) return;
  }
  epi_bias<BN, WN, FN / 2>(sB, wn, g, epi);
  if (ABL(8) && acc[0][0][0] != 12345.f) return;
  conv_epilogue<T, TH, TW, BN, WM, WN, FM, FN, POOL>(d, acc, epi, pre, b, oy0, ox0, wm, lr);
}

// ---------------------------------------------------------------------------------------------------------
// Multi-tile form of the 8 x 16 tiled kernel for layers with ONE K chunk (32 input channels: the full-resolution maps, 3x3 / stride 1):
// a workgroup walks P.ntw consecutive tiles; the filter rows are staged once, and the NEXT tile's patch is requested before the MFMAs
// of the current tile, so that it is in flight during this tile's MFMAs and epilogue (mask loads, stores).  These layers are bound by
// how many bytes a CU keeps in flight (four workgroups x one patch each), not by the MFMAs.  No split K, no fused pool.
// ---------------------------------------------------------------------------------------------------------
template <int DT, int BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_min_waves(DT, 128, BN))))
void conv_fwd_mt_kernel(const ConvK P) {
  using T = typename DtSel<DT>::type;
  using TT = Tr<T>;
  constexpr int TH = 8, TW = 16, WM = 4, WN = 1, KH = 3, KW = 3, BM = TH * TW;
  constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW, NT = 9;
  constexpr int PIECES = TT::PIECES, EPP = TT::EPP, RSTR = TT::RSTR;
  constexpr int PATCH_BYTES = ((NPIX * RSTR + 15) / 16) * 16;
  constexpr int NPP = (NPIX * PIECES + 255) / 256;
  constexpr int NWP = (NT * BN * PIECES + 255) / 256;
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;
  char* sW = smem + PATCH_BYTES;
  float* sB = reinterpret_cast<float*>(smem + PATCH_BYTES + NT * BN * RSTR);

  seg_conv_desc d = P.d;
  conv_signal(d);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave, wn = 0;
  const int lr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * BN;
  select_dst(d, n0);
  const float bias_r = bias_fetch<BN>(d, n0, tid);
  const int ntiles = d.B * P.tiles_y * P.tiles_x;
  const int t_first = blockIdx.x * P.ntw;
  const int t_last = t_first + P.ntw < ntiles ? t_first + P.ntw : ntiles;

  int p_lds[NPP], p_off[NPP];                     // staging descriptors of the tile being REQUESTED (p_off: -1 = zero fill)
  const T* src = nullptr;
  auto tile_coords = [&](int t, int& b, int& oy0, int& ox0) {
    const int tx = t % P.tiles_x; t /= P.tiles_x;
    const int ty = t % P.tiles_y; b = t / P.tiles_y;
    oy0 = ty * TH; ox0 = tx * TW;
  };
  u32x4 rp[NPP];
  auto request = [&](int t) {                      // descriptors + loads of tile t's patch
    int b, oy0, ox0;
    tile_coords(t, b, oy0, ox0);
    src = reinterpret_cast<const T*>(d.src0.ptr) + (int64_t)b * d.src0.H * d.src0.W * d.src0.cs;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int idx = tid + i * 256;
      p_off[i] = -1;
      rp[i] = u32x4{0, 0, 0, 0};
      if (idx < NPIX * PIECES) {
        const int q = idx / PIECES, h = idx % PIECES;
        const int py = q / PW, px = q % PW;
        const int iy = oy0 - d.pad_t + py, ix = ox0 - d.pad_l + px;
        if (iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi) {
          p_off[i] = ((iy + d.src0.oy) * d.src0.W + ix + d.src0.ox) * d.src0.cs + d.src0.coff + h * EPP;
          rp[i] = *reinterpret_cast<const u32x4*>(src + p_off[i]);
        }
      }
    }
  };
#pragma unroll
  for (int i = 0; i < NPP; ++i) { const int idx = tid + i * 256; p_lds[i] = idx < NPIX * PIECES ? TT::lds_off(idx / PIECES, idx % PIECES) : -1; }
  request(t_first);
  {
    // the filter rows of the single chunk: staged once
    const T* wp = reinterpret_cast<const T*>(d.w_packed);
    u32x4 rw[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int idx = tid + i * 256;
      if (NT * BN * PIECES % 256 == 0 || idx < NT * BN * PIECES) {
        const int tap = idx / (BN * PIECES), row = (idx / PIECES) % BN, h = idx % PIECES;
        rw[i] = *reinterpret_cast<const u32x4*>(wp + ((int64_t)tap * d.n_total + d.n_off + n0 + row) * 32 + h * EPP);
      }
    }
    bias_to_lds<BN>(bias_r, tid, sB);
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int idx = tid + i * 256;
      if (NT * BN * PIECES % 256 == 0 || idx < NT * BN * PIECES) {
        const int tap = idx / (BN * PIECES), row = (idx / PIECES) % BN, h = idx % PIECES;
        *reinterpret_cast<u32x4*>(sW + TT::lds_off(tap * BN + row, h)) = rw[i];
      }
    }
  }
  int a_addr[FN];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) a_addr[fn] = frag_addr<T>(wn * (BN / WN) + fn * 16 + lr, g);
  int b_addr[NT][FM];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int m = wm * (BM / WM) + fm * 16 + lr;
    const int py = m / TW, px = m % TW;
#pragma unroll
    for (int u = 0; u < KH; ++u)
#pragma unroll
      for (int v = 0; v < KW; ++v) b_addr[u * KW + v][fm] = frag_addr<T>((py + u) * PW + px + v, g);
  }
  for (int t = t_first; t < t_last; ++t) {
    int b, oy0, ox0;
    tile_coords(t, b, oy0, ox0);
    __syncthreads();                               // every wave is behind its LDS reads of the previous tile
#pragma unroll
    for (int i = 0; i < NPP; ++i)
      if (p_lds[i] >= 0) *reinterpret_cast<u32x4*>(sP + p_lds[i]) = rp[i];
    __syncthreads();
    EpiCtx<FN / 2> epi;
    epi_setup<BN, WN, FN / 2>(d, n0, wn, g, epi, sB);
    EpiPre<T, FN / 2, FM> pre;
    epi_issue<T, TH, TW, BN, WM, FM, FN>(d, epi, b, oy0, ox0, wm, lr, pre);
    if (t + 1 < t_last) request(t + 1);            // in flight during this tile's MFMAs and stores
    f32x4 acc[FN][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0, 0, 0, 0};
    {
      Frag<T> fa[2][FN], fb[2][FM];
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) fa[0][fn] = lds_read_frag_at<T>(sW + a_addr[fn]);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) fb[0][fm] = lds_read_frag_at<T>(sP + b_addr[0][fm]);
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        if (tap + 1 < NT) {
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) fa[(tap + 1) & 1][fn] = lds_read_frag_at<T>(sW + a_addr[fn] + (tap + 1) * BN * RSTR);
#pragma unroll
          for (int fm = 0; fm < FM; ++fm) fb[(tap + 1) & 1][fm] = lds_read_frag_at<T>(sP + b_addr[tap + 1 < NT ? tap + 1 : 0][fm]);
        }
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
          for (int fm = 0; fm < FM; ++fm) mma32(acc[fn][fm], fa[tap & 1][fn], fb[tap & 1][fm]);
      }
    }
    epi_bias<BN, WN, FN / 2>(sB, wn, g, epi);
    conv_epilogue<T, TH, TW, BN, WM, WN, FM, FN>(d, acc, epi, pre, b, oy0, ox0, wm, lr);
  }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 fast path: same tiling, but the patch and filter rows go global -> LDS directly (global_load_lds_dwordx4,
// no staging VGPRs, no ds_write), into TWO LDS buffers: chunk c+1 is in flight while chunk c feeds the MFMAs, one
// barrier per chunk.  The LDS image of a load is lane-linear (wave base + lane*16), so the bank swizzle of
// Tr<bf16>::lds_off is applied to the SOURCE address of each lane; out-of-image pixels read a 16-byte zero word.
// ---------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint32_t g_zero16[4] = {0, 0, 0, 0};

// (Round 4 tried the fills as inline asm: with the builtin hipcc waits lgkmcnt(0) in front of every use of a ds_read result once an
// LDS-DMA is pending -- 40 full waits and not one counted wait in this kernel's .s -- and the asm form restores lgkmcnt(1) / (2).  In
// the C2 train step it measured 1.5 % SLOWER (0.981 against 0.967 ms, three interleaved rounds, profiles/r04_ab_glds_asm_256.txt):
// the asm statements pin the fills in front of the MFMAs, where the builtin lets the scheduler spread them.)
SEG_DEV void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// NBUF = 2: double-buffered (chunk c+1 in flight during chunk c, ~1 workgroup/CU); NBUF = 1: single LDS buffer, loads are
// not overlapped inside a workgroup but 4-5 small workgroups per CU overlap each other (no staging VGPRs -> 4+ waves/SIMD).
// (Deeper rings and a weight-stationary persistent form were measured slower in the overlapped step and removed: DESIGN.md.)
template <int TH, int TW, int BN, int WM, int WN, int KH, int KW, int S, int NBUF>
__global__ __launch_bounds__(64 * WM * WN) void conv_fwd_glds_kernel(const ConvK P) {
  using T = bf16_t;
  using TT = Tr<T>;
  constexpr int BM = TH * TW;
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, NPIX = PH * PW;
  constexpr int NT = KH * KW;
  constexpr int RSTR = TT::RSTR;
  constexpr int PINST = (NPIX * 4 + 63) / 64;             // wave-instructions (1 KiB each) for the patch
  constexpr int WINST = NT * BN * 4 / 64;                 // ... for the filter rows
  constexpr int NW = WM * WN;                                 // waves per workgroup: 4, or 8 for the 512-pixel tile
  constexpr int PPW = (PINST + NW - 1) / NW, WPW = (WINST + NW - 1) / NW;   // per wave
  constexpr int PATCH_BYTES = PINST * 1024;
  constexpr int BUF = PATCH_BYTES + WINST * 1024;
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  static_assert((NW == 4 || NW == 8) && FN % 2 == 0, "wave layout");
  static_assert((NT * BN * 4) % 64 == 0, "filter rows fill whole wave-instructions");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  seg_conv_desc d = P.d;
  conv_signal(d);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, g = lane >> 4;

  int t = blockIdx.x;
  const int tx = t % P.tiles_x; t /= P.tiles_x;
  const int ty = t % P.tiles_y; const int b = t / P.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int n0 = blockIdx.y * BN;
  select_dst(d, n0);
  constexpr bool SBIAS = NBUF * BUF + BN * 4 <= 160 * 1024;      // (the deepest rings fill the LDS: bias from global there)
  float* sB = SBIAS ? reinterpret_cast<float*>(smem + NBUF * BUF) : nullptr;
  float bias_r = 0.f;
  if (SBIAS) bias_r = bias_fetch<BN>(d, n0, tid);

  // ---- per-lane source descriptors: instruction i of this wave covers LDS pieces [(i*4+wave)*64, +64) ----
  int p_off0[PPW], p_off1[PPW];                 // element offsets inside image b; -1 = zero word
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = (i * NW + wave) * 64 + lane;
    const int q = piece >> 2, h = (piece & 3) ^ ((q >> 1) & 2);          // un-swizzle: which 8 channels land here
    p_off0[i] = -1; p_off1[i] = -1;
    if (q < NPIX) {
      const int py = q / PW, px = q % PW;
      const int iy = oy0 * S - d.pad_t + py, ix = ox0 * S - d.pad_l + px;
      if (iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi) {
        p_off0[i] = ((iy + d.src0.oy) * d.src0.W + ix + d.src0.ox) * d.src0.cs + d.src0.coff + h * 8;
        p_off1[i] = ((iy + d.src1.oy) * d.src1.W + ix + d.src1.ox) * d.src1.cs + d.src1.coff + h * 8;
      }
    }
  }
  int w_off[WPW];                               // element offset inside one (tap-major) chunk slab of the packed filters
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int piece = (i * NW + wave) * 64 + lane;
    const int rowg = piece >> 2, h = (piece & 3) ^ ((rowg >> 1) & 2);    // rowg = tap*BN + row
    const int tap = rowg / BN, row = rowg % BN;
    w_off[i] = (i * NW + wave < WINST) ? ((tap * P.nchunks) * d.n_total + d.n_off + n0 + row) * 32 + h * 8 : -1;
  }
  const T* src0 = reinterpret_cast<const T*>(d.src0.ptr) + (int64_t)b * d.src0.H * d.src0.W * d.src0.cs;
  const T* src1 = reinterpret_cast<const T*>(d.src1.ptr) + (int64_t)b * d.src1.H * d.src1.W * d.src1.cs;
  const T* wp = reinterpret_cast<const T*>(d.w_packed);

  auto issue = [&](int c, char* buf) {
    const bool first = c < P.nchunks0;
    const T* sb = first ? src0 + c * 32 : src1 + (c - P.nchunks0) * 32;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (i * NW + wave < PINST) {
        const int off = p_off1[i] ^ ((p_off0[i] ^ p_off1[i]) & -(int)first);   // (a select of two registers; hipcc turned `?:` into a scratch table)
        const void* gp = off >= 0 ? (const void*)(sb + off) : (const void*)g_zero16;
        glds16(gp, buf + (i * NW + wave) * 1024);
      }
    }
    const T* wc = wp + (int64_t)c * d.n_total * 32;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      if (i * NW + wave < WINST) glds16(wc + w_off[i], buf + PATCH_BYTES + (i * NW + wave) * 1024);
    }
  };

  int a_addr[FN];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn) a_addr[fn] = PATCH_BYTES + frag_addr<T>(wn * (BN / WN) + fn * 16 + lr, g);
  int b_addr[NT][FM];
#pragma unroll
  for (int fm = 0; fm < FM; ++fm) {
    const int m = wm * (BM / WM) + fm * 16 + lr;
    const int py = (m / TW) * S, px = (m % TW) * S;
#pragma unroll
    for (int u = 0; u < KH; ++u)
#pragma unroll
      for (int v = 0; v < KW; ++v) b_addr[u * KW + v][fm] = frag_addr<T>((py + u) * PW + px + v, g);
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int fn = 0; fn < FN; ++fn)
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0, 0, 0, 0};

  static_assert(NBUF == 1 || NBUF == 2, "single- or double-buffered");
  constexpr int PD = NBUF - 1;                             // chunks in flight ahead of the one being computed
  // split K (d.ksplit > 1): this workgroup walks chunks [c0, c1) of the tile; see splitk_combine
  const int ksn = d.ksplit > 1 ? d.ksplit : 1, kz = d.ksplit > 1 ? (int)blockIdx.z : 0;
  const int c0 = (int)((long)P.nchunks * kz / ksn), c1 = (int)((long)P.nchunks * (kz + 1) / ksn);
  if (NBUF >= 2) {
#pragma unroll
    for (int i = 0; i < PD; ++i) if (c0 + i < c1) issue(c0 + i, smem + i * BUF);
  }
  if (SBIAS && NBUF >= 2) bias_to_lds<BN>(bias_r, tid, sB);   // after the first fills were issued; visible after the K loop's barriers
  int slot = 0;                                            // c % NBUF
  for (int c = c0; c < c1; ++c) {
    char* cur = smem + (NBUF >= 2 ? slot * BUF : 0);
    if (NBUF == 1) {
      if (c > c0) __syncthreads();                         // everyone finished reading the previous chunk
      issue(c, smem);
      if (SBIAS && c == c0) bias_to_lds<BN>(bias_r, tid, sB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of chunk c has landed
    __syncthreads();                                      // ... everyone's has, and nobody still reads the stage refilled next
    if (NBUF >= 2 && c + PD < c1) {
      const int ns = slot == 0 ? NBUF - 1 : slot - 1;      // (c + PD) % NBUF == (c - 1) % NBUF
      issue(c + PD, smem + ns * BUF);
    }
    slot = slot + 1 == NBUF ? 0 : slot + 1;
    // fragments of tap t+1 are read from LDS before the MFMAs of tap t (explicit software pipeline)
    Frag<T> fa[2][FN], fb[2][FM];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) fa[0][fn] = lds_read_frag_at<T>(cur + a_addr[fn]);
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) fb[0][fm] = lds_read_frag_at<T>(cur + b_addr[0][fm]);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      if (tap + 1 < NT) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) fa[(tap + 1) & 1][fn] = lds_read_frag_at<T>(cur + a_addr[fn] + (tap + 1) * BN * RSTR);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) fb[(tap + 1) & 1][fm] = lds_read_frag_at<T>(cur + b_addr[tap + 1 < NT ? tap + 1 : 0][fm]);
      }
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) mma32(acc[fn][fm], fa[tap & 1][fn], fb[tap & 1][fm]);
    }
  }
  if (ksn > 1) {
    __syncthreads();                                 // every wave is behind its last LDS read: the first stage's bytes are free
    int* s_flag = reinterpret_cast<int*>(smem);
    if (!splitk_combine<FN, FM>(d, acc, (int)(blockIdx.x + gridDim.x * blockIdx.y), kz, tid, 64 * WM * WN, s_flag)) return;
  }
  EpiCtx<FN / 2> epi;                              // all bias loads issued together (one latency), then the mask loads
  epi_setup<BN, WN, FN / 2>(d, n0, wn, g, epi, sB);
  if (SBIAS) epi_bias<BN, WN, FN / 2>(sB, wn, g, epi);
  conv_epilogue<T, TH, TW, BN, WM, WN, FM, FN>(d, acc, epi, b, oy0, ox0, wm, lr);
}

thread_local char* g_name_out = nullptr;   // when set, launches are dry: only the kernel name is reported
thread_local int g_name_cap = 0;
struct TilePlan { int bm, bn; long wgs; };
thread_local TilePlan* g_plan_out = nullptr;   // when set, launches are dry: the tile shape and grid are reported (split-K planning)

// grid.z and argument checks of a split-K launch (seg_conv_desc.ksplit > 1)
inline int splitk_grid(const ConvK& P, int* gz) {
  *gz = 1;
  if (P.d.ksplit <= 1) return SEG_OK;
  if (!P.d.splitk_ws || !P.d.splitk_tickets) { seg_set_error("conv: ksplit %d without workspace / tickets", P.d.ksplit); return SEG_ERR_ARG; }
  if (P.d.ksplit > P.nchunks || P.d.ksplit > 64) { seg_set_error("conv: ksplit %d exceeds the layer's %d K chunks", P.d.ksplit, P.nchunks); return SEG_ERR_ARG; }
  *gz = P.d.ksplit;
  return SEG_OK;
}

// Window of a linearised tile for an Ho x Wo map: nx column strips of tc = ceil(Wo / nx) pixels, tr rows with tr * tc <= 128 slots
// and (tr + 2) * (tc + 2) <= LIN_CAP patch pixels, rows balanced over the strips; fewest tiles wins, then the smaller halo.
inline long lin_pick(int Ho, int Wo, int* tr_out, int* tc_out) {
  long best = -1, best_halo = 0;
  for (int nx = 1; nx <= 16; ++nx) {
    const int tc = cdiv(Wo, nx);
    if (tc > 128) continue;
    int tr = 128 / tc; if (tr > Ho) tr = Ho;
    while (tr >= 1 && (tr + 2) * (tc + 2) > LIN_CAP) --tr;
    if (tr < 1) continue;
    const int ny = cdiv(Ho, tr);
    tr = cdiv(Ho, ny);
    const long tiles = (long)ny * nx, halo = tiles * (tr + 2) * (tc + 2);
    if (best < 0 || tiles < best || (tiles == best && halo < best_halo)) { best = tiles; best_halo = halo; *tr_out = tr; *tc_out = tc; }
    if (tc <= 8) break;
  }
  return best < 0 ? -1 : best * 128;
}

template <typename T, int TH, int TW, int BN, int WM, int WN, int KH, int KW, int S, bool POOL = false, bool LIN = false, bool SPLIT = false>
int launch_cfg(const ConvK& P0, hipStream_t st) {
  using TT = Tr<T>;
  if (g_name_out) {
    // (the fused-pool instance reports under the same name: same tile, same main loop, one more store per window)
    if (SPLIT) snprintf(g_name_out, g_name_cap, "conv_fwd_kernel<%s,%d,%d,%d,%d,%d,%d,%d,%d,split>", sizeof(T) == 2 ? "bf16" : "f32", TH, TW, BN, WM, WN, KH, KW, S);
    else if (LIN) snprintf(g_name_out, g_name_cap, "conv_fwd_kernel<%s,lin128,%d,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", BN, WM, WN, KH, KW, S);
    else snprintf(g_name_out, g_name_cap, "conv_fwd_kernel<%s,%d,%d,%d,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", TH, TW, BN, WM, WN, KH, KW, S);
    return SEG_OK;
  }
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
  constexpr int PATCH_BYTES = (((LIN ? LIN_CAP : PH * PW) * TT::RSTR + 15) / 16) * 16;
  constexpr int LDS = PATCH_BYTES + KH * KW * BN * TT::RSTR + BN * 4;   // patch, filter rows, bias
  ConvK P = P0;
  P.tiles_x = cdiv(P.d.Wo, TW);
  P.tiles_y = cdiv(P.d.Ho, TH);
  P.lin_tr = P.lin_tc = 0;
  if (LIN) {
    if (lin_pick(P.d.Ho, P.d.Wo, &P.lin_tr, &P.lin_tc) < 0) { seg_set_error("conv: no linearised tile for a %d x %d map", P.d.Ho, P.d.Wo); return SEG_ERR_ARG; }
    P.tiles_x = cdiv(P.d.Wo, P.lin_tc);
    P.tiles_y = cdiv(P.d.Ho, P.lin_tr);
  }
  if (SPLIT ? (P.d.n_count != BN || P.d.n_split * 2 != BN) : (P.d.n_count % BN != 0 || P.d.n_split % BN != 0)) { seg_set_error("conv: n_count %d / n_split %d not a multiple of BN %d", P.d.n_count, P.d.n_split, BN); return SEG_ERR_ARG; }
  if (g_plan_out) { g_plan_out->bm = TH * TW; g_plan_out->bn = BN; g_plan_out->wgs = (long)P.d.B * P.tiles_y * P.tiles_x * (P.d.n_count / BN); return SEG_OK; }
  int gz;
  if (int rc = splitk_grid(P, &gz)) return rc;
  auto kern = conv_fwd_kernel<Tr<T>::DT, TH, TW, BN, WM, WN, KH, KW, S, POOL, LIN, SPLIT>;
  static bool attr_done = false;
  if (!attr_done && LDS > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      seg_set_error("conv: cannot raise dynamic LDS to %d", LDS); return SEG_ERR_LAUNCH;
    }
    attr_done = true;
  }
  dim3 grid(P.d.B * P.tiles_y * P.tiles_x, P.d.n_count / BN, gz);
  SEG_LAUNCH(kern, grid, dim3(256), LDS, st, P);
  return seg_check_launch("conv_fwd");
}

template <typename T, int BN>
int launch_mt(const ConvK& P0, int ntw, hipStream_t st) {
  using TT = Tr<T>;
  if (g_name_out) { snprintf(g_name_out, g_name_cap, "conv_fwd_mt_kernel<%s,%d>", sizeof(T) == 2 ? "bf16" : "f32", BN); return SEG_OK; }
  constexpr int PATCH_BYTES = ((10 * 18 * TT::RSTR + 15) / 16) * 16;
  constexpr int LDS = PATCH_BYTES + 9 * BN * TT::RSTR + BN * 4;
  ConvK P = P0;
  P.tiles_x = cdiv(P.d.Wo, 16);
  P.tiles_y = cdiv(P.d.Ho, 8);
  P.lin_tr = P.lin_tc = 0;
  P.ntw = ntw;
  if (P.d.n_count % BN != 0 || P.d.n_split % BN != 0 || P.d.ksplit > 1 || ntw < 1 || P.nchunks != 1) { seg_set_error("conv (multi-tile): n_count %d / n_split %d / ksplit %d / %d K chunks", P.d.n_count, P.d.n_split, P.d.ksplit, P.nchunks); return SEG_ERR_ARG; }
  const long ntiles = (long)P.d.B * P.tiles_y * P.tiles_x;
  if (g_plan_out) { g_plan_out->bm = 128; g_plan_out->bn = BN; g_plan_out->wgs = ntiles * (P.d.n_count / BN); return SEG_OK; }
  auto kern = conv_fwd_mt_kernel<Tr<T>::DT, BN>;
  static bool attr_done = false;
  if (!attr_done && LDS > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      seg_set_error("conv: cannot raise dynamic LDS to %d", LDS); return SEG_ERR_LAUNCH;
    }
    attr_done = true;
  }
  dim3 grid((unsigned)cdiv(ntiles, ntw), P.d.n_count / BN, 1);
  SEG_LAUNCH(kern, grid, dim3(256), LDS, st, P);
  return seg_check_launch("conv_fwd_mt");
}

template <int TH, int TW, int BN, int WM, int WN, int KH, int KW, int S, int NBUF = 2>
int launch_glds(const ConvK& P0, hipStream_t st) {
  if (g_name_out) {
    snprintf(g_name_out, g_name_cap, "conv_fwd_glds_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d>", TH, TW, BN, WM, WN, KH, KW, S, NBUF);
    return SEG_OK;
  }
  constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
  constexpr int PINST = (PH * PW * 4 + 63) / 64, WINST = KH * KW * BN * 4 / 64;
  constexpr int STAGES = NBUF * (PINST + WINST) * 1024;
  constexpr int LDS = STAGES + (STAGES + BN * 4 <= 160 * 1024 ? BN * 4 : 0);   // stages, bias
  static_assert(LDS <= 160 * 1024, "LDS budget");
  ConvK P = P0;
  P.tiles_x = cdiv(P.d.Wo, TW);
  P.tiles_y = cdiv(P.d.Ho, TH);
  if (P.d.n_count % BN != 0 || P.d.n_split % BN != 0) { seg_set_error("conv: n_count %d / n_split %d not a multiple of BN %d", P.d.n_count, P.d.n_split, BN); return SEG_ERR_ARG; }
  if (g_plan_out) { g_plan_out->bm = TH * TW; g_plan_out->bn = BN; g_plan_out->wgs = (long)P.d.B * P.tiles_y * P.tiles_x * (P.d.n_count / BN); return SEG_OK; }
  int gz;
  if (int rc = splitk_grid(P, &gz)) return rc;
  auto kern = conv_fwd_glds_kernel<TH, TW, BN, WM, WN, KH, KW, S, NBUF>;
  static bool attr_done = false;
  if (!attr_done && LDS > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      seg_set_error("conv: cannot raise dynamic LDS to %d", LDS); return SEG_ERR_LAUNCH;
    }
    attr_done = true;
  }
  dim3 grid(P.d.B * P.tiles_y * P.tiles_x, P.d.n_count / BN, gz);
  SEG_LAUNCH(kern, grid, dim3(64 * WM * WN), LDS, st, P);
  return seg_check_launch("conv_fwd_glds");
}


// padded output area for a tile shape (smaller = less wasted MFMA work)
inline long waste(int Ho, int Wo, int th, int tw) { return (long)cdiv(Ho, th) * th * cdiv(Wo, tw) * tw; }

template <typename T, int KH, int KW, int S>
int launch_k(const ConvK& P, hipStream_t st) {
  const seg_conv_desc& d = P.d;
  int cfg = d.cfg;
  const int mode_hint = cfg < 0 ? -cfg : 0;      // cfg < 0: automatic tile choice with staging mode |cfg| (see below)
  if (d.pool.ptr) {
    // fused max-pool: only the 8x16 register-staged tiles carry it (bf16, 3x3/s1); anything else is refused so that the
    // caller keeps the separate pool launch (it asks seg_conv2d_kernel_name first)
    if constexpr (sizeof(T) == 2 && KH == 3 && S == 1) {
      const long w1 = waste(d.Ho, d.Wo, 8, 16), w3 = waste(d.Ho, d.Wo, 8, 8);
      if (!d.up2 && !d.accum && !d.out_f32 && d.n_split == 0 && !d.mask.ptr && (cfg == 0 || cfg == 1 || cfg == 2) && 4 * w1 <= 5 * w3) {
        if ((cfg == 0 || cfg == 1) && d.n_count % 64 == 0) return launch_cfg<T, 8, 16, 64, 4, 1, KH, KW, S, true>(P, st);
        if (cfg != 1) return launch_cfg<T, 8, 16, 32, 4, 1, KH, KW, S, true>(P, st);
      }
    }
    seg_set_error("conv: fused max-pool is not available for this layer / tile"); return SEG_ERR_UNSUPPORTED;
  }
  if constexpr (KH == 3 && KW == 3 && S == 1) {
    // two destinations of 32 channels each: one 64-channel block per tile (cfg 9; SEG_CONV_SPLIT2=0: two 32-channel blocks)
    static const bool split2 = !(seg_env("SEG_CONV_SPLIT2") && atoi(seg_env("SEG_CONV_SPLIT2")) == 0);
    // (automatic for bf16 on big maps: 512^2 step 4.068 -> 4.033 ms, C2 -- 83 k pixels -- unchanged; the f32 instance needs 272 VGPRs)
    if ((cfg == 9 || (cfg <= 0 && split2 && sizeof(T) == 2 && (long)d.B * d.Ho * d.Wo >= 400000)) && d.n_split == 32 && d.n_count == 64) return launch_cfg<T, 8, 16, 64, 4, 1, KH, KW, S, false, false, true>(P, st);
    if (cfg == 9) { seg_set_error("conv: cfg 9 is the [32 | 32]-channel two-destination block"); return SEG_ERR_ARG; }
  }
  if (cfg <= 0) {
    // Tile choice by a two-term cost model calibrated on MI355X micro-benchmarks: a workgroup costs its MACs per
    // K step (BM*BN) plus a fixed part (prologue, first-load latency, epilogue ~ 5000 MAC-equivalents), and the
    // grid runs in rounds of 768 resident workgroups (3 per CU).  Small / deep layers therefore prefer the 32-channel
    // tiles that double the workgroup count; big maps prefer 128x64.
    static const int TH_[4] = {8, 8, 8, 8}, TW_[4] = {16, 16, 8, 8}, BN_[4] = {64, 32, 64, 32};
    long best = -1; int bi = 1;
    for (int c = 0; c < 4; ++c) {
      if (d.n_count % BN_[c] || d.n_split % BN_[c]) continue;
      const long nwg = (long)d.B * cdiv(d.Ho, TH_[c]) * cdiv(d.Wo, TW_[c]) * (d.n_count / BN_[c]);
      const long cost = ((nwg + 767) / 768) * ((long)TH_[c] * TW_[c] * BN_[c] + 5000);
      if (best < 0 || cost < best) { best = cost; bi = c; }
    }
    cfg = bi + 1;
    bool lin = false;
    if constexpr (KH == 3 && KW == 3 && S == 1) {
      // linearised tiles where the map pads badly into every fixed tile (round 4): at least SEG_CONV_LIN_PCT % (default 13) fewer slots
      static const int lin_pct = seg_env("SEG_CONV_LIN_PCT") ? atoi(seg_env("SEG_CONV_LIN_PCT")) : 13;
      int tr, tc;
      const long sl = lin_pct > 0 && sizeof(T) == 2 ? lin_pick(d.Ho, d.Wo, &tr, &tc) : -1;      // (bf16: the f32 64-channel instance needs 264 VGPRs)
      if (sl > 0 && sl * 100 <= waste(d.Ho, d.Wo, TH_[bi], TW_[bi]) * (100 - lin_pct)) { cfg = BN_[bi] == 64 ? 7 : 8; lin = true; }
    }
    if constexpr (KH == 3 && KW == 3 && S == 1 && sizeof(T) == 2) {
      // multi-tile walk (conv_fwd_mt_kernel) for the maps with one K chunk and at least SEG_CONV_NTW_MIN workgroups: SEG_CONV_NTW tiles
      // per workgroup (0 / 1 = off).  512^2 step 3.993 -> 3.948 ms with 2 (4: 3.946), C2 0.9695 -> 0.9667, inference 256^2 x 32 +0.8 %,
      // FCN-8s / DeconvModel unchanged (profiles/r04_ab_mt_*.txt)
      static const int ntw_env = seg_env("SEG_CONV_NTW") ? atoi(seg_env("SEG_CONV_NTW")) : 2;
      static const int ntw_min = seg_env("SEG_CONV_NTW_MIN") ? atoi(seg_env("SEG_CONV_NTW_MIN")) : 512;
      const long nwg = (long)d.B * cdiv(d.Ho, 8) * cdiv(d.Wo, 16) * (d.n_count / BN_[bi]);
      if (ntw_env > 1 && !lin && (cfg == 1 || cfg == 2) && !d.up2 && d.ksplit <= 1 && P.nchunks == 1 && nwg >= ntw_min && !g_plan_out)
        return cfg == 1 ? launch_mt<T, 64>(P, ntw_env, st) : launch_mt<T, 32>(P, ntw_env, st);
    }
    if (sizeof(T) == 2 && !lin) {
      // bf16: direct-to-LDS double-buffered variants.  SEG_CONV_MODE: 0 = never, 1 = always, 2 = always + 256-pixel
      // tile on maps >= 32 wide, 3 (default) = only for the 64-pixel tiles, 4 = single-buffered direct-to-LDS everywhere (fastest stand-alone:
      // the host asks for it on the forward pass; in backward the dgrads share the chip with the filter gradients and 3 wins)
      static const int env_mode = getenv("SEG_CONV_MODE") ? atoi(getenv("SEG_CONV_MODE")) : -1;
      const int mode = env_mode >= 0 ? env_mode : (mode_hint ? mode_hint : 3);
      if (mode == 1 || mode == 2 || (mode == 3 && cfg >= 3)) cfg += 10;
      if (mode == 4) cfg += 20;          // single-buffered direct-to-LDS (default: fastest on every measured layer)
      if (mode == 5) cfg = (cfg == 1 || cfg == 2) ? 22 : 24;   // ... and always 32-channel tiles
      if (mode == 2 && cfg == 11 && d.Ho >= 32 && d.Wo >= 32) cfg = 15;
    }
  }
  switch (cfg) {
    case 1: return launch_cfg<T, 8, 16, 64, 4, 1, KH, KW, S>(P, st);   // 128 px x 64 ch
    case 2: return launch_cfg<T, 8, 16, 32, 4, 1, KH, KW, S>(P, st);   // 128 px x 32 ch
    case 3: return launch_cfg<T, 8, 8, 64, 2, 2, KH, KW, S>(P, st);    //  64 px x 64 ch
    case 4: return launch_cfg<T, 8, 8, 32, 4, 1, KH, KW, S>(P, st);    //  64 px x 32 ch
    case 5: return launch_cfg<T, 16, 16, 64, 4, 1, KH, KW, S>(P, st);  // 256 px x 64 ch
    case 6: return launch_cfg<T, 16, 16, 32, 4, 1, KH, KW, S>(P, st);  // 256 px x 32 ch
    default: break;
  }
  if constexpr (KH == 3 && KW == 3 && S == 1) {
    // 32 / 34 / 38: the multi-tile walk for one-chunk layers, 32-channel blocks, 2 / 4 / 8 tiles per workgroup; 62 / 64 / 68: 64-channel blocks
    if (cfg == 32 || cfg == 34 || cfg == 38) return launch_mt<T, 32>(P, cfg - 30, st);
    if (cfg == 62 || cfg == 64 || cfg == 68) return launch_mt<T, 64>(P, cfg - 60, st);
    if (cfg == 7) return launch_cfg<T, 8, 16, 64, 4, 1, KH, KW, S, false, true>(P, st);   // 128 linearised slots x 64 ch
    if (cfg == 8) return launch_cfg<T, 8, 16, 32, 4, 1, KH, KW, S, false, true>(P, st);   // 128 linearised slots x 32 ch
  }
  if (sizeof(T) == 2) {
    switch (cfg) {      // bf16 direct-to-LDS double-buffered variants
      case 11: return launch_glds<8, 16, 64, 4, 1, KH, KW, S>(P, st);
      case 12: return launch_glds<8, 16, 32, 4, 1, KH, KW, S>(P, st);
      case 13: return launch_glds<8, 8, 64, 2, 2, KH, KW, S>(P, st);
      case 14: return launch_glds<8, 8, 32, 4, 1, KH, KW, S>(P, st);
      case 15: return launch_glds<16, 16, 64, 4, 1, KH, KW, S>(P, st);  // 256 px x 64 ch
      case 21: return launch_glds<8, 16, 64, 4, 1, KH, KW, S, 1>(P, st);  // single-buffered variants
      case 22: return launch_glds<8, 16, 32, 4, 1, KH, KW, S, 1>(P, st);
      case 23: return launch_glds<8, 8, 64, 2, 2, KH, KW, S, 1>(P, st);
      case 24: return launch_glds<8, 8, 32, 4, 1, KH, KW, S, 1>(P, st);
      default: break;
    }
  }
  {
    seg_set_error("conv: unknown cfg %d", cfg); return SEG_ERR_ARG;
  }
}

template <typename T>
int launch_t(const ConvK& P, hipStream_t st) {
  const seg_conv_desc& d = P.d;
  if (d.KH == 3 && d.KW == 3 && d.stride == 1) return launch_k<T, 3, 3, 1>(P, st);
  if (d.KH == 1 && d.KW == 1 && d.stride == 1) return launch_k<T, 1, 1, 1>(P, st);
  if (d.KH == 2 && d.KW == 2 && d.stride == 2) return launch_k<T, 2, 2, 2>(P, st);
  seg_set_error("conv: unsupported kernel %dx%d stride %d", d.KH, d.KW, d.stride);
  return SEG_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int seg_conv2d(const seg_conv_desc* dp, void* stream);
extern "C" int seg_conv2d_kernel_name(const seg_conv_desc* dp, char* buf, int32_t cap) {
  if (!buf || cap <= 0) { seg_set_error("kernel_name: bad buffer"); return SEG_ERR_ARG; }
  buf[0] = 0;
  g_name_out = buf; g_name_cap = cap;
  const int rc = seg_conv2d(dp, nullptr);
  g_name_out = nullptr;
  return rc;
}

extern "C" int seg_conv2d_splitk_plan(const seg_conv_desc* dp, int32_t* ksplit, int64_t* ws_bytes, int32_t* tickets) {
  if (!dp || !ksplit || !ws_bytes || !tickets) { seg_set_error("splitk_plan: null argument"); return SEG_ERR_ARG; }
  *ksplit = 1; *ws_bytes = 0; *tickets = 0;
  if (dp->cfg >= 200) return SEG_OK;                 // an instance of the persistent kernel (conv_ring.hip) is forced: it has no K split
  TilePlan tp = {0, 0, 0};
  seg_conv_desc d = *dp;
  d.ksplit = 0;
  g_plan_out = &tp;
  const int rc = seg_conv2d(&d, nullptr);
  g_plan_out = nullptr;
  if (rc != SEG_OK) return rc;
  if (tp.wgs <= 0) return SEG_OK;                    // (a kernel without a tile plan: the persistent form)
  const int nch = dp->src0.c / 32 + (dp->src1.ptr ? dp->src1.c / 32 : 0);
  int ks = 1;
  if (dp->ksplit > 0) {
    ks = dp->ksplit < nch ? dp->ksplit : nch;
  } else {
    // (no automatic split: measured slower on every C2 layer but one, header comment of seg_conv_desc.ksplit)
  }
  if (ks > 1) {
    *ksplit = ks;
    *ws_bytes = (int64_t)tp.wgs * ks * tp.bm * tp.bn * 4;
    *tickets = (int32_t)tp.wgs;
  }
  return SEG_OK;
}

extern "C" int seg_conv2d(const seg_conv_desc* dp, void* stream) {
  if (!dp) { seg_set_error("conv: null descriptor"); return SEG_ERR_ARG; }
  seg_conv_desc d_local;
  const seg_conv_desc* dq = dp;
  const seg_conv_desc& d0 = *dp;
  if (d0.cfg == 204 || d0.cfg == 208) {
    // a forced tile class of conv_ring.hip: where that kernel does not take the layer (32-channel blocks, f32, 1x1 ...) the
    // automatic choice runs instead
    int rc = SEG_OK;
    char dry[4];                                     // (a name buffer makes the call a dry run)
    if (seg_conv_ring(d0, dry, sizeof(dry), nullptr, &rc) == 0) { d_local = d0; d_local.cfg = 0; dq = &d_local; }
  }
  const seg_conv_desc& d = *dq;
  if (!d.src0.ptr || !d.dst.ptr || !d.w_packed) { seg_set_error("conv: null pointer"); return SEG_ERR_ARG; }
  if (d.src0.c <= 0 || d.src0.c % 32 || (d.src1.ptr && (d.src1.c <= 0 || d.src1.c % 32))) {
    seg_set_error("conv: source channels must be positive multiples of 32 (got %d,%d)", d.src0.c, d.src1.c); return SEG_ERR_ARG;
  }
  if (d.n_count <= 0 || d.n_count % 32 || d.n_off % 32 || d.n_off + d.n_count > d.n_total) {
    seg_set_error("conv: bad n range off %d count %d total %d", d.n_off, d.n_count, d.n_total); return SEG_ERR_ARG;
  }
  if (d.B <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.Hi <= 0 || d.Wi <= 0) { seg_set_error("conv: empty extent"); return SEG_ERR_ARG; }
  if (d.thin_src) {
    if (d.src0.cs != 8 || d.src0.coff != 0 || d.src0.c != 32 || (d.src1.ptr && (d.src1.cs != 8 || d.src1.coff != 0 || d.src1.c != 32))) {
      seg_set_error("conv: a thin source has cs 8, coff 0, c 32 (got cs %d coff %d c %d)", d.src0.cs, d.src0.coff, d.src0.c); return SEG_ERR_ARG;
    }
    if (d.src0.oy + d.Hi > d.src0.H || d.src0.ox + d.Wi > d.src0.W || (d.src1.ptr && (d.src1.oy + d.Hi > d.src1.H || d.src1.ox + d.Wi > d.src1.W))) {
      seg_set_error("conv: source window exceeds its buffer"); return SEG_ERR_ARG;
    }
  } else if (d.src0.oy + d.Hi > d.src0.H || d.src0.ox + d.Wi > d.src0.W || d.src0.coff + d.src0.c > d.src0.cs ||
      (d.src1.ptr && (d.src1.oy + d.Hi > d.src1.H || d.src1.ox + d.Wi > d.src1.W || d.src1.coff + d.src1.c > d.src1.cs))) {
    seg_set_error("conv: source window exceeds its buffer"); return SEG_ERR_ARG;
  }
  if (d.n_store != 0 && (d.n_store != 8 || d.n_split != 0 || d.pool.ptr || d.accum || d.dst.cs % 8 || d.dst.coff % 8)) {
    seg_set_error("conv: n_store is 0 or 8 and excludes n_split / pool / accum"); return SEG_ERR_ARG;
  }
  if (d.pool.ptr) {
    if (d.pool_h != d.Ho / 2 || d.pool_w != d.Wo / 2 || d.pool_h < 1 || d.pool_w < 1 || d.pool.oy + d.pool_h > d.pool.H || d.pool.ox + d.pool_w > d.pool.W ||
        d.pool.coff + d.n_count > d.pool.cs || d.pool.cs % 8 || d.pool.coff % 8) { seg_set_error("conv: bad pooled destination"); return SEG_ERR_ARG; }
  }
  if (d.n_split != 0) {
    if (d.n_split < 0 || d.n_split >= d.n_count || d.n_split % 32 || d.up2 || d.accum || d.out_f32 || !d.dst1.ptr) { seg_set_error("conv: bad n_split %d (needs 0 < n_split < n_count, multiple of 32, no up2/accum/out_f32)", d.n_split); return SEG_ERR_ARG; }
    const int n1 = d.n_count - d.n_split;
    if (d.dst1.oy + d.Ho > d.dst1.H || d.dst1.ox + d.Wo > d.dst1.W || d.dst1.coff + n1 > d.dst1.cs || d.dst.coff + d.n_split > d.dst.cs) { seg_set_error("conv: split destination window exceeds its buffer"); return SEG_ERR_ARG; }
    if (d.mask1.ptr && (d.mask1.oy + d.Ho > d.mask1.H || d.mask1.ox + d.Wo > d.mask1.W || d.mask1.coff + n1 > d.mask1.cs)) { seg_set_error("conv: split mask window exceeds its buffer"); return SEG_ERR_ARG; }
  }
  {
    const int sc = d.up2 ? 2 : 1;
    const int nch = d.n_store > 0 ? d.n_store : (d.up2 ? d.up_cout : (d.n_split > 0 ? d.n_split : d.n_count));
    if (d.dst.oy + sc * d.Ho > d.dst.H || d.dst.ox + sc * d.Wo > d.dst.W || d.dst.coff + nch > d.dst.cs) {
      seg_set_error("conv: destination window exceeds its buffer"); return SEG_ERR_ARG;
    }
    if (d.up2 && (d.up_cout <= 0 || d.up_cout % 8)) { seg_set_error("conv: up_cout must be a multiple of 8"); return SEG_ERR_ARG; }
    if (d.mask.ptr && (d.mask.oy + sc * d.Ho > d.mask.H || d.mask.ox + sc * d.Wo > d.mask.W || d.mask.coff + nch > d.mask.cs)) {
      seg_set_error("conv: mask window exceeds its buffer"); return SEG_ERR_ARG;
    }
  }
  {
    // bf16 3x3 / stride 1: the persistent 8-wave kernel (conv_ring.hip) where it is asked for (cfg 204 / 208, SEG_CONV_IMPL=ring)
    int rc = SEG_OK;
    if (!g_plan_out && seg_conv_ring(d, g_name_out, g_name_cap, reinterpret_cast<hipStream_t>(stream), &rc)) return rc;
  }
  ConvK P;
  P.d = d;
  if (!d.src1.ptr) { P.d.src1 = d.src0; P.d.src1.c = 0; }
  P.nchunks0 = d.src0.c / 32;
  P.nchunks = P.nchunks0 + (d.src1.ptr ? d.src1.c / 32 : 0);
  P.tiles_x = P.tiles_y = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d.dtype == SEG_F32) return launch_t<float>(P, st);
  if (d.dtype == SEG_BF16) return launch_t<bf16_t>(P, st);
  seg_set_error("conv: bad dtype %d", d.dtype);
  return SEG_ERR_ARG;
}
