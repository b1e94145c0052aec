// conv_sweep.hip -- 3x3 / stride-1 NHWC convolution (forward, and the data gradient as a correlation with the flipped packed
// filters) on MFMA, bf16, in the wave-specialised form of wgrad_sweep.hip (round 3).  Replaces conv_fwd_kernel for the layers
// it accepts (slim.convolution2d sites of /root/reference/models/unet.py:111-166, models/fcn.py:110-128 and their
// Conv2DBackpropInput); conv_fwd.hip keeps f32, 1x1, 2x2/s2, the transposed-conv scatter and the fused max-pool.
//
// Why another kernel (profiles/r02_sq_counters.json: 31 % of the tiled kernel's wave-cycles issuing,
// 37 % issue-stalled -- a third of that on LDS -- and 31 % parked at two barriers per 32-channel chunk):
//   * 8 waves = 4 MFMA waves + 4 loader waves.  The loaders move the next K chunk (input patch + all 9 taps of the filter rows)
//     global -> registers -> LDS while the MFMA waves work on the current one: ONE s_barrier per chunk, no staging or commit
//     instructions in the MFMA waves, two LDS stages + one register stage = two chunks of look-ahead.
//   * Workgroup tile = up to 384 output pixels x 64 channels; an MFMA wave owns 96 pixels x 64 channels (6 x 4 fragments):
//     10 fragment reads per 24 MFMAs = 0.42 KB of LDS reads per MFMA (the 4-wave tile: 0.75), and the 36 KB of filter rows a
//     chunk needs are amortised over three times the pixels.
//   * The pixel tile is a LINEARISED rectangle of output pixels (the B operand takes one LDS address per pixel): 24 x 24 maps
//     are cut into 12 x 24 windows instead of padded 8 x 16 tiles.
//   * Patch rows are 96 bytes apart in LDS (64 of data): any 16 consecutive rows are conflict-free for ds_read_b128 at every
//     tap shift and the address stays linear -- tap (u, v) is one scalar add (u) and an instruction offset (v).
#include "common.h"
#include <stdlib.h>
#include <string.h>

int seg_conv_sweep(const seg_conv_desc& d, char* name_out, int name_cap, hipStream_t st, int* rc);

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_cs[4] = {0, 0, 0, 0};

struct CsK {
  seg_conv_desc d;
  int TR, TC, PC, NPX, NPATCH;     // window rows / columns, patch columns, window / patch pixels
  uint32_t m_pc, m_tc;             // reciprocals: q / d == umulhi(q, 2^32 / d + 1) for q < 2^16, d >= 2
  int wy_n, wx_n;                  // windows per image
  int nchunks0, nchunks;           // 32-channel K chunks of src0 / total
  int nblk, ntiles;                // channel blocks; tiles = windows x channel blocks (channel block fastest)
};

constexpr int CS_PROW = 96;        // LDS bytes per patch pixel (64 of data)

#ifdef SEG_STAMPS
// debug builds only (tools/stamp_conv.py): s_memrealtime stamps (100 MHz) of wave 0 (MFMA) and wave 4 (loader) of every workgroup;
// row 0: entry, then per chunk (barrier passed, computed) and per tile (flushed); row 1: loader events (fetch issued, committed)
__device__ long long* g_csstamps = nullptr;
#define CSSTAMP(row, idx) do { if (csst && (idx) < 64) csst[(row) * 64 + (idx)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CSSTAMP(row, idx) do { } while (0)
#endif

SEG_DEV void cs_signal(const seg_conv_desc& d) {
  if (d.signal != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(d.signal, d.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// FM: 16-pixel fragments per MFMA wave (window = 64 FM pixels at most); BN: output channels per workgroup; XPXMAX: patch capacity.
// PERSISTENT: one workgroup per compute unit walks tiles (window x channel block); the loader waves run up to two K chunks
// ahead of the MFMA waves ACROSS tile boundaries, so only a workgroup's first tile pays the ~5 us from kernel entry to the first
// MFMA (first-touch latency of its loads), and the MFMA waves' epilogue overlaps the loads of the next tile.  Tiles are handed
// out by an atomic ticket (seg_conv_desc.sched; the last workgroup to finish resets it), because in a train step the
// filter-gradient kernels hold some of the compute units and a static split would wait for the workgroups that start late.
template <int FM, int BN, int XPXMAX>
__global__ __launch_bounds__(512) void conv_sweep_kernel(const CsK P) {
  using T = bf16_t;
  constexpr int FN = BN / 16, NJ = FN / 2;
  constexpr int PATCH_BYTES = XPXMAX * CS_PROW, W_BYTES = 9 * BN * 64;
  constexpr int STAGE = PATCH_BYTES + W_BYTES;
  static_assert(2 * STAGE + 64 <= 160 * 1024, "LDS budget");
  static_assert(XPXMAX % 64 == 0, "whole loader rounds of the patch");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* ring = reinterpret_cast<volatile int*>(smem + 2 * STAGE);      // tile id of this workgroup's k-th tile (k & 7), -1 = no more
  const seg_conv_desc& d = P.d;
  cs_signal(d);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
  const int nch = P.nchunks;
  const int nwin_img = P.wy_n * P.wx_n;
#ifdef SEG_STAMPS
  long long* csst = (g_csstamps && blockIdx.x < 512 && lane == 0 && (wave == 0 || wave == 4)) ? g_csstamps + (int64_t)blockIdx.x * 128 : nullptr;
  int nst = 1;
  if (wave == 0) CSSTAMP(0, 0);
#endif
  struct Tile { int b, oy0, ox0, n0; };
  auto decode = [&](int t) {
    Tile r; const int nb = t % P.nblk; int w = t / P.nblk;
    r.n0 = nb * BN; r.b = w / nwin_img; w -= r.b * nwin_img;
    const int wy = w / P.wx_n; r.oy0 = wy * P.TR; r.ox0 = (w - wy * P.wx_n) * P.TC;
    return r;
  };

  if (wave < 4) {
    // =========================================== MFMA waves ===========================================
    const int lr = lane & 15, g = lane >> 4;
    int pb[FM];                                                    // LDS byte offset of this lane's pixel (tap 0,0), + its K quarter
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      const uint32_t p = wave * (FM * 16) + fm * 16 + lr;
      uint32_t r = 0, c = 0;
      if ((int)p < P.NPX) { r = __umulhi(p, P.m_tc); c = p - r * (uint32_t)P.TC; }
      pb[fm] = (int)(r * (uint32_t)P.PC + c) * CS_PROW + 16 * g;
    }
    int a_addr[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) a_addr[fn] = PATCH_BYTES + frag_addr<T>(fn * 16 + lr, g);
    const int rowb = __builtin_amdgcn_readfirstlane(P.PC * CS_PROW);   // bytes per patch row

    f32x4 acc[FN][FM];
    auto compute = [&](const char* sb) {
      // per tap: FN filter fragments (next tap's are read while this tap computes) and FM pixel fragments (fragment fm of the
      // NEXT tap is read right behind the MFMAs that consumed it: FM - 1 fragment steps of look-ahead without a second set)
      Frag<T> fa[2][FN], fb[FM];
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) fa[0][fn] = lds_read_frag_at<T>(sb + a_addr[fn]);
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) fb[fm] = lds_read_frag_at<T>(sb + pb[fm]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int du1 = (tap + 1) / 3, dv1 = (tap + 1) % 3;
        const char* nb = sb + du1 * rowb + dv1 * CS_PROW;
        if (tap + 1 < 9) {
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) fa[(tap + 1) & 1][fn] = lds_read_frag_at<T>(sb + a_addr[fn] + (tap + 1) * BN * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) mma32(acc[fn][fm], fa[tap & 1][fn], fb[fm]);
          if (tap + 1 < 9) fb[fm] = lds_read_frag_at<T>(nb + pb[fm]);
          // order pinned: left alone the scheduler sinks every read to just in front of its MFMAs (one register set for all
          // FM fragments, s_waitcnt lgkmcnt(0) per 4 MFMAs: 3.4 us per chunk instead of 2.2)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    int tile = blockIdx.x, seq = 0;
    for (int k = 0; tile >= 0; ++k) {
      const Tile tl = decode(tile);
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) acc[fn][fm] = f32x4{0, 0, 0, 0};
      for (int c = 0; c < nch; ++c, ++seq) {
        barrier();                                 // this chunk is in its stage (loaders); everybody has finished the previous one
#ifdef SEG_STAMPS
        CSSTAMP(0, nst); ++nst;
#endif
        compute(smem + (seq & 1) * STAGE);
#ifdef SEG_STAMPS
        CSSTAMP(0, nst); ++nst;
#endif
      }
      // the loaders named this workgroup's next tile before the barrier of the chunk just computed (they run two chunks ahead)
      const int next = __builtin_amdgcn_readfirstlane(ring[(k + 1) & 7]);

      // ---- epilogue: bias, ReLU, accumulate, ReLU-grad mask; a lane stores 8 consecutive channels of one pixel ----
      // (packed-filter row order: fragments 2j, 2j+1 of 32-block j hold channels 8g..8g+3 / 8g+4..8g+7 of the lane's group g)
      seg_view dst = d.dst, msk = d.mask;
      if (d.n_split > 0 && tl.n0 >= d.n_split) {   // data gradient of a channel-concat input: this block's channels live in dst1
        dst = d.dst1; dst.coff -= d.n_split;
        msk = d.mask1; msk.coff -= d.n_split;
      }
      const bool has_mask = msk.ptr != nullptr, has_acc = d.accum != 0;
      const float lo = d.relu ? 0.f : -INFINITY;
      const int64_t dbase = ((int64_t)(tl.b * dst.H + dst.oy) * dst.W + dst.ox) * dst.cs + dst.coff;
      const T* mbase = reinterpret_cast<const T*>(msk.ptr) + ((int64_t)(tl.b * msk.H + msk.oy) * msk.W + msk.ox) * msk.cs + msk.coff;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nl = tl.n0 + j * 32 + 8 * g;
        if (nl >= d.n_count) continue;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int np = d.n_off + nl + e; bv[e] = (d.bias != nullptr && np < d.bias_n) ? d.bias[np] : 0.f; }
        Vec8<T> mk[FM], old[FM];
        int poff_d[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          const uint32_t p = wave * (FM * 16) + fm * 16 + lr;      // (window coordinates recomputed here: not worth 2 FM registers in the K loop)
          const uint32_t r = __umulhi(p, P.m_tc), c = p - r * (uint32_t)P.TC;
          const int oy = tl.oy0 + (int)r, ox = tl.ox0 + (int)c;
          const bool ok = (int)p < P.NPX && oy < d.Ho && ox < d.Wo;
          poff_d[fm] = ok ? (oy * dst.W + ox) * dst.cs : -1;
          if (has_mask && ok) mk[fm].load(mbase + (oy * msk.W + ox) * msk.cs + nl);
          if (has_acc && ok) old[fm].load(reinterpret_cast<const T*>(dst.ptr) + dbase + poff_d[fm] + nl);
        }
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
          if (poff_d[fm] < 0) continue;
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaxf(acc[2 * j][fm][e] + bv[e], lo);
            v[4 + e] = fmaxf(acc[2 * j + 1][fm][e] + bv[4 + e], lo);
          }
          if (has_acc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += old[fm].get(e);
          }
          if (has_mask) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = mk[fm].get(e) > 0.f ? v[e] : 0.f;
          }
          const int64_t doff = dbase + poff_d[fm] + nl;
          if (d.out_f32) {
            Vec8<float> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.set(e, v[e]);
            o.store(reinterpret_cast<float*>(dst.ptr) + doff);
          } else {
            Vec8<T> o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.set(e, v[e]);
            o.store(reinterpret_cast<T*>(dst.ptr) + doff);
          }
        }
      }
      tile = next;
#ifdef SEG_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CSSTAMP(0, nst); ++nst;
#endif
    }
    return;
  }

  // ============================================= loader waves =============================================
  // 16-byte pieces; four consecutive lanes = the 64 bytes a pixel (a filter row) contributes to a 32-channel chunk.  Plain
  // loads (the compiler counts them); chunk s+2 of the workgroup's chunk sequence is in registers while chunk s+1 sits in the
  // other LDS stage -- across tile boundaries.
  const int ll = (wave - 4) * 64 + lane;                 // 0..255
  constexpr int NPJ = XPXMAX * 4 / 256, NWJ = (9 * BN * 4 + 255) / 256;
  const T* wp = reinterpret_cast<const T*>(d.w_packed);
  const char* const zero = reinterpret_cast<const char*>(g_zero16_cs);
  const bool two = P.nchunks0 < nch;
  int* sched = reinterpret_cast<int*>(d.sched);
  const char* cur[NPJ];                                   // this lane's source address per patch slot for the NEXT chunk to fetch
  uint32_t okm = 0;                                       // bit j: slot j lies inside the image (else it reads the zero word and never advances)
  int woff0 = 0, wstr = 0;                                // byte offset of this lane's filter piece in round 0 / per round, inside a chunk slab
  Tile ft;                                                // the tile being fetched
  auto point_at = [&](const seg_view& sv) {               // cur[] <- chunk 0 of source sv for tile ft
    const int iy0 = ft.oy0 - d.pad_t, ix0 = ft.ox0 - d.pad_l;
    const T* base = reinterpret_cast<const T*>(sv.ptr) + ((int64_t)(ft.b * sv.H + iy0 + sv.oy) * sv.W + ix0 + sv.ox) * sv.cs + sv.coff;
    okm = 0;
#pragma unroll
    for (int j = 0; j < NPJ; ++j) {
      const uint32_t i = j * 256 + ll, q = i >> 2, pc4 = i & 3;
      const uint32_t qq = (int)q < P.NPATCH ? q : 0;
      const uint32_t py = __umulhi(qq, P.m_pc), px = qq - py * (uint32_t)P.PC;
      const bool ok = ((int)q < P.NPATCH) & ((unsigned)(iy0 + (int)py) < (unsigned)d.Hi) & ((unsigned)(ix0 + (int)px) < (unsigned)d.Wi);
      cur[j] = ok ? reinterpret_cast<const char*>(base + ((int)py * sv.W + (int)px) * sv.cs + pc4 * 8) : zero;
      okm |= (ok ? 1u : 0u) << j;
    }
  };
  auto setup_tile = [&](int t) {
    ft = decode(t);
    point_at(d.src0);
    // filter pieces: round j holds rows j * 64 .. of the (tap-major) chunk slab -- one tap per round (BN = 64) or two (BN = 32)
    const int r0 = ll >> 2, tap0 = r0 / BN, row0 = r0 % BN;
    woff0 = ((tap0 * nch * d.n_total + d.n_off + ft.n0 + row0) * 32 + (ll & 3) * 8) * 2;
    wstr = (64 / BN) * nch * d.n_total * 64;
  };
  const int pl0 = (ll >> 2) * CS_PROW + (ll & 3) * 16;                       // LDS address of slot 0; slot j is 64 pixels further
  const int wl0 = PATCH_BYTES + Tr<T>::lds_off(ll >> 2, ll & 3);             // ... 64 rows further (bit 2 of the row is unchanged)
  u32x4 rp[NPJ], rw[NWJ];
  // Tile sequence of this workgroup: tile 0 = its own index, tile 1 = index + grid size, every further one 2 x grid size + an
  // atomic ticket (static stride without a counter).  Lane 0 of the first loader wave names tile k+1 in ring[(k+1) & 7] (-1: no
  // more) while tile k is being OPENED -- at least one barrier before any other wave needs it: the other loader waves when they
  // open tile k+1, the MFMA waves after the last chunk of tile k.
  int fk = 0, fc = 0, nfetched = 0;                       // tile sequence number / chunk inside it of the next fetch; chunks fetched so far
  int cur_tile = (int)blockIdx.x;
  bool ended = false;
  auto announce_next = [&]() {                            // (first loader wave only) names tile fk + 1
    int id;
    if (fk == 0) id = (int)(blockIdx.x + gridDim.x);
    else if (sched) {
      int v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      id = 2 * (int)gridDim.x + __builtin_amdgcn_readfirstlane(v);
    } else id = cur_tile + (int)gridDim.x;
    if (lane == 0) ring[(fk + 1) & 7] = id < P.ntiles ? id : -1;
  };
  auto try_fetch = [&]() {
    if (ended) return;
    if (fc == nch) {                              // the next chunk opens a new tile
      ++fk; fc = 0;
      int t = fk == 1 ? (int)(blockIdx.x + gridDim.x) : __builtin_amdgcn_readfirstlane(ring[fk & 7]);
      if (t >= P.ntiles) t = -1;
      if (t < 0) { ended = true; return; }
      cur_tile = t;
      setup_tile(t);
      if (wave == 4) announce_next();
    }
    if (two && fc == P.nchunks0) point_at(d.src1);       // channel concat: the second source starts here
    const char* wc = reinterpret_cast<const char*>(wp) + (int64_t)fc * d.n_total * 64 + woff0;
#pragma unroll
    for (int j = 0; j < NPJ; ++j) { rp[j] = *reinterpret_cast<const u32x4*>(cur[j]); cur[j] += ((okm >> j) & 1) ? 64 : 0; }
#pragma unroll
    for (int j = 0; j < NWJ; ++j) rw[j] = *reinterpret_cast<const u32x4*>(j * 256 + ll < 9 * BN * 4 ? wc + (int64_t)j * wstr : zero);   // (BN = 32: the last round is half full)
    ++fc; ++nfetched;
#ifdef SEG_STAMPS
    CSSTAMP(1, nst); ++nst;
#endif
  };
  auto commit = [&](char* sbase) {
#pragma unroll
    for (int j = 0; j < NPJ; ++j) *reinterpret_cast<u32x4*>(sbase + pl0 + j * 64 * CS_PROW) = rp[j];
#pragma unroll
    for (int j = 0; j < NWJ; ++j) if (j * 256 + ll < 9 * BN * 4) *reinterpret_cast<u32x4*>(sbase + wl0 + j * 64 * 64) = rw[j];
  };
  setup_tile(blockIdx.x);
  if (wave == 4) announce_next();
  try_fetch(); commit(smem);
  try_fetch();
  for (int s = 0; s < nfetched; ++s) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my LDS stores of chunk s (and the ring entry) are done
    barrier();
    if (s + 1 < nfetched) commit(smem + ((s + 1) & 1) * STAGE);
#ifdef SEG_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CSSTAMP(1, nst); ++nst;
#endif
    try_fetch();
  }
  // the last workgroup to leave resets the ticket counter for the next launch that uses this slot
  if (sched && wave == 4 && lane == 0) {
    const int done = __hip_atomic_fetch_add(sched + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (int)gridDim.x - 1) {
      __hip_atomic_store(sched, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sched + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
struct CsGeom { int TR, TC, PC, NPX, NPATCH, wy_n, wx_n; double cost; };

// window for an Ho x Wo map under a capacity of `maxpx` output pixels / `maxpatch` patch pixels: least padded work
bool cs_window(int Ho, int Wo, int maxpx, int maxpatch, CsGeom* out) {
  bool found = false; out->cost = 1e30;
  auto consider = [&](int TR, int TC) {
    if (TR < 1 || TC < 2 || TR > Ho || TC > Wo) return;
    const int PR = TR + 2, PC = TC + 2;
    if (TR * TC > maxpx || PR * PC > maxpatch) return;
    const int wy = cdiv(Ho, TR), wx = cdiv(Wo, TC);
    // cost ~ windows x (capacity-sized compute + a fixed part): prefers few, well-filled windows
    const double cost = (double)wy * wx * (maxpx + 0.25 * maxpx) + 1e-3 * (wy * wx * PR * PC);
    if (cost < out->cost) { out->TR = TR; out->TC = TC; out->PC = PC; out->NPX = TR * TC; out->NPATCH = PR * PC; out->wy_n = wy; out->wx_n = wx; out->cost = cost; found = true; }
  };
  for (int tc = 2; tc <= Wo && tc <= maxpx; ++tc) {
    if (tc != Wo && tc % 4) continue;
    for (int tr = 1; tr <= Ho && tr * tc <= maxpx; ++tr) consider(tr, tc);
  }
  return found;
}

template <int FM, int BN, int XPXMAX>
int cs_launch(const CsK& P, char* name_out, int name_cap, hipStream_t st) {
  constexpr int LDS = 2 * (XPXMAX * CS_PROW + 9 * BN * 64) + 64;
  if (name_out) { snprintf(name_out, name_cap, "conv_sweep_kernel<%d,%d,%d>", FM, BN, XPXMAX); return SEG_OK; }
  auto kern = conv_sweep_kernel<FM, BN, XPXMAX>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      seg_set_error("conv_sweep: cannot raise dynamic LDS to %d", LDS); return SEG_ERR_LAUNCH;
    }
    attr_done = true;
  }
  // persistent: one workgroup per compute unit (the LDS footprint admits one), fewer when there are fewer tiles
  static int ncu = 0;
  if (ncu == 0) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
  SEG_LAUNCH(kern, dim3(P.ntiles < ncu ? P.ntiles : ncu), dim3(512), LDS, st, P);
  return seg_check_launch("conv_sweep");
}

}  // namespace

#ifdef SEG_STAMPS
extern "C" int seg_dbg_set_csstamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_csstamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// 1 = handled (rc holds the result), 0 = not eligible (the caller runs the 4-wave tiled kernel)
int seg_conv_sweep(const seg_conv_desc& d, char* name_out, int name_cap, hipStream_t st, int* rc) {
  // OPT-IN (SEG_CONV_IMPL=sweep, or a forced tile class cfg 102 / 104 / 106): measured in r03 (profiles/r03_conv_sweep_micro.txt) this
  // kernel is level with the 4-wave tiled kernel stand-alone at 512 x 512 (conv4_2 83 us vs 84, conv6_1 102 vs 112, conv3_2 94 vs 90,
  // conv2_2 130 vs 110), slower on the 256 x 256 layers (its ~5 us from entry to the first MFMA is paid per workgroup, the tiled
  // kernel hides it behind two co-resident workgroups), and 10 % slower inside the train step at both sizes -- so the tiled
  // kernel stays the default.
  static const char* impl = getenv("SEG_CONV_IMPL");
  const bool opt_in = impl && !strcmp(impl, "sweep");
  if (!opt_in && d.cfg < 100) return 0;
  if (d.dtype != SEG_BF16 || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.up2 || d.pool.ptr || d.n_store || d.thin_src) return 0;
  if (d.cfg > 0 && d.cfg < 100) return 0;                     // an explicit tile of the 4-wave kernel
  if (d.n_count % 32 || (d.n_split % 32)) return 0;
  const int BN = (d.n_count % 64 == 0 && d.n_split % 64 == 0) ? 64 : 32;
  // all element offsets inside one image of any tensor stay below 2^31 (the kernel's 32-bit pixel offsets)
  auto small = [](const seg_view& v) { return !v.ptr || (int64_t)v.H * v.W * v.cs < ((int64_t)1 << 30); };
  if (!small(d.src0) || !small(d.src1) || !small(d.dst) || !small(d.dst1) || !small(d.mask) || !small(d.mask1)) return 0;
  if ((int64_t)9 * (d.src0.c + (d.src1.ptr ? d.src1.c : 0)) * d.n_total * 2 >= ((int64_t)1 << 31)) return 0;
  // tile class: pixels per workgroup 384 / 256 / 128.  A workgroup costs ~5 us before its first MFMA plus the epilogue, so small
  // layers prefer more, smaller workgroups until the chip is full; big ones the biggest tile (filter rows amortised best).
  static const int FMs[] = {6, 4, 2}, XPs[] = {448, 320, 192};
  int pick = -1; CsGeom best; double bestc = 1e30;
  const int nchunks = (d.src0.c + (d.src1.ptr ? d.src1.c : 0)) / 32;
  int f_fm = 0;
  if (d.cfg >= 100) f_fm = d.cfg - 100;                       // cfg 102 / 104 / 106: forced tile class (tests, tools)
  if (const char* e = getenv("SEG_CONV_SWEEP_FM")) f_fm = atoi(e);
  for (int k = 0; k < 3; ++k) {
    if (f_fm && FMs[k] != f_fm) continue;
    CsGeom g;
    if (!cs_window(d.Ho, d.Wo, FMs[k] * 64, XPs[k], &g)) continue;
    const long wgs = (long)d.B * g.wy_n * g.wx_n * (d.n_count / BN);
    const double t_mfma = 9.0 * FMs[k] * (BN / 16) * 16.0 / 1550.0, t_load = 0.1 * (XPs[k] * 4 / 256 + 9 * BN * 4 / 256) + 0.15;
    const double t_wg = 5.0 + nchunks * (t_mfma > t_load ? t_mfma : t_load) + 1.0 + 0.004 * FMs[k] * 64;
    const double cost = (double)((wgs + 255) / 256) * t_wg;
    if (cost < bestc) { bestc = cost; pick = k; best = g; }
  }
  if (pick < 0) return 0;
  CsK P;
  P.d = d;
  if (!d.src1.ptr) { P.d.src1 = d.src0; P.d.src1.c = 0; }
  P.TR = best.TR; P.TC = best.TC; P.PC = best.PC; P.NPX = best.NPX; P.NPATCH = best.NPATCH;
  auto magic = [](int dv) { return (uint32_t)((((uint64_t)1) << 32) / (uint64_t)dv + 1); };
  P.m_pc = magic(best.PC); P.m_tc = magic(best.TC);
  P.wy_n = best.wy_n; P.wx_n = best.wx_n;
  P.nchunks0 = d.src0.c / 32; P.nchunks = nchunks;
  P.nblk = d.n_count / BN; P.ntiles = d.B * best.wy_n * best.wx_n * P.nblk;
  if (BN == 64) {
    if (pick == 0) *rc = cs_launch<6, 64, 448>(P, name_out, name_cap, st);
    else if (pick == 1) *rc = cs_launch<4, 64, 320>(P, name_out, name_cap, st);
    else *rc = cs_launch<2, 64, 192>(P, name_out, name_cap, st);
  } else {
    if (pick == 0) *rc = cs_launch<6, 32, 448>(P, name_out, name_cap, st);
    else if (pick == 1) *rc = cs_launch<4, 32, 320>(P, name_out, name_cap, st);
    else *rc = cs_launch<2, 32, 192>(P, name_out, name_cap, st);
  }
  return 1;
}
