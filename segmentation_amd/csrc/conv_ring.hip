// conv_ring.hip -- 3x3 / stride-1 NHWC convolution (forward, and the data gradient as a correlation with the flipped packed filters)
// on MFMA, bf16: the round-4 main loop.  An alternative to conv_fwd_kernel on the layers it accepts (it replaces round 3's conv_sweep_kernel, which it beats everywhere)
// (slim.convolution2d sites of /root/reference/models/unet.py:111-166, models/fcn.py:110-128 and their Conv2DBackpropInput);
// conv_fwd.hip keeps f32, 1x1, 2x2/s2 and the transposed-convolution scatter.
//
// What the r04 evidence says about the two older kernels (profiles/r04_sq_counters.json, r04_conv_sweep_stamps.txt, r04_calibration.json):
//   * all three round-3 forms plateau at ~700 TF/s on the C >= 128 layers of the 512^2 step.  The 4-wave tile (128 px x 64 ch) moves
//     48 KB global -> LDS per 4.7 MFLOP chunk = 97 FLOP/B: at the ~7.3 TB/s that 4 loading waves per CU pull out of L2
//     (profiles/r01_fill_bw.txt) that IS 700 TF/s; its MFMA pipes are 40 % busy.
//   * the persistent kernel's chunk loop runs 144 MFMAs in 1.95 us = 59 % of the bare MFMA rate of this box (2.0 PF on random
//     data): its four loader waves push every byte through VGPRs and ds_write_b128 (13 LDS-path cycles per KB) beside the MFMA
//     waves' fragment reads; then 3.6 us of epilogue per 16 us tile with the matrix pipes idle, and 3.5 tiles per workgroup.
// So: (1) more FLOP per staged byte -- 512 px x 64 ch (248 FLOP/B) or 256 px x 64 ch workgroup tiles; (2) no loader waves: all 8
// waves compute, the stage is filled by global_load_lds (no VGPRs, no ds_write), chunk c+1 in flight while chunk c computes, ONE
// barrier per chunk; (3) fewer LDS fragment reads per MFMA: a wave owns 4 output rows x 16 columns and walks the patch ROWS -- the
// pixel fragment of patch row r at column shift v serves the output rows r, r-1, r-2 (filter rows 0, 1, 2): 3 x 6 pixel reads per
// chunk instead of 9 x 4, with the filter fragments of one filter COLUMN held in registers (0.375 KB per MFMA against 0.75);
// (4) persistent workgroups with tile tickets whose next tile's first stage is in flight during the epilogue.
#include "common.h"
#include <stdlib.h>
#include <string.h>

int seg_conv_ring(const seg_conv_desc& d, char* name_out, int name_cap, hipStream_t st, int* rc);

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero16_cr[4] = {0, 0, 0, 0};

struct CrK {
  seg_conv_desc d;
  int WR, WC;                      // waves along the rows / columns of the pixel tile (WR * WC = pixel waves)
  int TH, TW, PW, NPATCH;          // tile extent, patch row length, patch pixels
  uint32_t m_pw;                   // reciprocal: q / PW == umulhi(q, m_pw) for q < 2^16
  int tiles_x, tiles_y, nblk, ntiles;
  int nchunks0, nchunks;
  int abl;                         // debug (SEG_RING_ABL): 1 no patch fills, 2 no filter fills, 4 no fragment reads / MFMAs -- results are garbage
};

// global -> LDS fill of 1 KiB per wave instruction (lane l: 16 bytes from its own address to LDS byte lds_wave_base + 16 l), as INLINE
// ASM: with the __builtin_amdgcn_global_load_lds form hipcc (ROCm 7.2) knows that an LDS-DMA is pending and from then on waits
// lgkmcnt(0) in front of every use of every ds_read result -- no counted waits, no fragment read in flight behind an MFMA (the .s of
// conv_fwd_glds_kernel shows the same: 40 x lgkmcnt(0), not one counted wait; the register-staged kernel has lgkmcnt(1) / (2)).
// Hidden in asm, the fills are ordered by hand: "s_waitcnt vmcnt(0)" + s_barrier before a stage is read (the compiler's own vmcnt
// counts only become more conservative with operations it does not know about).
SEG_DEV int cr_swz(int L) { return L ^ ((L >> 3) & 32); }

SEG_DEV void cr_glds16(const void* gsrc, uint32_t lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_wave_base), "v"(gsrc) : "memory");
}

#ifdef SEG_STAMPS
__device__ long long* g_crstamps = nullptr;
#define CRSTAMP(idx) do { if (crst && (idx) < 32) { crst[(idx)] = __builtin_amdgcn_s_memrealtime(); crst[32 + (idx)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define CRSTAMPN() do { CRSTAMP(nst); ++nst; } while (0)
#else
#define CRSTAMP(idx) do { } while (0)
#define CRSTAMPN() do { } while (0)
#endif

// NWAVE = PXW x CHW waves: PXW pixel groups (R rows x 16 columns each) x CHW channel groups (FN 16-channel fragments each).
// R = 4 with 8 waves (two per SIMD) or R = 8 with 4 waves (one per SIMD, twice the registers): the wave that owns 8 rows reads
// 9 FN + 3 x 10 fragments per 72 FN MFMAs instead of 9 FN + 3 x 6 per 36 FN -- the 8-wave form spends 1.2 us per chunk on LDS reads
// alone (432 KB) of which only 40 % hide behind its 2.5 us of MFMAs (profiles/r04_ring_ablate.txt).
// NPIXCAP: patch capacity in pixels (a multiple of 16).  POOL: also write the 2x2/s2 max-pool of the output.
template <int NWAVE, int R, int CHW, int FN, int NPIXCAP, bool POOL>
__global__ __launch_bounds__(64 * NWAVE, NWAVE == 4 ? 2 : 1) void conv_ring_kernel(const CrK P) {
  // (4 waves: "2 workgroups per CU" only caps the registers at 256 architectural VGPRs -- given 512, hipcc parks the accumulators in
  // AGPRs and copies them through v_accvgpr_write in front of every MFMA; the LDS footprint admits one workgroup per CU anyway)
  constexpr int PXW = NWAVE / CHW;
  using T = bf16_t;
  constexpr int BN = CHW * FN * 16, NJ = FN / 2;
  constexpr int P_BYTES = NPIXCAP * 64, W_BYTES = 9 * BN * 64, STAGE = P_BYTES + W_BYTES;
  constexpr int NPI = NPIXCAP / 16, NPJ = (NPI + NWAVE - 1) / NWAVE;         // patch fills (1 KiB wave instructions): all / per wave
  constexpr int NWI = 9 * BN / 16, NWJ = (NWI + NWAVE - 1) / NWAVE;          // filter fills
  constexpr int WPT = BN / 16;                                   // filter fills per tap
  static_assert((NWAVE == 8 || NWAVE == 4) && PXW * CHW == NWAVE && FN % 2 == 0 && NPIXCAP % 16 == 0 && R % 2 == 0, "wave layout");
  constexpr int NS = 3 * (R + 2);                                // steps (patch row x filter column) per chunk
  static_assert(2 * STAGE + 64 <= 160 * 1024, "LDS budget");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* ring = reinterpret_cast<volatile int*>(smem + 2 * STAGE);     // tile ids of this workgroup's tiles k & 3
  const seg_conv_desc& d = P.d;
  if (d.signal != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(d.signal, d.signal_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / CHW, wn = wave % CHW;                    // pixel group / channel group
  const int wr = wp / P.WC, wc = wp - wr * P.WC;                 // its place in the tile
  const int lr = lane & 15, g = lane >> 4;
  const int nch = P.nchunks;
#ifdef SEG_STAMPS
  long long* crst = (g_crstamps && blockIdx.x < 512 && tid == 0) ? g_crstamps + (int64_t)blockIdx.x * 64 : nullptr;
  int nst = 1;
  CRSTAMP(0);
#endif

  // XCD-aware first tile: workgroups b and b + 8 share an XCD (round-robin dispatch); give each XCD a contiguous run of tile ids
  // (channel block fastest: the workgroups of an XCD then read the same patches, and every XCD's L2 holds the whole filter)
  const int G = (int)gridDim.x;
  int vb = (int)blockIdx.x;
  if ((G & 7) == 0) vb = (vb & 7) * (G >> 3) + (vb >> 3);

  struct Tile { int b, oy0, ox0, n0; };
  auto decode = [&](int t) {
    Tile r; const int nb = t % P.nblk; int w = t / P.nblk;
    const int per = P.tiles_y * P.tiles_x;
    r.n0 = nb * BN; r.b = w / per; w -= r.b * per;
    const int ty = w / P.tiles_x; r.oy0 = ty * P.TH; r.ox0 = (w - ty * P.tiles_x) * P.TW;
    return r;
  };

  // ---- fill descriptors: fill k of a stage covers its LDS bytes [k * 1024, +1024) = 16 records x 4 pieces; wave w issues k = w, w + 8, ..
  // A record's 16-byte pieces are stored at position (piece ^ f(record)), f = bit 2 of the record index << 1 (conflict-free
  // ds_read_b128 of 16 consecutive records, common.h); global_load_lds writes lane-linear, so the swizzle picks the SOURCE piece.
  const T* img0 = nullptr; const T* img1 = nullptr;
  int f_iy0 = 0, f_ix0 = 0;                                      // input coordinates of patch record (0, 0) of the tile being filled
  auto setup = [&](const Tile& t) {
    f_iy0 = t.oy0 - d.pad_t; f_ix0 = t.ox0 - d.pad_l;
    img0 = reinterpret_cast<const T*>(d.src0.ptr) + (int64_t)t.b * d.src0.H * d.src0.W * d.src0.cs + d.src0.coff;
    img1 = reinterpret_cast<const T*>(d.src1.ptr) + (int64_t)t.b * d.src1.H * d.src1.W * d.src1.cs + d.src1.coff;
  };
  const T* wpk = reinterpret_cast<const T*>(d.w_packed);
  const int w_lane = (lane >> 2) * 32 + ((lane & 3) ^ (((lane >> 2) >> 1) & 2)) * 8;       // element offset of this lane's piece inside a fill's 16 rows
  const char* const zero = reinterpret_cast<const char*>(g_zero16_cr);
  const int npi_used = (P.NPATCH + 15) >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // The NPJ + NWJ fills a wave contributes to a stage are issued ONE AT A TIME between the MFMA groups of the chunk that computes
  // meanwhile (compute(): a burst of them right behind the barrier kept the matrix pipes idle for ~0.45 us per chunk).
  struct Fill { const T* sb; const T* wc0; int64_t tap_str; uint32_t stage; int vW, vcs, voff; bool on; };
  auto fill_ctx = [&](int c, int n0, int stage_idx, bool on) {
    Fill f;
    const bool first = c < P.nchunks0;
    f.sb = first ? img0 + c * 32 : img1 + (c - P.nchunks0) * 32;
    const seg_view& sv = first ? d.src0 : d.src1;
    f.vW = sv.W; f.vcs = sv.cs; f.voff = (sv.oy * sv.W + sv.ox) * sv.cs;
    f.wc0 = wpk + ((int64_t)c * d.n_total + d.n_off + n0) * 32 + w_lane;
    f.tap_str = (int64_t)nch * d.n_total * 32;
    f.stage = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)stage_idx * STAGE);
    f.on = on;
    return f;
  };
  auto issue_one = [&](const Fill& f, int i) {                   // i: 0 .. NPJ + NWJ - 1 (compile-time after unrolling)
    if (!f.on) return;
    if (i < NPJ) {
      const int j = i;
      if (j * NWAVE + wave < npi_used && !(P.abl & 1)) {
        // (the record's coordinates are recomputed per fill: a dozen VALU instructions beside 16+ MFMAs instead of a register per fill)
        // (laundered lane id: otherwise hipcc hoists the coordinates of every fill out of the chunk loop -- three registers per fill --
        // and spills them)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t q = (uint32_t)((j * NWAVE + wave) * 16 + (ln >> 2));
        const int h = (ln & 3) ^ ((q >> 1) & 2);                   // the piece that belongs at this lane's LDS position (swizzle on the source)
        const uint32_t py = __umulhi(q, P.m_pw), px = q - py * (uint32_t)P.PW;
        const int iy = f_iy0 + (int)py, ix = f_ix0 + (int)px;
        const bool ok = (int)q < P.NPATCH && (unsigned)iy < (unsigned)d.Hi && (unsigned)ix < (unsigned)d.Wi;
        const int off = (iy * f.vW + ix) * f.vcs + f.voff + h * 8;
        const void* gp = ok ? (const void*)(f.sb + off) : (const void*)zero;
        cr_glds16(gp, f.stage + (j * NWAVE + wave) * 1024);
      }
    } else {
      const int k = (i - NPJ) * NWAVE + wave;
      if (k < NWI && !(P.abl & 2)) {
        const int tap = k / WPT, sub = k - tap * WPT;
        cr_glds16(f.wc0 + tap * f.tap_str + sub * 512, f.stage + P_BYTES + k * 1024);
      }
    }
  };
  auto issue = [&](int c, int n0, int stage_idx) {               // all of a stage at once (a workgroup's very first stage)
    const Fill f = fill_ctx(c, n0, stage_idx, true);
#pragma unroll
    for (int i = 0; i < NPJ + NWJ; ++i) issue_one(f, i);
  };

  // ---- fragment addresses (bytes inside a stage) ----
  // pixel fragment of patch row wr*R + ri at column shift v: record (wr*R + ri) * PW + wc*16 + v + lr, byte address L ^ ((L >> 3) & 32)
  // with L = record * 64 + 16 g (the piece swizzle is bit 2 of the record).  Recomputed per read from ONE register (4 VALU instructions
  // beside >= 4 MFMAs): a table of the 3 (R + 2) addresses cost 18 - 30 registers that the 8-row wave does not have.
  const int pl0 = ((wr * R) * P.PW + wc * 16 + lr) * 64 + g * 16;
  const int prow = __builtin_amdgcn_readfirstlane(P.PW * 64);
#define CR_PADDR(ri_, v_) cr_swz(pl + (ri_) * prow + (v_) * 64)
  // filter fragment fn: rows wn * FN * 16 + fn * 16 + lr -- 16 rows further is 1024 bytes further (bit 2 of the row, which swizzles the
  // pieces, is bit 2 of lr)
  const int a_addr0 = P_BYTES + frag_addr<T>(wn * (FN * 16) + lr, g);
#define CR_AADDR(fn_) (a_addr0 + (fn_) * 1024)

  f32x4 acc[FN][R];
#if defined(SEG_RING_NOREAD)
#define CR_READ(dst, addr) asm volatile("" : "+v"(dst.v))
#else
#define CR_READ(dst, addr) dst = lds_read_frag_at<T>(addr)
#endif
#if defined(SEG_RING_NOMMA)
#define CR_MMA(acc_, a_, b_) asm volatile("" :: "v"(a_.v), "v"(b_.v))
#else
#define CR_MMA(acc_, a_, b_) mma32(acc_, a_, b_)
#endif
  auto compute = [&](const char* sb, const Fill& fl) {
    // filter column v: the 3 x FN filter fragments of taps (0..2, v) are held while the six patch rows pass.  Filter row u is last
    // used by patch row 3 + u, and its registers are refilled with column v + 1 right behind those MFMAs (needed again 12 / 8 / 12
    // MFMAs later): one register set of filter fragments, no bubble between the columns of a chunk.  Pixel fragments are read two
    // steps (>= 8 MFMAs) ahead.  At the start only filter row 0 and the first pixel fragment are waited for.
    // The ORDER is pinned (sched_barrier): left alone, hipcc sinks every fragment read to just in front of the MFMAs that consume
    // it and waits lgkmcnt(0) there -- no read is in flight behind any MFMA (first build: 4.0 us per chunk for 2.4 us of MFMAs).
    Frag<T> wf[3][FN], pf[3];
    int pl = pl0;
    asm volatile("" : "+v"(pl));                                  // (keeps the 3 (R + 2) fragment addresses from being hoisted out of the chunk loop)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) CR_READ(wf[0][fn], sb + CR_AADDR(fn));
    CR_READ(pf[0], sb + CR_PADDR(0, 0));
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) CR_READ(wf[1][fn], sb + CR_AADDR(fn) + 3 * BN * 64);
    CR_READ(pf[1], sb + CR_PADDR(1, 0));
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) CR_READ(wf[2][fn], sb + CR_AADDR(fn) + 6 * BN * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < 3; ++v) {
#pragma unroll
      for (int ri = 0; ri < R + 2; ++ri) {
        const int step = v * (R + 2) + ri;
        if (step + 2 < NS) CR_READ(pf[(step + 2) % 3], sb + CR_PADDR((step + 2) % (R + 2), (step + 2) / (R + 2)));
        __builtin_amdgcn_sched_barrier(0);
        const bool refill = v + 1 < 3 && ri >= R - 1;
        const int ur = ri - (R - 1);                              // the filter row whose last use is this patch row
        if (refill) {
          // its MFMAs first (the last output row), then the refill reads, then the other filter rows
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) CR_MMA(acc[fn][R - 1], wf[ur][fn], pf[step % 3]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) CR_READ(wf[ur][fn], sb + CR_AADDR(fn) + (ur * 3 + v + 1) * BN * 64);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int fm = ri - u;
          if (fm < 0 || fm > R - 1 || (refill && u == ur)) continue;
#pragma unroll
          for (int fn = 0; fn < FN; ++fn) CR_MMA(acc[fn][fm], wf[u][fn], pf[step % 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (step < NPJ + NWJ) {
          issue_one(fl, step);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    static_assert(NPJ + NWJ <= NS, "one fill per step");
  };

  // ---- tile sequence: tile 0 = own index, tile 1 = index + grid, further ones 2 x grid + an atomic ticket (seg_conv_desc.sched);
  // wave 0 names tile k + 2 in ring[(k + 2) & 3] while tile k is OPENED -- at least one barrier before anybody reads it
  int* sched = reinterpret_cast<int*>(d.sched);
  auto announce = [&](int k, int cur_id) {                        // (wave 0) names tile k + 2
    int id;
    if (sched) {
      int vt = 0;
      if (lane == 0) vt = __hip_atomic_fetch_add(sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      id = 2 * G + __builtin_amdgcn_readfirstlane(vt);
    } else id = cur_id + 2 * G;
    if (lane == 0) ring[(k + 2) & 3] = id < P.ntiles ? id : -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  int tile = vb, k = 0, seq = 0;
  if (tile >= P.ntiles) tile = -1;
  Tile cur = decode(tile < 0 ? 0 : tile);
  if (tile >= 0) {
    setup(cur);
    issue(0, cur.n0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  while (tile >= 0) {
    if (wave == 0) announce(k, tile);
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int fm = 0; fm < R; ++fm) acc[fn][fm] = f32x4{0, 0, 0, 0};
    int next = -1;
    Tile nt = cur;
    for (int c = 0; c < nch; ++c, ++seq) {
      // this wave's fills of chunk c have landed: waited for at the end of the previous chunk (below) -- for a tile's first chunk
      // BEFORE the previous tile's epilogue, so that its stores need not have completed here
      __builtin_amdgcn_s_barrier();                               // everybody's fills have landed, and nobody reads the other stage any more
      CRSTAMPN();
      const int other = (seq + 1) & 1;
      Fill fl;
      if (c + 1 < nch) fl = fill_ctx(c + 1, cur.n0, other, true);
      else {
        // last chunk of this tile: the next tile's first stage flies during these MFMAs and the epilogue
        next = k == 0 ? vb + G : __builtin_amdgcn_readfirstlane(ring[(k + 1) & 3]);
        if (next >= P.ntiles) next = -1;
        if (next >= 0) { nt = decode(next); setup(nt); }
        fl = fill_ctx(0, nt.n0, other, next >= 0);
      }
      if (!(P.abl & 4)) compute(smem + (seq & 1) * STAGE, fl);
      else {
#pragma unroll
        for (int i = 0; i < NPJ + NWJ; ++i) issue_one(fl, i);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the fills of the next chunk (issued >= 7 steps of MFMAs ago)
    }
    CRSTAMPN();

    // ---- epilogue: bias, ReLU, accumulate, ReLU-grad mask; a lane stores 8 consecutive channels of one pixel ----
    {
      // (the epilogue's descriptor fields are re-read from the kernel arguments here: hoisted out of the tile loop they cost ~200
      // spilled SGPRs, i.e. v_readlane traffic inside the chunk loop)
#if defined(__HIP_DEVICE_COMPILE__)
      const __attribute__((address_space(4))) CrK* pp = (const __attribute__((address_space(4))) CrK*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(pp));
      const __attribute__((address_space(4))) seg_conv_desc& d = pp->d;
#endif
#define CR_VIEW(dst_, src_) do { dst_.ptr = src_.ptr; dst_.H = src_.H; dst_.W = src_.W; dst_.cs = src_.cs; dst_.coff = src_.coff; dst_.oy = src_.oy; dst_.ox = src_.ox; dst_.c = src_.c; } while (0)
      seg_view dst, msk;
      if (d.n_split > 0 && cur.n0 >= d.n_split) { CR_VIEW(dst, d.dst1); dst.coff -= d.n_split; CR_VIEW(msk, d.mask1); msk.coff -= d.n_split; }
      else { CR_VIEW(dst, d.dst); CR_VIEW(msk, d.mask); }
      const bool has_mask = msk.ptr != nullptr, has_acc = d.accum != 0;
      const float lo = d.relu ? 0.f : -INFINITY;
      const int64_t dbase = ((int64_t)(cur.b * dst.H + dst.oy) * dst.W + dst.ox) * dst.cs + dst.coff;
      const T* mbase = reinterpret_cast<const T*>(msk.ptr) + ((int64_t)(cur.b * msk.H + msk.oy) * msk.W + msk.ox) * msk.cs + msk.coff;
      const int ox = cur.ox0 + wc * 16 + lr;
      const int oyb = cur.oy0 + wr * R;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nl = cur.n0 + wn * (FN * 16) + j * 32 + 8 * g;
        if (nl >= d.n_count) continue;
        float bv[8];
        {
          const int np = d.n_off + nl;
          if (d.bias != nullptr && np + 8 <= d.bias_n) {           // (bias rows start on 16-byte boundaries of the arena: np is a multiple of 8)
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + np), b1 = *reinterpret_cast<const f32x4*>(d.bias + np + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = (d.bias != nullptr && np + e < d.bias_n) ? d.bias[np + e] : 0.f;
          }
        }
        Vec8<T> mk[R], old[R];
        int poff[R];
#pragma unroll
        for (int fm = 0; fm < R; ++fm) {
          const int oy = oyb + fm;
          const bool ok = oy < d.Ho && ox < d.Wo;
          poff[fm] = ok ? (oy * dst.W + ox) * dst.cs : -1;
          if (has_mask && ok) mk[fm].load(mbase + (oy * msk.W + ox) * msk.cs + nl);
          if (has_acc && ok) old[fm].load(reinterpret_cast<const T*>(dst.ptr) + dbase + poff[fm] + nl);
        }
        T* pbase = nullptr;
        if constexpr (POOL) pbase = reinterpret_cast<T*>(d.pool.ptr) + ((int64_t)(cur.b * d.pool.H + d.pool.oy) * d.pool.W + d.pool.ox) * d.pool.cs + d.pool.coff;
        // row pairs: two output rows are finished, stored and (POOL) reduced before the next two are touched (few live registers)
#pragma unroll
        for (int hp = 0; hp < R / 2; ++hp) {
          Vec8<T> o[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int fm = 2 * hp + h2;
            float vv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              vv[e] = fmaxf(acc[2 * j][fm][e] + bv[e], lo);
              vv[4 + e] = fmaxf(acc[2 * j + 1][fm][e] + bv[4 + e], lo);
            }
            if (has_acc && poff[fm] >= 0) {
#pragma unroll
              for (int e = 0; e < 8; ++e) vv[e] += old[fm].get(e);
            }
            if (has_mask && poff[fm] >= 0) {
#pragma unroll
              for (int e = 0; e < 8; ++e) vv[e] = mk[fm].get(e) > 0.f ? vv[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[h2].set(e, vv[e]);
            if (poff[fm] >= 0) o[h2].store(reinterpret_cast<T*>(dst.ptr) + dbase + poff[fm] + nl);
          }
          if constexpr (POOL) {
            // fused 2x2/s2 VALID max-pool (slim.max_pool2d behind the layer): window = fragment rows (2 hp, 2 hp + 1) x lanes (lr, lr^1);
            // pooling the bf16-ROUNDED values gives the bits of the separate pool kernel
            Vec8<T> m;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float t = fmaxf(o[0].get(e), o[1].get(e));
              t = fmaxf(t, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true)));   // lane ^ 1
              m.set(e, t);
            }
            const int py = (oyb >> 1) + hp, px = ox >> 1;
            if ((lr & 1) == 0 && py < d.pool_h && px < d.pool_w) m.store(pbase + ((int64_t)py * d.pool.W + px) * d.pool.cs + nl);
          }
        }
      }
    }
    CRSTAMPN();
    tile = next; cur = nt; ++k;
  }
  // the last workgroup to leave resets the ticket counter for the next launch that uses this slot
  if (sched && tid == 0) {
    const int done = __hip_atomic_fetch_add(sched + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == G - 1) {
      __hip_atomic_store(sched, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sched + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int NWAVE, int R, int CHW, int FN, int NPIXCAP, bool POOL>
int cr_launch(const CrK& P, char* name_out, int name_cap, hipStream_t st) {
  constexpr int BN = CHW * FN * 16;
  constexpr int LDS = 2 * (NPIXCAP * 64 + 9 * BN * 64) + 64;
  if (name_out) { snprintf(name_out, name_cap, "conv_ring_kernel<%d,%d,%d,%d,%d,%s>", NWAVE, R, CHW, FN, NPIXCAP, POOL ? "true" : "false"); return SEG_OK; }
  auto kern = conv_ring_kernel<NWAVE, R, CHW, FN, NPIXCAP, POOL>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      seg_set_error("conv_ring: cannot raise dynamic LDS to %d", LDS); return SEG_ERR_LAUNCH;
    }
    attr_done = true;
  }
  static int ncu = 0;
  if (ncu == 0) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
  int grid = P.ntiles < ncu ? P.ntiles : ncu;
  SEG_LAUNCH(kern, dim3(grid), dim3(64 * NWAVE), LDS, st, P);
  return seg_check_launch("conv_ring");
}

struct CrShape { int WR, WC; };

}  // namespace

#ifdef SEG_STAMPS
extern "C" int seg_dbg_set_crstamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_crstamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// 1 = handled (rc holds the result), 0 = not eligible (the caller runs another kernel)
// cfg 208 / 204: forced tile class (512 / 256 pixels x 64 channels); cfg 0 with SEG_CONV_IMPL=ring or the library default: cost model.
int seg_conv_ring(const seg_conv_desc& d, char* name_out, int name_cap, hipStream_t st, int* rc) {
  static const char* impl = getenv("SEG_CONV_IMPL");
  static const int impl_mode = impl ? (!strcmp(impl, "ring") ? 1 : (!strcmp(impl, "tiled") ? -1 : 0)) : 0;
  const bool forced = d.cfg == 208 || d.cfg == 204;
  if (!forced && (d.cfg != 0 || impl_mode <= 0)) return 0;
  if (d.dtype != SEG_BF16 || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.up2 || d.n_store || d.thin_src || d.out_f32 || d.ksplit > 1) return 0;
  if (d.n_count % 64 || d.n_split % 64) return 0;             // 64-channel blocks (the 32-channel layers stay on the tiled kernel)
  if (d.pool.ptr && (d.accum || d.n_split || d.mask.ptr)) return 0;
  auto small = [](const seg_view& v) { return !v.ptr || (int64_t)v.H * v.W * v.cs < ((int64_t)1 << 30); };
  if (!small(d.src0) || !small(d.src1) || !small(d.dst) || !small(d.dst1) || !small(d.mask) || !small(d.mask1) || !small(d.pool)) return 0;
  const int nchunks = (d.src0.c + (d.src1.ptr ? d.src1.c : 0)) / 32;
  if (!forced && !((int64_t)d.B * d.Ho * d.Wo >= 200000 && nchunks <= 4 && d.n_count <= 128)) return 0;   // where it measured faster (profiles/r04_ring_micro512.txt)
  if ((int64_t)9 * nchunks * 32 * d.n_total * 2 >= ((int64_t)1 << 31)) return 0;
  // tile class and shape: least (rounds of tiles over the chip) x (time of a tile).  Classes: 0 = 8 waves x 4 rows, 512 px (cfg 208);
  // 1 = 8 waves = 4 pixel x 2 channel groups x 4 rows, 256 px (cfg 204); 2 = 4 waves x 8 rows, 512 px (cfg 209)
  static const int PXWs[3] = {8, 4, 4}, RWs[3] = {4, 4, 8}, CAPs[3] = {672, 400, 672};
  static const double t_chunk[3] = {3.7, 2.2, 4.8};              // us per 32-channel chunk (r04 stamps)
  int best_cls = -1; CrShape best_s = {0, 0}; double best_cost = 1e30;
  const int nblk = d.n_count / 64;
  for (int cls = 0; cls < 2; ++cls) {                           // (class 2 = 4 waves x 8 rows, one wave per SIMD, measured 4.8 us per chunk against 3.7: not built)
    if (d.cfg == 208 && cls != 0) continue;
    if (d.cfg == 204 && cls != 1) continue;
    const int pxw = PXWs[cls];
    for (int wr = 1; wr <= pxw; wr *= 2) {
      const int wc = pxw / wr;
      const int TH = RWs[cls] * wr, TW = 16 * wc;
      if ((TH + 2) * (TW + 2) > CAPs[cls]) continue;
      const long tiles = (long)d.B * cdiv(d.Ho, TH) * cdiv(d.Wo, TW) * nblk;
      const double rounds = (double)((tiles + 255) / 256);
      // a workgroup's first tile pays ~4 us until its first stage has landed; every tile ~3 us of epilogue and tile switch
      const double cost = 4.0 + rounds * (nchunks * t_chunk[cls] + 3.0);
      if (cost < best_cost) { best_cost = cost; best_cls = cls; best_s.WR = wr; best_s.WC = wc; }
    }
  }
  if (best_cls < 0) return 0;
  CrK P;
  P.d = d;
  if (!d.src1.ptr) { P.d.src1 = d.src0; P.d.src1.c = 0; }
  static const int RW2[3] = {4, 4, 8};
  P.WR = best_s.WR; P.WC = best_s.WC; P.TH = RW2[best_cls] * P.WR; P.TW = 16 * P.WC; P.PW = P.TW + 2; P.NPATCH = (P.TH + 2) * P.PW;
  P.m_pw = (uint32_t)((((uint64_t)1) << 32) / (uint64_t)P.PW + 1);
  P.tiles_x = cdiv(d.Wo, P.TW); P.tiles_y = cdiv(d.Ho, P.TH); P.nblk = nblk; P.ntiles = d.B * P.tiles_y * P.tiles_x * nblk;
  P.nchunks0 = d.src0.c / 32; P.nchunks = nchunks;
#ifdef SEG_STAMPS
  static const int abl = getenv("SEG_RING_ABL") ? atoi(getenv("SEG_RING_ABL")) : 0;      // (debug builds: tools/stamp_ring.py)
  P.abl = abl;
#else
  P.abl = 0;
#endif
  const bool pool = d.pool.ptr != nullptr;
  if (best_cls == 0) *rc = pool ? cr_launch<8, 4, 1, 4, 672, true>(P, name_out, name_cap, st) : cr_launch<8, 4, 1, 4, 672, false>(P, name_out, name_cap, st);
  else *rc = pool ? cr_launch<8, 4, 2, 2, 400, true>(P, name_out, name_cap, st) : cr_launch<8, 4, 2, 2, 400, false>(P, name_out, name_cap, st);
  return 1;
}
