// common.h -- shared device helpers for the gfx950 kernels (wave64, MFMA 16x16, LDS layouts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/seg_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SEG_DEV __device__ __forceinline__
// hipGetLastError() is sticky across unrelated runtime calls: clear it before every launch we check.
#define SEG_LAUNCH(k, ...) do { (void)hipGetLastError(); seg_note_kernel(#k); hipLaunchKernelGGL(k, __VA_ARGS__); } while (0)

void seg_set_error(const char* fmt, ...);
const char* seg_env(const char* name);                      // getenv, read once per name (elementwise.hip)
void seg_note_kernel(const char* launch_site_spelling);     // -> seg_last_kernel_name() (thread-local)
int seg_check_launch(const char* what);

// ---------------------------------------------------------------------------------------------
// Per-dtype traits.  A "row" is 32 K-elements of one pixel (or one packed weight row); a "piece"
// is 16 bytes of it.  LDS rows are laid out so that the MFMA operand read -- lane l reads the
// 8 K-elements [8*(l>>4), 8*(l>>4)+8) of row (l&15) -- is bank-conflict free for ds_read_b128:
//   bf16: 64-B rows, piece index XOR-swizzled with bit 2 of the row index (brute-forced over the
//         ds_read_b128 lane groups of gfx950: conflict-free at all 9 tap shifts for 16-wide tiles);
//   f32 : 128-B rows padded to 144 B (2-way; parity mode only).
// ---------------------------------------------------------------------------------------------
template <typename T> struct Tr;
template <> struct Tr<bf16_t> {
  static constexpr int DT = SEG_BF16;
  static constexpr int PIECES = 4;      // 16-B pieces per 32-element row
  static constexpr int EPP = 8;         // elements per piece
  static constexpr int RSTR = 64;       // LDS row stride, bytes
  static SEG_DEV int lds_off(int row, int piece) { return row * 64 + ((piece ^ ((row >> 1) & 2)) << 4); }
};
template <> struct Tr<float> {
  static constexpr int DT = SEG_F32;
  static constexpr int PIECES = 8;
  static constexpr int EPP = 4;
  static constexpr int RSTR = 144;
  static SEG_DEV int lds_off(int row, int piece) { return row * 144 + (piece << 4); }
};

// Kernel templates take the dtype as an INT (SEG_F32 / SEG_BF16) rather than a type: a `__bf16` template argument
// mangles to DF16b, which GNU/rocprof demanglers garble, and the kernel names must stay readable in rocprofv3 output.
template <int DT> struct DtSel;
template <> struct DtSel<SEG_F32> { typedef float type; };
template <> struct DtSel<SEG_BF16> { typedef bf16_t type; };

// MFMA operand fragment: the 8 K-elements a lane owns.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { f32x4 lo, hi; };

// lane-group g = lane>>4 owns K-elements [8g, 8g+8) of the 32-chunk.
template <typename T> SEG_DEV Frag<T> lds_read_frag(const char* lds_row_base, int row, int g);
template <> SEG_DEV Frag<bf16_t> lds_read_frag<bf16_t>(const char* base, int row, int g) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(base + Tr<bf16_t>::lds_off(row, g));
  return f;
}
template <> SEG_DEV Frag<float> lds_read_frag<float>(const char* base, int row, int g) {
  Frag<float> f;
  const char* p = base + Tr<float>::lds_off(row, 2 * g);
  f.lo = *reinterpret_cast<const f32x4*>(p);
  f.hi = *reinterpret_cast<const f32x4*>(p + 16);
  return f;
}
// Same, from a precomputed byte address (row and piece already folded in).
template <typename T> SEG_DEV Frag<T> lds_read_frag_at(const char* p);
template <> SEG_DEV Frag<bf16_t> lds_read_frag_at<bf16_t>(const char* p) {
  Frag<bf16_t> f; f.v = *reinterpret_cast<const bf16x8*>(p); return f;
}
template <> SEG_DEV Frag<float> lds_read_frag_at<float>(const char* p) {
  Frag<float> f; f.lo = *reinterpret_cast<const f32x4*>(p); f.hi = *reinterpret_cast<const f32x4*>(p + 16); return f;
}
template <typename T> SEG_DEV int frag_addr(int row, int g);
template <> SEG_DEV int frag_addr<bf16_t>(int row, int g) { return Tr<bf16_t>::lds_off(row, g); }
template <> SEG_DEV int frag_addr<float>(int row, int g) { return Tr<float>::lds_off(row, 2 * g); }

// D(16x16) += A(16 x 32) * B(32 x 16): A row = lane&15, B col = lane&15, K slice by lane>>4.
// D: col = lane&15, row = 4*(lane>>4) + reg.
SEG_DEV void mma32(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
SEG_DEV void mma32(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
  // exact-f32 MFMA (k-ordered fmaf chain); K step s uses element s of each lane-group's 8.
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[0], b.lo[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[1], b.lo[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[2], b.lo[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[3], b.lo[3], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[0], b.hi[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[1], b.hi[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[2], b.hi[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[3], b.hi[3], acc, 0, 0, 0);
}

SEG_DEV float to_f32(float x) { return x; }
SEG_DEV float to_f32(bf16_t x) { return (float)x; }
template <typename T> SEG_DEV T from_f32(float x);
template <> SEG_DEV float from_f32<float>(float x) { return x; }
template <> SEG_DEV bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// 8 consecutive elements <-> registers
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  bf16x8 v;
  SEG_DEV void load(const void* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  SEG_DEV void store(void* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
  SEG_DEV float get(int i) const { return (float)v[i]; }
  SEG_DEV void set(int i, float x) { v[i] = (bf16_t)x; }
  SEG_DEV void zero() { for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f; }
};
template <> struct Vec8<float> {
  f32x4 lo, hi;
  SEG_DEV void load(const void* p) { lo = reinterpret_cast<const f32x4*>(p)[0]; hi = reinterpret_cast<const f32x4*>(p)[1]; }
  SEG_DEV void store(void* p) const { reinterpret_cast<f32x4*>(p)[0] = lo; reinterpret_cast<f32x4*>(p)[1] = hi; }
  SEG_DEV float get(int i) const { return i < 4 ? lo[i] : hi[i - 4]; }
  SEG_DEV void set(int i, float x) { if (i < 4) lo[i] = x; else hi[i - 4] = x; }
  SEG_DEV void zero() { lo = f32x4{0, 0, 0, 0}; hi = f32x4{0, 0, 0, 0}; }
};

// element offset of logical pixel (b,y,x) channel 0 in a view
SEG_DEV int64_t view_off(const seg_view& v, int b, int y, int x) {
  return ((int64_t)(b * v.H + y + v.oy) * v.W + (x + v.ox)) * v.cs + v.coff;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
