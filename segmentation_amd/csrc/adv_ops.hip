// adv_ops.hip -- the ops adversarial segmentation training needs beyond the model kernels
// (/root/reference/models/basemodel.py:215-355): one-hot "real" maps and softmax "fake" maps for the
// adversary, the softmax backward that carries the adversary's input gradient into the segmentation logits, flatten
// between the padded NHWC layout and feature rows, batch norm over feature rows (slim.batch_norm on a [B, F] tensor: any F),
// and the 2-class cross-entropy of the adversary's verdicts.  The adversary's convolutions, pools, resize and dense layers
// run on the kernels of deconv_ops.hip.  HBM-bound byte movers: coalesced 16-byte access over channels / features.
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

// one_hot(labels) over a label window -> activation (pad channels 0).  models/basemodel.py:283 (`input_y` to the adversary)
template <typename T>
__global__ void onehot_kernel(const uint8_t* labels, int LH, int LW, int ly0, int lx0, int B, int H, int W, seg_view dst, int C8) {
  const int64_t total = (int64_t)B * H * W * C8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8); int64_t t = i / C8;
    const int x = t % W; t /= W;
    const int y = t % H; const int b = t / H;
    const int lab = labels[((int64_t)b * LH + y + ly0) * LW + x + lx0];
    Vec8<T> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.set(e, c8 * 8 + e == lab ? 1.f : 0.f);
    o.store(reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, y, x) + c8 * 8);
  }
}

// softmax over the class axis of float logits -> activation (pad channels 0).  models/basemodel.py:285 (`y_hat` to the adversary)
// (NCP = classes rounded up to 4 / 8 / 16 / 32 bounds the unrolled loops; the float logits are read as 16-byte vectors)
template <int NCP>
SEG_DEV float softmax_row(const float* z, int nc, float (&zv)[NCP]) {
  float m = -INFINITY;
#pragma unroll
  for (int c4 = 0; c4 < NCP / 4; ++c4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(z + c4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int c = c4 * 4 + e; zv[c] = c < nc ? v[e] : -INFINITY; m = fmaxf(m, zv[c]); }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCP; ++c) { zv[c] = c < nc ? expf(zv[c] - m) : 0.f; s += zv[c]; }
  return 1.f / s;
}

template <typename T, int NCP>
__global__ void softmax_probs_kernel(seg_view lg, int B, int H, int W, int nc, seg_view dst, int C8) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int x = t % W; t /= W;
    const int y = t % H; const int b = t / H;
    float zv[NCP];
    const float rs = softmax_row<NCP>(reinterpret_cast<const float*>(lg.ptr) + view_off(lg, b, y, x), nc, zv);
    T* o = reinterpret_cast<T*>(dst.ptr) + view_off(dst, b, y, x);
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      if (c8 >= C8) break;
      Vec8<T> ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov.set(e, c8 * 8 + e < NCP ? zv[c8 * 8 + e < NCP ? c8 * 8 + e : 0] * rs : 0.f);
      ov.store(o + c8 * 8);
    }
  }
}

// dlogits += scale * p * (dp - sum_c dp_c p_c): the adversary's input gradient through the softmax, added to the
// x-entropy gradient already in dlogits.  (p is recomputed from the float logits: the stored copy is rounded.)
template <typename T, int NCP>
__global__ void softmax_bwd_add_kernel(seg_view lg, seg_view dp, int B, int H, int W, int nc, float scale, seg_view dl, int C8) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int x = t % W; t /= W;
    const int y = t % H; const int b = t / H;
    float zv[NCP];
    const float rs = softmax_row<NCP>(reinterpret_cast<const float*>(lg.ptr) + view_off(lg, b, y, x), nc, zv);
    const T* g = reinterpret_cast<const T*>(dp.ptr) + view_off(dp, b, y, x);
    float gv[NCP];
    float dot = 0.f;
#pragma unroll
    for (int c8 = 0; c8 < (NCP + 7) / 8; ++c8) {
      Vec8<T> v; v.load(g + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        if (c < NCP) { gv[c] = v.get(e); zv[c] *= rs; dot = fmaf(gv[c], zv[c], dot); }
      }
    }
    T* o = reinterpret_cast<T*>(dl.ptr) + view_off(dl, b, y, x);
#pragma unroll
    for (int c8 = 0; c8 < (NCP + 7) / 8; ++c8) {
      Vec8<T> ov; ov.load(o + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        if (c < NCP && c < nc) ov.set(e, ov.get(e) + scale * zv[c < NCP ? c : 0] * (gv[c < NCP ? c : 0] - dot));
      }
      ov.store(o + c8 * 8);
    }
  }
}

// slim.flatten: [B,H,W,C logical] (padded NHWC) <-> feature rows [B, F = H*W*C] (padded to Fp); MODE 0 gather, 1 scatter back
// (pad channels of the NHWC side are written 0 by the scatter)
template <typename T, int MODE>
__global__ void flatten_kernel(seg_view a, int B, int H, int W, int C, int Cp, seg_view f, int F, int Fp) {
  if (MODE == 0) {
    const int64_t total = (int64_t)B * Fp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int j = (int)(i % Fp), b = (int)(i / Fp);
      float v = 0.f;
      if (j < F) {
        const int c = j % C, p = j / C;
        v = to_f32(reinterpret_cast<const T*>(a.ptr)[view_off(a, b, p / W, p % W) + c]);
      }
      reinterpret_cast<T*>(f.ptr)[view_off(f, b, 0, 0) + j] = from_f32<T>(v);
    }
  } else {
    const int64_t total = (int64_t)B * H * W * Cp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int c = (int)(i % Cp); int64_t t = i / Cp;
      const int x = t % W; t /= W;
      const int y = t % H; const int b = t / H;
      float v = 0.f;
      if (c < C) v = to_f32(reinterpret_cast<const T*>(f.ptr)[view_off(f, b, 0, 0) + (y * W + x) * C + c]);
      reinterpret_cast<T*>(a.ptr)[view_off(a, b, y, x) + c] = from_f32<T>(v);
    }
  }
}

// slim.batch_norm (beta only, training mode) over feature rows [B, Fp]: one thread per feature walks the B rows (coalesced
// across the wave, fixed order, double-precision sums).  stats = [mean | rstd]; moving = [mean | variance].
template <typename T>
__global__ void bn_rows_fwd_kernel(seg_view a, seg_view y, const float* beta, float* moving, float* stats, int B, int Fp, int F, float decay, float eps) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Fp) return;
  const T* ap = reinterpret_cast<const T*>(a.ptr) + view_off(a, 0, 0, 0) + j;
  const int64_t rs = (int64_t)a.H * a.W * a.cs;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < B; ++b) { const double v = (double)to_f32(ap[b * rs]); s1 += v; s2 += v * v; }
  const double m = s1 / B;
  double vv = s2 / B - m * m; if (vv < 0.0) vv = 0.0;
  float mean = (float)m, var = (float)vv;
  if (j < F && moving != nullptr) {
    moving[j] = decay * moving[j] + (1.f - decay) * mean;
    moving[Fp + j] = decay * moving[Fp + j] + (1.f - decay) * var;
  }
  if (j >= F) { mean = 0.f; var = 1.f; }
  const float rstd = 1.f / sqrtf(var + eps), bt = j < F ? beta[j] : 0.f;
  stats[j] = mean; stats[Fp + j] = rstd;
  T* yp = reinterpret_cast<T*>(y.ptr) + view_off(y, 0, 0, 0) + j;
  const int64_t ys = (int64_t)y.H * y.W * y.cs;
  for (int b = 0; b < B; ++b) yp[b * ys] = from_f32<T>(j < F ? (to_f32(ap[b * rs]) - mean) * rstd + bt : 0.f);
}

// gradient of the above wrt its input and beta; relu_mask: the input is a ReLU output and dz is also gated by a > 0
template <typename T>
__global__ void bn_rows_bwd_kernel(seg_view a, seg_view dy, seg_view dz, const float* stats, float* dbeta, int dbeta_add, int B, int Fp, int F, int relu_mask) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Fp) return;
  const T* ap = reinterpret_cast<const T*>(a.ptr) + view_off(a, 0, 0, 0) + j;
  const T* gp = reinterpret_cast<const T*>(dy.ptr) + view_off(dy, 0, 0, 0) + j;
  T* zp = reinterpret_cast<T*>(dz.ptr) + view_off(dz, 0, 0, 0) + j;
  const int64_t as = (int64_t)a.H * a.W * a.cs, gs = (int64_t)dy.H * dy.W * dy.cs, zs = (int64_t)dz.H * dz.W * dz.cs;
  const float mean = stats[j], rstd = stats[Fp + j];
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < B; ++b) { const double g = (double)to_f32(gp[b * gs]); s1 += g; s2 += g * (double)((to_f32(ap[b * as]) - mean) * rstd); }
  if (j < F) dbeta[j] = dbeta_add ? dbeta[j] + (float)s1 : (float)s1;
  const float m1 = (float)(s1 / B), m2 = (float)(s2 / B);
  for (int b = 0; b < B; ++b) {
    const float av = to_f32(ap[b * as]), xh = (av - mean) * rstd;
    float v = rstd * (to_f32(gp[b * gs]) - m1 - xh * m2);
    if (j >= F || (relu_mask && !(av > 0.f))) v = 0.f;
    zp[b * zs] = from_f32<T>(v);
  }
}

// tf.nn.softmax_cross_entropy_with_logits of the adversary's 2 logits per image against one_hot(label): mean over the batch
// stored (not added) to *loss_out, gradient of that mean times gscale to dl.  One wave, fixed order.  basemodel.py:291-297
template <typename T>
__global__ __launch_bounds__(64) void bce2_kernel(seg_view lg, int B, int label, float gscale, float* loss_out, seg_view dl) {
  float local = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) {
    const T* z = reinterpret_cast<const T*>(lg.ptr) + view_off(lg, b, 0, 0);
    const float z0 = to_f32(z[0]), z1 = to_f32(z[1]);
    const float m = fmaxf(z0, z1), e0 = expf(z0 - m), e1 = expf(z1 - m), s = e0 + e1;
    local += logf(s) - ((label ? z1 : z0) - m);
    Vec8<T> o; o.zero();
    o.set(0, (e0 / s - (label == 0 ? 1.f : 0.f)) * gscale / B);
    o.set(1, (e1 / s - (label == 1 ? 1.f : 0.f)) * gscale / B);
    T* op = reinterpret_cast<T*>(dl.ptr) + view_off(dl, b, 0, 0);
    o.store(op);
    Vec8<T> zr; zr.zero();
    for (int c8 = 1; c8 < dl.c / 8; ++c8) zr.store(op + c8 * 8);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
  if (threadIdx.x == 0) *loss_out = local / B;
}

}  // namespace

#define ADV_DISPATCH(dtype, F32, BF16, what) do { if ((dtype) == SEG_F32) { F32; } else if ((dtype) == SEG_BF16) { BF16; } \
    else { seg_set_error(what ": bad dtype"); return SEG_ERR_ARG; } } while (0)

extern "C" int seg_onehot(const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0, int32_t B, int32_t H, int32_t W,
                          const seg_view* dst, int32_t dtype, void* stream) {
  if (!labels || !dst || !view_ok(*dst, H, W, dst->c) || ly0 < 0 || lx0 < 0 || ly0 + H > LH || lx0 + W > LW) { seg_set_error("onehot: bad arguments"); return SEG_ERR_ARG; }
  const int C8 = dst->c / 8;
  const int g = grid_for((int64_t)B * H * W * C8);
  hipStream_t st = (hipStream_t)stream;
  ADV_DISPATCH(dtype, SEG_LAUNCH(onehot_kernel<float>, dim3(g), dim3(256), 0, st, labels, LH, LW, ly0, lx0, B, H, W, *dst, C8),
               SEG_LAUNCH(onehot_kernel<bf16_t>, dim3(g), dim3(256), 0, st, labels, LH, LW, ly0, lx0, B, H, W, *dst, C8), "onehot");
  return seg_check_launch("onehot");
}

extern "C" int seg_softmax_probs(const seg_view* logits, int32_t B, int32_t H, int32_t W, int32_t n_classes, const seg_view* dst, int32_t dtype, void* stream) {
  if (!logits || !logits->ptr || !dst || !view_ok(*dst, H, W, dst->c) || n_classes < 1 || n_classes > 32 || n_classes > dst->c || dst->c > 32 || n_classes > logits->cs) {
    seg_set_error("softmax_probs: bad arguments (1..32 classes)"); return SEG_ERR_ARG;
  }
  const int g = grid_for((int64_t)B * H * W);
  hipStream_t st = (hipStream_t)stream;
#define SMX(TT, NCP) SEG_LAUNCH((softmax_probs_kernel<TT, NCP>), dim3(g), dim3(256), 0, st, *logits, B, H, W, n_classes, *dst, dst->c / 8)
#define SMX_N(TT) do { if (n_classes <= 4) SMX(TT, 4); else if (n_classes <= 8) SMX(TT, 8); else if (n_classes <= 16) SMX(TT, 16); else SMX(TT, 32); } while (0)
  ADV_DISPATCH(dtype, SMX_N(float), SMX_N(bf16_t), "softmax_probs");
#undef SMX_N
#undef SMX
  return seg_check_launch("softmax_probs");
}

extern "C" int seg_softmax_bwd_add(const seg_view* logits, const seg_view* dprobs, int32_t B, int32_t H, int32_t W, int32_t n_classes, float scale,
                                   const seg_view* dlogits, int32_t dtype, void* stream) {
  if (!logits || !logits->ptr || !dprobs || !dlogits || !view_ok(*dprobs, H, W, dprobs->c) || !view_ok(*dlogits, H, W, dlogits->c) || n_classes < 1 ||
      n_classes > 32 || n_classes > dlogits->c || dlogits->c > 32 || dprobs->c != dlogits->c || n_classes > logits->cs) {
    seg_set_error("softmax_bwd_add: bad arguments (1..32 classes)"); return SEG_ERR_ARG;
  }
  const int g = grid_for((int64_t)B * H * W);
  hipStream_t st = (hipStream_t)stream;
#define SMB(TT, NCP) SEG_LAUNCH((softmax_bwd_add_kernel<TT, NCP>), dim3(g), dim3(256), 0, st, *logits, *dprobs, B, H, W, n_classes, scale, *dlogits, dlogits->c / 8)
#define SMB_N(TT) do { if (n_classes <= 4) SMB(TT, 4); else if (n_classes <= 8) SMB(TT, 8); else if (n_classes <= 16) SMB(TT, 16); else SMB(TT, 32); } while (0)
  ADV_DISPATCH(dtype, SMB_N(float), SMB_N(bf16_t), "softmax_bwd_add");
#undef SMB_N
#undef SMB
  return seg_check_launch("softmax_bwd_add");
}

extern "C" int seg_flatten(const seg_view* a, int32_t B, int32_t H, int32_t W, int32_t C, const seg_view* f, int32_t backward, int32_t dtype, void* stream) {
  if (!a || !f || C < 1 || !view_ok(*a, H, W, a->c) || C > a->c || !view_ok(*f, 1, 1, f->c) || H * W * C > f->c) { seg_set_error("flatten: bad arguments"); return SEG_ERR_ARG; }
  const int F = H * W * C, Fp = f->c, Cp = a->c;
  hipStream_t st = (hipStream_t)stream;
  if (!backward) {
    const int g = grid_for((int64_t)B * Fp);
    ADV_DISPATCH(dtype, SEG_LAUNCH((flatten_kernel<float, 0>), dim3(g), dim3(256), 0, st, *a, B, H, W, C, Cp, *f, F, Fp),
                 SEG_LAUNCH((flatten_kernel<bf16_t, 0>), dim3(g), dim3(256), 0, st, *a, B, H, W, C, Cp, *f, F, Fp), "flatten");
  } else {
    const int g = grid_for((int64_t)B * H * W * Cp);
    ADV_DISPATCH(dtype, SEG_LAUNCH((flatten_kernel<float, 1>), dim3(g), dim3(256), 0, st, *a, B, H, W, C, Cp, *f, F, Fp),
                 SEG_LAUNCH((flatten_kernel<bf16_t, 1>), dim3(g), dim3(256), 0, st, *a, B, H, W, C, Cp, *f, F, Fp), "flatten");
  }
  return seg_check_launch("flatten");
}

extern "C" int seg_bn_rows_fwd(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, int32_t B, int32_t F,
                               float decay, float eps, int32_t dtype, void* stream) {
  if (!a || !y || !beta || !stats || B < 1 || F < 1 || !view_ok(*a, 1, 1, a->c) || !view_ok(*y, 1, 1, y->c) || a->c != y->c || F > a->c) { seg_set_error("bn_rows_fwd: bad arguments"); return SEG_ERR_ARG; }
  const int Fp = a->c, g = (Fp + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  ADV_DISPATCH(dtype, SEG_LAUNCH(bn_rows_fwd_kernel<float>, dim3(g), dim3(256), 0, st, *a, *y, beta, moving, stats, B, Fp, F, decay, eps),
               SEG_LAUNCH(bn_rows_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, *a, *y, beta, moving, stats, B, Fp, F, decay, eps), "bn_rows_fwd");
  return seg_check_launch("bn_rows_fwd");
}

extern "C" int seg_bn_rows_bwd(const seg_view* a, const seg_view* dy, const seg_view* dz, const float* stats, float* dbeta, int32_t dbeta_add, int32_t B,
                               int32_t F, int32_t relu_mask, int32_t dtype, void* stream) {
  if (!a || !dy || !dz || !stats || !dbeta || B < 1 || F < 1 || !view_ok(*a, 1, 1, a->c) || !view_ok(*dy, 1, 1, a->c) || !view_ok(*dz, 1, 1, a->c) ||
      dy->c != a->c || dz->c != a->c || F > a->c) { seg_set_error("bn_rows_bwd: bad arguments"); return SEG_ERR_ARG; }
  const int Fp = a->c, g = (Fp + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  ADV_DISPATCH(dtype, SEG_LAUNCH(bn_rows_bwd_kernel<float>, dim3(g), dim3(256), 0, st, *a, *dy, *dz, stats, dbeta, dbeta_add, B, Fp, F, relu_mask),
               SEG_LAUNCH(bn_rows_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, *a, *dy, *dz, stats, dbeta, dbeta_add, B, Fp, F, relu_mask), "bn_rows_bwd");
  return seg_check_launch("bn_rows_bwd");
}

extern "C" int seg_bce2(const seg_view* logits, int32_t B, int32_t label, float grad_scale, float* loss_out, const seg_view* dlogits, int32_t dtype, void* stream) {
  if (!logits || !dlogits || !loss_out || B < 1 || (label != 0 && label != 1) || !view_ok(*logits, 1, 1, logits->c) || !view_ok(*dlogits, 1, 1, dlogits->c) ||
      logits->c < 8 || dlogits->c < 8) { seg_set_error("bce2: bad arguments"); return SEG_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  ADV_DISPATCH(dtype, SEG_LAUNCH(bce2_kernel<float>, dim3(1), dim3(64), 0, st, *logits, B, label, grad_scale, loss_out, *dlogits),
               SEG_LAUNCH(bce2_kernel<bf16_t>, dim3(1), dim3(64), 0, st, *logits, B, label, grad_scale, loss_out, *dlogits), "bce2");
  return seg_check_launch("bce2");
}
