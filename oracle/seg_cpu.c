/* seg_cpu.c -- plain-C (OpenMP) restatement of the ops on the U-Net hot path: libseg_cpu.so.
 *
 * TEST INFRASTRUCTURE (see oracle/__init__.py): a third, independent implementation beside oracle/np_ops.py (numpy) and
 * oracle/torch_ref.py (torch autograd).  tests/test_oracle.py checks it against the numpy restatement; bench.py times it as
 * the `cpu_baseline` ("port": the repo's TensorFlow CPU path cannot run here, SURVEY 8(c),(d)).  Nothing under
 * segmentation_amd/ loads it: the product has no CPU path.
 *
 * Semantics (TF-1.x / slim, as listed at the head of SURVEY section 8), float32, NHWC, cross-correlation:
 *   conv            slim.convolution2d            models/unet.py:111-166     filter HWIO [kh,kw,Cin,Cout]
 *   convT2x2s2      slim.convolution2d_transpose  models/unet.py:138-159     filter [2,2,Cout,Cin], stride 2, VALID
 *   maxpool2x2      slim.max_pool2d(x, 2)         models/unet.py:120-132     stride 2, VALID, first maximum wins
 *   softmax_xent    softmax_cross_entropy_with_logits + reduce_mean          models/basemodel.py:59-70,360
 *   adam            tf.train.AdamOptimizer        models/basemodel.py:321    eps outside the bias correction
 * Every function is a straightforward loop nest; the innermost loop runs over contiguous output channels so that gcc
 * vectorises it (-O3 -mavx2 -mfma).  Threads: OpenMP over (image, output row) -- segcpu_set_threads / segcpu_threads.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int segcpu_version(void) { return 100; }

void segcpu_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int segcpu_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* y[b,oy,ox,:] = act( bias + sum_{u,v,c} x[b, oy*s+u-pt, ox*s+v-pl, c] * w[u,v,c,:] ).  x may be a window of a larger
 * buffer: xs_h / xs_w are the buffer's row / pixel strides in floats (dense: W*Cin and Cin). */
void segcpu_conv2d_fwd(const float* x, int64_t xs_b, int64_t xs_h, int64_t xs_w, const float* w, const float* bias, float* y, int B, int H, int W,
                       int Cin, int Cout, int k, int s, int pt, int pl, int Ho, int Wo, int relu) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < Ho; ++oy) {
      float* yr = y + ((int64_t)(b * Ho + oy) * Wo) * Cout;
      for (int ox = 0; ox < Wo; ++ox) {
        float* acc = yr + (int64_t)ox * Cout;
        if (bias) memcpy(acc, bias, sizeof(float) * Cout); else memset(acc, 0, sizeof(float) * Cout);
        for (int u = 0; u < k; ++u) {
          const int iy = oy * s + u - pt;
          if (iy < 0 || iy >= H) continue;
          for (int v = 0; v < k; ++v) {
            const int ix = ox * s + v - pl;
            if (ix < 0 || ix >= W) continue;
            const float* xp = x + b * xs_b + iy * xs_h + ix * xs_w;
            const float* wp = w + ((int64_t)(u * k + v) * Cin) * Cout;
            for (int c = 0; c < Cin; ++c) {
              const float xv = xp[c];
              const float* wr = wp + (int64_t)c * Cout;
              for (int o = 0; o < Cout; ++o) acc[o] += xv * wr[o];
            }
          }
        }
        if (relu) for (int o = 0; o < Cout; ++o) acc[o] = acc[o] > 0.f ? acc[o] : 0.f;
      }
    }
}

/* dx[b,iy,ix,c] = sum_{u,v,o} dz[b,oy,ox,o] * w[u,v,c,o]   with iy = oy*s+u-pt, ix = ox*s+v-pl  (gather form) */
void segcpu_conv2d_dgrad(const float* dz, const float* w, float* dx, int B, int H, int W, int Cin, int Cout, int k, int s, int pt, int pl,
                         int Ho, int Wo) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int iy = 0; iy < H; ++iy)
      for (int ix = 0; ix < W; ++ix) {
        float* acc = dx + ((int64_t)(b * H + iy) * W + ix) * Cin;
        memset(acc, 0, sizeof(float) * Cin);
        for (int u = 0; u < k; ++u) {
          const int ty = iy + pt - u;
          if (ty < 0 || ty % s) continue;
          const int oy = ty / s;
          if (oy >= Ho) continue;
          for (int v = 0; v < k; ++v) {
            const int tx = ix + pl - v;
            if (tx < 0 || tx % s) continue;
            const int ox = tx / s;
            if (ox >= Wo) continue;
            const float* zp = dz + ((int64_t)(b * Ho + oy) * Wo + ox) * Cout;
            const float* wp = w + ((int64_t)(u * k + v) * Cin) * Cout;
            for (int c = 0; c < Cin; ++c) {
              const float* wr = wp + (int64_t)c * Cout;
              float a = 0.f;
              for (int o = 0; o < Cout; ++o) a += zp[o] * wr[o];
              acc[c] += a;
            }
          }
        }
      }
}

/* dw[u,v,c,o] = sum_{b,oy,ox} x[...] * dz[b,oy,ox,o];  db[o] = sum dz.  Parallel over (tap, c): every output element has
 * one owner and a fixed summation order. */
void segcpu_conv2d_wgrad(const float* x, int64_t xs_b, int64_t xs_h, int64_t xs_w, const float* dz, float* dw, float* db, int B, int H, int W,
                         int Cin, int Cout, int k, int s, int pt, int pl, int Ho, int Wo) {
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int t = 0; t < k * k; ++t)
    for (int c = 0; c < Cin; ++c) {
      const int u = t / k, v = t % k;
      float* acc = dw + ((int64_t)t * Cin + c) * Cout;
      memset(acc, 0, sizeof(float) * Cout);
      for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < Ho; ++oy) {
          const int iy = oy * s + u - pt;
          if (iy < 0 || iy >= H) continue;
          for (int ox = 0; ox < Wo; ++ox) {
            const int ix = ox * s + v - pl;
            if (ix < 0 || ix >= W) continue;
            const float xv = x[b * xs_b + iy * xs_h + ix * xs_w + c];
            const float* zp = dz + ((int64_t)(b * Ho + oy) * Wo + ox) * Cout;
            for (int o = 0; o < Cout; ++o) acc[o] += xv * zp[o];
          }
        }
    }
  if (db) {
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
      double a = 0.0;
      for (int64_t p = 0; p < (int64_t)B * Ho * Wo; ++p) a += dz[p * Cout + o];
      db[o] = (float)a;
    }
  }
}

/* 2x2 / stride 2 transposed conv (VALID): y[b,2i+a,2j+c,o] = act( bias[o] + sum_ci x[b,i,j,ci] * w[a,c,o,ci] ) */
void segcpu_convT2x2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int relu) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < W; ++j) {
        const float* xp = x + ((int64_t)(b * H + i) * W + j) * Cin;
        for (int a = 0; a < 2; ++a)
          for (int c = 0; c < 2; ++c) {
            float* yp = y + ((int64_t)(b * 2 * H + 2 * i + a) * 2 * W + 2 * j + c) * Cout;
            const float* wp = w + ((int64_t)(a * 2 + c) * Cout) * Cin;
            for (int o = 0; o < Cout; ++o) {
              const float* wr = wp + (int64_t)o * Cin;
              float s = bias ? bias[o] : 0.f;
              for (int ci = 0; ci < Cin; ++ci) s += xp[ci] * wr[ci];
              yp[o] = relu ? (s > 0.f ? s : 0.f) : s;
            }
          }
      }
}

/* dx[b,i,j,ci] = sum_{a,c,o} dz[b,2i+a,2j+c,o] * w[a,c,o,ci];  dw[a,c,o,ci] = sum x[b,i,j,ci] * dz[b,2i+a,2j+c,o];  db[o] = sum dz */
void segcpu_convT2x2_bwd(const float* x, const float* w, const float* dz, float* dx, float* dw, float* db, int B, int H, int W, int Cin, int Cout) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < W; ++j) {
        float* dp = dx + ((int64_t)(b * H + i) * W + j) * Cin;
        memset(dp, 0, sizeof(float) * Cin);
        for (int a = 0; a < 2; ++a)
          for (int c = 0; c < 2; ++c) {
            const float* zp = dz + ((int64_t)(b * 2 * H + 2 * i + a) * 2 * W + 2 * j + c) * Cout;
            const float* wp = w + ((int64_t)(a * 2 + c) * Cout) * Cin;
            for (int o = 0; o < Cout; ++o) {
              const float zv = zp[o];
              const float* wr = wp + (int64_t)o * Cin;
              for (int ci = 0; ci < Cin; ++ci) dp[ci] += zv * wr[ci];
            }
          }
      }
#pragma omp parallel for collapse(2) schedule(static)
  for (int t = 0; t < 4; ++t)
    for (int o = 0; o < Cout; ++o) {
      const int a = t / 2, c = t % 2;
      float* acc = dw + ((int64_t)t * Cout + o) * Cin;
      memset(acc, 0, sizeof(float) * Cin);
      for (int b = 0; b < B; ++b)
        for (int i = 0; i < H; ++i)
          for (int j = 0; j < W; ++j) {
            const float zv = dz[((int64_t)(b * 2 * H + 2 * i + a) * 2 * W + 2 * j + c) * Cout + o];
            const float* xp = x + ((int64_t)(b * H + i) * W + j) * Cin;
            for (int ci = 0; ci < Cin; ++ci) acc[ci] += zv * xp[ci];
          }
    }
#pragma omp parallel for schedule(static)
  for (int o = 0; o < Cout; ++o) {
    double a = 0.0;
    for (int64_t p = 0; p < (int64_t)B * 4 * H * W; ++p) a += dz[p * Cout + o];
    db[o] = (float)a;
  }
}

void segcpu_maxpool2x2_fwd(const float* x, float* y, uint8_t* idx, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < Ho; ++oy)
      for (int ox = 0; ox < Wo; ++ox)
        for (int c = 0; c < C; ++c) {
          float m = 0.f; int mi = -1;
          for (int q = 0; q < 4; ++q) {
            const float v = x[((int64_t)(b * H + 2 * oy + q / 2) * W + 2 * ox + q % 2) * C + c];
            if (mi < 0 || v > m) { m = v; mi = q; }
          }
          const int64_t o = ((int64_t)(b * Ho + oy) * Wo + ox) * C + c;
          y[o] = m; idx[o] = (uint8_t)mi;
        }
}

void segcpu_maxpool2x2_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  memset(dx, 0, sizeof(float) * (size_t)B * H * W * C);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < Ho; ++oy)
      for (int ox = 0; ox < Wo; ++ox)
        for (int c = 0; c < C; ++c) {
          const int64_t o = ((int64_t)(b * Ho + oy) * Wo + ox) * C + c;
          const int q = idx[o];
          dx[((int64_t)(b * H + 2 * oy + q / 2) * W + 2 * ox + q % 2) * C + c] = dy[o];
        }
}

/* mean over pixels of softmax_cross_entropy_with_logits(one_hot(label), z); dlogits = (softmax - onehot) / npix.  Returns the loss. */
double segcpu_softmax_xent(const float* z, const uint8_t* labels, float* dz, int64_t npix, int C) {
  double loss = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : loss)
  for (int64_t p = 0; p < npix; ++p) {
    const float* zp = z + p * C;
    float m = zp[0];
    for (int c = 1; c < C; ++c) m = zp[c] > m ? zp[c] : m;
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += exp((double)(zp[c] - m));
    const int lab = labels[p];
    const int valid = lab < C;
    if (valid) loss += log(s) - (double)(zp[lab] - m);
    for (int c = 0; c < C; ++c)
      dz[p * C + c] = valid ? (float)((exp((double)(zp[c] - m)) / s - (c == lab ? 1.0 : 0.0)) / (double)npix) : 0.f;
  }
  return loss / (double)npix;
}

/* tf.train.AdamOptimizer step t (1-based), in place */
void segcpu_adam(float* p, const float* g, float* m, float* v, int64_t n, int t, float lr, float b1, float b2, float eps) {
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float gi = g[i];
    m[i] = b1 * m[i] + (1.f - b1) * gi;
    v[i] = b2 * v[i] + (1.f - b2) * gi * gi;
    p[i] -= (float)lr_t * m[i] / (sqrtf(v[i]) + eps);
  }
}

/* out = relu-grad mask: dz = dy * (y > 0), elementwise */
void segcpu_relu_grad(const float* dy, const float* y, float* dz, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}
