/* seg_hip.h -- C-ABI of libseg_hip.so: the MI355X (gfx950) kernels behind the
 * UNetModel / FCNModel / BaseModel.train_step() / .infer() hot path.
 *
 * The reference (nathanin/segmentation) has no FFI boundary of its own: all of
 * its arithmetic is TensorFlow-1.x / tf.contrib.slim graph ops.  Each entry point
 * below replaces the TF op(s) invoked at the cited reference site (paths relative
 * to /root/reference); the Python host (segmentation_amd/) binds them with ctypes.
 *
 * Conventions
 *   - activations NHWC; every activation buffer carries a channel stride `cs`
 *     (elements per pixel) that is a multiple of 32 with zero-filled pad channels;
 *   - raw device pointers, int32 dims, `stream` is a hipStream_t passed as void*;
 *   - returns 0 on success, <0 on error (message: seg_last_error(), thread-local);
 *   - never allocates, frees or synchronises; safe to capture into a hipGraph;
 *   - dtype: SEG_F32 (parity mode: f32 storage, exact-f32 MFMA) or SEG_BF16
 *     (bf16 storage, fp32 accumulate on v_mfma_f32_16x16x32_bf16).
 */
#ifndef SEG_HIP_H
#define SEG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SEG_F32 0
#define SEG_BF16 1

#define SEG_OK 0
#define SEG_ERR_ARG (-1)
#define SEG_ERR_LAUNCH (-2)
#define SEG_ERR_UNSUPPORTED (-3)

/* A window into an NHWC activation buffer: logical pixel (y,x), channel c lives at
 * element ((b*H + y+oy)*W + x+ox)*cs + coff + c. */
typedef struct seg_view {
  void* ptr;
  int32_t H, W;      /* buffer spatial extent            */
  int32_t cs, coff;  /* channel stride / offset (elems)  */
  int32_t oy, ox;    /* window (crop) origin             */
  int32_t c;         /* channels used, multiple of 32 (padded) */
} seg_view;

/* Implicit-GEMM convolution descriptor (forward, dgrad and transposed-conv all run on it).
 * GEMM view: M = B*Ho*Wo pixels, N = n_count output channels, K = KH*KW*(src0.c+src1.c). */
typedef struct seg_conv_desc {
  seg_view src0, src1;     /* src1.ptr==NULL: single input; else channel-concat [src0|src1] by view */
  int32_t B, Hi, Wi;       /* logical input extent (of the windows above)  */
  int32_t KH, KW, stride;  /* 3x3/s1, 1x1/s1, 2x2/s2                        */
  int32_t pad_t, pad_l;    /* zero padding before (VALID: 0; SAME 3x3: 1; dgrad of VALID 3x3: 2) */
  int32_t Ho, Wo;          /* logical output extent                          */
  const void* w_packed;    /* [KH*KW][K/32][n_total][32] in `dtype` (seg_pack_weights) */
  int32_t n_total;         /* rows in the packed weights (multiple of 32)    */
  int32_t n_off, n_count;  /* this launch computes packed rows [n_off, n_off+n_count) (n_count%32==0) */
  const float* bias;       /* nullable; bias[j] for j<bias_n, logical index = out channel (mod up_cout if up2) */
  int32_t bias_n;
  seg_view dst;            /* out channel j (0-based within this launch) -> dst.coff + j */
  int32_t up2;             /* 1: 2x2/s2 transposed-conv scatter: packed row n=(a*2+c)*up_cout+co -> dst pixel (2y+a,2x+c) */
  int32_t up_cout;
  seg_view mask;           /* nullable ptr: ReLU-grad mask source, same logical extent/channels as dst; out=0 where mask<=0 */
  int32_t relu;            /* fused ReLU                                     */
  int32_t out_f32;         /* store float even when dtype==SEG_BF16 (logits)  */
  int32_t dtype;
  int32_t cfg;             /* 0 = auto tile choice; else forced config id (tuning/tests): 1-6 fixed tiles, 7 / 8 linearised 128-slot tile x 64 / 32
                            * channels (3x3/s1), 9 the [32 | 32] two-destination block, 32 / 34 / 38 and 62 / 64 / 68 the multi-tile walk (2 / 4 / 8 tiles x 32 / 64 channels, one K chunk), 11-24 direct-to-LDS forms, 204 / 208 conv_ring; < 0: auto with staging mode |cfg| */
  int32_t accum;           /* 1: add to what dst already holds (second consumer of a tensor in backward) */
  int32_t n_split;         /* > 0: two destinations in one launch (dgrad of a channel-concat input): out channels
                            * [0, n_split) -> dst / mask, channels [n_split, n_count) -> dst1 / mask1 (as j - n_split).
                            * n_split must be a multiple of 64 or 32 (the channel tile) */
  seg_view dst1, mask1;
  seg_view pool;           /* nullable ptr: also write the 2x2/s2 VALID max-pool of the output ([pool_h = Ho/2, pool_w = Wo/2], same
                            * channels) -- slim.max_pool2d fused into the producing conv; SEG_ERR_UNSUPPORTED when the layer's
                            * tile cannot carry it (ask seg_conv2d_kernel_name first, or fall back to seg_maxpool2x2_fwd) */
  int32_t pool_h, pool_w;
  uint32_t* signal;        /* nullable: the launch stores signal_value to *signal (system-scope relaxed store by its first workgroup)
                            * when it STARTS.  On an in-order stream that means every earlier launch of the stream is complete and
                            * released, so another stream can wait for it with hipStreamWaitValue32(signal >= value) -- a fork that
                            * puts no packet on the producing stream (an event record costs it 4-7 us; tools/micro/sigwait.hip) */
  uint32_t signal_value;
  int32_t* sched;          /* nullable: two zero-initialised int32 words of device memory owned by this launch site (not shared with a
                            * launch that may run concurrently).  The persistent bf16 3x3 kernel hands its tiles out through them (an
                            * atomic ticket) and leaves them zero again when it ends; without them tiles are split statically. */
  int32_t ksplit;          /* 0 / 1: one workgroup per output tile.  > 1 (tiled kernels; ask seg_conv2d_splitk_plan): the K chunks of a
                            * tile are shared by `ksplit` workgroups; each stores its f32 partial tile to splitk_ws and takes a ticket;
                            * the LAST to arrive adds the partials in split order (fixed association: reproducible bit for bit,
                            * whichever workgroup does it) and runs the whole epilogue.  Meant for the layers whose grid leaves <= 2
                            * workgroups per CU on maps of 8..28 pixels (conv4_x..conv7_x at 256^2 inputs and their dgrads); measured
                            * slower than the unsplit launch on all of them but one, so seg_conv2d_splitk_plan only proposes a split
                            * when asked to (SEG_CONV_SPLITK=n) -- DESIGN.md section 5. */
  int32_t n_store;         /* 0: every computed channel is stored.  8: THIN destination -- a tensor of <= 8 logical channels kept at a
                            * channel stride of 8 (dst.cs == 8) instead of 32: only the lanes of channels [0, 8) store (per
                            * transposed-conv tap with up2); mask must then be thin as well.  No n_split / pool / accum. */
  float* splitk_ws;        /* ksplit > 1: workspace of seg_conv2d_splitk_plan's size, owned by this launch site */
  int32_t* splitk_tickets; /* ksplit > 1: one zero-initialised int32 per (output tile x channel block); left zero again by the launch */
  int32_t thin_src;        /* 1: THIN source(s) -- src0 (and src1) have cs == 8, coff == 0 and c == 32: the kernel reads 32 channels
                            * per pixel, i.e. the pixel's 8 and the next pixels' values, and relies on the packed filters being ZERO
                            * for every channel >= 8 (they are: pad channels of a layer with <= 8 logical inputs).  The buffer must
                            * extend 64 elements behind its last pixel and hold finite values throughout.  The DeconvModel's 2-class
                            * tail (deconv3_0 / bn8 / conv_out at 512^2) moves a quarter of the bytes this way. */
  int32_t thin_pad_;
} seg_conv_desc;

/* slim.convolution2d / conv2d_transpose fwd, Conv2DBackpropInput: models/unet.py:111-166,
 * models/fcn.py:110-128,192,195 (forward); TF autodiff of the same sites (dgrad). */
int seg_conv2d(const seg_conv_desc* d, void* stream);
/* Reports (without launching) the kernel template instance seg_conv2d would run for d, e.g.
 * "conv_fwd_kernel<bf16,8,16,64,4,1,3,3,1>" -- used by bench.py to key per-kernel roofline numbers
 * to the names rocprofv3 prints. */
int seg_conv2d_kernel_name(const seg_conv_desc* d, char* buf, int32_t cap);
/* Reports (without launching) whether seg_conv2d would gain from sharing the K loop of d among several workgroups per output
 * tile: *ksplit (1 = no), the bytes of seg_conv_desc.splitk_ws and the number of int32 words of splitk_tickets a launch with that
 * ksplit needs.  d->ksplit > 0 on entry asks for that split (clamped to the K chunks the layer has). */
int seg_conv2d_splitk_plan(const seg_conv_desc* d, int32_t* ksplit, int64_t* ws_bytes, int32_t* tickets);

/* Filter gradient (Conv2DBackpropFilter) + bias gradient (BiasAddGrad) for the same sites.
 * dw[tap][k][n] = sum_pixels src[b, y*s+u-pad_t, x*s+v-pad_l, k] * dz[b,y,x,n]   (f32, overwritten).
 * dw layout [KH*KW][k_logical][n_logical] (= TF HWIO for conv; = TF [kh,kw,Cout,Cin] for the transposed
 * conv when src:=dz_big, dz:=x_small); k_logical maps padded concat channels back to logical ones via
 * (src0_clog, src1_clog).  Two launches: workgroups write per-split partial slabs into `ws` with plain
 * stores, then a reduce kernel sums the ksplit slabs in a fixed order (deterministic, no atomics).
 * bias_mode 1: db[n] = sum_pixels dz[...,n] (conv); 2: db[k] = sum over pixels and taps of src (transposed conv).
 * Call seg_conv2d_wgrad_plan first to learn ksplit and the workspace size the launch needs. */
typedef struct seg_wgrad_desc {
  seg_view src0, src1;
  int32_t src0_clog, src1_clog;   /* logical (unpadded) channels of each source */
  int32_t B, Hi, Wi;
  int32_t KH, KW, stride, pad_t, pad_l;
  int32_t Ho, Wo;
  seg_view dz;                    /* [B,Ho,Wo,n] window; dz.c = padded n */
  int32_t n_log;                  /* logical n (columns of dw)            */
  float* dw;
  int32_t dtype;
  int32_t cfg;
  float* ws;                      /* partial-slab workspace                 */
  int64_t ws_bytes;
  int32_t ksplit;                 /* 0 = auto                               */
  int32_t bias_mode;              /* 0 none, 1 sum dz, 2 sum src            */
  float* db;
  int32_t bias_n;
  int32_t phase;                  /* 0 = partial sums + reduce (two launches); 1 = partial sums only; 2 = reduce only */
  /* First layer (models/unet.py:111 conv1_1, models/fcn.py:110 conv1): im2col_x != NULL makes src0 VIRTUAL -- the 3x3
   * im2col of this dense float32 NHWC image [B, im2col_h, im2col_w, im2col_cin], gathered while the tiles are staged
   * (channel tap*cin + ci of output pixel (y,x) = image[b, y+u-pad, x+v-pad, ci], zero outside).  Then KH = KW = 1,
   * src0.c = 32, src0_clog = 9*cin, Hi = Ho, Wi = Wo; src0.ptr is not dereferenced (any non-NULL value) and dw comes
   * out as the [3][3][cin][n] HWIO filter gradient. */
  const float* im2col_x;
  int32_t im2col_h, im2col_w, im2col_cin, im2col_pad;
  /* ... and (bf16) pool_y.ptr != NULL makes dz VIRTUAL as well: the gradient of the layer's own activation pool_y [B,Ho,Wo,n]
   * rebuilt while the tiles are staged from the 2x2/s2 max-pool that consumes it -- exactly seg_maxpool2x2_bwd:
   *   dz = (pool_y > 0) * (pool_dp routed to the first maximum of each window + pool_add inside its window)
   * pool_dp [B,Ho/2,Wo/2,n] (ptr NULL: no routing); pool_add (ptr NULL: none) = gradient from the layer's other consumer, a
   * [pool_add_h, pool_add_w] window whose pixel (0,0) is pixel (pool_add_y0, pool_add_x0) of the map (the U-Net's conv1_2 data
   * gradient).  dz.ptr is then not dereferenced (dz.c = n still describes the channel padding). */
  seg_view pool_y, pool_dp, pool_add;
  int32_t pool_add_h, pool_add_w, pool_add_y0, pool_add_x0;
  /* Workgroups the automatic K split aims for (0 = the library default, 128: half the chip, because filter gradients normally
   * share it with the data gradients of the critical stream).  The last filter gradients of a backward pass run after the
   * critical stream has finished: the caller marks them with 256. */
  int32_t target_wgs;
  /* THIN operands (see seg_conv_desc.thin_src): bit 0 = src0 (and src1), bit 1 = dz have cs == 8, coff == 0 and c == 32; the
   * channels read beyond 8 only meet filter-gradient entries that are never stored (src*_clog, n_log <= 8). */
  int32_t thin;
} seg_wgrad_desc;
int seg_conv2d_wgrad(const seg_wgrad_desc* d, void* stream);
int seg_conv2d_wgrad_plan(const seg_wgrad_desc* d, int32_t* ksplit, int64_t* ws_bytes);
int seg_conv2d_wgrad_kernel_name(const seg_wgrad_desc* d, char* buf, int32_t cap);
/* Batched slab reduction for a whole backward segment (one launch instead of one per layer): run the layers'
 * wgrads with phase = 1, then reduce all of their slabs at once.  _plan writes one opaque 96-byte job record per
 * descriptor with ksplit > 1 into host memory (jobs_host, cap_bytes) and reports the job count and grid size; the
 * caller copies the records to device memory once and replays seg_wgrad_reduce_batch.  Same fixed summation order
 * as the per-layer reduction (bitwise identical results). */
#define SEG_REDUCE_JOB_BYTES 96
int seg_wgrad_reduce_batch_plan(const seg_wgrad_desc* const* descs, int32_t n, void* jobs_host, int64_t cap_bytes,
                                int32_t* njobs, int32_t* total_blocks);
int seg_wgrad_reduce_batch(const void* jobs_dev, int32_t njobs, int32_t total_blocks, void* stream);

/* First layer (Cin = input_channel <= 4, never padded to 32): models/unet.py:111-116 conv1_1,
 * models/fcn.py:110-115 conv1.  x is float32 NHWC [B,H,W,cin] dense. */
int seg_conv_first_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin,
                       const float* w_hwio, const float* bias, int32_t cout, int32_t pad,
                       const seg_view* dst, int32_t Ho, int32_t Wo, int32_t relu, int32_t dtype, void* stream);
/* A small KH x KW filter at any stride on the raw image in one pass (bf16, cin <= 3, cout <= 64; 5x5/s2, 3x3/s1, 3x3/s2, 7x7/s2
 * are instantiated): slim.convolution2d(images, n_kernels, 5, 2, padding='SAME') of models/deconvolution.py:44-46 without the
 * im2col tensor.  pad_t / pad_l: the leading pads (TF SAME: total // 2); w_hwio float32 [KH][KW][cin][cout]. */
int seg_conv_first_gen(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const float* w_hwio, const float* bias,
                       int32_t cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_t, int32_t pad_l, const seg_view* dst,
                       int32_t Ho, int32_t Wo, int32_t relu, int32_t dtype, void* stream);
/* The same launch also leaving the statistics rows of the batch norm that consumes the layer (5x5 / stride 2 only): bn_ws is that
 * batch norm's workspace (seg_bn_ws_bytes(bn_C)); seg_bn_fwd_rows(..., rows = seg_conv_first_gen_rows(B, Ho, Wo, cout)) finishes it
 * without reading the activation for its statistics (models/deconvolution.py:44-50 conv1_0 -> bn1). */
int32_t seg_conv_first_gen_rows(int32_t B, int32_t Ho, int32_t Wo, int32_t cout);
int seg_conv_first_gen_bn(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const float* w_hwio, const float* bias,
                          int32_t cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_t, int32_t pad_l, const seg_view* dst,
                          int32_t Ho, int32_t Wo, int32_t relu, float* bn_ws, int32_t bn_C, int32_t dtype, void* stream);
/* Filter + bias gradient of that layer (tf.gradients through slim.convolution2d(images, n_kernels, 5, 2), models/deconvolution.py:44-46)
 * straight from the image -- no im2col tensor: MFMAs over the pixel dimension with the patch read transposed out of LDS.  bf16, 5x5 /
 * stride 2, cin <= 3, cout <= 64.  dw_hwio float32 [5][5][cin][cout], db [cout] or NULL; ws: seg_conv_first_gen_wgrad_ws_bytes
 * (cout) bytes of partial rows, summed in a fixed order by a second launch (deterministic). */
int64_t seg_conv_first_gen_wgrad_ws_bytes(int32_t cout);
int seg_conv_first_gen_wgrad(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, const seg_view* dz, int32_t Ho, int32_t Wo,
                             int32_t cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad_t, int32_t pad_l, float* dw_hwio, float* db,
                             void* ws, int64_t ws_bytes, int32_t dtype, void* stream);

/* Same layer fused with the 2x2/s2 VALID max-pool that consumes it (models/unet.py pool1 over conv1_1, models/fcn.py:116
 * pool1 over conv1): one pass writes the activation and its pooled map [Hp = Ho/2, Wp = Wo/2].  bf16, cin <= 3, cout <= 64. */
int seg_conv_first_pool_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin,
                            const float* w_hwio, const float* bias, int32_t cout, int32_t pad,
                            const seg_view* dst, int32_t Ho, int32_t Wo, int32_t relu,
                            const seg_view* pool, int32_t Hp, int32_t Wp, int32_t dtype, void* stream);

/* im2col of the raw float input (3x3 window, k = (u*3+v)*cin + c, zero-padded to 32 channels): lets the first
 * layer's Conv2DBackpropFilter run as a 1x1 seg_conv2d_wgrad on the MFMA path (dw comes out in HWIO order). */
int seg_im2col3x3(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t pad, const seg_view* dst,
                  int32_t Ho, int32_t Wo, int32_t dtype, void* stream);
/* The same for any window / stride / leading pads (k = (u*KW+v)*cin + c, zero-filled up to dst->c, a multiple of 32): the
 * DeconvModel's 5x5/s2 SAME first layer (models/deconvolution.py) as a 1x1 convolution on the MFMA kernels. */
int seg_im2col(const float* x, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t KH, int32_t KW, int32_t stride,
               int32_t pad_t, int32_t pad_l, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t dtype, void* stream);

/* The DeconvModel's 5x5/s2 transposed convolutions on the MFMA kernels (models/deconvolution.py deconv1_0 / deconv2_0 /
 * deconv2_1): a transposed convolution with filter [kh,kw,Cout,Cin] is the adjoint of the strided convolution whose HWIO filter
 * is the same memory, and that one is a 1x1 convolution over the strided im2col of its input.
 * seg_im2col_act: dst[b,oy,ox,(u*KW+v)*C + c] = src[b, oy*s - pad_t + u, ox*s - pad_l + v, c] (zero outside, zero-filled up to
 * dst->c, a multiple of 32).  seg_col2im: the adjoint gather, dst[b,Y,X,c] = relu?(bias[c] + sum of col[b,(Y+pad_t-u)/s,
 * (X+pad_l-v)/s,(u*KW+v)*C + c] over the taps that divide).  Any C (a multiple of 8 takes the 16-byte path).  The same pair runs
 * the adversary's 3x3/s2 convolutions (models/basemodel.py:228-246) as 1x1 convolutions over an im2col: engine.Net.sconv_*. */
int seg_im2col_act(const seg_view* src, int32_t B, int32_t Hs, int32_t Ws, int32_t C, int32_t KH, int32_t KW, int32_t stride,
                   int32_t pad_t, int32_t pad_l, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t dtype, void* stream);
int seg_col2im(const seg_view* col, int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t KH, int32_t KW, int32_t stride,
               int32_t pad_t, int32_t pad_l, const float* bias, int32_t relu, const seg_view* dst, int32_t Ho, int32_t Wo,
               int32_t dtype, void* stream);

/* slim.max_pool2d(x, 2) (kernel 2, stride 2, VALID): models/unet.py:120,124,128,132; models/fcn.py:116-126.
 * idx (nullable): uint8 plane [B,Ho,Wo,C] with the first-max position 0..3 in window order. */
int seg_maxpool2x2_fwd(const seg_view* src, const seg_view* dst, uint8_t* idx,
                       int32_t B, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, void* stream);
/* MaxPoolGrad fused with (a) the zero-padded add of a skip-connection gradient (crop grad,
 * models/unet.py:139-141 backward) and (b) the ReLU-grad mask of the producing conv:
 *   dz[y,x,c] = ( route(dpool)[y,x,c] + (in window ? add[y-ay, x-ax, c] : 0) ) * (y_act[y,x,c] > 0)
 * over the full H x W extent of y_act (odd trailing row/col get only the add term).
 * Routing recomputes the first maximum of each window from y_act (bit-identical to an index plane).
 * dpool nullable (then only the add term), add nullable. */
int seg_maxpool2x2_bwd(const seg_view* y_act, const seg_view* dpool, const seg_view* add,
                       int32_t add_h, int32_t add_w, int32_t add_y0, int32_t add_x0,
                       const seg_view* dz, int32_t B, int32_t H, int32_t W, int32_t C,
                       int32_t dtype, void* stream);

/* Per-pixel softmax cross-entropy + its gradient: models/basemodel.py:59-70 (spec), :194, :360.
 * logits: float32 view; labels: uint8 [B,LH,LW] read at (y+ly0, x+lx0) (centre crop of input_y,
 * models/unet.py:171-174).  loss_sum += sum_pixels xent * inv_n (atomic, caller zeroes);
 * dlogits (dtype view, padded channels written as 0) = (softmax - onehot) * inv_n * grad_scale. */
int seg_softmax_xent(const seg_view* logits, const uint8_t* labels, int32_t LH, int32_t LW,
                     int32_t ly0, int32_t lx0, int32_t B, int32_t H, int32_t W, int32_t n_classes,
                     float inv_n, float grad_scale, float* loss_sum, const seg_view* dlogits,
                     int32_t dtype, void* stream);
/* The same launch also writing softmax(logits) as a tensor of the compute dtype (probs; NULL: seg_softmax_xent): the adversary's
 * "fake" input (models/basemodel.py:285) without a second pass over the float logits. */
int seg_softmax_xent_probs(const seg_view* logits, const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0,
                           int32_t B, int32_t H, int32_t W, int32_t n_classes, float inv_n, float grad_scale, float* loss_sum,
                           const seg_view* dlogits, const seg_view* probs, int32_t dtype, void* stream);

/* Fused U-Net training head: the 1x1 'output' convolution (models/unet.py:166; float32 logits), the per-pixel softmax
 * cross-entropy + gradient above, and the 1x1 convolution's input gradient masked by its input's ReLU (dact), in one
 * pass.  act: the conv's input (32 or 64 padded channels, `cin` logical); w_hwio: its fp32 filter [cin][n_classes].
 * dw_ws == NULL: dlogits is written and feeds the filter gradient (seg_conv2d_wgrad) exactly as after seg_softmax_xent.
 * dw_ws != NULL (n_classes <= 8; seg_head_xent_ws_bytes of device memory): the filter and bias gradients of the 1x1
 * convolution are accumulated in the same pass as per-workgroup partial sums, dlogits is not written (its view may
 * carry a NULL ptr), and seg_head_dw_reduce -- same shape arguments -- finishes them into dw [cin][n_classes] / db. */
int64_t seg_head_xent_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t cin_pad, int32_t n_classes);   /* 0: not covered */
int seg_head_xent(const seg_view* act, const float* w_hwio, const float* bias, int32_t cin, const uint8_t* labels,
                  int32_t LH, int32_t LW, int32_t ly0, int32_t lx0, int32_t B, int32_t H, int32_t W, int32_t n_classes,
                  float inv_n, float* loss_sum, const seg_view* logits, const seg_view* dlogits, const seg_view* dact,
                  void* dw_ws, int64_t dw_ws_bytes, int32_t dtype, void* stream);
int seg_head_dw_reduce(const void* dw_ws, int64_t dw_ws_bytes, int32_t B, int32_t H, int32_t W, int32_t cin_pad,
                       int32_t cin, int32_t n_classes, float* dw, float* db, void* stream);

/* tf.nn.sigmoid + tf.argmax(axis=3) + expand_dims + cast: models/unet.py:75-79, models/fcn.py:74-78.
 * sig: dense float32 [B,H,W,n_classes]; out: dense float32 [B,H,W,1]; argmax taken over the float32
 * sigmoid values, first maximum wins. */
int seg_sigmoid_argmax(const seg_view* logits, int32_t B, int32_t H, int32_t W, int32_t n_classes,
                       float* sig, float* out, void* stream);

/* BiasAddGrad: db[c] = sum over B*H*W of dz[...,c] for c < n_log (overwritten; one workgroup per 8 channels, fixed order). */
int seg_bias_grad(const seg_view* dz, int32_t B, int32_t H, int32_t W, int32_t n_log, float* db,
                  int32_t dtype, void* stream);
/* 3x3 / stride-1 convolution between THIN tensors (<= 8 channels, cs == 8, coff == 0: seg_conv_desc.thin_src) on the vector ALU:
 * dst = relu?(bias + conv(src, w_hwio[3][3][cin][cout])) with `pad` zeros around the input (Ho = Hi + 2 pad - 2), float output
 * when out_f32; dgrad != 0: the data gradient of that layer instead -- src is dZ [.., cout], dst is dX [.., cin], pad is
 * 2 - (the layer's pad), mask (nullable) the layer input's ReLU output.  The DeconvModel's conv_out (models/deconvolution.py:170). */
int seg_thin_conv3x3(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const float* w_hwio, const float* bias, int32_t cin,
                     int32_t cout, int32_t pad, int32_t relu, int32_t dgrad, const seg_view* mask, const seg_view* dst, int32_t Ho,
                     int32_t Wo, int32_t out_f32, int32_t dtype, void* stream);
/* Forward of seg_thin_conv3x3 on slim.batch_norm(src) without that tensor (models/deconvolution.py:168-170: bn8 -> conv_out): src is
 * the PRE-batch-norm activation, bn_stats the 8-channel batch norm's [mean[8] | rstd[8]] (the `stats` of seg_bn_fwd; pass y = NULL
 * there for the statistics alone) and bn_beta its beta[cin]; every value read is normalised and rounded exactly as seg_bn_fwd would
 * have stored it, so the output is bit-identical. */
int seg_thin_conv3x3_bn(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const float* w_hwio, const float* bias, int32_t cin,
                        int32_t cout, int32_t pad, int32_t relu, const seg_view* dst, int32_t Ho, int32_t Wo, int32_t out_f32,
                        const float* bn_stats, const float* bn_beta, int32_t dtype, void* stream);
/* Filter and bias gradient of that layer (tf.gradients through slim.convolution2d(net, n_classes, 3, 1), models/deconvolution.py:170)
 * from a thin source and a thin dZ on the vector ALU: dw_hwio float32 [3][3][cin][cout], db [cout] (NULL: none); two launches
 * (partial rows in ws, seg_thin_wgrad3x3_ws_bytes(cin, cout) bytes; a fixed-order sum of the rows), deterministic. */
int64_t seg_thin_wgrad3x3_ws_bytes(int32_t cin, int32_t cout);
int seg_thin_wgrad3x3(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const seg_view* dz, int32_t Ho, int32_t Wo, int32_t cin,
                      int32_t cout, int32_t pad, float* dw_hwio, float* db, void* ws, int64_t ws_bytes, int32_t dtype, void* stream);
/* seg_thin_wgrad3x3 with the batch norm of `src` applied on load (src = the pre-batch-norm activation; see seg_thin_conv3x3_bn). */
int seg_thin_wgrad3x3_bn(const seg_view* src, int32_t B, int32_t Hi, int32_t Wi, const seg_view* dz, int32_t Ho, int32_t Wo, int32_t cin,
                         int32_t cout, int32_t pad, float* dw_hwio, float* db, void* ws, int64_t ws_bytes, const float* bn_stats,
                         const float* bn_beta, int32_t dtype, void* stream);
/* 2x2 / stride-2 transposed convolution from a [H,W,cin] tensor INTO a thin [2H,2W,cout <= 8] tensor (dgrad == 0: `big` written,
 * bias + optional ReLU) and its data gradient (dgrad != 0: `small` written from the thin gradient `big`, optional ReLU-grad mask
 * over `small`'s layout), filter in the TF layout [2,2,cout,cin] -- the DeconvModel's deconv3_0 on the vector ALU. */
int seg_thin_up2x2(const seg_view* small, const seg_view* big, int32_t B, int32_t H, int32_t W, const float* w_tf, const float* bias,
                   int32_t cin, int32_t cout, int32_t relu, int32_t dgrad, const seg_view* mask, int32_t dtype, void* stream);
/* Forward of seg_thin_up2x2 + the statistics rows ([rows][8][2] floats, rows = seg_thin_up2x2_rows(B, H, W)) of the batch norm that
 * consumes `big` (models/deconvolution.py:166-168 deconv3_0 -> bn8), for seg_bn_fwd_rows. */
int32_t seg_thin_up2x2_rows(int32_t B, int32_t H, int32_t W);
int seg_thin_up2x2_bn(const seg_view* small, const seg_view* big, int32_t B, int32_t H, int32_t W, const float* w_tf, const float* bias,
                      int32_t cin, int32_t cout, int32_t relu, float* bn_ws, int32_t dtype, void* stream);
/* The same in two stages for big maps (512 workgroups of partial sums + a fixed-order final pass; bitwise reproducible):
 * ws of seg_bias_grad_ws_bytes(dz->c) bytes (0: this channel count is not supported, use seg_bias_grad). */
int64_t seg_bias_grad_ws_bytes(int32_t C);
int seg_bias_grad_ws(const seg_view* dz, int32_t B, int32_t H, int32_t W, int32_t n_log, float* db, float* ws, int64_t ws_bytes,
                     int32_t dtype, void* stream);

/* tf.train.AdamOptimizer(lr, name='segAdam') on one flat fp32 parameter arena: models/basemodel.py:321,366.
 * TF variant: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr_t * m / (sqrt(v) + eps)   (eps OUTSIDE the bias correction).
 * t = *step_dev + 1 is read on the device (so a captured hipGraph replays with the live step);
 * g is multiplied by grad_scale first (1/world for data-parallel averaging). */
int seg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
             float eps, float grad_scale, const int64_t* step_dev, void* stream);
/* global_step assign_add 1: models/basemodel.py:88-89,368. */
int seg_step_increment(int64_t* step_dev, void* stream);
/* The same assign_add folded into the START of a train step, together with zeroing the loss accumulator (one tiny
 * launch on the auxiliary stream instead of two on the critical path): step2 = {global_step, completed}:
 * completed := global_step; global_step += 1; *loss_sum = 0.  seg_adam is then given &step2[1] (t = completed + 1). */
int seg_step_begin(int64_t* step2_dev, float* loss_sum, void* stream);

/* Weight re-layout (+cast) from the fp32 TF-layout master copy into the packed MFMA operand layout
 * [taps][K/32][n_total][32].  One launch handles a whole table of layers. */
#define SEG_PACK_CONV_FWD 0   /* src HWIO [kh,kw,Cin,Cout]: K=cin(padded concat), N=cout               */
#define SEG_PACK_CONV_DGRAD 1 /* src HWIO: taps flipped, K=cout, N=cin(padded concat)                  */
#define SEG_PACK_UP_FWD 2     /* src [2,2,Cout,Cin]: taps=1, K=cin, N=(a*2+c)*cout_pad+co               */
#define SEG_PACK_UP_DGRAD 3   /* src [2,2,Cout,Cin]: taps=4 (a,c), K=cout, N=cin                        */
typedef struct seg_pack_entry {
  int64_t src_off;      /* element offset into the fp32 arena              */
  int64_t dst_off;      /* element offset into the packed arena            */
  int32_t mode;
  int32_t KH, KW;
  int32_t cin, cout;    /* logical sizes of the source tensor              */
  int32_t seg0_c, seg0_cp, seg1_c, seg1_cp; /* cin concat segments: logical / padded widths (seg1 may be 0) */
  int32_t cout_pad;
  int32_t k_pad, n_total; /* packed K (multiple of 32) and N                */
  int64_t n_elems;      /* taps*k_pad*n_total                              */
  int64_t blk_start;    /* first 256-element block of this entry in the launch */
} seg_pack_entry;
int seg_pack_weights(const float* arena, void* packed, const seg_pack_entry* table_dev, int32_t n_entries,
                     int64_t total_blocks, int32_t dtype, void* stream);

/* seg_pack_weights with ONE read of the fp32 arena for both packed copies (same packed bytes as seg_pack_weights): each 32x32
 * source tile is written as its forward tile and, transposed, as its dgrad tile.  fwd_table_dev: the SEG_PACK_CONV_FWD /
 * SEG_PACK_UP_FWD entries only, blk_start counted over them in 32x32 tiles; dgrad_dst_off_dev[i]: packed offset of entry i's
 * dgrad copy or -1. */
int seg_pack_weights_dual(const float* arena, void* packed, const seg_pack_entry* fwd_table_dev,
                          const int64_t* dgrad_dst_off_dev, int32_t n_entries, int64_t total_tiles, int32_t dtype, void* stream);

/* Depthwise bilinear transposed conv = tf.nn.conv2d_transpose(x, bilinear_upsample_weights(f,C), SAME)
 * (models/fcn.py:199-216; filter bank utils/upsampling.py:27-46, channel-diagonal => depthwise),
 * fused with the following resize_image_with_crop_or_pad and skip addition:
 *   dst[y,x,c] = (add? add[y,x,c] : 0) + up(src)[y+cy, x+cx, c]   (0 outside the upsampled extent)
 * filt: float32 [k][k] tent (k = 2f - f%2), pad_before = (k-f)/2.  bwd is the adjoint wrt src. */
int seg_bilinear_up_fwd(const seg_view* src, int32_t Hs, int32_t Ws, int32_t factor, const float* filt,
                        const seg_view* add, const seg_view* dst, int32_t Hd, int32_t Wd,
                        int32_t cy, int32_t cx, int32_t B, int32_t C, int32_t dst_f32, int32_t dtype, void* stream);
/* mask_act / dz_masked (both or neither): ALSO store the gradient behind the ReLU of mask_act, dz_masked = dsrc * (mask_act > 0)
 * -- the score convolutions of the FCN skip fusion keep their ReLU (models/fcn.py:149-170), and their backward wants exactly
 * that tensor; saves a relu-grad launch per fusion level. */
int seg_bilinear_up_bwd(const seg_view* ddst, int32_t Hd, int32_t Wd, int32_t cy, int32_t cx,
                        int32_t factor, const float* filt, const seg_view* dsrc, int32_t Hs, int32_t Ws,
                        int32_t B, int32_t C, int32_t ddst_f32, const seg_view* mask_act, const seg_view* dz_masked,
                        int32_t dtype, void* stream);

/* The FCN training head in one launch: logits = the (cropped) bilinear up-sampling of the score map, softmax x-entropy against
 * the labels, dlogits -- models/fcn.py:199-218 + models/basemodel.py:59-70 -- without materialising the float logits
 * (logits_out non-NULL: also store them, for `y_hat` / tests).  Arguments as seg_bilinear_up_fwd (src .. cy, cx) followed by
 * those of seg_softmax_xent. */
int seg_bilinear_xent(const seg_view* src, int32_t Hs, int32_t Ws, int32_t factor, const float* filt, int32_t cy, int32_t cx,
                      const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0, int32_t B, int32_t H, int32_t W,
                      int32_t n_classes, float inv_n, float grad_scale, float* loss_sum, const seg_view* dlogits,
                      const seg_view* logits_out, int32_t dtype, void* stream);
/* seg_bilinear_up_bwd in separable form (the tent filter bank is an outer product, utils/upsampling.py:6-24): a horizontal pass
 * into ws (float [B][Hd][Ws][C], seg_bilinear_up_bwd_ws_bytes) and a vertical one -- 2k taps per source pixel instead of k*k.
 * Same result up to float summation order. */
int64_t seg_bilinear_up_bwd_ws_bytes(int32_t B, int32_t Hd, int32_t Ws, int32_t C);
int seg_bilinear_up_bwd_sep(const seg_view* ddst, int32_t Hd, int32_t Wd, int32_t cy, int32_t cx, int32_t factor, const float* filt,
                            const seg_view* dsrc, int32_t Hs, int32_t Ws, int32_t B, int32_t C, int32_t ddst_f32, float* ws,
                            int64_t ws_bytes, const seg_view* mask_act, const seg_view* dz_masked, int32_t dtype, void* stream);

/* out = a * (relu_mask>0) elementwise over a [B,H,W,C] window (ReLU-grad where no conv epilogue can do it). */
int seg_relu_grad(const seg_view* dy, const seg_view* y_act, const seg_view* dz,
                  int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);

/* slim.dropout-style mask: y = x * Bernoulli(keep)/keep, counter-based RNG (seed, offset); build-defined
 * placement.  Same call regenerates the same mask (used by bwd). */
int seg_dropout(const seg_view* x, const seg_view* y, int32_t B, int32_t H, int32_t W, int32_t C,
                float keep, uint64_t seed, uint64_t offset, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * DeconvModel (models/deconvolution.py:101-178): the ops its graph needs beyond the U-Net / FCN set.
 * ------------------------------------------------------------------------------------------------------------------- */

/* Direct correlation for the layers the MFMA tiles do not cover (5x5 stride-2 conv, 5x5 stride-2 transposed conv:
 * models/deconvolution.py:109-116,147-157).  One descriptor, three launches:
 *   seg_dconv_fwd       y[b,oy,ox,n] = act( bias[n] + sum_{u,v,k} x[b, oy*s+u-pad_t, ox*s+v-pad_l, k] * W(u,v,k,n) )
 *   seg_dconv_bwd_data  x[b,iy,ix,k] = act( bias[k] + sum_{u,v,n : (iy+pad_t-u) % s == 0, (ix+pad_l-v) % s == 0}
 *                                                      y[b, (iy+pad_t-u)/s, (ix+pad_l-v)/s, n] * W(u,v,k,n) )
 *   seg_dconv_wgrad     dW(u,v,k,n) = sum_{b,oy,ox} x[b, oy*s+u-pad_t, ox*s+v-pad_l, k] * y[b,oy,ox,n]   (f32, overwritten;
 *                       fixed summation order); db_mode 1: db[n] = sum y, 2: db[k] = sum x over its full extent
 * with W(u,v,k,n) = w[u*w_su + v*w_sv + k*w_sk + n] read from the fp32 master copy (n contiguous).
 * slim.convolution2d (HWIO, k = Cin, n = Cout): forward = _fwd, input gradient = _bwd_data, filter gradient = _wgrad.
 * slim.convolution2d_transpose (filter [kh,kw,Cout,Cin]; x := the LARGE output map, y := the small input map, k = Cout,
 * n = Cin): forward = _bwd_data (with bias / ReLU), input gradient = _fwd, filter gradient = _wgrad (db_mode 2).
 * `mask` (nullable): ReLU-grad mask on the written tensor (out = 0 where mask <= 0), same extent as the written tensor. */
typedef struct seg_dconv_desc {
  seg_view x; int32_t xc;            /* xc / yc: logical channels (loops stop there; pad channels are written as 0) */
  seg_view y; int32_t yc;
  int32_t B, Hx, Wx, Hy, Wy;
  int32_t KH, KW, stride, pad_t, pad_l;
  const float* w; int64_t w_su, w_sv, w_sk;
  const float* bias; int32_t bias_n;
  int32_t relu;
  seg_view mask;
  int32_t dtype;
} seg_dconv_desc;
int seg_dconv_fwd(const seg_dconv_desc* d, void* stream);
int seg_dconv_bwd_data(const seg_dconv_desc* d, void* stream);
/* _wgrad: `ws` = seg_dconv_wgrad_ws_bytes(d) bytes of scratch (may be 0 / NULL): with it, layers with few (tap, k, n) tiles split
 * their pixels over several workgroups per tile and add the partial slabs in a fixed order; without it the unsplit form runs. */
int64_t seg_dconv_wgrad_ws_bytes(const seg_dconv_desc* d);
int seg_dconv_wgrad(const seg_dconv_desc* d, float* dw, float* db, int32_t db_mode, float* ws, int64_t ws_bytes, void* stream);

/* slim.max_pool2d(x, k, k) (kernel k, stride k, VALID; models/deconvolution.py:118,131,140: k = 2, 3, 3) on a tensor that is
 * NOT a ReLU output (it follows a batch norm): no mask fusion.  bwd routes to the first maximum in row-major window order. */
int seg_maxpool_k_fwd(const seg_view* src, const seg_view* dst, int32_t k, int32_t B, int32_t Ho, int32_t Wo, int32_t C,
                      int32_t dtype, void* stream);
int seg_maxpool_k_bwd(const seg_view* src, const seg_view* dpool, const seg_view* dsrc, int32_t k, int32_t B, int32_t H, int32_t W,
                      int32_t C, int32_t dtype, void* stream);

/* slim.batch_norm with its defaults (decay 0.999, center=True -> beta, scale=False -> no gamma, epsilon 0.001) behind a
 * ReLU'd convolution: models/deconvolution.py:116,124,138,146,150,155,158,165; updates ride on UPDATE_OPS
 * (models/basemodel.py:364-365).  a = the ReLU output [B,H,W,C]; statistics over B*H*W per channel (population variance).
 *   seg_bn_fwd  training != 0: batch mean / variance -> stats[0:C] = mean, stats[C:2C] = 1/sqrt(var + eps); when moving != NULL
 *               also moving[0:C] = decay*moving + (1-decay)*mean, moving[C:2C] likewise with var.
 *               training == 0: mean / variance come from moving[] (the test() path).
 *               y = (a - mean) * rstd + beta.   ws: >= seg_bn_ws_bytes(C) bytes of scratch (partial sums, fixed order).
 *   seg_bn_relu_bwd  (training statistics)  dbeta[c] = sum dy;  da = rstd * (dy - mean(dy) - xhat * mean(dy * xhat));
 *               dz = da where a > 0 else 0  (the ReLU-grad of the producing convolution, fused: dz feeds its wgrad / dgrad).
 *               dbeta_add != 0: dbeta += (a second pass through the same layer, e.g. the adversary's real and fake batches). */
int64_t seg_bn_ws_bytes(int32_t C);
int seg_bn_fwd(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, int32_t training, float decay,
               float eps, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t dtype, void* stream);
/* seg_bn_fwd (training statistics) whose statistics pass was done by the launch that produced `a`: ws holds `rows` (<= 1024) rows
 * [C][2] of per-channel (sum, sum of squares) over disjoint pixel sets. */
int seg_bn_fwd_rows(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, float decay, float eps,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t rows, int32_t dtype, void* stream);
/* slim.batch_norm + the k x k / stride-k slim.max_pool2d behind it (models/deconvolution.py:50-75: bn1 -> pool 2x2, bn2 / bn3 -> pool
 * 3x3) in one pass over `a`: pooled [H/k, W/k] is bit for bit what seg_bn_fwd + seg_maxpool_k_fwd write (normalisation and rounding
 * are increasing maps, so they commute with the maximum) and the normalised full-resolution tensor does not exist;
 * seg_maxpool_k_bwd then takes `a` as its source: it routes to the first STRICT maximum of the raw activation, which is a maximum of
 * the normalised tensor too -- but where rounding to bf16 collapses distinct values of `a` into a tie, the unfused pair routes to the
 * first of the tied pixels and this one to the largest `a` among them (same forward bits, a different -- equally valid -- subgradient;
 * float32 has no such ties in practice).  rows: 0, or the statistics rows already in ws. */
int seg_bn_pool_fwd(const seg_view* a, const seg_view* pooled, const float* beta, float* moving, float* stats, int32_t training,
                    float decay, float eps, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t rows,
                    int32_t k, int32_t dtype, void* stream);
/* Backward of that pair: seg_maxpool_k_bwd (its source = `a`) + seg_bn_relu_bwd in two passes over `a`, the pool's full-resolution
 * gradient never written.  dz = masked pre-activation gradient in front of the batch norm, dbeta as seg_bn_relu_bwd. */
int seg_bn_pool_relu_bwd(const seg_view* a, const seg_view* dpool, const seg_view* dz, const float* stats, float* dbeta, int32_t dbeta_add,
                         int32_t k, int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t dtype, void* stream);
int seg_bn_relu_bwd(const seg_view* a, const seg_view* dy, const seg_view* dz, const float* stats, float* dbeta, int32_t dbeta_add,
                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t c_log, float* ws, int32_t dtype, void* stream);

/* tf.image.resize_bilinear(x, [Hd, Wd]) with align_corners=False (models/deconvolution.py:160): source coordinate =
 * dst * (Hs/Hd), lerp of the 2x2 neighbours (upper index clamped).  bwd is the adjoint, in gather form (fixed order). */
int seg_resize_bilinear_fwd(const seg_view* src, int32_t Hs, int32_t Ws, const seg_view* dst, int32_t Hd, int32_t Wd, int32_t B,
                            int32_t C, int32_t dtype, void* stream);
int seg_resize_bilinear_bwd(const seg_view* ddst, int32_t Hd, int32_t Wd, const seg_view* dsrc, int32_t Hs, int32_t Ws, int32_t B,
                            int32_t C, int32_t dtype, void* stream);

/* seg_dropout whose counter offset is offset + (*step_dev << 40), read on the device: a training step captured in a hipGraph
 * draws a fresh mask at every replay; the backward launch (dy -> dz with the same arguments) regenerates the forward mask. */
int seg_dropout_step(const seg_view* x, const seg_view* y, int32_t B, int32_t H, int32_t W, int32_t C, float keep, uint64_t seed,
                     uint64_t offset, const int64_t* step_dev, int32_t dtype, void* stream);

/* ---- adversarial segmentation training (models/basemodel.py:215-355; csrc/adv_ops.hip).  The adversary's convolutions,
 * pools, resize and dense layers are seg_dconv_*, seg_maxpool_k_*, seg_resize_bilinear_*; these are the remaining pieces. ----
 *   seg_onehot          tf.one_hot(input_y) over the output window -> the adversary's "real" map           (basemodel.py:283)
 *   seg_softmax_probs   softmax(y_hat) -> the adversary's "fake" map (float logits in, compute dtype out)   (basemodel.py:285)
 *   seg_softmax_bwd_add dlogits += scale * p * (dp - sum dp p): lambda * d l_bce_fake_one / d y_hat         (basemodel.py:334)
 *   seg_flatten         slim.flatten (NHWC order of the logical channels) and its adjoint                   (basemodel.py:251)
 *   seg_bn_rows_fwd/bwd slim.batch_norm on [B, F] feature rows (any F; training statistics; relu_mask gates
 *                       dz by a > 0 when the rows are a ReLU output)                                        (basemodel.py:252,256)
 *   seg_bce2            mean_b softmax_cross_entropy_with_logits(one_hot(label), adversary logits) stored to *loss_out,
 *                       its gradient * grad_scale to dlogits                                                (basemodel.py:287-297) */
int seg_onehot(const uint8_t* labels, int32_t LH, int32_t LW, int32_t ly0, int32_t lx0, int32_t B, int32_t H, int32_t W,
               const seg_view* dst, int32_t dtype, void* stream);
int seg_softmax_probs(const seg_view* logits, int32_t B, int32_t H, int32_t W, int32_t n_classes, const seg_view* dst, int32_t dtype, void* stream);
int seg_softmax_bwd_add(const seg_view* logits, const seg_view* dprobs, int32_t B, int32_t H, int32_t W, int32_t n_classes, float scale,
                        const seg_view* dlogits, int32_t dtype, void* stream);
int seg_flatten(const seg_view* a, int32_t B, int32_t H, int32_t W, int32_t C, const seg_view* f, int32_t backward, int32_t dtype, void* stream);
int seg_bn_rows_fwd(const seg_view* a, const seg_view* y, const float* beta, float* moving, float* stats, int32_t B, int32_t F,
                    float decay, float eps, int32_t dtype, void* stream);
int seg_bn_rows_bwd(const seg_view* a, const seg_view* dy, const seg_view* dz, const float* stats, float* dbeta, int32_t dbeta_add, int32_t B,
                    int32_t F, int32_t relu_mask, int32_t dtype, void* stream);
int seg_bce2(const seg_view* logits, int32_t B, int32_t label, float grad_scale, float* loss_out, const seg_view* dlogits, int32_t dtype, void* stream);

/* Diagnostic (tests/test_precision_gpu.py: which rounding of the bf16 mode moves the gradients): rounds a float32 window / a flat
 * float32 array to the bf16 value grid, result still float32 (dst may be src).  With these the f32-mode plans can be run with
 * bf16-rounded activations, gradients or filter-gradient operands one at a time. */
int seg_round_bf16(const seg_view* src, const seg_view* dst, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int seg_round_bf16_flat(const float* src, float* dst, int64_t n, void* stream);

/* float32 NHWC -> dtype NHWC with channel padding (feeding placeholder inputs). */
int seg_cast_pad(const float* x, int64_t npix, int32_t c, const seg_view* dst_dense, int32_t dtype, void* stream);

const char* seg_last_error(void);
int seg_version(void);
/* Name of the FIRST kernel the calling host thread has launched through this library since the previous call of this function (as
 * spelled at its launch site, blanks dropped, e.g. "conv_first_win_kernel<1,true,true>"; "" when no launch site recorded one: the
 * templated convolution / filter-gradient dispatchers are queried with seg_conv2d_kernel_name / seg_conv2d_wgrad_kernel_name).
 * bench.py labels launches whose instance the C side picks (first layer, thresholds) with what actually ran. */
const char* seg_last_kernel_name(void);
/* The library reads its environment switches (SEG_FIRST_IMPL, SEG_CONV_MODE, SEG_CONV_IMPL, SEG_WGRAD_WGS ...) once; this makes the
 * next use re-read those that go through the cache (a test hook: the first-layer test flips SEG_FIRST_IMPL between launches). */
void seg_dbg_reload_env(void);

/* ---- A whole launch plan from one host call (models/basemodel.py:480-489 train_step body; :527-531 infer) -----------------
 * The host side compiles the launches of a train step / forward pass once into an array of seg_plan_op -- the entry point of each
 * launch (seg_plan_fn_id), its arguments as seg_arg words WITHOUT the trailing stream, the index of the stream it goes to, and the
 * cross-stream edges: EVENT forks (record on `stream`, `stream2` waits) and the signal forks of seg_conv_desc.signal (the
 * convolution with signal_slot > 0 stores signal_base + slot to *signal_flag when it starts; a WAIT_VALUE op makes `stream` wait for
 * that value with hipStreamWaitValue32) -- and replays it with ONE call per step instead of ~130 interpreter iterations and ctypes
 * calls (0.64 ms of host time per 0.98 ms U-Net step in round 3).  Returns 0 or the first failing launch's code (*failed_op = its
 * index).  seg_plan_op.event is owned by the library (zero-initialise; created on first use, one per fork op). */
typedef union seg_arg { int64_t i; float f32; void* p; } seg_arg;
#define SEG_OP_LAUNCH 0
#define SEG_OP_EVENT_FORK 1
#define SEG_OP_WAIT_VALUE 2
typedef struct seg_plan_op {
  int32_t kind;
  int32_t fn;              /* SEG_OP_LAUNCH: seg_plan_fn_id() of the entry point */
  int32_t stream, stream2; /* indices into the streams array */
  int32_t nargs;           /* seg_arg words at args (checked against the entry point) */
  int32_t signal_slot;     /* LAUNCH of seg_conv2d: > 0 = announce signal_base + slot (0: no signal); WAIT_VALUE: the slot waited for */
  const seg_arg* args;
  void* event;
} seg_plan_op;
int seg_plan_fn_id(const char* name);
int seg_plan_run(seg_plan_op* ops, int32_t n, void* const* streams, int32_t n_streams, uint32_t* signal_flag, uint32_t signal_base,
                 int32_t* failed_op);
/* Hands the fork events of a plan that is going away back to the library's pool (no HIP call: safe from a finaliser at any time). */
int seg_plan_destroy_events(seg_plan_op* ops, int32_t n);

#ifdef __cplusplus
}
#endif
#endif
