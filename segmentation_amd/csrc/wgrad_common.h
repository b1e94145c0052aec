// wgrad_common.h -- what the two filter-gradient translation units (conv_wgrad.hip: register-staged tile walk, every dtype /
// kernel shape; wgrad_sweep.hip: the wave-specialised bf16 3x3 walk) share: the slab-reduction arguments and the query modes
// of seg_conv2d_wgrad (name / plan / job capture).
#pragma once
#include "common.h"

// Sums the ksplit slabs (fixed association => bitwise reproducible) and scatters to the logical TF layout.
// slab layout: [taps][k_pad][n_pad] floats followed by bias[max(k_pad, n_pad)].
struct RedArgs {
  const float* ws; float* dw; float* db;
  int64_t slab;
  int ksplit, taps, k_pad, n_pad, seg0_c, seg0_cp, seg1_c, n_log, bias_mode, bias_n;
};
// One job of the batched reduction (seg_wgrad_reduce_batch): RedArgs + the job's block range and kernel form.
struct RedJob { RedArgs a; int first_block, nblocks, flags, pad_; int64_t pad2_; };
static_assert(sizeof(RedJob) == 96, "RedJob is an opaque 96-byte record in the C-ABI");

// query modes of seg_conv2d_wgrad (all null: launch)
struct WgQuery {
  char* name_out; int name_cap;            // report the kernel instance, launch nothing
  int32_t* plan_ks; int64_t* plan_bytes;   // report ksplit / workspace bytes, launch nothing
  RedJob* job_out;                         // describe the slab reduction, launch nothing
};

// conv_wgrad.hip: launches (or, with q.job_out, describes) the reduction of `ks` slabs
int seg_wgrad_reduce_launch(const RedArgs& RA, int ks, const WgQuery& q, hipStream_t st);

// wgrad_sweep.hip: 1 = handled (rc holds the result), 0 = not eligible (the caller runs the register-staged kernel)
int seg_wgrad_sweep(const seg_wgrad_desc& d, const WgQuery& q, hipStream_t st, int* rc);
