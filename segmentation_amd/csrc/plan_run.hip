// plan_run.hip -- seg_plan_run: a compiled launch plan enqueued by ONE host call (include/seg_hip.h, "A whole launch plan ...").
// Replaces the per-launch Python interpreter loop of engine.Plan.run for the train step of /root/reference/models/basemodel.py:480-489
// (forward + loss + backward + Adam as ~100 launches on three streams): VERDICT r03 item 4a.
#include "common.h"
#include <string.h>

struct SegThunk { const char* name; int (*fn)(const seg_arg*, void*); int nargs; };
#include "plan_thunks.inc"

#include <mutex>
#include <vector>

static const int N_THUNKS = (int)(sizeof(g_thunks) / sizeof(g_thunks[0]));

// Fork events are never destroyed: a plan that goes away hands them back to this pool (its owner may be garbage-collected at any
// time -- e.g. while another thread captures a hipGraph, where hipEventDestroy is not a safe call) and later plans take them again.
// Re-recording a pooled event cannot disturb a wait issued earlier: a wait refers to the record that preceded it.
static std::mutex g_ev_mutex;
static std::vector<hipEvent_t> g_ev_free;
static int g_conv2d_id = -2;

extern "C" int seg_plan_fn_id(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < N_THUNKS; ++i) if (!strcmp(g_thunks[i].name, name)) return i;
  return -1;
}

extern "C" int seg_plan_run(seg_plan_op* ops, int32_t n, void* const* streams, int32_t n_streams, uint32_t* signal_flag, uint32_t signal_base,
                            int32_t* failed_op) {
  if (failed_op) *failed_op = -1;
  if (!ops || n < 0 || !streams || n_streams <= 0) { seg_set_error("plan_run: bad arguments"); return SEG_ERR_ARG; }
  if (g_conv2d_id == -2) g_conv2d_id = seg_plan_fn_id("seg_conv2d");
  for (int i = 0; i < n; ++i) {
    seg_plan_op& op = ops[i];
    if (op.stream < 0 || op.stream >= n_streams || (op.kind == SEG_OP_EVENT_FORK && (op.stream2 < 0 || op.stream2 >= n_streams))) {
      if (failed_op) *failed_op = i;
      seg_set_error("plan_run: op %d names stream %d / %d of %d", i, op.stream, op.stream2, n_streams); return SEG_ERR_ARG;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(streams[op.stream]);
    if (op.kind == SEG_OP_LAUNCH) {
      if (op.fn < 0 || op.fn >= N_THUNKS || op.nargs != g_thunks[op.fn].nargs || !op.args) {
        if (failed_op) *failed_op = i;
        seg_set_error("plan_run: op %d: entry point %d takes %d arguments, got %d", i, op.fn, op.fn >= 0 && op.fn < N_THUNKS ? g_thunks[op.fn].nargs : -1, op.nargs);
        return SEG_ERR_ARG;
      }
      if (op.fn == g_conv2d_id) {
        // (the descriptor is host memory of the plan's owner and is reused by every run: always rewritten)
        seg_conv_desc* d = reinterpret_cast<seg_conv_desc*>(op.args[0].p);
        d->signal = op.signal_slot > 0 ? signal_flag : nullptr;
        d->signal_value = op.signal_slot > 0 ? ((signal_base + (uint32_t)op.signal_slot) & 0x7fffffffu) : 0u;
      }
      const int rc = g_thunks[op.fn].fn(op.args, streams[op.stream]);
      if (rc != SEG_OK) { if (failed_op) *failed_op = i; return rc; }
    } else if (op.kind == SEG_OP_EVENT_FORK) {
      hipEvent_t ev = reinterpret_cast<hipEvent_t>(op.event);
      if (!ev) {
        {
          std::lock_guard<std::mutex> lk(g_ev_mutex);
          if (!g_ev_free.empty()) { ev = g_ev_free.back(); g_ev_free.pop_back(); }
        }
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { if (failed_op) *failed_op = i; seg_set_error("plan_run: cannot create an event"); return SEG_ERR_LAUNCH; }
        op.event = ev;
      }
      if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(reinterpret_cast<hipStream_t>(streams[op.stream2]), ev, 0) != hipSuccess) {
        if (failed_op) *failed_op = i;
        seg_set_error("plan_run: event fork failed at op %d", i); return SEG_ERR_LAUNCH;
      }
    } else if (op.kind == SEG_OP_WAIT_VALUE) {
      const uint32_t v = (signal_base + (uint32_t)op.signal_slot) & 0x7fffffffu;
      if (!signal_flag || hipStreamWaitValue32(st, signal_flag, v, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
        if (failed_op) *failed_op = i;
        seg_set_error("plan_run: hipStreamWaitValue32 failed at op %d", i); return SEG_ERR_LAUNCH;
      }
    } else {
      if (failed_op) *failed_op = i;
      seg_set_error("plan_run: op %d has unknown kind %d", i, op.kind); return SEG_ERR_ARG;
    }
  }
  return SEG_OK;
}

extern "C" int seg_plan_destroy_events(seg_plan_op* ops, int32_t n) {
  if (!ops) return SEG_OK;
  std::lock_guard<std::mutex> lk(g_ev_mutex);
  for (int i = 0; i < n; ++i) if (ops[i].kind == SEG_OP_EVENT_FORK && ops[i].event) { g_ev_free.push_back(reinterpret_cast<hipEvent_t>(ops[i].event)); ops[i].event = nullptr; }
  return SEG_OK;
}
