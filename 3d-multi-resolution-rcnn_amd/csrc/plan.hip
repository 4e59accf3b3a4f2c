// Plan executor: one native call walks a host-side list of mr3d_op records and enqueues every
// kernel on the given stream (no allocation, no synchronisation: capturable in a hipGraph).
#include "mr3d_common.h"
#include "conv3d.h"
#include <string.h>
#include <mutex>

namespace {

long long i64_of(const int32_t *lo_hi) {
  return (long long)(((unsigned long long)(uint32_t)lo_hi[1] << 32) | (uint32_t)lo_hi[0]);
}

// shared_ws: the partial-sum workspace of weight-gradient records may be used (records that can run concurrently on
// DIFFERENT streams must not share it: see mr3d_plan_run)
int run_one(const mr3d_op &o, mr3d_stream_t s, bool shared_ws = true, const unsigned char *x_live = nullptr) {
  const int32_t *i = o.i;
  void *const *p = o.p;
  const bool bf16 = i[25] == MR3D_DTYPE_BF16;        // storage type of the activations the record touches
  switch (o.kind) {
    case MR3D_OP_CONV_FWD: {
      mr3d_conv3d_desc d;
      d.x = (const float *)p[0]; d.w = (const float *)p[1]; d.y = (float *)p[2]; d.shift = (const float *)p[3];
      d.add1 = (const float *)p[4]; d.add2 = (const float *)p[5]; d.up = (const float *)p[6]; d.mask = (const float *)p[7];
      memcpy(&d.n, i, sizeof(int) * 25);
      d.workspace = nullptr; d.workspace_bytes = 0;      // transformed weights come from the plan's WINO2D_PREP record, if any
      d.x_live = x_live;                                 // armed by the VOXEL_LIVENESS record in front of this one, if any
      return bf16 ? mr3d_conv3d_forward_bf16(&d, s) : mr3d_conv3d_forward(&d, s);
    }
    case MR3D_OP_CONV_DGRAD_S2: {
      mr3d_conv3d_desc d;
      d.x = (const float *)p[0]; d.w = (const float *)p[1]; d.y = (float *)p[2]; d.shift = nullptr;
      d.add1 = (const float *)p[4]; d.add2 = (const float *)p[5]; d.up = nullptr; d.mask = (const float *)p[7];
      memcpy(&d.n, i, sizeof(int) * 25);
      d.workspace = nullptr; d.workspace_bytes = 0;
      d.x_live = nullptr;
      return bf16 ? mr3d_conv3d_dgrad_s2k3_bf16(&d, s) : mr3d_conv3d_dgrad_s2k3(&d, s);
    }
    case MR3D_OP_CONV_WGRAD: {
      mr3d_conv3d_wgrad_desc d;
      d.x = (const float *)p[0]; d.dy = (const float *)p[1]; d.dw = (float *)p[2]; d.dbias = (float *)p[3];
      memcpy(&d.n, i, sizeof(int) * 15);
      d.flags = i[17];
      d.workspace = shared_ws ? (float *)p[4] : nullptr; d.workspace_bytes = d.workspace ? i64_of(i + 15) : 0;
      return bf16 ? mr3d_conv3d_wgrad_bf16(&d, s) : mr3d_conv3d_wgrad(&d, s);
    }
    case MR3D_OP_VOXEL_LIVENESS:
      return mr3d_voxel_liveness(p[0], bf16 ? 1 : 0, i[0], i[1], i[2], i[3], i[4], (unsigned *)p[1], s);
    case MR3D_OP_PW_PAIR: {
      mr3d_pw_pair_desc d;
      d.x = (const float *)p[0]; d.w1 = (const float *)p[1]; d.add1 = (const float *)p[2]; d.mask1 = (const float *)p[3];
      d.y1 = (float *)p[4]; d.w2 = (const float *)p[5]; d.mask2 = (const float *)p[6]; d.y2 = (float *)p[7];
      d.c0 = i[0]; d.c1 = i[1]; d.c2 = i[2]; d.relu1 = i[3]; d.relu2 = i[4];
      d.positions = i64_of(i + 5);
      d.shift1 = (const float *)(uintptr_t)i64_of(i + 7);
      d.shift2 = (const float *)(uintptr_t)i64_of(i + 9);
      return mr3d_pointwise_pair(&d, s);
    }
    case MR3D_OP_MAXPOOL_FWD:
      if (bf16) return mr3d_maxpool3d_forward_bf16(p[0], p[1], (uint32_t *)p[2], i[0], i[1], i[2], i[3], i[4], s);
      return mr3d_maxpool3d_forward((const float *)p[0], (float *)p[1], (uint32_t *)p[2], i[0], i[1], i[2], i[3], i[4], s);
    case MR3D_OP_MAXPOOL_BWD:
      if (bf16) return mr3d_maxpool3d_backward_bf16(p[0], (const uint32_t *)p[1], p[2], p[3], i[0], i[1], i[2], i[3], i[4], s);
      return mr3d_maxpool3d_backward((const float *)p[0], (const uint32_t *)p[1], (const float *)p[2], (float *)p[3], i[0],
                                     i[1], i[2], i[3], i[4], s);
    case MR3D_OP_ZERO_STUFF2:
      if (bf16) return mr3d_zero_stuff2_bf16(p[0], p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], s);
      return mr3d_zero_stuff2((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], s);
    case MR3D_OP_UPSAMPLE_ADJ:
      if (bf16) return mr3d_upsample_nearest_adjoint_add_bf16(p[0], p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], s);
      return mr3d_upsample_nearest_adjoint_add((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6],
                                               i[7], s);
    case MR3D_OP_SUBSAMPLE_FWD:
      if (bf16) return mr3d_subsample2_forward_bf16(p[0], p[1], i[0], i[1], i[2], i[3], i[4], s);
      return mr3d_subsample2_forward((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], s);
    case MR3D_OP_SUBSAMPLE_BWD:
      if (bf16) return mr3d_subsample2_backward_add_bf16(p[0], p[1], i[0], i[1], i[2], i[3], i[4], s);
      return mr3d_subsample2_backward_add((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], s);
    case MR3D_OP_VOLUME_PACK:
      return mr3d_volume_to_ndhwc4((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], s);
    case MR3D_OP_MEMSET0:
      return hipMemsetAsync(p[0], 0, (size_t)i64_of(i), mr3d_stream(s)) == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
    case MR3D_OP_AXPY:
      return mr3d_axpy((float *)p[0], (const float *)p[1], o.f[0], i64_of(i), s);
    case MR3D_OP_PREP_WEIGHTS:
      return mr3d_prepare_weights((const mr3d_layer_desc *)p[0], i[0], (const float *)p[1], (float *)p[2], s);
    case MR3D_OP_FINISH_GRADS:
      return mr3d_finish_grads((const mr3d_layer_desc *)p[0], i[0], i[1], (const float *)p[1], (const float *)p[2],
                               (float *)p[3], s);
    case MR3D_OP_PREP_WEIGHTS_BF16:
      return mr3d_prepare_weights_bf16((const mr3d_layer_desc *)p[0], i[0], (const float *)p[1], p[2], s);
    case MR3D_OP_STEM_PACK_BF16:
      return mr3d_stem_pack_weights_bf16((const float *)p[0], p[1], s);
    case MR3D_OP_STEM_FWD:
      if (bf16 && i[5] == 1) return mr3d_stem_conv_forward_bf16((const float *)p[0], p[1], (const float *)p[2], p[3], i[0], i[1], i[2], i[3], i[4], s);
      if (bf16) return mr3d_stem_conv_forward_bf16out((const float *)p[0], (const float *)p[1], (const float *)p[2], p[3], i[0], i[1],
                                                      i[2], i[3], i[4], s);
      return mr3d_stem_conv_forward((const float *)p[0], (const float *)p[1], (const float *)p[2], (float *)p[3], i[0], i[1], i[2],
                                    i[3], i[4], s);
    case MR3D_OP_STEM_WGRAD:
      if (bf16 && i[6] == 1 && i[7] == 1)
        return mr3d_stem_conv_wgrad_bf16_pooled((const float *)p[0], p[1], (const uint32_t *)p[5], p[6], (float *)p[2], (float *)p[3],
                                                i[0], i[1], i[2], i[3], s);
      if (bf16 && i[6] == 1) return mr3d_stem_conv_wgrad_bf16((const float *)p[0], p[1], (float *)p[2], (float *)p[3], i[0], i[1], i[2], i[3], s);
      if (bf16) return mr3d_stem_conv_wgrad_bf16dy((const float *)p[0], p[1], (float *)p[2], (float *)p[3], i[0], i[1], i[2], i[3], s);
      if (i[7] == 1)
        return mr3d_stem_conv_wgrad_pooled((const float *)p[0], (const float *)p[1], (const uint32_t *)p[5], (const float *)p[6],
                                           (float *)p[2], (float *)p[3], i[0], i[1], i[2], i[3], shared_ws ? (float *)p[4] : nullptr,
                                           (shared_ws && p[4]) ? i64_of(i + 4) : 0, s);
      return mr3d_stem_conv_wgrad_ws((const float *)p[0], (const float *)p[1], (float *)p[2], (float *)p[3], i[0], i[1], i[2], i[3],
                                     shared_ws ? (float *)p[4] : nullptr, (shared_ws && p[4]) ? i64_of(i + 4) : 0, s);
  }
  return MR3D_ERR_INVALID_ARG;
}

}  // namespace

namespace {

// Side stream for weight-gradient kernels.  In the backward pass the data-gradient chain is the critical path;
// every wgrad only READS (x, dy) and adds into its own scratch slice, so it can run beside the chain.  Many of
// them are small (layer3/4: tens of workgroups) and would otherwise leave most of the 256 CUs idle one after
// another.  Fork: event recorded on the caller's stream right before the op -> side stream waits on it.
// Join: before FINISH_GRADS and at the end of every call.
struct SideStream {
  hipStream_t stream = nullptr;
  // second side queue: CONV_WGRAD records alternate between the two (they only read finished tensors and add into their own
  // scratch slices), every other side record stays on `stream`.  One in-order queue ran the many small weight gradients of the
  // large maps one at a time -- 25-85 us kernels that fill a fraction of the chip -- and fell behind the main chain once that got
  // faster: 0.5 ms of them ran alone, with 6 us launch gaps, after the stem weight gradient (profiles/r03).
  hipStream_t stream2 = nullptr;
  hipEvent_t ev12 = nullptr, ev21 = nullptr;
  hipEvent_t join_ev = nullptr;
  hipEvent_t fork[16] = {};
  hipEvent_t mark[8] = {};          // MR3D_OP_SYNC slots: a point of the side queue the caller's stream can wait for
  // a slot is ARMED while its event stands for a point of a real (not captured) run that nobody outside the plan has consumed
  // yet: mr3d_plan_stream_wait_mark_checked only waits on an armed slot and tells its caller when it did not
  bool mark_armed[8] = {};
  int next = 0;
  bool ok = false;
  bool init() {
    if (ok) return true;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // The side stream asks for the HIGH priority class so that the runtime gives it a hardware queue of its own: a
    // default-priority stream can be folded onto the same hardware queue as the caller's stream once other
    // libraries hold streams too -- measured with an RCCL communicator alive: 17.4 instead of 15.6 ms/step, all of
    // the weight-gradient overlap gone (kernel durations unchanged).
    // (Round 3, headline step: high 11.10 ms, normal 11.23, low 11.24 -- the weight gradients finishing early matters more
    // than the main chain's kernels keeping every CU.)
    // Round 5 again (8.9 ms step): high 8.97, middle 8.89, low 8.99 on a bare process -- with the side queues in the high class the
    // dispatcher serves a long weight gradient's workgroups first and 20 us convolutions of the main chain take 100-300 us beside it
    // -- but with an RCCL communicator alive the middle class folds onto the caller's queue (10.30 against 9.02 ms), so high stays.
    // Measured and not kept to get both: GPU_MAX_HW_QUEUES = 8 / 16 (13.4-14.2 ms with the communicator alive); the main chain on a
    // third high-class queue of the executor's own (backbone step 8.90 against 8.85 ms, the two-stage step 18-22 against 12.4 ms: one
    // more stream of the class and the runtime folds queues elsewhere); the caller running the whole step on a high-priority torch
    // stream (-0.04 ms, -0.07 with the communicator: a caller's choice).
    // MR3D_TUNE_SIDE_PRIORITY (set BEFORE the first plan run of the process): 0 = high (the default), 1 = the middle of the range, 2 = low
    const int want = g_mr3d_tuning[MR3D_TUNE_SIDE_PRIORITY];
    const int prio = want == 2 ? lo : (want == 1 ? (lo + hi) / 2 : hi);
    if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio) != hipSuccess) return false;
    if (hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, prio) != hipSuccess) return false;
    // Fork / join events order kernels of ONE device across its queues: the kernels' own agent-scope release / acquire (every
    // dispatch packet carries them -- consecutive kernels of one queue on different XCDs depend on the same) is all the visibility
    // a consumer on another queue needs.  Without hipEventDisableSystemFence the event's packet also writes the caches back to
    // system scope for a HOST reader that never comes, and the main-chain kernel behind it starts later: bf16 backbone step
    // 4.82 -> 4.72 ms, fp32 9.79 -> 9.73 (MR3D_TUNE_FORK_SYSTEM_FENCE = 1 before the first plan run restores the fence: A/B).
    // The `mark` events stay as they were: other libraries' streams (RCCL) wait on them.
    const unsigned fork_flags = hipEventDisableTiming | (g_mr3d_tuning[MR3D_TUNE_FORK_SYSTEM_FENCE] ? 0u : hipEventDisableSystemFence);
    if (hipEventCreateWithFlags(&join_ev, fork_flags) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&ev12, fork_flags) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&ev21, fork_flags) != hipSuccess) return false;
    for (auto &e : fork)
      if (hipEventCreateWithFlags(&e, fork_flags) != hipSuccess) return false;
    for (auto &e : mark)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    ok = true;
    return true;
  }
};
// one set of side streams / events per DEVICE (a process that drives several GPUs must not fork onto another device's
// stream), and one enqueue at a time per device (the fork-event ring is shared by the plans of that device)
constexpr int kMaxDevices = 16;
SideStream g_sides[kMaxDevices];
std::mutex g_side_mu[kMaxDevices];
int g_side_enabled = 1;

// records that may run beside the caller's stream: they read finished tensors and write buffers nobody on the main chain
// touches before the next join / wait (weight gradients, the table-driven weight preparation and gradient finishing, the
// zero-fill of the weight-gradient scratch)
bool wants_side(const mr3d_op &o) {
  // ... and, since round 3, convolutions / subsampling / liveness bitmaps of pyramid levels whose results the main chain
  // only needs behind a SYNC record: the P3..P5 output convolutions (and their data gradients) run beside the P2 one
  return o.i[26] == 1 && (o.kind == MR3D_OP_CONV_WGRAD || o.kind == MR3D_OP_STEM_WGRAD || o.kind == MR3D_OP_PREP_WEIGHTS ||
                          o.kind == MR3D_OP_PREP_WEIGHTS_BF16 || o.kind == MR3D_OP_MEMSET0 || o.kind == MR3D_OP_FINISH_GRADS ||
                          o.kind == MR3D_OP_CONV_FWD || o.kind == MR3D_OP_SUBSAMPLE_FWD || o.kind == MR3D_OP_VOXEL_LIVENESS);
}

}  // namespace

extern "C" void mr3d_plan_set_side_stream(int enable) { g_side_enabled = enable; }

namespace {

int plan_run_impl(const mr3d_op *ops, int num_ops, int *failed_index, mr3d_stream_t stream, const int *marks, int num_marks,
                  mr3d_event_pair *pairs) {
  if (!ops || num_ops < 0) return MR3D_ERR_INVALID_ARG;
  hipStream_t main_s = mr3d_stream(stream);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  std::lock_guard<std::mutex> lock(g_side_mu[dev]);
  SideStream &g_side = g_sides[dev];
  // a single-op call (used to time one kernel) always runs on the caller's stream
  const bool use_side = g_side_enabled && num_ops > 1 && g_side.init();
  bool side_busy = false;
  bool s2_busy = false;          // the second side queue holds work the first has not waited for
  bool s1_ahead = false;         // the first side queue holds a non-wgrad record (scratch zero-fill, ...) the second has not waited for
  bool use2 = false;             // which side queue the next CONV_WGRAD record takes
  bool main_dirty = true;        // the caller's stream got work since the last fork event
  hipEvent_t last_fork = nullptr;
  // a plan that is being captured into a hipGraph keeps to ONE side queue (mr3d_plan_capture: the capture of a second forked
  // stream that is folded back through the first crashed inside the runtime of this image)
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  const bool two_side_queues = hipStreamIsCapturing(main_s, &cap_status) == hipSuccess && cap_status == hipStreamCaptureStatusNone;
  const bool capturing = !two_side_queues;
  auto fold2 = [&]() {           // the first side queue waits for everything queued on the second
    if (!s2_busy) return true;
    s2_busy = false;
    return hipEventRecord(g_side.ev21, g_side.stream2) == hipSuccess && hipStreamWaitEvent(g_side.stream, g_side.ev21, 0) == hipSuccess;
  };
  auto join = [&]() {
    if (!side_busy) return true;
    side_busy = false;
    return fold2() && hipEventRecord(g_side.join_ev, g_side.stream) == hipSuccess &&
           hipStreamWaitEvent(main_s, g_side.join_ev, 0) == hipSuccess;
  };
  // WINO2D_PREP record at k0: the weight transforms of the two-dimensional Winograd convolutions (conv3d_wino2d.hip) of the
  // records behind it, up to the next weight-writing record, in ONE launch into the plan-owned workspace (one launch in
  // front of every such convolution cost 0.5 ms of an 11.8 ms step)
  auto prepare_wino2d = [&](int k0, float *ws, long long ws_bytes) {
    const float *w[32];
    int cin[32], cout[32], n = 0;
    for (int k1 = k0 + 1; k1 < num_ops; ++k1) {
      const mr3d_op &o = ops[k1];
      if (o.kind == MR3D_OP_PREP_WEIGHTS || o.kind == MR3D_OP_PREP_WEIGHTS_BF16 || o.kind == MR3D_OP_WINO2D_PREP) break;
      if (o.kind != MR3D_OP_CONV_FWD || o.i[25] == MR3D_DTYPE_BF16 || n == 32) continue;
      mr3d_conv3d_desc d{};
      d.w = (const float *)o.p[1];
      memcpy(&d.n, o.i, sizeof(int) * 25);
      d.workspace_bytes = -1;
      mr3d_conv3d_variant v;
      if (mr3d_conv3d_forward_variant(&d, &v) != MR3D_OK || v.winograd != 8) continue;
      bool seen = false;
      for (int j = 0; j < n; ++j) seen = seen || w[j] == d.w;
      if (!seen) { w[n] = d.w; cin[n] = d.cin; cout[n] = d.cout; ++n; }
    }
    return mr3d_conv3d_wino2d_prepare(w, cin, cout, n, ws, ws_bytes, main_s);
  };
  struct PrepEnd { ~PrepEnd() { mr3d_conv3d_wino2d_prepare_end(); } } prep_end;
  int next_mark = 0;
  auto fail = [&](int k, int st) {
    if (failed_index) *failed_index = k;
    join();
    return st;
  };
  constexpr int kMaxDwSeen = 256;
  const void *dw_seen[kMaxDwSeen];
  bool dw_queue[kMaxDwSeen];
  int n_dw_seen = 0;
  const int skip_mask = g_mr3d_tuning[MR3D_TUNE_PLAN_SKIP];
  const unsigned char *armed_live = nullptr;          // set by a VOXEL_LIVENESS record with arm = 1, taken by the next CONV_FWD
  for (int k = 0; k < num_ops; ++k) {
    int st = MR3D_OK;
    if (ops[k].kind == MR3D_OP_SYNC) {
      // 0: the caller's stream waits for everything queued on the side stream so far; 1: remember this point of the side
      // queue in slot i[1]; 2: the caller's stream waits for the point remembered in slot i[1]
      const int mode = ops[k].i[0], slot = ops[k].i[1] & 7;
      // a mark of this run is only worth waiting for if the event is really recorded NOW: not with the side stream off (the
      // record is skipped) and not under capture (the record becomes a graph node; a replay does not touch the event)
      if (mode == 1) g_side.mark_armed[slot] = use_side && !capturing;
      if (use_side) {
        bool ok = true;
        if (mode != 1) main_dirty = true;          // the caller's stream took on a dependency: later forks must carry it
        if (mode == 0) ok = join();
        // (a mark is a point of the FIRST side queue: the records it stands for -- zero-fill, finishing, CONV_FWD -- run there, and a
        // finishing record has already waited for the second queue's weight gradients)
        else if (mode == 1) ok = hipEventRecord(g_side.mark[slot], g_side.stream) == hipSuccess;
        else if (mode == 2) ok = hipStreamWaitEvent(main_s, g_side.mark[slot], 0) == hipSuccess;
        if (!ok) return fail(k, MR3D_ERR_LAUNCH);
      }
      continue;
    }
    if (ops[k].kind == MR3D_OP_WINO2D_PREP) {
      mr3d_conv3d_wino2d_prepare_end();
      main_dirty = true;
      st = prepare_wino2d(k, (float *)ops[k].p[0], i64_of(ops[k].i));
      if (st != MR3D_OK) return fail(k, st);
      continue;
    }
    if (skip_mask) {          // MR3D_TUNE_PLAN_SKIP: sensitivity runs (what would the step cost without this class of records?)
      const mr3d_op &o = ops[k];
      const int32_t *i = o.i;
      bool skip = false;
      if (o.kind == MR3D_OP_CONV_WGRAD) {
        skip = (skip_mask & 4) || ((skip_mask & 1) && i[6] == 3 && i[4] == 16 && i[5] == 16) ||
               ((skip_mask & 2) && i[6] == 3 && i[4] == 64 && i[5] == 64) || ((skip_mask & 64) && i[6] == 1);
      } else if (o.kind == MR3D_OP_MAXPOOL_FWD) skip = skip_mask & 8;
      else if (o.kind == MR3D_OP_FINISH_GRADS) skip = skip_mask & 16;
      else if (o.kind == MR3D_OP_STEM_WGRAD) skip = skip_mask & 32;
      else if (o.kind == MR3D_OP_STEM_FWD) skip = skip_mask & 256;
      else if (o.kind == MR3D_OP_ZERO_STUFF2 || o.kind == MR3D_OP_UPSAMPLE_ADJ || o.kind == MR3D_OP_MAXPOOL_BWD) skip = skip_mask & 512;
      else if (o.kind == MR3D_OP_CONV_FWD) skip = (skip_mask & 128) && i[6] == 3 && i[4] >= 64 && (long long)i[0] * i[1] * i[2] * i[3] < 32768;
      if (skip) continue;
    }
    const bool on_side = use_side && wants_side(ops[k]);
    bool on_second = false;
    // (the bf16 backbone step measured 5.50 ms on one side queue and 5.57 on two in round 4 -- its queue had slack; at 4.3 ms it ends
    //  with the side queue's last weight gradients: MR3D_TUNE_BF16_ONE_SIDE_QUEUE)
    if (on_side && two_side_queues && ops[k].kind == MR3D_OP_CONV_WGRAD &&
        (ops[k].i[25] != MR3D_DTYPE_BF16 || g_mr3d_tuning[MR3D_TUNE_BF16_ONE_SIDE_QUEUE] == 0)) {
      // records that accumulate into the SAME dw must stay on one queue (the first queue may reduce a lent partial-sum workspace
      // into dw without atomics, the second adds atomically: two queues on one dw would race) -- e.g. the RPN head's five
      // levels share dw1 / dw3.  A new dw takes the queues in turn.
      const void *dw = ops[k].p[2];
      int seen = -1;
      for (int j = 0; j < n_dw_seen; ++j)
        if (dw_seen[j] == dw) { seen = j; break; }
      if (seen >= 0) {
        on_second = dw_queue[seen];
      } else {
        on_second = use2;
        use2 = !use2;
        if (n_dw_seen < kMaxDwSeen) { dw_seen[n_dw_seen] = dw; dw_queue[n_dw_seen++] = on_second; }
        else on_second = false;          // table full (never with the shipped plans): the first queue is always safe
      }
    }
    hipStream_t run_s = on_side ? (on_second ? g_side.stream2 : g_side.stream) : main_s;
    const bool marked = marks && next_mark < num_marks && marks[next_mark] == k;
    if (on_side) {
      // Fork: the side queue waits for the caller's stream as of NOW.  Consecutive side records share one event: an event record
      // is a packet in the caller's queue, and the kernel behind it starts ~6 us later than it would back to back (visible as
      // device-idle gaps in front of main-chain kernels once the side queues had slack: profiles/r03).
      if (main_dirty || !last_fork) {
        last_fork = g_side.fork[g_side.next++ & 15];
        if (hipEventRecord(last_fork, main_s) != hipSuccess) return fail(k, MR3D_ERR_LAUNCH);
        main_dirty = false;
      }
      if (hipStreamWaitEvent(run_s, last_fork, 0) != hipSuccess) return fail(k, MR3D_ERR_LAUNCH);
      if (on_second) {
        if (s1_ahead) {          // e.g. the zero-fill of the weight-gradient scratch
          s1_ahead = false;
          if (hipEventRecord(g_side.ev12, g_side.stream) != hipSuccess || hipStreamWaitEvent(g_side.stream2, g_side.ev12, 0) != hipSuccess)
            return fail(k, MR3D_ERR_LAUNCH);
        }
        s2_busy = true;
      } else if (ops[k].kind != MR3D_OP_CONV_WGRAD) {
        if (!fold2()) return fail(k, MR3D_ERR_LAUNCH);          // finishing / zero-fill / ... behind every weight gradient
        s1_ahead = true;
      }
      side_busy = true;
    } else if (ops[k].kind == MR3D_OP_FINISH_GRADS && !join()) {
      return fail(k, MR3D_ERR_LAUNCH);
    }
    if (marked) {        // timing events on the stream the record really runs on; the caller reads them after a sync
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return fail(k, MR3D_ERR_LAUNCH);
      pairs[next_mark].start = a; pairs[next_mark].stop = b;
      if (hipEventRecord(a, run_s) != hipSuccess) return fail(k, MR3D_ERR_LAUNCH);
    }
    const unsigned char *x_live = nullptr;
    if (ops[k].kind == MR3D_OP_CONV_FWD) { x_live = armed_live; armed_live = nullptr; }
    if (!on_side) main_dirty = true;
    st = run_one(ops[k], (mr3d_stream_t)run_s, !on_second, x_live);      // the lent partial-sum workspace belongs to the first side queue
    if (ops[k].kind == MR3D_OP_VOXEL_LIVENESS && ops[k].i[5] == 1) armed_live = (const unsigned char *)ops[k].p[1];
    if (marked) {
      if (hipEventRecord((hipEvent_t)pairs[next_mark].stop, run_s) != hipSuccess && st == MR3D_OK) st = MR3D_ERR_LAUNCH;
      ++next_mark;
    }
    if (st != MR3D_OK) return fail(k, st);
  }
  return join() ? MR3D_OK : MR3D_ERR_LAUNCH;
}

}  // namespace

extern "C" int mr3d_plan_run(const mr3d_op *ops, int num_ops, int *failed_index, mr3d_stream_t stream) {
  return plan_run_impl(ops, num_ops, failed_index, stream, nullptr, 0, nullptr);
}

extern "C" int mr3d_plan_run_marked(const mr3d_op *ops, int num_ops, int *failed_index, mr3d_stream_t stream, const int *marks,
                                    int num_marks, mr3d_event_pair *pairs) {
  if (num_marks < 0 || (num_marks > 0 && (!marks || !pairs))) return MR3D_ERR_INVALID_ARG;
  for (int j = 0; j < num_marks; ++j) {
    if (marks[j] < 0 || marks[j] >= num_ops || (j > 0 && marks[j] <= marks[j - 1]) || ops[marks[j]].kind == MR3D_OP_SYNC ||
        ops[marks[j]].kind == MR3D_OP_WINO2D_PREP)
      return MR3D_ERR_INVALID_ARG;
    pairs[j].start = pairs[j].stop = nullptr;
  }
  return plan_run_impl(ops, num_ops, failed_index, stream, marks, num_marks, pairs);
}

// Lets a stream of the CALLER (e.g. the communication stream of the gradient exchange) wait for a point of the side queue
// that a SYNC(mark) record remembered during the last plan run on this device.
extern "C" int mr3d_plan_stream_wait_mark(int slot, mr3d_stream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || slot < 0 || slot > 7) return MR3D_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_side_mu[dev]);
  SideStream &g_side = g_sides[dev];
  if (!g_side.ok) return MR3D_OK;               // no side stream was ever used: everything ran on the caller's stream
  return hipStreamWaitEvent(mr3d_stream(stream), g_side.mark[slot], 0) == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
}

extern "C" int mr3d_plan_stream_wait_mark_checked(int slot, mr3d_stream_t stream, int *waited) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || slot < 0 || slot > 7 || !waited) return MR3D_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_side_mu[dev]);
  SideStream &g_side = g_sides[dev];
  *waited = 0;
  if (!g_side.ok || !g_side.mark_armed[slot]) return MR3D_OK;
  if (hipStreamWaitEvent(mr3d_stream(stream), g_side.mark[slot], 0) != hipSuccess) return MR3D_ERR_LAUNCH;
  g_side.mark_armed[slot] = false;          // one consumer per recorded mark: the next run has to record it again
  *waited = 1;
  return MR3D_OK;
}

extern "C" int mr3d_event_pair_elapsed_ms(const mr3d_event_pair *pair, float *ms) {
  if (!pair || !pair->start || !pair->stop || !ms) return MR3D_ERR_INVALID_ARG;
  return hipEventElapsedTime(ms, (hipEvent_t)pair->start, (hipEvent_t)pair->stop) == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
}

extern "C" void mr3d_event_pair_destroy(mr3d_event_pair *pair) {
  if (!pair) return;
  if (pair->start) (void)hipEventDestroy((hipEvent_t)pair->start);
  if (pair->stop) (void)hipEventDestroy((hipEvent_t)pair->stop);
  pair->start = pair->stop = nullptr;
}

// ---------------------------------------------------------------------------------------------- hipGraph replay
// A plan touches only pre-allocated buffers and passes every scalar by value, so its whole launch sequence --
// side-stream forks and joins included -- can be captured ONCE and replayed with a single hipGraphLaunch: the
// ~170 / ~330 launches of a forward / backward pass then cost no per-kernel host work and the device sees the
// dependency edges instead of two in-order queues.
struct mr3d_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

extern "C" int mr3d_plan_capture(const mr3d_op *ops, int num_ops, mr3d_graph **out) {
  if (!ops || num_ops <= 0 || !out) return MR3D_ERR_INVALID_ARG;
  static hipStream_t cap = nullptr;
  if (!cap && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) return MR3D_ERR_LAUNCH;
  if (hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return MR3D_ERR_LAUNCH; }
  int failed = -1;
  const int st = mr3d_plan_run(ops, num_ops, &failed, (mr3d_stream_t)cap);
  mr3d_graph *g = new mr3d_graph();
  const hipError_t e = hipStreamEndCapture(cap, &g->graph);
  if (st != MR3D_OK || e != hipSuccess || !g->graph) {
    (void)hipGetLastError();
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return st != MR3D_OK ? st : MR3D_ERR_LAUNCH;
  }
  if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipGraphDestroy(g->graph);
    delete g;
    return MR3D_ERR_LAUNCH;
  }
  *out = g;
  return MR3D_OK;
}

extern "C" int mr3d_graph_launch(mr3d_graph *g, mr3d_stream_t stream) {
  if (!g || !g->exec) return MR3D_ERR_INVALID_ARG;
  return hipGraphLaunch(g->exec, mr3d_stream(stream)) == hipSuccess ? MR3D_OK : MR3D_ERR_LAUNCH;
}

extern "C" void mr3d_graph_destroy(mr3d_graph *g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}
