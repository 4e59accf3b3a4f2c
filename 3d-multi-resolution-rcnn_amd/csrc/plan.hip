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
int run_one(const mr3d_op &o, mr3d_stream_t s, bool shared_ws = true) {
  const int32_t *i = o.i;
  void *const *p = o.p;
  const bool bf16 = i[25] == MR3D_DTYPE_BF16;        // storage type of the activations the record touches
  switch (o.kind) {
    case MR3D_OP_CONV_FWD: {
      mr3d_conv3d_desc d;
      d.x = (const float *)p[0]; d.w = (const float *)p[1]; d.y = (float *)p[2]; d.shift = (const float *)p[3];
      d.add1 = (const float *)p[4]; d.add2 = (const float *)p[5]; d.up = (const float *)p[6]; d.mask = (const float *)p[7];
      memcpy(&d.n, i, sizeof(int) * 25);
      return bf16 ? mr3d_conv3d_forward_bf16(&d, s) : mr3d_conv3d_forward(&d, s);
    }
    case MR3D_OP_CONV_DGRAD_S2: {
      mr3d_conv3d_desc d;
      d.x = (const float *)p[0]; d.w = (const float *)p[1]; d.y = (float *)p[2]; d.shift = nullptr;
      d.add1 = (const float *)p[4]; d.add2 = (const float *)p[5]; d.up = nullptr; d.mask = (const float *)p[7];
      memcpy(&d.n, i, sizeof(int) * 25);
      return mr3d_conv3d_dgrad_s2k3(&d, s);
    }
    case MR3D_OP_CONV_WGRAD: {
      mr3d_conv3d_wgrad_desc d;
      d.x = (const float *)p[0]; d.dy = (const float *)p[1]; d.dw = (float *)p[2]; d.dbias = (float *)p[3];
      memcpy(&d.n, i, sizeof(int) * 15);
      d.reserved = 0;
      d.workspace = shared_ws ? (float *)p[4] : nullptr; d.workspace_bytes = d.workspace ? i64_of(i + 15) : 0;
      return bf16 ? mr3d_conv3d_wgrad_bf16(&d, s) : mr3d_conv3d_wgrad(&d, s);
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
      if (bf16 && i[6] == 1) return mr3d_stem_conv_wgrad_bf16((const float *)p[0], p[1], (float *)p[2], (float *)p[3], i[0], i[1], i[2], i[3], s);
      if (bf16) return mr3d_stem_conv_wgrad_bf16dy((const float *)p[0], p[1], (float *)p[2], (float *)p[3], i[0], i[1], i[2], i[3], s);
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
  hipStream_t streams[4] = {};
  int nstreams = 1;
  hipEvent_t joins[4] = {};
  hipEvent_t fork[16] = {};
  int next = 0;
  bool ok = false;
  bool init() {
    if (ok) return true;
    if (const char *e = getenv("MR3D_SIDE_STREAMS")) nstreams = atoi(e) < 1 ? 1 : (atoi(e) > 4 ? 4 : atoi(e));
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // The side stream asks for the HIGH priority class so that the runtime gives it a hardware queue of its own: a
    // default-priority stream can be folded onto the same hardware queue as the caller's stream once other
    // libraries hold streams too -- measured with an RCCL communicator alive: 17.4 instead of 15.6 ms/step, all of
    // the weight-gradient overlap gone (kernel durations unchanged).  MR3D_SIDE_PRIO=0 / 1 select normal / low.
    int prio = hi;
    if (const char *e = getenv("MR3D_SIDE_PRIO")) prio = atoi(e) > 0 ? lo : (atoi(e) < 0 ? hi : 0);
    for (int i = 0; i < nstreams; ++i) {
      if (hipStreamCreateWithPriority(&streams[i], hipStreamNonBlocking, prio) != hipSuccess) return false;
      if (hipEventCreateWithFlags(&joins[i], hipEventDisableTiming) != hipSuccess) return false;
    }
    for (auto &e : fork)
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

bool wants_side(const mr3d_op &o) {
  return o.i[26] == 1 && (o.kind == MR3D_OP_CONV_WGRAD || o.kind == MR3D_OP_STEM_WGRAD);
}

}  // namespace

extern "C" void mr3d_plan_set_side_stream(int enable) { g_side_enabled = enable; }

extern "C" int mr3d_plan_run(const mr3d_op *ops, int num_ops, int *failed_index, mr3d_stream_t stream) {
  if (!ops || num_ops < 0) return MR3D_ERR_INVALID_ARG;
  hipStream_t main_s = mr3d_stream(stream);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  std::lock_guard<std::mutex> lock(g_side_mu[dev]);
  SideStream &g_side = g_sides[dev];
  // a single-op call (used to time one kernel) always runs on the caller's stream
  const bool use_side = g_side_enabled && num_ops > 1 && g_side.init();
  bool side_busy = false;
  int nside = 0;
  auto join = [&]() {
    if (!side_busy) return true;
    side_busy = false;
    for (int i = 0; i < g_side.nstreams; ++i)
      if (hipEventRecord(g_side.joins[i], g_side.streams[i]) != hipSuccess || hipStreamWaitEvent(main_s, g_side.joins[i], 0) != hipSuccess)
        return false;
    return true;
  };
  // Weight transforms of the two-dimensional Winograd convolutions (conv3d_wino2d.hip): ONE launch per stretch of records
  // between weight-writing records instead of one in front of every such convolution (16 per step: 0.5 ms of 11.8).
  int prepared_upto = 0;
  auto prepare_wino2d = [&](int k0) {
    int k1 = k0;
    const float *w[32];
    int cin[32], cout[32], n = 0;
    for (; k1 < num_ops; ++k1) {
      const mr3d_op &o = ops[k1];
      if (k1 > k0 && (o.kind == MR3D_OP_PREP_WEIGHTS || o.kind == MR3D_OP_PREP_WEIGHTS_BF16)) break;      // they rewrite weights
      if (o.kind != MR3D_OP_CONV_FWD || o.i[25] == MR3D_DTYPE_BF16 || n == 32) continue;
      mr3d_conv3d_desc d{};
      d.w = (const float *)o.p[1];
      memcpy(&d.n, o.i, sizeof(int) * 25);
      mr3d_conv3d_variant v;
      if (mr3d_conv3d_forward_variant(&d, &v) != MR3D_OK || v.winograd != 8) continue;
      bool seen = false;
      for (int j = 0; j < n; ++j) seen = seen || w[j] == d.w;
      if (!seen) { w[n] = d.w; cin[n] = d.cin; cout[n] = d.cout; ++n; }
    }
    prepared_upto = k1;
    return mr3d_conv3d_wino2d_prepare(w, cin, cout, n, main_s);
  };
  struct PrepEnd { ~PrepEnd() { mr3d_conv3d_wino2d_prepare_end(); } } prep_end;
  for (int k = 0; k < num_ops; ++k) {
    int st;
    if (ops[k].kind == MR3D_OP_CONV_FWD && k >= prepared_upto && num_ops > 1) {
      st = prepare_wino2d(k);
      if (st != MR3D_OK) {
        if (failed_index) *failed_index = k;
        join();
        return st;
      }
    }
    if (use_side && wants_side(ops[k])) {
      hipEvent_t ev = g_side.fork[g_side.next++ & 15];
      hipStream_t ss = g_side.streams[nside++ % g_side.nstreams];
      if (hipEventRecord(ev, main_s) != hipSuccess || hipStreamWaitEvent(ss, ev, 0) != hipSuccess) {
        if (failed_index) *failed_index = k;
        return MR3D_ERR_LAUNCH;
      }
      // one side stream: its records run in order and may share the partial-sum workspace; with several they could overlap
      st = run_one(ops[k], (mr3d_stream_t)ss, g_side.nstreams == 1);
      side_busy = true;
    } else {
      if (ops[k].kind == MR3D_OP_FINISH_GRADS && !join()) st = MR3D_ERR_LAUNCH;
      else st = run_one(ops[k], stream);
    }
    if (st != MR3D_OK) {
      if (failed_index) *failed_index = k;
      join();
      return st;
    }
  }
  return join() ? MR3D_OK : MR3D_ERR_LAUNCH;
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
