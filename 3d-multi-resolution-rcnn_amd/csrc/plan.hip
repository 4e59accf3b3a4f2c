// Plan executor: one native call walks a host-side list of mr3d_op records and enqueues every
// kernel on the given stream (no allocation, no synchronisation: capturable in a hipGraph).
#include "mr3d_common.h"
#include <string.h>

namespace {

long long i64_of(const int32_t *lo_hi) {
  return (long long)(((unsigned long long)(uint32_t)lo_hi[1] << 32) | (uint32_t)lo_hi[0]);
}

int run_one(const mr3d_op &o, mr3d_stream_t s) {
  const int32_t *i = o.i;
  void *const *p = o.p;
  switch (o.kind) {
    case MR3D_OP_CONV_FWD: {
      mr3d_conv3d_desc d;
      d.x = (const float *)p[0]; d.w = (const float *)p[1]; d.y = (float *)p[2]; d.shift = (const float *)p[3];
      d.add1 = (const float *)p[4]; d.add2 = (const float *)p[5]; d.up = (const float *)p[6]; d.mask = (const float *)p[7];
      memcpy(&d.n, i, sizeof(int) * 25);
      return mr3d_conv3d_forward(&d, s);
    }
    case MR3D_OP_CONV_WGRAD: {
      mr3d_conv3d_wgrad_desc d;
      d.x = (const float *)p[0]; d.dy = (const float *)p[1]; d.dw = (float *)p[2]; d.dbias = (float *)p[3];
      memcpy(&d.n, i, sizeof(int) * 15);
      return mr3d_conv3d_wgrad(&d, s);
    }
    case MR3D_OP_MAXPOOL_FWD:
      return mr3d_maxpool3d_forward((const float *)p[0], (float *)p[1], (uint32_t *)p[2], i[0], i[1], i[2], i[3], i[4], s);
    case MR3D_OP_MAXPOOL_BWD:
      return mr3d_maxpool3d_backward((const float *)p[0], (const uint32_t *)p[1], (const float *)p[2], (float *)p[3], i[0],
                                     i[1], i[2], i[3], i[4], s);
    case MR3D_OP_ZERO_STUFF2:
      return mr3d_zero_stuff2((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], s);
    case MR3D_OP_UPSAMPLE_ADJ:
      return mr3d_upsample_nearest_adjoint_add((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6],
                                               i[7], s);
    case MR3D_OP_SUBSAMPLE_FWD:
      return mr3d_subsample2_forward((const float *)p[0], (float *)p[1], i[0], i[1], i[2], i[3], i[4], s);
    case MR3D_OP_SUBSAMPLE_BWD:
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
    case MR3D_OP_STEM_FWD:
      return mr3d_stem_conv_forward((const float *)p[0], (const float *)p[1], (const float *)p[2], (float *)p[3], i[0], i[1], i[2],
                                    i[3], i[4], s);
    case MR3D_OP_STEM_WGRAD:
      return mr3d_stem_conv_wgrad((const float *)p[0], (const float *)p[1], (float *)p[2], (float *)p[3], i[0], i[1], i[2], i[3], s);
  }
  return MR3D_ERR_INVALID_ARG;
}

}  // namespace

extern "C" int mr3d_plan_run(const mr3d_op *ops, int num_ops, int *failed_index, mr3d_stream_t stream) {
  if (!ops || num_ops < 0) return MR3D_ERR_INVALID_ARG;
  for (int k = 0; k < num_ops; ++k) {
    const int st = run_one(ops[k], stream);
    if (st != MR3D_OK) {
      if (failed_index) *failed_index = k;
      return st;
    }
  }
  return MR3D_OK;
}
