// Two chained pointwise (1x1x1, stride 1) convolutions in ONE launch, for the small feature maps of layer3 / layer4.
//
// Between two bottleneck blocks of a ResNet3D stage (mmdet/models/backbones/resnet3d.py:198-233) the forward pass runs
//     out = relu(bn3(conv3(a2)) + identity)            C0 -> C1   (64 -> 256 / 128 -> 512)
//     a1' = relu(bn1'(conv1'(out)))                    C1 -> C2   (256 -> 64 / 512 -> 128)
// and the backward pass the mirror image (conv1'^T with the residual gradient added and the ReLU mask of `out`, then conv3^T
// with the mask of a2).  Neither convolution needs a halo, so a workgroup that owns 32 positions can run both: the C1-wide
// intermediate goes to global memory once (the next block's residual / the weight gradients need it) and stays in LDS for
// the second product instead of being read back by another launch.  On the 10x32x32 / 5x16x16 maps these launches are
// latency-bound (10-20 us for < 3 us of matrix work: global load -> LDS -> a short MFMA loop -> epilogue, with nothing to
// overlap), and each one saved is a launch, a cold start and a round trip of the intermediate.
//
// Workgroup = 32 positions, 4 waves.
//   stage 1  [32 x C0] x [C0 x C1]: v_mfma_f32_32x32x2_f32, A from the x tile in LDS (row stride C0 + 1), a wave owns
//            C1 / 128 column tiles of 32; B fragments straight from L2 (rows of C1 are contiguous), ALL of them (both
//            products) requested at the top of the kernel;
//            epilogue: + shift, + residual, ReLU / mask -> y1 (global) and the y1 tile in LDS (row stride C1 + 4).
//   stage 2  [32 x C1] x [C1 x C2]: v_mfma_f32_16x16x4_f32 (C2 / 64 column tiles of 16 per wave, both row tiles), A from the
//            y1 tile; epilogue: + shift, ReLU / mask -> y2.
#include "mr3d_common.h"
#include "conv3d.h"

namespace {

constexpr int kPairThreads = 256;

struct PwPairArgs {
  const float *x, *w1, *shift1, *add1, *mask1;
  float *y1;
  const float *w2, *shift2, *mask2;
  float *y2;
  long long P;
  int relu1, relu2;
};

template <int C0_, int C1_, int C2_>
struct PairCfg {
  static constexpr int C0 = C0_, C1 = C1_, C2 = C2_;
  static constexpr int MP = 32;
  static constexpr int XS = C0 + 1, YS = C1 + 4;
  static constexpr int T1 = C1 / 32 / 4;          // 32-wide stage-1 column tiles per wave
  static constexpr int T2 = C2 / 16 / 4;          // 16-wide stage-2 column tiles per wave
  static constexpr int KS1 = C0 / 2, KS2 = C1 / 4;
  static constexpr int LDS_BYTES = (MP * XS + 2 * MP * YS) * 4;      // x tile, y1 tile (first holds the residual), mask tile
  // ALL B fragments of a product are requested at once (64 registers per product at 64 / 256 / 64), the first product's at
  // the top of the kernel, the second's right behind the first product's MFMAs: fetching them a chunk ahead of their MFMAs
  // stalled every chunk for an L2 round trip (a chunk of 8 k-steps is 0.25-0.5 us of matrix work) and made the fused launch
  // slower than the two it replaces (39 vs 35 us)
  static_assert(C1 % 128 == 0 && C2 % 64 == 0 && C0 % 16 == 0 && KS1 * T1 <= 64 && KS2 * T2 <= 64, "shape");
};

template <class C>
__global__ __launch_bounds__(kPairThreads) void pw_pair_kernel(PwPairArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *xs = lds, *ys = lds + C::MP * C::XS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long p0 = (long long)blockIdx.x * C::MP;

  // ---- all weight fragments of this wave, both products: in flight from here on
  const int li = lane & 31, lk = lane >> 5;
  const int n1 = wave * C::T1 * 32;                                  // this wave's first stage-1 column
  const int li2 = lane & 15, lk2 = lane >> 4;
  const int n2 = wave * C::T2 * 16;
  float b1[C::KS1][C::T1], b2[C::KS2][C::T2];
  {
    const float *w1l = p.w1 + (size_t)lk * C::C1 + n1 + li;
#pragma unroll
    for (int kk = 0; kk < C::KS1; ++kk)
#pragma unroll
      for (int t = 0; t < C::T1; ++t) b1[kk][t] = w1l[(size_t)kk * 2 * C::C1 + t * 32];
  }
  // ---- x tile -> LDS
  {
    constexpr int Q = C::C0 / 4;
    for (int e = tid; e < C::MP * Q; e += kPairThreads) {
      const int row = e / Q, q = e - row * Q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p0 + row < p.P) v = *reinterpret_cast<const float4 *>(p.x + (size_t)(p0 + row) * C::C0 + q * 4);
      float *d = xs + row * C::XS + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  // ---- residual and mask tiles of the stage-1 outputs -> LDS with 16-byte coalesced loads (the accumulator layout of the
  // 32x32 MFMA would make them 32 + 32 four-byte loads per lane); the residual lands in the y1 tile, where the epilogue
  // replaces it value by value
  float *ms = ys + C::MP * C::YS;
  {
    constexpr int Q1 = C::C1 / 4;
    for (int e = tid; e < C::MP * Q1; e += kPairThreads) {
      const int row = e / Q1, q = e - row * Q1;
      const bool in = p0 + row < p.P;
      const size_t g = (size_t)(p0 + row) * C::C1 + q * 4;
      const float4 a = (in && p.add1) ? *reinterpret_cast<const float4 *>(p.add1 + g) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 m = (in && p.mask1) ? *reinterpret_cast<const float4 *>(p.mask1 + g) : make_float4(1.f, 1.f, 1.f, 1.f);
      *reinterpret_cast<float4 *>(ys + row * C::YS + q * 4) = a;
      *reinterpret_cast<float4 *>(ms + row * C::YS + q * 4) = m;
    }
  }
  // ---- stage 1
  f32x16 acc1[C::T1];
#pragma unroll
  for (int t = 0; t < C::T1; ++t)
    for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < C::KS1; ++kk) {
    const float a = xs[li * C::XS + kk * 2 + lk];
#pragma unroll
    for (int t = 0; t < C::T1; ++t) acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[kk][t], acc1[t], 0, 0, 0);
  }
  // the second product's fragments: requested as soon as the first product has issued (they take over its registers --
  // holding both sets at once left one workgroup per CU) and land during the epilogue and the barrier behind it
  {
    const float *w2l = p.w2 + (size_t)lk2 * C::C2 + n2 + li2;
#pragma unroll
    for (int kk = 0; kk < C::KS2; ++kk)
#pragma unroll
      for (int t = 0; t < C::T2; ++t) b2[kk][t] = w2l[(size_t)kk * 4 * C::C2 + t * 16];
  }
  // ---- epilogue 1: y1 = act(acc + shift + residual) into the LDS tile, then the tile -> global with 16-byte stores
#pragma unroll
  for (int t = 0; t < C::T1; ++t) {
    const int col = n1 + t * 32 + li;
    const float sh = p.shift1 ? p.shift1[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      float v = acc1[t][r] + sh + ys[row * C::YS + col];
      if (p.relu1) v = fmaxf(v, 0.f);
      v = ms[row * C::YS + col] > 0.f ? v : 0.f;
      ys[row * C::YS + col] = p0 + row < p.P ? v : 0.f;
    }
  }
  __syncthreads();
  {
    constexpr int Q1 = C::C1 / 4;
    for (int e = tid; e < C::MP * Q1; e += kPairThreads) {
      const int row = e / Q1, q = e - row * Q1;
      if (p0 + row < p.P)
        *reinterpret_cast<float4 *>(p.y1 + (size_t)(p0 + row) * C::C1 + q * 4) = *reinterpret_cast<const float4 *>(ys + row * C::YS + q * 4);
    }
  }
  // ---- stage 2
  f32x4 acc2[C::T2][2];
#pragma unroll
  for (int t = 0; t < C::T2; ++t)
    for (int m = 0; m < 2; ++m) acc2[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < C::KS2; ++kk) {
    const int k = kk * 4 + lk2;
    const float a0 = ys[li2 * C::YS + k], a1 = ys[(16 + li2) * C::YS + k];
#pragma unroll
    for (int t = 0; t < C::T2; ++t) {
      acc2[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b2[kk][t], acc2[t][0], 0, 0, 0);
      acc2[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b2[kk][t], acc2[t][1], 0, 0, 0);
    }
  }
  // ---- epilogue 2
#pragma unroll
  for (int t = 0; t < C::T2; ++t) {
    const int col = n2 + t * 16 + li2;
    const float sh = p.shift2 ? p.shift2[col] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long pos = p0 + m * 16 + lk2 * 4 + r;
        if (pos < p.P) {
          const size_t idx = (size_t)pos * C::C2 + col;
          float v = acc2[t][m][r] + sh;
          if (p.relu2) v = fmaxf(v, 0.f);
          if (p.mask2) v = p.mask2[idx] > 0.f ? v : 0.f;
          p.y2[idx] = v;
        }
      }
  }
}

template <class C>
int launch_pair(const PwPairArgs &a, hipStream_t stream) {
  static bool attr_done = false;
  if (!attr_done && C::LDS_BYTES > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(pw_pair_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MR3D_ERR_LAUNCH;
    }
    attr_done = true;
  }
  const long long wgs = (a.P + C::MP - 1) / C::MP;
  pw_pair_kernel<C><<<(unsigned)wgs, kPairThreads, C::LDS_BYTES, stream>>>(a);
  return mr3d_launch_status();
}

}  // namespace

namespace {
// the HBM-bound regime (>= 32k positions): the streaming kernel of pointwise.hip with its second product (16 -> 64 -> 16, layer1)
Conv3dFwdArgs stream_args(const mr3d_pw_pair_desc *d) {
  Conv3dFwdArgs a{};
  a.x = d->x; a.w = d->w1; a.y = d->y1; a.shift = d->shift1; a.add1 = d->add1; a.add2 = nullptr; a.up = nullptr; a.mask = d->mask1;
  a.N = 1; a.D = 1; a.H = 1; a.W = (int)d->positions; a.Cin = d->c0; a.Do = 1; a.Ho = 1; a.Wo = (int)d->positions; a.Cout = d->c1;
  a.pd = a.ph = a.pw = 0; a.Du = a.Hu = a.Wu = 0; a.relu = d->relu1; a.up_even = 0;
  a.ysd = a.ysh = a.ysw = 1; a.yD = 1; a.yH = 1; a.yW = (int)d->positions;
  a.w2 = d->w2; a.shift2 = d->shift2; a.mask2 = d->mask2; a.y2 = d->y2; a.relu2 = d->relu2;
  return a;
}
}  // namespace

extern "C" int mr3d_pointwise_pair_supported(int c0, int c1, int c2, long long positions) {
  if (positions <= 0 || positions > 0x7fffffffLL) return 0;
  if (positions >= 32768) {
    Conv3dFwdArgs a{};
    a.N = 1; a.D = a.H = a.Do = a.Ho = 1; a.W = a.Wo = (int)positions; a.Cin = c0; a.Cout = c1;
    return mr3d_pointwise_pair_stream_try(a, c2, nullptr, true) == MR3D_OK;
  }
  // the small maps: the workgroup-tile kernel below
  // layer3 only: at 128 / 512 / 128 (layer4: 40 workgroups of 32 positions) a wave cannot hold its weight fragments and
  // streaming 512 KB of weights per workgroup chunk by chunk measured 70 us against 22 us for the two separate launches
  return c0 == 64 && c1 == 256 && c2 == 64;
}

extern "C" int mr3d_pointwise_pair(const mr3d_pw_pair_desc *d, mr3d_stream_t stream_) {
  if (!d || !d->x || !d->w1 || !d->y1 || !d->w2 || !d->y2 || d->positions <= 0) return MR3D_ERR_INVALID_ARG;
  if (!mr3d_pointwise_pair_supported(d->c0, d->c1, d->c2, d->positions)) return MR3D_ERR_UNSUPPORTED;
  if (d->positions >= 32768) return mr3d_pointwise_pair_stream_try(stream_args(d), d->c2, mr3d_stream(stream_));
  PwPairArgs a;
  a.x = d->x; a.w1 = d->w1; a.shift1 = d->shift1; a.add1 = d->add1; a.mask1 = d->mask1; a.y1 = d->y1;
  a.w2 = d->w2; a.shift2 = d->shift2; a.mask2 = d->mask2; a.y2 = d->y2;
  a.P = d->positions; a.relu1 = d->relu1; a.relu2 = d->relu2;
  hipStream_t stream = mr3d_stream(stream_);
  return launch_pair<PairCfg<64, 256, 64>>(a, stream);
}
