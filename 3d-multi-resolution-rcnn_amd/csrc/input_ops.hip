// Input preparation on the device.  The reference does this per SLICE on the host
// (mmdet/datasets/coco_3d_2scales.py:245-263,267-289: grey -> np.repeat to 3 channels -> mmcv.imnormalize ->
// impad_to_multiple -> stack -> transpose to [3, D, H, W]; transforms.py:33-51) and builds the second pathway with
// skimage.transform.resize of the whole volume (coco_3d_2scales.py:219), then ships 3 fp32 channels over PCIe.
// Here the grey volume goes up once (1 channel, u8 or f32, the .npy's own [H, W, D] axis order or [D, H, W]) and
// one kernel writes the normalised, zero-padded planar [3, D', Hp, Wp] tensor the stem kernel reads -- optionally
// resampled exactly as skimage.transform.resize does with its defaults when up-scaling (order 1, half-pixel centres,
// mode 'reflect' = ndimage 'mirror', anti-aliasing sigma 0): pinned against scipy.ndimage.map_coordinates, the routine
// skimage calls (tests/test_input_prepare_cpu.py); scikit-image itself is not installed.
// HBM-bound: bytes = |grey| read + 12 * D'*Hp*Wp written.
#include "mr3d_common.h"

namespace {

constexpr int kT = 256;

struct PrepArgs {
  const void *src;
  float *dst;
  int src_u8, src_hwd;        // dtype / axis order of the source
  int D, H, W;                // source extents
  int Do, Ho, Wo;             // extents after the optional resample
  int Hp, Wp;                 // padded output extents (rows / columns beyond Ho / Wo are zero)
  float fd, fh, fw;           // source step per output step (1 / scale factor)
  int resample;
  float mean[3], inv_unused[3], stdv[3];
};

__device__ __forceinline__ float load_src(const PrepArgs &p, int d, int h, int w) {
  const size_t i = p.src_hwd ? ((size_t)h * p.W + w) * p.D + d : ((size_t)d * p.H + h) * p.W + w;
  return p.src_u8 ? (float)((const uint8_t *)p.src)[i] : ((const float *)p.src)[i];
}

// skimage.transform.resize (scikit-image 0.18, n-dimensional branch: transform/_warps.py) samples the volume at
// c = (o + 0.5) * in/out - 0.5 with scipy.ndimage.map_coordinates(order=1, mode='mirror'): its default mode 'reflect' is
// ndimage's 'mirror' (_shared/utils.py mode table) -- the signal is mirrored about the CENTRE of the edge voxel, so a sample
// up to half a voxel outside mixes the edge voxel with its inner neighbour (index -1 -> 1, index n -> n - 2).
__device__ __forceinline__ int mirror_index(int i, int n) {
  if (n == 1) return 0;
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

__device__ __forceinline__ void axis_taps(int o, float step, int n, int &i0, int &i1, float &t) {
  const float c = ((float)o + 0.5f) * step - 0.5f;
  const int f = (int)floorf(c);
  t = c - (float)f;
  i0 = mirror_index(f, n);
  i1 = mirror_index(f + 1, n);
}

// one thread per output voxel (w fastest: coalesced planar stores); the [H,W,D] source is read through a 32x32 LDS
// transpose of the (w, d) plane when no resampling is requested, gathered through L2 otherwise
__global__ __launch_bounds__(kT) void prepare_kernel(PrepArgs p) {
  const long long plane = (long long)p.Hp * p.Wp;
  const long long total = (long long)p.Do * plane;
  for (long long e = (long long)blockIdx.x * kT + threadIdx.x; e < total; e += (long long)gridDim.x * kT) {
    const int w = (int)(e % p.Wp), h = (int)((e / p.Wp) % p.Hp), d = (int)(e / plane);
    float v[3] = {0.f, 0.f, 0.f};
    if (h < p.Ho && w < p.Wo) {
      float g;
      if (!p.resample) {
        g = load_src(p, d, h, w);
      } else {
        int d0, d1, h0, h1, w0, w1;
        float td, th, tw;
        axis_taps(d, p.fd, p.D, d0, d1, td);
        axis_taps(h, p.fh, p.H, h0, h1, th);
        axis_taps(w, p.fw, p.W, w0, w1, tw);
        const float c00 = load_src(p, d0, h0, w0) * (1.f - tw) + load_src(p, d0, h0, w1) * tw;
        const float c01 = load_src(p, d0, h1, w0) * (1.f - tw) + load_src(p, d0, h1, w1) * tw;
        const float c10 = load_src(p, d1, h0, w0) * (1.f - tw) + load_src(p, d1, h0, w1) * tw;
        const float c11 = load_src(p, d1, h1, w0) * (1.f - tw) + load_src(p, d1, h1, w1) * tw;
        g = (c00 * (1.f - th) + c01 * th) * (1.f - td) + (c10 * (1.f - th) + c11 * th) * td;
      }
      for (int c = 0; c < 3; ++c) v[c] = (g - p.mean[c]) / p.stdv[c];          // mmcv.imnormalize: (img - mean) / std
    }
    for (int c = 0; c < 3; ++c) p.dst[(size_t)c * total + e] = v[c];
  }
}

// [H,W,D] source, no resample: transpose (w,d) planes through LDS so that both sides move in contiguous runs
__global__ __launch_bounds__(kT) void prepare_hwd_kernel(PrepArgs p) {
  __shared__ float tile[32][33];
  const int h = blockIdx.z;
  const int w0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int w = w0 + j, d = d0 + tx;
    float g = 0.f;
    if (h < p.Ho && w < p.Wo && d < p.D) g = load_src(p, d, h, w);
    tile[j][tx] = g;
  }
  __syncthreads();
  const long long plane = (long long)p.Hp * p.Wp, total = (long long)p.Do * plane;
  for (int j = ty; j < 32; j += 8) {
    const int d = d0 + j, w = w0 + tx;
    if (d < p.Do && w < p.Wp) {
      const bool in = h < p.Ho && w < p.Wo;
      const float g = tile[tx][j];
      const long long e = (long long)d * plane + (long long)h * p.Wp + w;
      for (int c = 0; c < 3; ++c) p.dst[(size_t)c * total + e] = in ? (g - p.mean[c]) / p.stdv[c] : 0.f;
    }
  }
}

}  // namespace

extern "C" int mr3d_volume_prepare(const void *grey, int src_is_u8, int src_is_hwd, int d, int h, int w, int out_d, int out_h,
                                   int out_w, int pad_h, int pad_w, const float *mean_host, const float *std_host, float *out,
                                   mr3d_stream_t s) {
  if (!grey || !out || !mean_host || !std_host || d <= 0 || h <= 0 || w <= 0 || out_d <= 0 || out_h <= 0 || out_w <= 0 ||
      pad_h < out_h || pad_w < out_w)
    return MR3D_ERR_INVALID_ARG;
  PrepArgs p;
  p.src = grey; p.dst = out; p.src_u8 = src_is_u8; p.src_hwd = src_is_hwd;
  p.D = d; p.H = h; p.W = w; p.Do = out_d; p.Ho = out_h; p.Wo = out_w; p.Hp = pad_h; p.Wp = pad_w;
  p.resample = !(out_d == d && out_h == h && out_w == w);
  p.fd = (float)d / (float)out_d; p.fh = (float)h / (float)out_h; p.fw = (float)w / (float)out_w;
  for (int c = 0; c < 3; ++c) {
    if (!(std_host[c] != 0.f)) return MR3D_ERR_INVALID_ARG;
    p.mean[c] = mean_host[c]; p.stdv[c] = std_host[c]; p.inv_unused[c] = 0.f;
  }
  hipStream_t st = mr3d_stream(s);
  if (!p.resample && src_is_hwd) {
    dim3 grid((unsigned)((pad_w + 31) / 32), (unsigned)((out_d + 31) / 32), (unsigned)pad_h);
    prepare_hwd_kernel<<<grid, kT, 0, st>>>(p);
  } else {
    const long long total = (long long)out_d * pad_h * pad_w;
    long long blocks = (total + kT - 1) / kT;
    if (blocks > 65536) blocks = 65536;
    prepare_kernel<<<(unsigned)blocks, kT, 0, st>>>(p);
  }
  return mr3d_launch_status();
}
