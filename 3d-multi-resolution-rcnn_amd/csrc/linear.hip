// Fully connected layers of the bbox heads (mmdet/models/bbox_heads/convfc_bbox_head_3d.py: shared_fcs 9408 -> 1024 ->
// 1024, fc_cls, fc_reg on 512 sampled rois): forward, data gradient and weight gradient as one fp32 MFMA GEMM kernel
//     C[M][N] = sum_k A(m, k) * B(k, n)            v_mfma_f32_32x32x2_f32, 128 x 128 tile, 4 waves x (64 x 64)
// with the three operand layouts the layer needs (no transposed copies):
//     forward   y  = x  w^T : A = x  [M][K]  (k contiguous)   B = w  [N][K] (k contiguous)
//     dgrad     dx = dy w   : A = dy [M][K]  (k contiguous)   B = w  [K][N] (n contiguous)
//     wgrad     dw = dy^T x : A = dy [K][M]  (m contiguous)   B = x  [K][N] (n contiguous)
// A k-step of 16 is staged per iteration: registers -> LDS [k][m] / [k][n] (fragment reads are conflict-free ds_read_b32
// whatever the global layout), the next step's global loads in flight during the 32 MFMAs of the current one, two LDS
// buffers, one barrier per step.  M = 512 x N = 1024 is only 32 tiles, so the K loop is split over blockIdx.z and the
// partial products are stored to a caller-lent workspace and summed in a fixed order by a second tiny kernel together with
// bias + ReLU (bit-reproducible; without a workspace they meet in C with fp32 atomics); shapes with enough tiles (dgrad / wgrad of fc1: 296 / 592) run unsplit with the epilogue fused.
#include "mr3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 128, kBK = 16;
constexpr int kLdsRow = kBM + 4;          // floats per k row (16-byte aligned rows for float4 stores)
constexpr int kGemmThreads = 256;

struct GemmArgs {
  const float *a, *b;
  float *c;
  const float *bias;        // [N] or null (only read when the launch is unsplit)
  int M, N, K;
  long long lda, ldb, ldc;  // leading dimensions (elements) of the operands as stored
  int relu, atomic;
  int ksteps_total;         // ceil(K / kBK)
  float *part;              // split launches with a workspace: split z stores its products to part[z][M][N] (plain stores)
  int fast_a, fast_b;       // the operand is read with whole float4 buffer loads (see gload_fast)
};

// stage one 16 x 128 operand block into registers.  KMAJ: stored [rows][K] (k contiguous); else stored [K][rows]
// TR = tile extent along the operand's non-k axis (128, or 64 for the narrow M tile); kBK * TR / 4 float4 per block
template <bool KMAJ, int TR>
__device__ __forceinline__ void gload(const float *__restrict__ p, long long ld, int row0, int rows, int k0, int K, int tid,
                                      float4 (&r)[kBK * TR / 4 / kGemmThreads]) {
#pragma unroll
  for (int u = 0; u < kBK * TR / 4 / kGemmThreads; ++u) {
    const int f = tid + u * kGemmThreads;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KMAJ) {
      const int m = f / (kBK / 4), kq = f % (kBK / 4);        // kBK / 4 lanes read the contiguous k run of a row
      const int row = row0 + m, k = k0 + 4 * kq;
      if (row < rows) {
        const float *src = p + (size_t)row * ld + k;
        if (k + 3 < K && (ld & 3) == 0) v = *reinterpret_cast<const float4 *>(src);
        else {
          if (k < K) v.x = src[0];
          if (k + 1 < K) v.y = src[1];
          if (k + 2 < K) v.z = src[2];
          if (k + 3 < K) v.w = src[3];
        }
      }
    } else {
      const int kr = f / (TR / 4), q = f % (TR / 4);          // a k row is TR contiguous floats
      const int k = k0 + kr, col = row0 + 4 * q;
      if (k < K) {
        const float *src = p + (size_t)k * ld + col;
        if (col + 3 < rows && (ld & 3) == 0) v = *reinterpret_cast<const float4 *>(src);
        else {
          if (col < rows) v.x = src[0];
          if (col + 1 < rows) v.y = src[1];
          if (col + 2 < rows) v.z = src[2];
          if (col + 3 < rows) v.w = src[3];
        }
      }
    }
    r[u] = v;
  }
}

// The same block through a buffer descriptor, for operands whose float4 groups are inside the matrix or outside it as a whole
// (contiguous extent and leading dimension multiples of 4, below 2^31 bytes): a group outside gets an offset past the end of the
// buffer and the hardware returns zeros.  No bounds test around a load -- in gload every load sits in its own divergent branch
// and the wave walks them -- so a step's three or four loads leave back to back.
template <bool KMAJ, int TR>
__device__ __forceinline__ void gload_fast(__amdgpu_buffer_rsrc_t rs, long long ld, int row0, int rows, int k0, int K, int tid,
                                           float4 (&r)[kBK * TR / 4 / kGemmThreads]) {
#pragma unroll
  for (int u = 0; u < kBK * TR / 4 / kGemmThreads; ++u) {
    const int f = tid + u * kGemmThreads;
    unsigned off;
    if constexpr (KMAJ) {
      const int m = f / (kBK / 4), kq = f % (kBK / 4);
      const int row = row0 + m, k = k0 + 4 * kq;
      off = (row < rows && k < K) ? (unsigned)((long long)row * ld + k) * 4u : 0x80000000u;
    } else {
      const int kr = f / (TR / 4), q = f % (TR / 4);
      const int k = k0 + kr, col = row0 + 4 * q;
      off = (k < K && col < rows) ? (unsigned)((long long)k * ld + col) * 4u : 0x80000000u;
    }
    r[u] = buffer_load4(rs, off, 0u);
  }
}

template <bool KMAJ, int TR>
__device__ __forceinline__ void lstore(float *__restrict__ t, int tid, const float4 (&r)[kBK * TR / 4 / kGemmThreads]) {
#pragma unroll
  for (int u = 0; u < kBK * TR / 4 / kGemmThreads; ++u) {
    const int f = tid + u * kGemmThreads;
    if constexpr (KMAJ) {
      const int m = f / (kBK / 4), kq = f % (kBK / 4);
      float *d = t + (4 * kq) * kLdsRow + m;
      d[0] = r[u].x; d[kLdsRow] = r[u].y; d[2 * kLdsRow] = r[u].z; d[3 * kLdsRow] = r[u].w;
    } else {
      const int kr = f / (TR / 4), q = f % (TR / 4);
      *reinterpret_cast<float4 *>(t + kr * kLdsRow + 4 * q) = r[u];
    }
  }
}

// BM = 128: waves 2 x 2, each 64 x 64 (2 x 2 MFMA blocks).  BM = 64: waves 1 x 4, each 64 x 32 (2 x 1 blocks) -- the narrow tile for
// shapes whose 128-row tiling leaves a long tail (296 / 592 tiles on 256 CUs x 2 resident workgroups).
template <bool AK, bool BK, int BM>
__global__ __launch_bounds__(kGemmThreads, 2) void linear_gemm_kernel(GemmArgs p) {
  constexpr int NBW = BM == 128 ? 2 : 1;                   // 32-column blocks per wave
  __shared__ __attribute__((aligned(16))) float lds[2][2][kBK * kLdsRow];       // [buffer][A / B][k][m or n]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BM == 128 ? wave >> 1 : 0, wn = BM == 128 ? wave & 1 : wave;
  const int l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * kBN;
  // this workgroup's share of the k-steps
  const int s_begin = (int)(((long long)p.ksteps_total * blockIdx.z) / gridDim.z);
  const int s_end = (int)(((long long)p.ksteps_total * (blockIdx.z + 1)) / gridDim.z);

  f32x16 acc[2][NBW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Operand blocks of the next TWO k-steps wait in registers: one step of 16-32 MFMAs per wave (1-2 thousand cycles per SIMD with
  // the partner workgroup's) is shorter than a trip to L2 / HBM, and with one step of lookahead every step ended in that wait.
  float4 ra[2][kBK * BM / 4 / kGemmThreads], rb[2][kBK * kBN / 4 / kGemmThreads];
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.a), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.b), 0, 0x7fffffff, 0x00020000);
  auto load_step = [&](int s, float4 (&qa)[kBK * BM / 4 / kGemmThreads], float4 (&qb)[kBK * kBN / 4 / kGemmThreads]) {
    if (p.fast_a) gload_fast<AK, BM>(rsa, p.lda, m0, p.M, s * kBK, p.K, tid, qa);
    else gload<AK, BM>(p.a, p.lda, m0, p.M, s * kBK, p.K, tid, qa);
    if (p.fast_b) gload_fast<BK, kBN>(rsb, p.ldb, n0, p.N, s * kBK, p.K, tid, qb);
    else gload<BK, kBN>(p.b, p.ldb, n0, p.N, s * kBK, p.K, tid, qb);
  };
  if (s_begin < s_end) load_step(s_begin, ra[0], rb[0]);
  if (s_begin + 1 < s_end) load_step(s_begin + 1, ra[1], rb[1]);
  auto step = [&](int s, float4 (&qa)[kBK * BM / 4 / kGemmThreads], float4 (&qb)[kBK * kBN / 4 / kGemmThreads]) {
    const int buf = (s - s_begin) & 1;
    lstore<AK, BM>(lds[buf][0], tid, qa);
    lstore<BK, kBN>(lds[buf][1], tid, qb);
    __syncthreads();
    if (s + 2 < s_end) load_step(s + 2, qa, qb);
    const float *ta = lds[buf][0] + h * kLdsRow + wm * 64 + l31;
    const float *tb = lds[buf][1] + h * kLdsRow + wn * (32 * NBW) + l31;
    // all fragments of the step are requested before its first MFMA (one LDS latency per step, not one per k pair)
    float fa[kBK / 2][2], fb[kBK / 2][NBW];
#pragma unroll
    for (int ks = 0; ks < kBK / 2; ++ks) {
      fa[ks][0] = ta[(2 * ks) * kLdsRow]; fa[ks][1] = ta[(2 * ks) * kLdsRow + 32];
#pragma unroll
      for (int j = 0; j < NBW; ++j) fb[ks][j] = tb[(2 * ks) * kLdsRow + 32 * j];
    }
    __builtin_amdgcn_sched_barrier(0);            // (the scheduler otherwise sinks each pair of reads next to its MFMAs)
#pragma unroll
    for (int ks = 0; ks < kBK / 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
    // the buffer written two steps from now is this one: the barrier of the NEXT step orders its reads before that
  };
  for (int s = s_begin; s < s_end; s += 2) {
    step(s, ra[0], rb[0]);
    if (s + 1 < s_end) step(s + 1, ra[1], rb[1]);
  }

  // C/D layout of 32x32: column = lane & 31, rows (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int col = n0 + wn * (32 * NBW) + j * 32 + l31;
      if (col >= p.N) continue;
      const float bs = (!p.atomic && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row >= p.M) continue;
        float *dst = p.c + (size_t)row * p.ldc + col;
        if (p.part) {
          p.part[((size_t)blockIdx.z * p.M + row) * p.N + col] = acc[i][j][e];
        } else if (p.atomic) {
          atomicAdd(dst, acc[i][j][e]);
        } else {
          float v = acc[i][j][e] + bs;
          if (p.relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
    }
}

__global__ void linear_finish_kernel(float *__restrict__ c, const float *__restrict__ part, int parts, const float *__restrict__ bias,
                                     int M, int N, long long ldc, int relu) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int col = (int)(idx % N);
  float *d = c + (idx / N) * ldc + col;
  float v;
  if (part) {
    v = 0.f;
    for (int z = 0; z < parts; ++z) v += part[(size_t)z * M * N + idx];
  } else {
    v = *d;
  }
  v += bias ? bias[col] : 0.f;
  if (relu) v = fmaxf(v, 0.f);
  *d = v;
}

// 64-row tiles (a) where the 128-row tiling has between one and four rounds of workgroups -- the tail of the last round
// dominates there (fc1 dgrad: 296 tiles, wgrad: 592) -- and (b) under a K split: twice the tiles need half the split, i.e.
// half the partial-product traffic (fc1 forward: 64 tiles x 4 instead of 32 x 8)
bool narrow_m(int M, int N) {
  const int tiles = ((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  return M > 64 && tiles < 1024;
}

int split_of(int M, int N, int K) {
  const int bm = narrow_m(M, N) ? 64 : kBM;
  const int tiles = ((M + bm - 1) / bm) * ((N + kBN - 1) / kBN), ksteps = (K + kBK - 1) / kBK;
  int split = 1;
  if (tiles < 256) {      // split the K loop until one workgroup per CU is in flight, keeping at least 8 k-steps per workgroup
    split = (256 + tiles - 1) / tiles;
    if (split > ksteps / 8) split = ksteps / 8;
    if (split < 1) split = 1;
  }
  return split;
}

template <bool AK, bool BK>
int launch_gemm(GemmArgs p, float *workspace, long long workspace_bytes, hipStream_t stream) {
  const bool narrow = narrow_m(p.M, p.N);
  const int bm = narrow ? 64 : kBM;
  const int tm = (p.M + bm - 1) / bm, tn = (p.N + kBN - 1) / kBN;
  p.ksteps_total = (p.K + kBK - 1) / kBK;
  // whole float4 groups inside or outside the operand, 32-bit byte offsets (gload_fast)
  auto fast = [](bool kmaj, long long ld, int rows, int K) {
    const long long extent = kmaj ? (long long)rows * ld : (long long)K * ld;
    return (ld & 3) == 0 && ((kmaj ? K : rows) & 3) == 0 && extent * 4 < 0x7fffffffLL;
  };
  p.fast_a = fast(AK, p.lda, p.M, p.K) && (reinterpret_cast<uintptr_t>(p.a) & 15) == 0;
  p.fast_b = fast(BK, p.ldb, p.N, p.K) && (reinterpret_cast<uintptr_t>(p.b) & 15) == 0;
  const int split = split_of(p.M, p.N, p.K);
  p.atomic = split > 1;
  p.part = (split > 1 && workspace && workspace_bytes >= (long long)split * p.M * p.N * 4) ? workspace : nullptr;
  if (p.atomic && !p.part) {
    if (hipMemset2DAsync(p.c, (size_t)p.ldc * 4, 0, (size_t)p.N * 4, (size_t)p.M, stream) != hipSuccess) return MR3D_ERR_LAUNCH;
  }
  dim3 grid((unsigned)tn, (unsigned)tm, (unsigned)split);
  if (narrow) linear_gemm_kernel<AK, BK, 64><<<grid, kGemmThreads, 0, stream>>>(p);
  else linear_gemm_kernel<AK, BK, 128><<<grid, kGemmThreads, 0, stream>>>(p);
  if (p.part || (p.atomic && (p.bias || p.relu))) {
    const long long n = (long long)p.M * p.N;
    linear_finish_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(p.c, p.part, split, p.bias, p.M, p.N, p.ldc, p.relu);
  }
  return mr3d_launch_status();
}

bool bad(const void *a, const void *b, const void *c, int m, int n, int k) { return !a || !b || !c || m <= 0 || n <= 0 || k <= 0; }

}  // namespace

// bytes of workspace with which the three entry points avoid atomics for this layer (0: none of its launches is split)
extern "C" size_t mr3d_linear_workspace_bytes(int rows, int in_features, int out_features) {
  if (rows <= 0 || in_features <= 0 || out_features <= 0) return 0;
  size_t need = 0;
  auto take = [&](int M, int N, int K) {
    const int sp = split_of(M, N, K);
    if (sp > 1) { const size_t b = (size_t)sp * M * N * 4; if (b > need) need = b; }
  };
  take(rows, out_features, in_features);      // forward
  take(rows, in_features, out_features);      // dgrad
  take(out_features, in_features, rows);      // wgrad
  return need;
}

extern "C" int mr3d_linear_forward(const float *x, const float *weight, const float *bias, float *y, int rows, int in_features,
                                   int out_features, int relu, float *workspace, long long workspace_bytes, mr3d_stream_t s) {
  if (bad(x, weight, y, rows, out_features, in_features)) return MR3D_ERR_INVALID_ARG;
  GemmArgs p{};
  p.a = x; p.lda = in_features; p.b = weight; p.ldb = in_features; p.c = y; p.ldc = out_features;
  p.bias = bias; p.M = rows; p.N = out_features; p.K = in_features; p.relu = relu;
  return launch_gemm<true, true>(p, workspace, workspace_bytes, mr3d_stream(s));
}

extern "C" int mr3d_linear_dgrad(const float *dy, const float *weight, float *dx, int rows, int in_features, int out_features,
                                 float *workspace, long long workspace_bytes, mr3d_stream_t s) {
  if (bad(dy, weight, dx, rows, in_features, out_features)) return MR3D_ERR_INVALID_ARG;
  GemmArgs p{};
  p.a = dy; p.lda = out_features; p.b = weight; p.ldb = in_features; p.c = dx; p.ldc = in_features;
  p.M = rows; p.N = in_features; p.K = out_features;
  return launch_gemm<true, false>(p, workspace, workspace_bytes, mr3d_stream(s));
}

extern "C" int mr3d_linear_wgrad(const float *dy, const float *x, float *dw, int rows, int in_features, int out_features,
                                 float *workspace, long long workspace_bytes, mr3d_stream_t s) {
  if (bad(dy, x, dw, out_features, in_features, rows)) return MR3D_ERR_INVALID_ARG;
  GemmArgs p{};
  p.a = dy; p.lda = out_features; p.b = x; p.ldb = in_features; p.c = dw; p.ldc = in_features;
  p.M = out_features; p.N = in_features; p.K = rows;
  return launch_gemm<false, false>(p, workspace, workspace_bytes, mr3d_stream(s));
}
