// MaxPool 3x3/s2 (+argmax), nearest-x2 upsample + skip concat + replicate pad, and a per-channel
// column sum (bias gradients).  All HBM-bound, NHWC, 4 channels (8/16 bytes) per lane.
// Replaces:
//   nn.MaxPool2d(3, 2, 1)                       resnet.py:122,206
//   F.interpolate(nearest, x2) + torch.cat       depth_encoder.py:126-133
//   the implicit replicate padding of upconv(i,1) depth_encoder.py:59 (materialised once here)
//   conv bias gradients (sum over N,H,W of dY)
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

// second tensor set of a two-problem max-pool launch: images n >= n0 of the flat index space belong to it
struct FsPoolPair { const void* in1; void* out1; uint8_t* idx1; const void* add1; int n0; };

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C,
                                                          int Ho, int Wo, FsPoolPair pr) {
  constexpr int VN = VecN<T>::N;           // one 16-byte lane per thread: 8 bf16 / 4 f32 channels
  const int CG = C / VN;
  const long total = (long)N * Ho * Wo * CG;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int cg = (int)(i % CG); long m = i / CG;
    int wo = (int)(m % Wo); long q = m / Wo; int ho = (int)(q % Ho); long n = q / Ho;
    const T* xs = x; T* ys = y; uint8_t* is = idx;
    if (n >= pr.n0) {        // images of the second tensor set (two problems in one launch)
      n -= pr.n0; m -= (long)pr.n0 * Ho * Wo;
      xs = reinterpret_cast<const T*>(pr.in1); ys = reinterpret_cast<T*>(pr.out1); is = pr.idx1;
    }
    float best[VN];
    int bi[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    bool first = true;
    for (int r = 0; r < 3; ++r) {
      int h = ho * 2 - 1 + r;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int s = 0; s < 3; ++s) {
        int w = wo * 2 - 1 + s;
        if ((unsigned)w >= (unsigned)W) continue;
        float v[VN];
        loadv<T>(xs + ((n * H + h) * W + w) * C + cg * VN, v);
#pragma unroll
        for (int j = 0; j < VN; ++j)
          if (first || v[j] > best[j]) { best[j] = v[j]; bi[j] = r * 3 + s; }  // first max wins (ATen)
        first = false;
      }
    }
    storev<T>(ys + m * C + cg * VN, best);
#pragma unroll
    for (int k = 0; k < VN / 4; ++k) {
      uint32_t packed = (uint32_t)bi[4 * k] | ((uint32_t)bi[4 * k + 1] << 8) | ((uint32_t)bi[4 * k + 2] << 16) |
                        ((uint32_t)bi[4 * k + 3] << 24);
      reinterpret_cast<uint32_t*>(is + m * C + cg * VN)[k] = packed;
    }
  }
}

// gather form: dx(h,w) = sum over the <=4 windows containing (h,w) whose argmax is (h,w), + addend
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          const T* __restrict__ addend, T* __restrict__ dx, int N,
                                                          int H, int W, int C, int Ho, int Wo, FsPoolPair pr) {
  constexpr int VN = VecN<T>::N;           // one 16-byte lane per thread: 8 bf16 / 4 f32 channels
  const int CG = C / VN;
  const long total = (long)N * H * W * CG;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int cg = (int)(i % CG); long m = i / CG;
    int w = (int)(m % W); long q = m / W; int h = (int)(q % H); long n = q / H;
    const T* dys = dy; T* dxs = dx; const uint8_t* is = idx; const T* ads = addend;
    if (n >= pr.n0) {
      n -= pr.n0; m -= (long)pr.n0 * H * W;
      dys = reinterpret_cast<const T*>(pr.in1); dxs = reinterpret_cast<T*>(pr.out1); is = pr.idx1;
      ads = reinterpret_cast<const T*>(pr.add1);
    }
    float acc[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) acc[j] = 0.f;
    if (ads) loadv<T>(ads + m * C + cg * VN, acc);
    for (int ho = (h - 1 + 1) / 2; ho <= (h + 1) / 2; ++ho) {   // windows with |h - 2ho| <= 1
      if (ho < 0 || ho >= Ho) continue;
      int r = h - (ho * 2 - 1);
      if (r < 0 || r > 2) continue;
      for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
        if (wo < 0 || wo >= Wo) continue;
        int s = w - (wo * 2 - 1);
        if (s < 0 || s > 2) continue;
        long mo = (n * Ho + ho) * Wo + wo;
        uint32_t packed[VN / 4];
#pragma unroll
        for (int k = 0; k < VN / 4; ++k) packed[k] = reinterpret_cast<const uint32_t*>(is + mo * C + cg * VN)[k];
        float g[VN];
        loadv<T>(dys + mo * C + cg * VN, g);
        int code = r * 3 + s;
#pragma unroll
        for (int j = 0; j < VN; ++j)
          if ((int)((packed[j >> 2] >> (8 * (j & 3))) & 0xff) == code) acc[j] += g[j];
      }
    }
    storev<T>(dxs + m * C + cg * VN, acc);
  }
}

// V consecutive channels per thread: 4 (8 bytes of bf16, any C % 4 == 0) or 8 (a full 16-byte bf16 lane)
template <typename T, int V> __device__ inline void ldc(const T* p, float* v) {
  if constexpr (V == 8) loadv<T>(p, v); else load4<T>(p, v);
}
template <typename T, int V> __device__ inline void stc(T* p, const float* v) {
  if constexpr (V == 8) storev<T>(p, v); else store4<T>(p, v);
}

// out[n, hp, wp, :] for the padded (2h+2)x(2w+2) grid = cat(up2(a), b) at the replicate-clamped pixel
template <typename T, int V>
__global__ __launch_bounds__(256) void upcat_pad_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                            T* __restrict__ out, int N, int h, int w, int Ca,
                                                            int Cb) {
  const int H = 2 * h, W = 2 * w, Hp = H + 2, Wp = W + 2, C = Ca + Cb, CG = C / V;
  const long total = (long)N * Hp * Wp * CG;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int cg = (int)(i % CG); long m = i / CG;
    int wp = (int)(m % Wp); long q = m / Wp; int hp = (int)(q % Hp); long n = q / Hp;
    int hi = min(max(hp - 1, 0), H - 1), wi = min(max(wp - 1, 0), W - 1);
    int c = cg * V;
    float v[V];
    if (c < Ca) ldc<T, V>(a + ((n * h + (hi >> 1)) * w + (wi >> 1)) * Ca + c, v);
    else ldc<T, V>(b + ((n * H + hi) * W + wi) * Cb + (c - Ca), v);
    stc<T, V>(out + m * C + c, v);
  }
}

// folded value of the padded gradient at interior pixel (hi, wi)
template <typename T, int V>
__device__ inline void fold_load(const T* dpad, long n, int hi, int wi, int H, int W, int C, int c, float* g) {
  const int Hp = H + 2, Wp = W + 2;
  int hs[2] = {hi + 1, 0}, ws[2] = {wi + 1, 0};
  int nh = 1, nw = 1;
  if (hi == 0) { hs[1] = 0; nh = 2; } else if (hi == H - 1) { hs[1] = H + 1; nh = 2; }
  if (wi == 0) { ws[1] = 0; nw = 2; } else if (wi == W - 1) { ws[1] = W + 1; nw = 2; }
#pragma unroll
  for (int j = 0; j < V; ++j) g[j] = 0.f;
  for (int x = 0; x < nh; ++x)
    for (int y = 0; y < nw; ++y) {
      float t[V];
      ldc<T, V>(dpad + ((n * Hp + hs[x]) * Wp + ws[y]) * C + c, t);
#pragma unroll
      for (int j = 0; j < V; ++j) g[j] += t[j];
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void upcat_pad_bwd_kernel(const T* __restrict__ dpad, T* __restrict__ da,
                                                            T* __restrict__ db, int N, int h, int w, int Ca,
                                                            int Cb) {
  const int H = 2 * h, W = 2 * w, C = Ca + Cb;
  const int CGa = Ca / V, CGb = Cb / V;
  const long ta = (long)N * h * w * CGa, tb = (long)N * H * W * CGb;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ta + tb; i += (long)gridDim.x * 256) {
    if (i < ta) {
      int cg = (int)(i % CGa); long m = i / CGa;
      int x = (int)(m % w); long q = m / w; int y = (int)(q % h); long n = q / h;
      float acc[V];
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = 0.f;
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          float g[V];
          fold_load<T, V>(dpad, n, 2 * y + dy, 2 * x + dx, H, W, C, cg * V, g);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += g[j];
        }
      stc<T, V>(da + m * Ca + cg * V, acc);
    } else {
      long k = i - ta;
      int cg = (int)(k % CGb); long m = k / CGb;
      int x = (int)(m % W); long q = m / W; int y = (int)(q % H); long n = q / H;
      float g[V];
      fold_load<T, V>(dpad, n, y, x, H, W, C, Ca + cg * V, g);
      stc<T, V>(db + m * Cb + cg * V, g);
    }
  }
}

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16>(float v) { return bf16_bits_to_f(f_to_bf16_bits(v)); }

// upcat_pad_bwd with the first pass of the BatchNorm backward it feeds (depth_encoder.py:126-133 backwards: the up-sampled
// half of the concatenation is y = relu(bn(c)) of the level's first ConvBnReLU): da is stored ReLU-masked (mask = y > 0) and
// the sums (sum g, sum g * xhat) of the masked, storage-rounded gradient go to the f64 slots bn_bwd_apply reads — what
// fs_bn_bwd_reduce computes from da in a launch of its own.  The first nA blocks: the da part, a thread keeps one channel
// group (256 % CGa == 0 and the stride nA * 256 is a multiple of CGa) — at most 512 of them: every block ends in 2 * Ca
// same-slot f64 atomics; the other blocks: the skip half, a plain copy.
template <typename T, int V>
__global__ __launch_bounds__(256) void upcat_pad_bwd_bn_kernel(const T* __restrict__ dpad, T* __restrict__ da,
                                                               T* __restrict__ db, int N, int h, int w, int Ca, int Cb,
                                                               const T* __restrict__ yact, const T* __restrict__ xraw,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, double* __restrict__ sums,
                                                               const int nA) {
  const int H = 2 * h, W = 2 * w, C = Ca + Cb;
  const int CGa = Ca / V, CGb = Cb / V;
  if ((int)blockIdx.x >= nA) {
    const long tb = (long)N * H * W * CGb;
    const int bx = (int)blockIdx.x - nA, nB = (int)gridDim.x - nA;
    for (long k = (long)bx * 256 + threadIdx.x; k < tb; k += (long)nB * 256) {
      int cg = (int)(k % CGb); long m = k / CGb;
      int x = (int)(m % W); long q = m / W; int y = (int)(q % H); long n = q / H;
      float g[V];
      fold_load<T, V>(dpad, n, y, x, H, W, C, Ca + cg * V, g);
      stc<T, V>(db + m * Cb + cg * V, g);
    }
    return;
  }
  __shared__ float red[2][V][256];
  const int cg = threadIdx.x % CGa, c = cg * V;
  float mu[V], is[V], s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; s1[j] = 0.f; s2[j] = 0.f; }
  const long ta = (long)N * h * w * CGa;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ta; i += (long)nA * 256) {
    long m = i / CGa;
    int x = (int)(m % w); long q = m / w; int y = (int)(q % h); long n = q / h;
    float acc[V], yy[V], xr[V];
    ldc<T, V>(yact + m * Ca + c, yy);
    ldc<T, V>(xraw + m * Ca + c, xr);
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) {
        float g[V];
        fold_load<T, V>(dpad, n, 2 * y + dy, 2 * x + dx, H, W, C, c, g);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += g[j];
      }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float g = yy[j] > 0.f ? round_to<T>(acc[j]) : 0.f;      // what the apply pass will read back from da
      acc[j] = g;
      s1[j] += g; s2[j] += g * (xr[j] - mu[j]) * is[j];
    }
    stc<T, V>(da + m * Ca + c, acc);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { red[0][j][threadIdx.x] = s1[j]; red[1][j][threadIdx.x] = s2[j]; }
  __syncthreads();
  // thread t < CGa * 2 * V finishes (channel group, kind, j) = one channel's one sum: PL = 256 / CGa partials
  const int PL = 256 / CGa;
  for (int t = threadIdx.x; t < CGa * 2 * V; t += 256) {
    const int g2 = t % CGa, kj = t / CGa, kind = kj / V, j = kj % V;
    float a = 0.f;
    for (int k = 0; k < PL; ++k) a += red[kind][j][k * CGa + g2];
    double* sl = sums + (long)(blockIdx.x % FS_STAT_SLOTS) * 2 * Ca;
    atomicAdd(sl + kind * Ca + g2 * V + j, (double)a);
  }
}

// out[c] += sum_m x[m][c]   (x dense [M][C], fp32 accumulate, one atomic per channel per block).
// 16-byte lanes when C allows (VL = 8 bf16 / 4 f32 channels per lane, else 4), four rows in flight per thread.
template <typename T, int VL>
__device__ __forceinline__ void channel_sum_body(const T* __restrict__ x, float* __restrict__ out, long M, int C, int Creal,
                                                 int CGB, int bx, int by, int gx) {
  __shared__ float red[VL][256];
  const int CG = C / VL;
  const int PL = 256 / CGB;
  const int cgl = threadIdx.x % CGB, pl = threadIdx.x / CGB;
  const int cg = by * CGB + cgl;
  float s[VL];
#pragma unroll
  for (int j = 0; j < VL; ++j) s[j] = 0.f;
  auto ld = [&](long m, float* v) {
    if constexpr (VL == 4) load4<T>(x + m * C + cg * 4, v);
    else loadv<T>(x + m * C + cg * VL, v);
  };
  if (cg < CG) {
    const long stride = (long)gx * PL;
    long m = (long)bx * PL + pl;
    for (; m + 3 * stride < M; m += 4 * stride) {
      float v0[VL], v1[VL], v2[VL], v3[VL];
      ld(m, v0); ld(m + stride, v1); ld(m + 2 * stride, v2); ld(m + 3 * stride, v3);
#pragma unroll
      for (int j = 0; j < VL; ++j) s[j] += (v0[j] + v1[j]) + (v2[j] + v3[j]);
    }
    for (; m < M; m += stride) {
      float v0[VL];
      ld(m, v0);
#pragma unroll
      for (int j = 0; j < VL; ++j) s[j] += v0[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VL; ++j) red[j][threadIdx.x] = s[j];
  __syncthreads();
  // thread (pl = j) of each channel group finishes channel j of the lane
  if (cg < CG) {
    for (int j = pl; j < VL; j += PL) {
      float a = 0.f;
      for (int k = 0; k < PL; ++k) a += red[j][k * CGB + cgl];
      if (cg * VL + j < Creal) atomicAdd(out + cg * VL + j, a);
    }
  }
}

template <typename T, int VL>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, float* __restrict__ out, long M,
                                                          int C, int Creal, int CGB) {
  channel_sum_body<T, VL>(x, out, M, C, Creal, CGB, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// up to FS_CSUM_MAX column sums in ONE launch (the bias gradients of a hand-off batch of weight gradients: each was a
// launch of its own behind its layer's weight gradient): block b belongs to the descriptor whose [start, start + gx * gy)
// range holds it
constexpr int FS_CSUM_MAX = 16;
struct CsumBatch { const void* x[FS_CSUM_MAX]; float* out[FS_CSUM_MAX]; long M[FS_CSUM_MAX]; int C[FS_CSUM_MAX], Creal[FS_CSUM_MAX],
                   CGB[FS_CSUM_MAX], gx[FS_CSUM_MAX], start[FS_CSUM_MAX + 1]; int n; };
template <typename T, int VL>
__global__ __launch_bounds__(256) void channel_sum_multi_kernel(const CsumBatch c) {
  int k = 0;
  while (k + 1 < c.n && (int)blockIdx.x >= c.start[k + 1]) ++k;
  const int lb = (int)blockIdx.x - c.start[k];
  const int gx = c.gx[k];
  channel_sum_body<T, VL>(reinterpret_cast<const T*>(c.x[k]), c.out[k], c.M[k], c.C[k], c.Creal[k], c.CGB[k], lb % gx, lb / gx, gx);
}

int grid_for(long items) {
  long b = (items + 255) / 256;
  return (int)std::max<long>(1, std::min<long>(b, 8192));
}

}  // namespace

#define DISPATCH(KERNEL, GRID, ...)                                                                   \
  if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(KERNEL<bf16>, GRID, dim3(256), 0, st, __VA_ARGS__);  \
  else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(KERNEL<float>, GRID, dim3(256), 0, st, __VA_ARGS__); \
  else return FS_EINVAL;

// x1 != NULL: a second tensor of N1 images with the same H, W, C in the same launch (the two encoders' stems)
extern "C" int fs_maxpool_fwd2(const void* x, void* y, uint8_t* idx, int N, const void* x1, void* y1, uint8_t* idx1, int N1,
                               int H, int W, int C, int dtype, void* stream) {
  const int vn = dtype == FS_DTYPE_BF16 ? 8 : 4;
  if (!x || !y || !idx || C % vn != 0 || N <= 0) return FS_EINVAL;
  if (x1 && (!y1 || !idx1 || N1 <= 0)) return FS_EINVAL;
  if (!x1) N1 = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const FsPoolPair pr{x1, y1, idx1, nullptr, x1 ? N : 0x7fffffff};
  dim3 grid(grid_for((long)(N + N1) * Ho * Wo * (C / vn)));
  if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (bf16*)y, idx, N + N1, H, W, C, Ho, Wo, pr);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, idx, N + N1, H, W, C, Ho, Wo, pr);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_maxpool_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int dtype,
                              void* stream) {
  return fs_maxpool_fwd2(x, y, idx, N, nullptr, nullptr, nullptr, 0, H, W, C, dtype, stream);
}

extern "C" int fs_maxpool_bwd2(const void* dy, const uint8_t* idx, const void* addend, void* dx, int N, const void* dy1,
                               const uint8_t* idx1, const void* addend1, void* dx1, int N1, int H, int W, int C, int dtype,
                               void* stream) {
  const int vn = dtype == FS_DTYPE_BF16 ? 8 : 4;
  if (!dy || !dx || !idx || C % vn != 0 || N <= 0) return FS_EINVAL;
  if (dy1 && (!dx1 || !idx1 || N1 <= 0)) return FS_EINVAL;
  if (!dy1) N1 = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const FsPoolPair pr{dy1, dx1, const_cast<uint8_t*>(idx1), addend1, dy1 ? N : 0x7fffffff};
  dim3 grid(grid_for((long)(N + N1) * H * W * (C / vn)));
  if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)dy, idx, (const bf16*)addend, (bf16*)dx, N + N1, H, W, C, Ho, Wo, pr);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, idx, (const float*)addend, (float*)dx, N + N1, H, W, C, Ho, Wo, pr);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_maxpool_bwd(const void* dy, const uint8_t* idx, const void* addend, void* dx, int N, int H, int W,
                              int C, int dtype, void* stream) {
  return fs_maxpool_bwd2(dy, idx, addend, dx, N, nullptr, nullptr, nullptr, nullptr, 0, H, W, C, dtype, stream);
}

extern "C" int fs_upcat_pad_fwd(const void* a, const void* b, void* out, int N, int h, int w, int Ca, int Cb,
                                int dtype, void* stream) {
  if (!a || !out || (Cb > 0 && !b) || Ca % 4 != 0 || Cb % 4 != 0) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = dtype == FS_DTYPE_BF16 && Ca % 8 == 0 && Cb % 8 == 0;     // 16-byte lanes
  dim3 grid(grid_for((long)N * (2 * h + 2) * (2 * w + 2) * ((Ca + Cb) / (wide ? 8 : 4))));
  if (wide)
    hipLaunchKernelGGL((upcat_pad_fwd_kernel<bf16, 8>), grid, dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)out, N, h, w, Ca, Cb);
  else if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL((upcat_pad_fwd_kernel<bf16, 4>), grid, dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)out, N, h, w, Ca, Cb);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL((upcat_pad_fwd_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, N, h, w, Ca, Cb);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_upcat_pad_bwd(const void* dpad, void* da, void* db, int N, int h, int w, int Ca, int Cb, int dtype,
                                void* stream) {
  if (!dpad || !da || (Cb > 0 && !db) || Ca % 4 != 0 || Cb % 4 != 0) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = dtype == FS_DTYPE_BF16 && Ca % 8 == 0 && Cb % 8 == 0;
  const int V = wide ? 8 : 4;
  dim3 grid(grid_for((long)N * h * w * (Ca / V) + (long)N * 4 * h * w * (Cb / V)));
  if (wide)
    hipLaunchKernelGGL((upcat_pad_bwd_kernel<bf16, 8>), grid, dim3(256), 0, st, (const bf16*)dpad, (bf16*)da, (bf16*)db, N, h, w, Ca, Cb);
  else if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL((upcat_pad_bwd_kernel<bf16, 4>), grid, dim3(256), 0, st, (const bf16*)dpad, (bf16*)da, (bf16*)db, N, h, w, Ca, Cb);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL((upcat_pad_bwd_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)dpad, (float*)da, (float*)db, N, h, w, Ca, Cb);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_upcat_pad_bwd_bn(const void* dpad, void* da, void* db, int N, int h, int w, int Ca, int Cb, const void* y,
                                   const void* x, const float* mean, const float* invstd, double* sums, int dtype,
                                   void* stream) {
  if (!dpad || !da || (Cb > 0 && !db) || !y || !x || !mean || !invstd || !sums || Ca % 4 != 0 || Cb % 4 != 0 || Ca <= 0)
    return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = dtype == FS_DTYPE_BF16 && Ca % 8 == 0 && Cb % 8 == 0;
  const int V = wide ? 8 : 4;
  const int CGa = Ca / V;
  if (CGa > 256 || 256 % CGa != 0) return FS_EINVAL;        // a thread keeps its channel group: powers of two up to 256
  const long ta = (long)N * h * w * CGa, tb = (long)N * 4 * h * w * (Cb / V);
  const int nA = (int)std::max<long>(1, std::min<long>((ta + 511) / 512, 512));
  const int nB = Cb > 0 ? grid_for(tb) : 0;
  dim3 grid(nA + nB);
  if (wide)
    hipLaunchKernelGGL((upcat_pad_bwd_bn_kernel<bf16, 8>), grid, dim3(256), 0, st, (const bf16*)dpad, (bf16*)da, (bf16*)db, N, h, w, Ca, Cb,
                       (const bf16*)y, (const bf16*)x, mean, invstd, sums, nA);
  else if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL((upcat_pad_bwd_bn_kernel<bf16, 4>), grid, dim3(256), 0, st, (const bf16*)dpad, (bf16*)da, (bf16*)db, N, h, w, Ca, Cb,
                       (const bf16*)y, (const bf16*)x, mean, invstd, sums, nA);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL((upcat_pad_bwd_bn_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)dpad, (float*)da, (float*)db, N, h, w, Ca, Cb,
                       (const float*)y, (const float*)x, mean, invstd, sums, nA);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_channel_sum_multi(const void* const* x, float* const* out, const int64_t* M, const int32_t* C,
                                    const int32_t* Creal, int n, int dtype, void* stream) {
  if (!x || !out || !M || !C || !Creal || n < 1 || n > FS_CSUM_MAX) return FS_EINVAL;
  const int es = dtype == FS_DTYPE_BF16 ? 2 : 4;
  const int VW = 16 / es;
  CsumBatch b;
  int blocks = 0;
  bool wide = true;
  for (int i = 0; i < n; ++i) {
    if (!x[i] || !out[i] || C[i] % 4 != 0 || M[i] <= 0) return FS_EINVAL;
    wide = wide && (C[i] % VW == 0 && VW != 4);
  }
  const int VL = wide ? VW : 4;
  for (int i = 0; i < n; ++i) {
    const int CG = C[i] / VL;
    int CGB = 1;
    while (CGB < CG && CGB < 64) CGB <<= 1;
    const int PL = 256 / CGB;
    const int gx = (int)std::max<long>(1, std::min<long>((M[i] + 4L * PL - 1) / (4L * PL), 256));
    const int gy = (CG + CGB - 1) / CGB;
    b.x[i] = x[i]; b.out[i] = out[i]; b.M[i] = M[i]; b.C[i] = C[i]; b.Creal[i] = Creal[i]; b.CGB[i] = CGB; b.gx[i] = gx;
    b.start[i] = blocks;
    blocks += gx * gy;
  }
  b.start[n] = blocks; b.n = n;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FS_DTYPE_BF16) {
    if (wide) hipLaunchKernelGGL((channel_sum_multi_kernel<bf16, 8>), dim3(blocks), dim3(256), 0, st, b);
    else hipLaunchKernelGGL((channel_sum_multi_kernel<bf16, 4>), dim3(blocks), dim3(256), 0, st, b);
  } else if (dtype == FS_DTYPE_F32) {
    hipLaunchKernelGGL((channel_sum_multi_kernel<float, 4>), dim3(blocks), dim3(256), 0, st, b);
  } else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_channel_sum(const void* x, float* out, int64_t M, int C, int Creal, int dtype, void* stream) {
  if (!x || !out || C % 4 != 0 || M <= 0) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int es = dtype == FS_DTYPE_BF16 ? 2 : 4;
  const int VW = 16 / es;                                  // channels per 16-byte lane
  const bool wide = C % VW == 0 && VW != 4;
  const int VL = wide ? VW : 4;
  const int CG = C / VL;
  int CGB = 1;
  while (CGB < CG && CGB < 64) CGB <<= 1;                   // channel lanes per block (power of two <= 64)
  const int PL = 256 / CGB;
  // <= 256 blocks per channel slab: every block ends in one same-address atomic per channel (~12 ns each)
  dim3 grid((unsigned)std::max<long>(1, std::min<long>((M + 4L * PL - 1) / (4L * PL), 256)), (CG + CGB - 1) / CGB);
  if (dtype == FS_DTYPE_BF16) {
    if (wide) hipLaunchKernelGGL((channel_sum_kernel<bf16, 8>), grid, dim3(256), 0, st, (const bf16*)x, out, (long)M, C, Creal, CGB);
    else hipLaunchKernelGGL((channel_sum_kernel<bf16, 4>), grid, dim3(256), 0, st, (const bf16*)x, out, (long)M, C, Creal, CGB);
  } else if (dtype == FS_DTYPE_F32) {
    hipLaunchKernelGGL((channel_sum_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)x, out, (long)M, C, Creal, CGB);
  } else return FS_EINVAL;
  return fs_launch_status();
}
