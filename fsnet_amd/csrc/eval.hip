// Depth evaluation on the device: the resize, mask, median scaling and the seven depth-error
// metrics of the reference's KITTI evaluation, so a validation pass never leaves the GPU between the network and
// the numbers.  Replaces:
//   cv2.resize(..., INTER_LINEAR) of the (inverse) depth     monodepth/pipeline_hooks/evaluation_hooks/base_evaluation_hooks.py:57
//                                                            monodepth/evaluation/kitti_unsupervised_eval.py:49
//   mask / Garg crop / median ratio / clamp                  kitti_unsupervised_eval.py:50-74
//   compute_errors (abs_rel, sq_rel, rmse, rmse_log, a1-a3)  monodepth/networks/utils/monodepth_utils.py:271-289
// cv2 is a third-party dependency that is absent from the reference tree: its INTER_LINEAR rule for
// single-channel float images is restated here from OpenCV's resize.cpp (pixel centres at (x+0.5)*scale-0.5, floor,
// fraction zeroed and index clamped at both borders).
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

__device__ __forceinline__ void lin_coord(int d, float scale, int n, int& s0, float& f) {
  float fx = ((float)d + 0.5f) * scale - 0.5f;
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= n - 1) { fx = 0.f; sx = n - 1; }
  s0 = sx; f = fx;
}

// dst[y][x] = bilinear(src) with cv2's INTER_LINEAR rule; invert: dst = 1 / bilinear(1 / src)
__device__ __forceinline__ float resample(const float* __restrict__ src, int h, int w, int H, int W, int y, int x,
                                          bool invert) {
  if (h == H && w == W) return src[(long)y * w + x];
  int x0, y0; float fx, fy;
  lin_coord(x, (float)w / (float)W, w, x0, fx);
  lin_coord(y, (float)h / (float)H, h, y0, fy);
  const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
  float v00 = src[(long)y0 * w + x0], v01 = src[(long)y0 * w + x1];
  float v10 = src[(long)y1 * w + x0], v11 = src[(long)y1 * w + x1];
  if (invert) { v00 = 1.f / v00; v01 = 1.f / v01; v10 = 1.f / v10; v11 = 1.f / v11; }
  // cv2 (float path): horizontal pass with (1-fx, fx), then vertical with (1-fy, fy)
  float top = v00 * (1.f - fx) + v01 * fx, bot = v10 * (1.f - fx) + v11 * fx;
  float v = top * (1.f - fy) + bot * fy;
  return invert ? 1.f / v : v;
}

__global__ __launch_bounds__(256) void resize_linear_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int h, int w, int H, int W, int invert) {
  const long total = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    dst[i] = resample(src, h, w, H, W, (int)(i / W), (int)(i % W), invert != 0);
}

// ---- one image per block (1024 threads): compact the valid (gt, pred) pairs once, then two medians by radix
//      select and 14 error sums over the compacted list (ground truth is sparse: 5-30 % of the pixels) -----------
struct EvalCtx {
  const float* pred; const float* gt;
  int h, w, H, W;
  int y0, y1, x0, x1;     // Garg crop
};

// k-th smallest (0-based) of v[0..n): 4 passes over 8-bit digits of the float bit pattern (all values are positive,
// so the patterns order like the floats).  v = interleaved pairs, element `which`.
__device__ float radix_select(const float2* __restrict__ v, long n, int which, long k, unsigned* hist,
                              unsigned* s_prefix, long* s_k) {
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
      float2 e = v[i];
      unsigned u = __float_as_uint(which ? e.y : e.x);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      long kk = k; unsigned d = 0;
      for (; d < 256; ++d) { if (kk < (long)hist[d]) break; kk -= hist[d]; }
      *s_prefix = prefix | (d << shift); *s_k = kk;
    }
    __syncthreads();
    prefix = *s_prefix; k = *s_k; mask |= 255u << shift;
    __syncthreads();
  }
  return __uint_as_float(prefix);
}

__device__ void error_sums(const float2* __restrict__ v, long n, float ratio, double* acc /*7*/, double* sh) {
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const float2 e = v[i];
    const float g = e.x;
    float p = e.y * ratio;
    p = fminf(fmaxf(p, 1e-3f), 80.0f);
    float th = fmaxf(g / p, p / g);
    float d = g - p, dl = logf(g) - logf(p);
    s[0] += (double)(fabsf(d) / g);            // abs_rel
    s[1] += (double)(d * d / g);               // sq_rel
    s[2] += (double)(d * d);                   // rmse^2
    s[3] += (double)(dl * dl);                 // rmse_log^2
    s[4] += th < 1.25f ? 1.0 : 0.0;
    s[5] += th < 1.25f * 1.25f ? 1.0 : 0.0;
    s[6] += th < 1.25f * 1.25f * 1.25f ? 1.0 : 0.0;
  }
  for (int j = 0; j < 7; ++j) {
    double t = wave_sum_d(s[j]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) { double a = 0; for (int q = 0; q < (int)(blockDim.x >> 6); ++q) a += sh[q]; acc[j] = a; }
    __syncthreads();
  }
}

// out[b][16] = { ratio, err[7] (median-scaled), abs_err[7] (unscaled), n_valid }; scratch[b] holds up to H*W pairs
__global__ __launch_bounds__(1024) void depth_eval_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          int B, int h, int w, int H, int W, float2* __restrict__ scratch,
                                                          double* __restrict__ out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix;
  __shared__ long s_k;
  __shared__ double sh[16];
  __shared__ double acc[7];
  __shared__ unsigned s_n;
  const int b = blockIdx.x;
  const float* pr = pred + (long)b * h * w;
  const float* g0 = gt + (long)b * H * W;
  float2* v = scratch + (long)b * H * W;
  // float64 like numpy: in float32 0.99189189 * 370 rounds up to 367.0 and the crop gains a row
  const int y0 = (int)(0.40810811 * (double)H), y1 = (int)(0.99189189 * (double)H);
  const int x0 = (int)(0.03594771 * (double)W), x1 = (int)(0.96405229 * (double)W);
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  for (int y = y0; y < y1; ++y)
    for (int x = x0 + threadIdx.x; x < x1; x += blockDim.x) {
      const float g = g0[(long)y * W + x];
      if (g > 1e-3f && g < 80.0f) {
        const float p = resample(pr, h, w, H, W, y, x, false);
        v[atomicAdd(&s_n, 1u)] = make_float2(g, p);       // order is irrelevant to medians and sums
      }
    }
  __syncthreads();
  const long n = s_n;
  double* o = out + (long)b * 16;
  if (n == 0) { if (threadIdx.x < 16) o[threadIdx.x] = 0.0; return; }
  // np.median: mean of the two middle order statistics (in the array's float32)
  float med[2];
  for (int which = 0; which < 2; ++which) {
    float lo = radix_select(v, n, which, (n - 1) / 2, hist, &s_prefix, &s_k);
    float hi = (n & 1) ? lo : radix_select(v, n, which, n / 2, hist, &s_prefix, &s_k);
    med[which] = (lo + hi) * 0.5f;
  }
  const float ratio = med[0] / med[1];
  for (int pass = 0; pass < 2; ++pass) {
    error_sums(v, n, pass == 0 ? ratio : 1.0f, acc, sh);
    if (threadIdx.x == 0) {
      double inv = 1.0 / (double)n;
      double* e = o + 1 + pass * 7;
      e[0] = acc[0] * inv; e[1] = acc[1] * inv; e[2] = sqrt(acc[2] * inv); e[3] = sqrt(acc[3] * inv);
      e[4] = acc[4] * inv; e[5] = acc[5] * inv; e[6] = acc[6] * inv;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { o[0] = (double)ratio; o[15] = (double)n; }
}

}  // namespace

extern "C" int fs_resize_linear(const float* src, float* dst, int h, int w, int H, int W, int invert, void* stream) {
  if (!src || !dst || h < 1 || w < 1 || H < 1 || W < 1) return FS_EINVAL;
  long total = (long)H * W;
  unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(resize_linear_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, h, w,
                     H, W, invert);
  return fs_launch_status();
}

extern "C" int fs_depth_eval(const float* pred, const float* gt, int B, int h, int w, int H, int W, void* scratch,
                             double* out16, void* stream) {
  if (!pred || !gt || !scratch || !out16 || B < 1 || h < 1 || w < 1 || H < 2 || W < 2) return FS_EINVAL;
  hipLaunchKernelGGL(depth_eval_kernel, dim3(B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), pred, gt, B, h, w, H,
                     W, static_cast<float2*>(scratch), out16);
  return fs_launch_status();
}
