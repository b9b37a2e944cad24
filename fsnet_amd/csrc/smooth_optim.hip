// Edge-aware smoothness loss (fwd + bwd, all scales per launch), the final loss assembly, and the
// fused gradient-norm / clip / Adam update over a flat parameter arena.
// Replaces:
//   get_smooth_loss + mean-normalised disparity        monodepth_utils.py:168-181, monodepth2_decoder.py:294-299
//   F.adaptive_avg_pool2d(original_image_0, (h, w))     monodepth2_decoder.py:214-219
//   loss bookkeeping (loss/s, smooth_loss/s, total)     monodepth2_decoder.py:292-304, 343
//   clip_grad_norm_ + torch.optim.Adam.step             base_training_hooks.py:46-49, optimizers.py:7-8
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

// colour pyramid: out_s[b][c][y][x] = mean of the (H/h)x(W/w) block (adaptive_avg_pool2d with integer ratio)
__global__ __launch_bounds__(256) void color_pyramid_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                            int B, int H, int W, int h, int w) {
  const int ry = H / h, rx = W / w;
  const long total = (long)B * 3 * h * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int x = (int)(i % w); long q = i / w; int y = (int)(q % h); long bc = q / h;
    const float* src = img + bc * H * W;
    float s = 0.f;
    for (int a = 0; a < ry; ++a)
      for (int c = 0; c < rx; ++c) s += src[(long)(y * ry + a) * W + x * rx + c];
    out[i] = s / (float)(ry * rx);
  }
}

// per (scale, batch) sum of disp -> sums[s][b]  (mean = sums / (h*w))
__global__ __launch_bounds__(256) void smooth_mean_kernel(const FsSmoothArgs p) {
  const int s = blockIdx.z, b = blockIdx.y;
  const long hw = (long)p.h[s] * p.w[s];
  const float* d = p.disp[s] + (long)b * hw;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) acc += d[i];
  __shared__ double sh[4];
  double tot = block_sum_d((double)acc, sh);
  if (threadIdx.x == 0) atomicAdd(p.disp_sum + s * p.B + b, tot);
}

__device__ __forceinline__ float edge_w(const float* __restrict__ col, long hw, long i0, long i1) {
  float g = fabsf(col[i0] - col[i1]) + fabsf(col[hw + i0] - col[hw + i1]) + fabsf(col[2 * hw + i0] - col[2 * hw + i1]);
  return expf(-g * (1.f / 3.f));
}

// forward: sx[s] = sum |nd(x)-nd(x+1)| e^{-gx},  sy[s] likewise (nd = disp / (mean + 1e-7))
__global__ __launch_bounds__(256) void smooth_fwd_kernel(const FsSmoothArgs p) {
  const int s = blockIdx.z, b = blockIdx.y;
  const int h = p.h[s], w = p.w[s];
  const long hw = (long)h * w;
  const float* d = p.disp[s] + (long)b * hw;
  const float* col = p.color[s] + (long)b * 3 * hw;
  const float inv = 1.f / ((float)(p.disp_sum[s * p.B + b] / (double)hw) + 1e-7f);
  float ax = 0.f, ay = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
    int x = (int)(i % w), y = (int)(i / w);
    float v = d[i] * inv;
    if (x + 1 < w) ax += fabsf(v - d[i + 1] * inv) * edge_w(col, hw, i, i + 1);
    if (y + 1 < h) ay += fabsf(v - d[i + w] * inv) * edge_w(col, hw, i, i + w);
  }
  __shared__ double sh[4];
  double tx = block_sum_d((double)ax, sh);
  double ty = block_sum_d((double)ay, sh);
  if (threadIdx.x == 0) {
    atomicAdd(p.sm_sums + (s * p.B + b) * 2, tx);
    atomicAdd(p.sm_sums + (s * p.B + b) * 2 + 1, ty);
  }
}

// d loss / d nd at pixel i (both neighbours on each axis), scaled by the loss weights
__device__ __forceinline__ float smooth_dnd(const float* __restrict__ d, const float* __restrict__ col, long hw, int h,
                                            int w, int y, int x, float inv, float kx, float ky) {
  long i = (long)y * w + x;
  float v = d[i] * inv, g = 0.f;
  auto sgn = [](float a) { return a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f); };
  if (x + 1 < w) g += kx * sgn(v - d[i + 1] * inv) * edge_w(col, hw, i, i + 1);
  if (x > 0) g -= kx * sgn(d[i - 1] * inv - v) * edge_w(col, hw, i - 1, i);
  if (y + 1 < h) g += ky * sgn(v - d[i + w] * inv) * edge_w(col, hw, i, i + w);
  if (y > 0) g -= ky * sgn(d[i - w] * inv - v) * edge_w(col, hw, i - w, i);
  return g;
}

// backward pass 1: dot[s][b] = sum_i dnd(i) * disp(i)   (coupling through the mean normalisation)
__global__ __launch_bounds__(256) void smooth_bwd_dot_kernel(const FsSmoothArgs p) {
  const int s = blockIdx.z, b = blockIdx.y;
  const int h = p.h[s], w = p.w[s];
  const long hw = (long)h * w;
  const float* d = p.disp[s] + (long)b * hw;
  const float* col = p.color[s] + (long)b * 3 * hw;
  const float inv = 1.f / ((float)(p.disp_sum[s * p.B + b] / (double)hw) + 1e-7f);
  const float gout = p.gout ? (float)*p.gout : 1.f;
  const float wgt = gout * 1e-5f / (float)(1 << p.scale_id[s]) / (float)p.S;
  const float kx = wgt / (float)((long)p.B * h * (w - 1)), ky = wgt / (float)((long)p.B * (h - 1) * w);
  float* dd = p.d_disp[s] + (long)b * hw;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
    int x = (int)(i % w), y = (int)(i / w);
    float g = smooth_dnd(d, col, hw, h, w, y, x, inv, kx, ky);
    dd[i] = g;                 // pass 2 only rescales and shifts it (no second walk over the colour edges)
    acc += g * d[i];
  }
  __shared__ double sh[4];
  double tot = block_sum_d((double)acc, sh);
  if (threadIdx.x == 0) atomicAdd(p.dot + s * p.B + b, tot);
}

// backward pass 2: d_disp(i) = dnd(i)*inv - dot * inv^2 / (h*w)
__global__ __launch_bounds__(256) void smooth_bwd_apply_kernel(const FsSmoothArgs p) {
  const int s = blockIdx.z, b = blockIdx.y;
  const int h = p.h[s], w = p.w[s];
  const long hw = (long)h * w;
  float* dd = p.d_disp[s] + (long)b * hw;
  const float inv = 1.f / ((float)(p.disp_sum[s * p.B + b] / (double)hw) + 1e-7f);
  const float corr = (float)p.dot[s * p.B + b] * inv * inv / (float)hw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256)
    dd[i] = dd[i] * inv - corr;        // dd holds dnd from pass 1
}

// loss assembly: out[0..S) = loss/s (f64), out[S..2S) = smooth_loss/s, out[2S] = total.  One wave: lane b sums batch
// element b (b, b + 64, ...), wave sums combine them — the single-thread version walked ~100 dependent loads (9 us on
// the serial path between the loss forward and backward).
__global__ __launch_bounds__(64) void loss_finalize_kernel(const double* __restrict__ loss_sums,
                                                           const double* __restrict__ mask_sum,
                                                           const double* __restrict__ sm_sums, const FsSmoothArgs p,
                                                           double* __restrict__ out, double* __restrict__ total_out) {
  const int lane = threadIdx.x;
  double msum = 0.0;
  for (int b = lane; b < p.B; b += 64) msum += mask_sum[b];
  msum = wave_sum_d(msum);
  double total = 0.0;
  for (int s = 0; s < p.S; ++s) {
    const int h = p.h[s], w = p.w[s];
    double ax = 0.0, ay = 0.0, ls = 0.0;
    for (int b = lane; b < p.B; b += 64) {
      ax += sm_sums[(s * p.B + b) * 2]; ay += sm_sums[(s * p.B + b) * 2 + 1]; ls += loss_sums[s * p.B + b];
    }
    ax = wave_sum_d(ax); ay = wave_sum_d(ay); ls = wave_sum_d(ls);
    float sx = (float)(ax / (double)((long)p.B * h * (w - 1)));
    float sy = (float)(ay / (double)((long)p.B * (h - 1) * w));
    float sm = (sx + sy) * 1e-5f / (float)(1 << p.scale_id[s]);   // fp32 like the reference's smooth_loss
    double l = ls / (msum + 1e-6) + (double)sm;
    if (lane == 0) { out[s] = l; out[p.S + s] = (double)sm; }
    total += l;
  }
  if (lane == 0) {
    out[2 * p.S] = total / (double)p.S;
    if (total_out) *total_out = total / (double)p.S;     // the scalar the caller differentiates: no device copy needed
  }
}

// ---------------------------------------------------------------------------------------------
// optimizer
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out,
                                                     int* __restrict__ step_counter) {
  // the optimizer's device-resident step count is bumped here (nothing in this kernel reads it; the Adam launch that
  // follows does): one serial 5-us node less at the tail of every step
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;
  double acc = 0.0;
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 256 * 4;
  for (; i + 3 + stride < n; i += 2 * stride) {          // two 16-byte loads in flight per thread
    float4 v = *reinterpret_cast<const float4*>(g + i), u = *reinterpret_cast<const float4*>(g + i + stride);
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    acc += (double)(u.x * u.x + u.y * u.y) + (double)(u.z * u.z + u.w * u.w);
  }
  for (; i + 3 < n; i += stride) {
    float4 v = *reinterpret_cast<const float4*>(g + i);
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  if (i < n) for (long k = i; k < n && k < i + 4; ++k) acc += (double)g[k] * g[k];
  // one f64 atomic per BLOCK: they all hit the same address (~12 ns each, serialised) — 8192 per-wave atomics
  // were 98 of this kernel's 114 us
  __shared__ double sh[4];
  acc = block_sum_d(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}

__global__ void counter_incr_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += 1; }

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   float max_norm, const double* __restrict__ sumsq,
                                                   float grad_scale, const int* __restrict__ step_ptr,
                                                   const float* __restrict__ lr_ptr) {
  if (step_ptr) {   // device-resident step count / learning rate: the launch can be replayed from a hipGraph
    const float st = (float)*step_ptr;
    bc1 = 1.f - powf(b1, st);
    bc2_sqrt = sqrtf(1.f - powf(b2, st));
  }
  if (lr_ptr) lr = *lr_ptr;
  float coef = grad_scale;
  if (sumsq && max_norm > 0.f) {
    float norm = (float)sqrt(*sumsq) * grad_scale;
    coef *= fminf(max_norm / (norm + 1e-6f), 1.f);
  }
  const float step = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gi = g[i] * coef;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = pi - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

}  // namespace

extern "C" int fs_color_pyramid(const float* img, float* out, int B, int H, int W, int h, int w, void* stream) {
  if (!img || !out || h <= 0 || w <= 0 || H % h != 0 || W % w != 0) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long total = (long)B * 3 * h * w;
  hipLaunchKernelGGL(color_pyramid_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, img, out, B, H, W, h, w);
  return fs_launch_status();
}

static bool smooth_valid(const FsSmoothArgs* a) {
  if (!a || a->S < 1 || a->S > 4 || a->B < 1 || !a->disp_sum) return false;
  for (int s = 0; s < a->S; ++s) if (!a->disp[s] || !a->color[s] || a->h[s] < 2 || a->w[s] < 2) return false;
  return true;
}
static dim3 smooth_grid(const FsSmoothArgs* a) {
  long hw = (long)a->h[0] * a->w[0];
  for (int s = 1; s < a->S; ++s) hw = std::max<long>(hw, (long)a->h[s] * a->w[s]);
  return dim3((unsigned)std::min<long>((hw + 255) / 256, 32), a->B, a->S);
}

extern "C" int fs_smooth_mean(const FsSmoothArgs* a, void* stream) {
  if (!smooth_valid(a)) return FS_EINVAL;
  hipLaunchKernelGGL(smooth_mean_kernel, smooth_grid(a), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  return fs_launch_status();
}
extern "C" int fs_smooth_fwd(const FsSmoothArgs* a, void* stream) {
  if (!smooth_valid(a) || !a->sm_sums) return FS_EINVAL;
  hipLaunchKernelGGL(smooth_fwd_kernel, smooth_grid(a), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  return fs_launch_status();
}
extern "C" int fs_smooth_bwd(const FsSmoothArgs* a, void* stream) {
  if (!smooth_valid(a) || !a->dot) return FS_EINVAL;
  for (int s = 0; s < a->S; ++s) if (!a->d_disp[s]) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(smooth_bwd_dot_kernel, smooth_grid(a), dim3(256), 0, st, *a);
  hipLaunchKernelGGL(smooth_bwd_apply_kernel, smooth_grid(a), dim3(256), 0, st, *a);
  return fs_launch_status();
}
extern "C" int fs_loss_finalize(const double* loss_sums, const double* mask_sum, const double* sm_sums,
                                const FsSmoothArgs* a, double* out, double* total_out, void* stream) {
  if (!loss_sums || !mask_sum || !sm_sums || !a || !out) return FS_EINVAL;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), loss_sums, mask_sum, sm_sums, *a, out,
                     total_out);
  return fs_launch_status();
}

extern "C" int fs_sumsq(const float* g, int64_t n, double* out, int* step_counter, void* stream) {
  if (!g || !out || n <= 0) return FS_EINVAL;
  // (one same-address f64 atomic per block at the end: 512 blocks keep that tail at ~6 us)
  long blocks = std::min<long>((n / 4 + 255) / 256 + 1, 512);
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, (long)n, out, step_counter);
  return fs_launch_status();
}

extern "C" int fs_counter_incr(int* counter, void* stream) {
  if (!counter) return FS_EINVAL;
  hipLaunchKernelGGL(counter_incr_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), counter);
  return fs_launch_status();
}

extern "C" int fs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float max_norm, const double* sumsq,
                            float grad_scale, const int* step_ptr, const float* lr_ptr, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || (step < 1 && !step_ptr)) return FS_EINVAL;
  if (step < 1) step = 1;
  float bc1 = 1.f - (float)pow((double)beta1, (double)step);
  float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  long blocks = std::min<long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v,
                     (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, max_norm, sumsq, grad_scale, step_ptr, lr_ptr);
  return fs_launch_status();
}
