// Mei unified fisheye camera model: the per-calibration ray table of MeiCameraProjection.image2cam and the
// per-batch mask staging of FishEyeDecoder (BASELINE configs[3]).
// Replaces (reference):
//   newton_methods / bisection_methods / whole_map_backtracking (numba-JIT CPU loops)   mei_fisheye_utils.py:66-120
//   MeiCameraProjection.image2cam cache fill (X, Y, Z, mask)                            mei_fisheye_utils.py:150-166
//   patched_mask * mask[:, 0]                                                           monodepth2_decoder.py:409
// The table is built once per calibration (the host caches it by the reference's own key) — one thread per pixel,
// both root finders in f64 exactly as numba types them (the f32 radius meets f64 calibration scalars).
#include "common.h"
#include "fsnet_hip_internal.h"

namespace {

__device__ __forceinline__ double radial(double k1, double k2, double r1, double r0) {
  const double r2 = r0 * r0;
  return r0 - r1 / (1.0 + k1 * r2 + k2 * (r2 * r2));
}
__device__ __forceinline__ double mirror(double r0, double xi, double Z) {
  return r0 * r0 - (1.0 - Z * Z) / ((xi + Z) * (xi + Z));
}

__global__ __launch_bounds__(256) void mei_lut_kernel(float* __restrict__ lut, int H, int W, float g1, float g2,
                                                      float u0, float v0, double k1, double k2, double xi) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long HW = (long)H * W;
  if (i >= HW) return;
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  float X = ((float)x - u0) / g1, Y = ((float)y - v0) / g2;
  const float r1f = sqrtf(X * X + Y * Y);
  // r1 = r0 (1 + k1 r0^2 + k2 r0^4)  ->  r0 by Newton with a forward-difference slope (tol 1e-6, <= 100 steps)
  const double tol = 1e-6, r1 = (double)r1f;
  double r0 = r1;
  for (int it = 0; it < 100; ++it) {
    const double f = radial(k1, k2, r1, r0);
    if (fabs(f) < tol) break;
    const double df = (radial(k1, k2, r1, r0 + tol) - f) / tol;
    r0 = r0 - f / df;
  }
  // r0^2 = (1 - Z^2) / (xi + Z)^2  ->  Z in [0, 1] by bisection; no sign change: invalid
  double x0 = 0.0, x1 = 1.0, Zd = -1.0;
  bool ok = !(mirror(r0, xi, x0) * mirror(r0, xi, x1) > 0.0);
  if (ok) {
    for (int it = 0; it < 100; ++it) {
      Zd = (x0 + x1) / 2;
      const double f = mirror(r0, xi, Zd);
      if (fabs(f) < tol) break;
      if (f * mirror(r0, xi, x0) < 0.0) x1 = Zd; else x0 = Zd;
    }
  }
  float Z = (float)Zd;
  float m = ok ? 1.f : 0.f;
  if (Z < 0.05f) m = 0.f;
  if (m == 0.f) { Z = -1.f; X = -1.f; Y = -1.f; }
  const float zx = Z + (float)xi;
  lut[i] = X * zx;
  lut[HW + i] = Y * zx;
  lut[2 * HW + i] = Z;
  lut[3 * HW + i] = m;
}

__global__ __launch_bounds__(256) void mei_mask_kernel(const float* const* __restrict__ lut_ptrs,
                                                       const double* __restrict__ patched_mask,
                                                       float* __restrict__ warp_mask, int B, long HW) {
  const int b = blockIdx.y;
  const float* m = lut_ptrs[b] + 3 * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    const double pm = patched_mask ? patched_mask[(long)b * HW + i] : 1.0;
    warp_mask[(long)b * HW + i] = (float)(pm * (double)m[i]);
  }
}

// points = ray table x norm  (image2cam :183-187; FishEyeDecoder.get_prediction reads z, monodepth2_decoder.py:413-420)
__global__ __launch_bounds__(256) void mei_points_kernel(const float* const* __restrict__ lut_ptrs,
                                                         const float* __restrict__ norm, float* __restrict__ points,
                                                         int B, long HW) {
  const int b = blockIdx.y;
  const float* t = lut_ptrs[b];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    const float n = norm[(long)b * HW + i];
    float* o = points + ((long)b * HW + i) * 3;
    o[0] = t[i] * n; o[1] = t[HW + i] * n; o[2] = t[2 * HW + i] * n;
  }
}

}  // namespace

extern "C" int fs_mei_lut(float* lut, int H, int W, float gamma1, float gamma2, float u0, float v0, double k1,
                          double k2, double xi, void* stream) {
  if (!lut || H < 1 || W < 1 || gamma1 == 0.f || gamma2 == 0.f) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long HW = (long)H * W;
  hipLaunchKernelGGL(mei_lut_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, st, lut, H, W, gamma1, gamma2,
                     u0, v0, k1, k2, xi);
  return fs_launch_status();
}

extern "C" int fs_mei_stage_mask(const float* const* lut_ptrs, const double* patched_mask, float* warp_mask, int B,
                                 int H, int W, void* stream) {
  if (!lut_ptrs || !warp_mask || B < 1 || H < 1 || W < 1) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long HW = (long)H * W;
  dim3 grid((unsigned)((HW + 1023) / 1024), B);
  hipLaunchKernelGGL(mei_mask_kernel, grid, dim3(256), 0, st, lut_ptrs, patched_mask, warp_mask, B, HW);
  return fs_launch_status();
}

extern "C" int fs_mei_points(const float* const* lut_ptrs, const float* norm, float* points, int B, int H, int W,
                             void* stream) {
  if (!lut_ptrs || !norm || !points || B < 1 || H < 1 || W < 1) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long HW = (long)H * W;
  dim3 grid((unsigned)((HW + 1023) / 1024), B);
  hipLaunchKernelGGL(mei_points_kernel, grid, dim3(256), 0, st, lut_ptrs, norm, points, B, HW);
  return fs_launch_status();
}
