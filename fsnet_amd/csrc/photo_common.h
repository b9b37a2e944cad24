// Device helpers shared by the photometric kernels (photometric.hip: staged kernels; photo_fused.hip: fused ones):
// per-pixel geometry (depth upsample -> backproject -> project, pinhole and Mei fisheye), bilinear sampler taps,
// tie-break noise.  Reference call sites: monodepth2_decoder.py:61-116,355-411; monodepth_utils.py:101-165;
// mei_fisheye_utils.py:14-51.
#pragma once
#include "common.h"
#include "fsnet_hip_internal.h"

namespace {


constexpr float C1 = 0.01f * 0.01f;
constexpr float C2 = 0.03f * 0.03f;
constexpr int GEO_STRIDE = 48;  // per batch element: invK[9] K[9] P[2][12]

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// ---------------------------------------------------------------------------------------------
// shared per-pixel geometry: depth upsample -> backproject -> project -> sample coordinates
// ---------------------------------------------------------------------------------------------
struct Geo {
  float D;             // upsampled depth (fisheye: ray norm)
  int y0, x0, y1, x1;  // low-res taps
  float ly, lx;        // low-res lambdas
  float r[3];          // K^-1 [x y 1]  (fisheye: the ray-table entry X, Y, Z of the pixel)
  float X, Y, Zp;      // projected point, Zp = Z + eps  (fisheye: the transformed point, no eps)
  float ixu, iyu;      // unnormalised (unclamped) sample coordinates
};

// Mei unified camera model, _cam2image + mei_distort (mei_fisheye_utils.py:14-51) in the reference's fp32 operation
// order.  m = {k1, k2, xi, gamma1, gamma2, u0, v0}.
__device__ __forceinline__ void mei_cam2image(const float* __restrict__ m, float qx, float qy, float qz, float& u,
                                              float& v) {
  const float eps = 1e-6f;
  float n = sqrtf(qx * qx + qy * qy + qz * qz);
  float s = n + eps;
  float x = qx / s, y = qy / s, z = qz / s;
  float a = z + m[2] + eps;
  x = x / a; y = y / a;
  float ro2 = x * x + y * y;
  float d = 1.f + m[0] * ro2 + m[1] * ro2 * ro2;
  u = m[3] * (x * d) + m[5];
  v = m[4] * (y * d) + m[6];
}

// reverse mode of mei_cam2image: (gu, gv) = d loss / d (u, v)  ->  d loss / d q
__device__ __forceinline__ void mei_cam2image_bwd(const float* __restrict__ m, float qx, float qy, float qz, float gu,
                                                  float gv, float (&dq)[3]) {
  const float eps = 1e-6f;
  float n = sqrtf(qx * qx + qy * qy + qz * qz);
  float s = n + eps;
  float xs = qx / s, ys = qy / s, zs = qz / s;
  float a = zs + m[2] + eps;
  float xm = xs / a, ym = ys / a;
  float ro2 = xm * xm + ym * ym;
  float d = 1.f + m[0] * ro2 + m[1] * ro2 * ro2;
  float g_xd = m[3] * gu, g_yd = m[4] * gv;
  float g_d = g_xd * xm + g_yd * ym;
  float g_ro2 = g_d * (m[0] + 2.f * m[1] * ro2);
  float g_xm = g_xd * d + g_ro2 * 2.f * xm, g_ym = g_yd * d + g_ro2 * 2.f * ym;
  float g_xs = g_xm / a, g_ys = g_ym / a;
  float g_zs = -(g_xm * xm + g_ym * ym) / a;
  float g_s = -(g_xs * xs + g_ys * ys + g_zs * zs) / s;
  float inv_n = n > 0.f ? 1.f / n : 0.f;
  dq[0] = g_xs / s + g_s * qx * inv_n;
  dq[1] = g_ys / s + g_s * qy * inv_n;
  dq[2] = g_zs / s + g_s * qz * inv_n;
}

__device__ __forceinline__ void upsample_taps(int y, int x, int H, int W, int h, int w, Geo& g) {
  // ATen upsample_bilinear2d, align_corners=True: scale = (in-1)/(out-1)
  float sh = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
  float sw = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
  float fy = sh * (float)y, fx = sw * (float)x;
  g.y0 = (int)fy; g.x0 = (int)fx;
  g.y1 = g.y0 + (g.y0 < h - 1 ? 1 : 0); g.x1 = g.x0 + (g.x0 < w - 1 ? 1 : 0);
  g.ly = fy - (float)g.y0; g.lx = fx - (float)g.x0;
}

// ray of pixel (y, x) and the depth (fisheye: ray norm) bilinearly upsampled from the scale's map -> g.D, g.r, taps
__device__ __forceinline__ void pixel_ray(const FsPhotoArgs& p, const float* __restrict__ depth, int b, int y, int x,
                                          int H, int W, int h, int w, const float* __restrict__ ge, Geo& g) {
  upsample_taps(y, x, H, W, h, w, g);
  const float* d = depth + (long)b * h * w;
  float d00 = d[g.y0 * w + g.x0], d01 = d[g.y0 * w + g.x1], d10 = d[g.y1 * w + g.x0], d11 = d[g.y1 * w + g.x1];
  g.D = (1.f - g.ly) * ((1.f - g.lx) * d00 + g.lx * d01) + g.ly * ((1.f - g.lx) * d10 + g.lx * d11);
  if (p.lut_ptrs) {
    const float* lut = p.lut_ptrs[b];
    const long HW = (long)H * W, o = (long)y * W + x;
    g.r[0] = lut[o]; g.r[1] = lut[HW + o]; g.r[2] = lut[2 * HW + o];
  } else {
    float px = (float)x, py = (float)y;
    g.r[0] = ge[0] * px + ge[1] * py + ge[2];
    g.r[1] = ge[3] * px + ge[4] * py + ge[5];
    g.r[2] = ge[6] * px + ge[7] * py + ge[8];
  }
}

// the point D * r through frame f's transform / projection -> sample coordinates in the source frame
__device__ __forceinline__ void project_ray(const FsPhotoArgs& p, int b, int H, int W, const float* __restrict__ ge,
                                            int f, Geo& g) {
  const float* P = ge + 18 + f * 12;
  if (p.lut_ptrs) {
    // FishEyeDecoder._generate_images_pred (monodepth2_decoder.py:355-387): point = ray table x norm, T, cam2image
    float cx = g.r[0] * g.D, cy = g.r[1] * g.D, cz = g.r[2] * g.D;
    g.X = P[0] * cx + P[1] * cy + P[2] * cz + P[3];
    g.Y = P[4] * cx + P[5] * cy + P[6] * cz + P[7];
    g.Zp = P[8] * cx + P[9] * cy + P[10] * cz + P[11];
    float u, v;
    mei_cam2image(p.mei + (long)b * 8, g.X, g.Y, g.Zp, u, v);
    float un = u / (float)max(W - 1, 1) * 2.f - 1.f, vn = v / (float)max(H - 1, 1) * 2.f - 1.f;
    g.ixu = (un + 1.f) * 0.5f * (float)(W - 1);
    g.iyu = (vn + 1.f) * 0.5f * (float)(H - 1);
    return;
  }
  float cx = g.D * g.r[0], cy = g.D * g.r[1], cz = g.D * g.r[2];
  g.X = P[0] * cx + P[1] * cy + P[2] * cz + P[3];
  g.Y = P[4] * cx + P[5] * cy + P[6] * cz + P[7];
  g.Zp = (P[8] * cx + P[9] * cy + P[10] * cz + P[11]) + 1e-7f;
  float u = g.X / g.Zp, v = g.Y / g.Zp;
  // Project3D normalisation followed by grid_sample's align_corners=True un-normalisation
  float un = (u / (float)(W - 1) - 0.5f) * 2.f, vn = (v / (float)(H - 1) - 0.5f) * 2.f;
  g.ixu = (un + 1.f) * 0.5f * (float)(W - 1);
  g.iyu = (vn + 1.f) * 0.5f * (float)(H - 1);
}

__device__ __forceinline__ void project_pixel(const FsPhotoArgs& p, const float* __restrict__ depth, int b, int y,
                                              int x, int H, int W, int h, int w, const float* __restrict__ ge, int f,
                                              Geo& g) {
  pixel_ray(p, depth, b, y, x, H, W, h, w, ge, g);
  project_ray(p, b, H, W, ge, f, g);
}

struct Taps {
  int x0, x1, y0, y1;
  float wx, wy;   // weight of the x1 / y1 side
  float mx, my;   // gradient multiplier of the border clamp (0 when clipped)
};
__device__ __forceinline__ void bilinear_taps(float ixu, float iyu, int H, int W, Taps& t) {
  float ix = ixu, iy = iyu;
  t.mx = 1.f; t.my = 1.f;
  if (!(ix > 0.f)) { ix = 0.f; t.mx = 0.f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); t.mx = 0.f; }
  if (!(iy > 0.f)) { iy = 0.f; t.my = 0.f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); t.my = 0.f; }
  float fx = floorf(ix), fy = floorf(iy);
  t.x0 = (int)fx; t.y0 = (int)fy;
  t.wx = ix - fx; t.wy = iy - fy;
  t.x1 = min(t.x0 + 1, W - 1); t.y1 = min(t.y0 + 1, H - 1);  // weight is 0 whenever the +1 tap is out of range
}

__device__ __forceinline__ uint32_t hash_u32(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16;
  return a;
}
__device__ __forceinline__ float tie_noise(int seed, uint32_t key) {
  // stand-in for the reference's torch.randn(...)*1e-5 tie-break noise (monodepth2_decoder.py:258-259): hashed
  // Box-Muller on the hardware transcendentals (v_log_f32 = log2, v_cos_f32 takes revolutions)
  if (seed < 0) return 0.f;
  uint32_t h1 = hash_u32(key * 2u + 0x9e3779b9u * (uint32_t)(seed + 1));
  uint32_t h2 = hash_u32(h1 ^ 0x85ebca6bu);
  float u1 = ((float)(h1 >> 8) + 1.f) * (1.f / 16777217.f), u2 = (float)(h2 >> 8) * (1.f / 16777216.f);
  return 1e-5f * __builtin_amdgcn_sqrtf(-2.f * 0.69314718f * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
}

}  // namespace
