// Training input pipeline on the device: one launch turns a batch of raw uint8 frames into the
// network / loss inputs.  Replaces, per sample and per frame, the DataLoader-worker chain of
// configs/kitti_wpose_example:129-155:
//   ConvertToFloat                                          vision_base/data/augmentations/augmentations.py:50-59
//   RandomWarpAffine (cv2.warpAffine, INTER_LINEAR for the frames, INTER_NEAREST for patched_mask)      :436-497
//   RandomMirror (images, mask)                                                                         :377-433
//   Shuffle[RandomBrightness :572-591, RandomContrast :545-569, ConvertColor/RandomSaturation :527-542, :200-226]
//   Normalize (mean/std for ('image', i); 0/1 for ('original_image', i))                                :91-109
//   ConvertToTensor (HWC -> CHW)                                                                        :62-88
// The random draws and the O(B) bookkeeping (P2, relative poses) stay on the host in the mirrored classes
// (fsnet_amd/vision_base/data/augmentations); this kernel receives them as a per-sample plan.
// cv2 is a third-party dependency absent from the reference tree: warpAffine follows OpenCV's imgwarp.cpp (matrix
// inverted in f64 by the host, 10-bit fixed-point coordinates, 1/32-pixel bilinear grid, BORDER_CONSTANT 0) and
// cvtColor its color_hsv RGB2HSV_f / HSV2RGB_f (hrange 360) — see oracle/augment_oracle.py ("parity unpinned").
// All pixel arithmetic is fp32 in the reference's operation order (the build has -ffp-contract=off, divisions are
// IEEE), so the result equals the numpy restatement bit for bit.
#include "common.h"
#include "fsnet_hip_internal.h"
#include <cstdint>

namespace {

constexpr int AB_BITS = 10, INTER_BITS = 5, INTER_TAB = 1 << INTER_BITS;
constexpr float FLT_EPS = 1.1920929e-07f;

__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
  v = fmaxf(fmaxf(r, g), b);
  const float vmin = fminf(fminf(r, g), b);
  float diff = v - vmin;
  s = diff / (fabsf(v) + FLT_EPS);
  diff = 60.f / (diff + FLT_EPS);
  if (v == r) h = (g - b) * diff;
  else if (v == g) h = (b - r) * diff + 120.f;
  else h = (r - g) * diff + 240.f;
  if (h < 0.f) h += 360.f;
}

__device__ __forceinline__ void hsv2rgb(float h, float s, float v, float& r, float& g, float& b) {
  if (s == 0.f) { r = g = b = v; return; }
  float hh = h * (6.f / 360.f);
  if (hh < 0.f) hh += 6.f;
  if (hh >= 6.f) hh -= 6.f;
  int sector = (int)floorf(hh);
  float f = hh - (float)sector;
  if ((unsigned)sector >= 6u) { sector = 0; f = 0.f; }
  const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * f), t3 = v * (1.f - s * (1.f - f));
  switch (sector) {           // (b, g, r) <- {1,3,0} {1,0,2} {3,0,1} {0,2,1} {0,1,3} {2,1,0}
    case 0: b = t1; g = t3; r = t0; break;
    case 1: b = t1; g = t0; r = t2; break;
    case 2: b = t3; g = t0; r = t1; break;
    case 3: b = t0; g = t2; r = t1; break;
    case 4: b = t0; g = t1; r = t3; break;
    default: b = t2; g = t1; r = t0; break;
  }
}

// Shuffle[RandomBrightness, RandomContrast, HSV round trip with RandomSaturation] in the drawn order (plan slots 0-2,
// applied-bits in slot 3) on one pixel, 0..255 scale
__device__ __forceinline__ void colour_chain(float (&c)[3], const int32_t* ip, const float* fp) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int op = ip[q];
    if (op == 0) {
      if (ip[3] & 1) { c[0] = c[0] + fp[0]; c[1] = c[1] + fp[0]; c[2] = c[2] + fp[0]; }
    } else if (op == 1) {
      if (ip[3] & 2) { c[0] = c[0] * fp[1]; c[1] = c[1] * fp[1]; c[2] = c[2] * fp[1]; }
    } else if (op == 2 && (ip[3] & 8)) {      // ConvertColor(HSV) .. ConvertColor(RGB) with or without a draw
      float h, s, v;
      rgb2hsv(c[0], c[1], c[2], h, s, v);
      if (ip[3] & 4) s = s * fp[2];
      hsv2rgb(h, s, v, c[0], c[1], c[2]);
    }
  }
}

__global__ __launch_bounds__(256) void augment_frames_kernel(const FsAugArgs p) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.H * p.W) return;
  const int y = i / p.W, xo = i - y * p.W;
  const int32_t* ip = p.iplan + b * FS_AUG_IPLAN;
  const float* fp = p.fplan + b * FS_AUG_FPLAN;
  const double* m = p.minv + b * 6;
  const int sh = ip[5], sw = ip[6];
  const int x = ip[4] ? p.W - 1 - xo : xo;                 // RandomMirror: out[:, xo] = warped[:, W-1-xo]

  // cv2.warpAffine coordinates: saturate_cast<int>(double) = round-half-even
  const long adelta = (long)rint(m[0] * (double)x * (double)(1 << AB_BITS));
  const long bdelta = (long)rint(m[3] * (double)x * (double)(1 << AB_BITS));
  const long X00 = (long)rint((m[1] * (double)y + m[2]) * (double)(1 << AB_BITS));
  const long Y00 = (long)rint((m[4] * (double)y + m[5]) * (double)(1 << AB_BITS));

  {   // patched_mask: ones warped with INTER_NEAREST -> 1 where the nearest source pixel exists
    const long rd = (1 << AB_BITS) / 2;
    const long sx = (X00 + rd + adelta) >> AB_BITS, sy = (Y00 + rd + bdelta) >> AB_BITS;
    if (p.mask) p.mask[((long)b * p.H + y) * p.W + xo] = (sx >= 0 && sx < sw && sy >= 0 && sy < sh) ? 1.0 : 0.0;
  }
  const long rd = (1 << AB_BITS) / INTER_TAB / 2;
  const long X = (X00 + rd + adelta) >> (AB_BITS - INTER_BITS), Y = (Y00 + rd + bdelta) >> (AB_BITS - INTER_BITS);
  const long sx = X >> INTER_BITS, sy = Y >> INTER_BITS;
  const float fx = (float)(X & (INTER_TAB - 1)) / (float)INTER_TAB, fy = (float)(Y & (INTER_TAB - 1)) / (float)INTER_TAB;
  const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
  const bool x0ok = sx >= 0 && sx < sw, x1ok = sx + 1 >= 0 && sx + 1 < sw;
  const bool y0ok = sy >= 0 && sy < sh, y1ok = sy + 1 >= 0 && sy + 1 < sh;
  const long HW = (long)p.H * p.W;

  for (int f = 0; f < p.F; ++f) {
    const uint8_t* src = p.src + ((long)b * p.F + f) * p.Hs * p.Ws * 3;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v00 = (y0ok && x0ok) ? (float)src[(sy * p.Ws + sx) * 3 + k] : 0.f;
      const float v01 = (y0ok && x1ok) ? (float)src[(sy * p.Ws + sx + 1) * 3 + k] : 0.f;
      const float v10 = (y1ok && x0ok) ? (float)src[((sy + 1) * p.Ws + sx) * 3 + k] : 0.f;
      const float v11 = (y1ok && x1ok) ? (float)src[((sy + 1) * p.Ws + sx + 1) * 3 + k] : 0.f;
      c[k] = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
    }
    const long o = (((long)f * p.B + b) * 3) * HW + (long)y * p.W + xo;
    if (p.original) {
#pragma unroll
      for (int k = 0; k < 3; ++k) p.original[o + k * HW] = c[k] / 255.0f;        // Normalize(mean 0, std 1)
    }
    if (p.image) {
      colour_chain(c, ip, fp);
#pragma unroll
      for (int k = 0; k < 3; ++k) p.image[o + k * HW] = (c[k] / 255.0f - p.mean[k]) / p.std[k];
    }
  }
}

// Resize-based inputs.  Validation (configs/kitti_wpose_example:156-166) and, with a per-sample plan, the training
// input of the Resize-based configs (configs/multi_dataset_example:178-205: Resize of all frames + INTER_NEAREST
// patched_mask, colour Shuffle, RandomMirror, two Normalizes).  Resize = cv2.resize(float image, INTER_LINEAR) to
// rh x rw, zero-padded / cropped to H x W (augmentations.py:112-198), then Normalize and CHW.  OpenCV's float
// path (resize.cpp resizeGeneric_ / HResizeLinear / VResizeLinear): source coordinate (d + 0.5) * scale - 0.5 in
// f64, rounded to f32, floor + fraction; fraction 0 and index clamped at both borders; horizontal pass, then vertical.
__device__ __forceinline__ void resize_coord(int d, double scale, int n, int& s0, float& f) {
  float fx = (float)(((double)d + 0.5) * scale - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= n - 1) { fx = 0.f; sx = n - 1; }
  s0 = sx; f = fx;
}

__global__ __launch_bounds__(256) void resize_frames_kernel(const FsResizeArgs p) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.H * p.W) return;
  const int y = i / p.W, xo = i - y * p.W;
  const int32_t* d = p.dims + b * 4;
  const int sh = d[0], sw = d[1], rh = d[2], rw = d[3];
  const int32_t* ip = p.iplan ? p.iplan + b * FS_AUG_IPLAN : nullptr;
  const float* fp = p.fplan ? p.fplan + b * FS_AUG_FPLAN : nullptr;
  const int x = (ip && ip[4]) ? p.W - 1 - xo : xo;         // RandomMirror of the padded / cropped image
  const bool inside = y < rh && x < rw;                    // outside: np.pad zeros (they pass the colour ops too)
  int x0 = 0, y0 = 0; float fx = 0.f, fy = 0.f;
  if (inside) {
    resize_coord(x, 1.0 / ((double)rw / (double)sw), sw, x0, fx);
    resize_coord(y, 1.0 / ((double)rh / (double)sh), sh, y0, fy);
  }
  const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
  const long HW = (long)p.H * p.W;
  // patched_mask: ones through cv2.resize(INTER_NEAREST) are ones; the padding is zero
  if (p.mask) p.mask[((long)b * p.H + y) * p.W + xo] = inside ? 1.0 : 0.0;
  for (int f = 0; f < p.F; ++f) {
    const uint8_t* src = p.src + ((long)b * p.F + f) * p.Hs * p.Ws * 3;
    const long o = (((long)f * p.B + b) * 3) * HW + (long)y * p.W + xo;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = 0.f;
      if (inside) {
        const float v00 = (float)src[((long)y0 * p.Ws + x0) * 3 + k], v01 = (float)src[((long)y0 * p.Ws + x1) * 3 + k];
        const float v10 = (float)src[((long)y1 * p.Ws + x0) * 3 + k], v11 = (float)src[((long)y1 * p.Ws + x1) * 3 + k];
        const float top = v00 * (1.f - fx) + v01 * fx, bot = v10 * (1.f - fx) + v11 * fx;
        v = top * (1.f - fy) + bot * fy;
      }
      c[k] = v;
    }
    if (p.original) {
#pragma unroll
      for (int k = 0; k < 3; ++k) p.original[o + k * HW] = c[k] / 255.0f;
    }
    if (ip) colour_chain(c, ip, fp);
#pragma unroll
    for (int k = 0; k < 3; ++k) p.image[o + k * HW] = (c[k] / 255.0f - p.mean[k]) / p.std[k];
  }
}

}  // namespace

extern "C" int fs_resize_frames(const FsResizeArgs* a, void* stream) {
  if (!a || !a->src || !a->dims || !a->image) return FS_EINVAL;
  if ((a->iplan != nullptr) != (a->fplan != nullptr)) return FS_EINVAL;
  if (a->B < 1 || a->F < 1 || a->H < 1 || a->W < 1 || a->Hs < 1 || a->Ws < 1) return FS_EINVAL;
  if ((long)a->Hs * a->Ws * 3 > 0x7fffffffL) return FS_EINVAL;
  for (int k = 0; k < 3; ++k) if (a->std[k] == 0.f) return FS_EINVAL;
  dim3 grid((unsigned)(((long)a->H * a->W + 255) / 256), (unsigned)a->B);
  hipLaunchKernelGGL(resize_frames_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  return fs_launch_status();
}

extern "C" int fs_augment_frames(const FsAugArgs* a, void* stream) {
  if (!a || !a->src || !a->minv || !a->iplan || !a->fplan || (!a->image && !a->original)) return FS_EINVAL;
  if (a->B < 1 || a->F < 1 || a->H < 1 || a->W < 1 || a->Hs < 1 || a->Ws < 1) return FS_EINVAL;
  if ((long)a->Hs * a->Ws * 3 > 0x7fffffffL) return FS_EINVAL;
  for (int k = 0; k < 3; ++k) if (a->std[k] == 0.f) return FS_EINVAL;
  dim3 grid((unsigned)(((long)a->H * a->W + 255) / 256), (unsigned)a->B);
  hipLaunchKernelGGL(augment_frames_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
  return fs_launch_status();
}
