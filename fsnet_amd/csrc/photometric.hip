// Photometric loss chain, forward + backward, as HBM-bound gather / stencil kernels.
// Replaces (reference, all eager ATen):
//   MonoDepth2Decoder._generate_images_pred          monodepth2_decoder.py:61-116
//     F.interpolate(bilinear, align_corners=True)     :68-69
//     K / pinv(K) on host                             :82-85
//     BackprojectDepth / Project3D                    monodepth_utils.py:101-165
//     F.grid_sample(bilinear, border) / (nearest)     monodepth2_decoder.py:98-101,110-116
//   compute_reprojection_loss (SSIM + L1)             monodepth2_decoder.py:118-128, monodepth_utils.py:184-215
//   compute_total_reprojection_loss (min / masks)     monodepth2_decoder.py:205-292
// All four scales run in ONE launch per stage (every stage works at full resolution; only the
// low-res depth map differs).  Images stay planar NCHW fp32 exactly as the data layer hands them over.
#include "photo_common.h"
#include <algorithm>

namespace {

// ---------------------------------------------------------------------------------------------
// setup: K, K^-1 (f64 adjugate == pinv for a regular K), P_f = (K T_f)[:3]; zero the accumulators
// ---------------------------------------------------------------------------------------------
__global__ void photo_setup_kernel(const float* __restrict__ P2, const float* __restrict__ T0,
                                   const float* __restrict__ T1, float* __restrict__ geo, int B,
                                   int* __restrict__ seed_counter, int fisheye) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (seed_counter && b == 0) *seed_counter += 1;      // tie-break noise seed of this step (read by the loss forward)
  if (b >= B) return;
  double k[3][3];
  // fisheye: the transform acts on the ray-table point directly (monodepth2_decoder.py:379-381), the camera model
  // follows it (cam2image) -> "K" is the identity here, P_f = T_f[:3], and fs_photo_pose_grad's K^T dP is dT itself
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) k[i][j] = fisheye ? (i == j ? 1.0 : 0.0) : (double)P2[b * 12 + i * 4 + j];
  double c00 = k[1][1] * k[2][2] - k[1][2] * k[2][1];
  double c01 = k[1][2] * k[2][0] - k[1][0] * k[2][2];
  double c02 = k[1][0] * k[2][1] - k[1][1] * k[2][0];
  double det = k[0][0] * c00 + k[0][1] * c01 + k[0][2] * c02;
  double inv[3][3];
  inv[0][0] = c00 / det; inv[1][0] = c01 / det; inv[2][0] = c02 / det;
  inv[0][1] = (k[0][2] * k[2][1] - k[0][1] * k[2][2]) / det;
  inv[1][1] = (k[0][0] * k[2][2] - k[0][2] * k[2][0]) / det;
  inv[2][1] = (k[0][1] * k[2][0] - k[0][0] * k[2][1]) / det;
  inv[0][2] = (k[0][1] * k[1][2] - k[0][2] * k[1][1]) / det;
  inv[1][2] = (k[0][2] * k[1][0] - k[0][0] * k[1][2]) / det;
  inv[2][2] = (k[0][0] * k[1][1] - k[0][1] * k[1][0]) / det;
  float* g = geo + (long)b * GEO_STRIDE;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { g[i * 3 + j] = (float)inv[i][j]; g[9 + i * 3 + j] = (float)k[i][j]; }
  for (int f = 0; f < 2; ++f) {
    const float* T = (f == 0 ? T0 : T1) + (long)b * 16;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) {
        float a = 0.f;
        for (int m = 0; m < 3; ++m) a += g[9 + i * 3 + m] * T[m * 4 + j];
        g[18 + f * 12 + i * 4 + j] = a;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// identity reprojection losses (scale independent) + patched-mask sum
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float reproj_at(const float* __restrict__ xp, const float* __restrict__ tp, int y, int x,
                                           int H, int W) {
  // xp, tp: planar [3][H][W] of one batch element; returns 0.85*mean_c SSIM + 0.15*mean_c |t - x|
  int ys[3] = {refl(y - 1, H), y, refl(y + 1, H)}, xs[3] = {refl(x - 1, W), x, refl(x + 1, W)};
  float ssim_sum = 0.f, l1 = 0.f;
  const long HW = (long)H * W;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float xv = xp[c * HW + (long)ys[a] * W + xs[b]], tv = tp[c * HW + (long)ys[a] * W + xs[b]];
        sx += xv; sy += tv; sxx += xv * xv; syy += tv * tv; sxy += xv * tv;
      }
    const float k = 1.f / 9.f;
    float mux = sx * k, muy = sy * k;
    float sgx = sxx * k - mux * mux, sgy = syy * k - muy * muy, sgxy = sxy * k - mux * muy;
    float n = (2.f * mux * muy + C1) * (2.f * sgxy + C2);
    float d = (mux * mux + muy * muy + C1) * (sgx + sgy + C2);
    ssim_sum += fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
    l1 += fabsf(tp[c * HW + (long)y * W + x] - xp[c * HW + (long)y * W + x]);
  }
  return 0.85f * (ssim_sum / 3.f) + 0.15f * (l1 / 3.f);
}

// One block = a 64 x 4 pixel tile of one sample: the nine planes (target + two source frames, three channels each)
// are staged ONCE with their reflected one-pixel halo into LDS and both identity terms read their 3x3 windows from
// there, in the same order and with the same arithmetic as reproj_at() above (bit-identical results).  The first
// version gathered 108 values per pixel from global memory (the target window twice): 154 us per step — as long as
// the whole fused forward of all four scales — for 65 MB of input.
constexpr int ID_TW = 64, ID_TH = 4, ID_HW = ID_TW + 2, ID_HH = ID_TH + 2;

__global__ __launch_bounds__(256) void photo_ident_kernel(const FsPhotoArgs p) {
  __shared__ float tile[9][ID_HH][ID_HW];
  const int b = blockIdx.y;
  const int tiles_x = (p.W + ID_TW - 1) / ID_TW;
  const int ty_i = blockIdx.x / tiles_x, tx_i = blockIdx.x - ty_i * tiles_x;
  const int y0 = ty_i * ID_TH, x0 = tx_i * ID_TW;
  const long HW = (long)p.H * p.W;
  const int t = threadIdx.x;
  for (int e = t; e < 9 * ID_HH * ID_HW; e += 256) {
    const int pl = e / (ID_HH * ID_HW), r = e - pl * (ID_HH * ID_HW);
    const int hy = r / ID_HW, hx = r - hy * ID_HW;
    // (rows / columns past the image only feed pixels that are not written; clamp keeps the address valid)
    const int yy = refl(min(y0 + hy - 1, p.H), p.H), xx = refl(min(x0 + hx - 1, p.W), p.W);
    const float* src = pl < 3 ? p.img0 : p.img_src[(pl - 3) / 3];
    tile[pl][hy][hx] = src[((long)b * 3 + pl % 3) * HW + (long)yy * p.W + xx];
  }
  __syncthreads();
  const int tx = t & (ID_TW - 1), ty = t / ID_TW;
  const int y = y0 + ty, x = x0 + tx;
  double msum = 0.0;
  if (y < p.H && x < p.W) {
    const long i = (long)y * p.W + x;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float ssim_sum = 0.f, l1 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const float xv = tile[3 + 3 * f + c][ty + a][tx + q], tv = tile[c][ty + a][tx + q];
            sx += xv; sy += tv; sxx += xv * xv; syy += tv * tv; sxy += xv * tv;
          }
        const float k = 1.f / 9.f;
        float mux = sx * k, muy = sy * k;
        float sgx = sxx * k - mux * mux, sgy = syy * k - muy * muy, sgxy = sxy * k - mux * muy;
        float n = (2.f * mux * muy + C1) * (2.f * sgxy + C2);
        float d = (mux * mux + muy * muy + C1) * (sgx + sgy + C2);
        ssim_sum += fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
        l1 += fabsf(tile[c][ty + 1][tx + 1] - tile[3 + 3 * f + c][ty + 1][tx + 1]);
      }
      p.ident[((long)b * 2 + f) * HW + i] = 0.85f * (ssim_sum / 3.f) + 0.15f * (l1 / 3.f);
    }
    msum = p.patched_mask ? p.patched_mask[(long)b * HW + i] : 1.0;
  }
  __shared__ double sh[4];
  msum = block_sum_d(msum, sh);
  if (threadIdx.x == 0) atomicAdd(p.mask_sum + b, msum);
}

// (Rounds 1-2 ran the chain as three staged kernels here — warp to HBM, loss forward, loss backward; the fused kernels of
// photo_fused.hip replaced them in round 3 and the staged path, kept behind a switch until round 5, is gone.)

// dT[f][b] (4x4, row 3 = 0) = K^T-contracted dP:  P = K3 * T[:3]  =>  dT[k][j] = sum_i K3[i][k] dP[i][j].
// One block per (b, f): sums the per-tile partials dP[s][b][tile][f][12] of photo_fused_bwd_kernel in a fixed order.
// A partial is three 16-byte lanes; thread (row lane r of 85, quarter q of 3) strides over the S * tiles rows, the
// 85 lanes are then added in f64 (first version: 4-byte loads, 366 dependent-latency iterations per thread — 33 us on
// the critical path between the loss backward and the pose chain's backward).
__global__ __launch_bounds__(256) void photo_pose_grad_kernel(const float* __restrict__ geo,
                                                              const float* __restrict__ dP, float* __restrict__ dT0,
                                                              float* __restrict__ dT1, int B, int S, int tiles) {
  constexpr int RL = 85;
  __shared__ double s_part[RL][12];
  __shared__ float s_g[12];
  const int b = blockIdx.x >> 1, f = blockIdx.x & 1;
  const int t = threadIdx.x, q = t % 3, r = t / 3;
  if (r < RL) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int rows = S * tiles;
#pragma unroll 4
    for (int j = r; j < rows; j += RL) {
      const int s = j / tiles, tl = j - s * tiles;
      const float4 v = *reinterpret_cast<const float4*>(dP + ((((long)s * B + b) * tiles + tl) * 2 + f) * 12 + q * 4);
      a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    s_part[r][q * 4 + 0] = a0; s_part[r][q * 4 + 1] = a1; s_part[r][q * 4 + 2] = a2; s_part[r][q * 4 + 3] = a3;
  }
  __syncthreads();
  if (t < 12) {
    double v = 0.0;
    for (int g = 0; g < RL; ++g) v += s_part[g][t];
    s_g[t] = (float)v;
  }
  __syncthreads();
  if (t < 16) {
    const float* K = geo + (long)b * GEO_STRIDE + 9;
    float* o = (f == 0 ? dT0 : dT1) + (long)b * 16;
    const int rr = t >> 2, j = t & 3;
    o[t] = rr < 3 ? K[0 * 3 + rr] * s_g[0 * 4 + j] + K[1 * 3 + rr] * s_g[1 * 4 + j] + K[2 * 3 + rr] * s_g[2 * 4 + j] : 0.f;
  }
}

bool valid(const FsPhotoArgs* a) {
  if (!a || !a->img0 || !a->img_src[0] || !a->img_src[1] || !a->geo) return false;
  if (a->S < 1 || a->S > 4 || a->B < 1 || a->H < 2 || a->W < 2) return false;
  if ((a->lut_ptrs != nullptr) != (a->mei != nullptr)) return false;   // the ray tables come with their parameters
  return true;
}

}  // namespace

extern "C" int fs_photo_setup(const float* P2, const float* T0, const float* T1, float* geo, int B, int* seed_counter,
                              int fisheye, void* stream) {
  if (!P2 || !T0 || !T1 || !geo) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(photo_setup_kernel, dim3((B + 63) / 64), dim3(64), 0, st, P2, T0, T1, geo, B, seed_counter,
                     fisheye);
  return fs_launch_status();
}

extern "C" int fs_photo_identity(const FsPhotoArgs* a, void* stream) {
  if (!valid(a) || !a->ident || !a->mask_sum) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)(((a->W + ID_TW - 1) / ID_TW) * ((a->H + ID_TH - 1) / ID_TH)), a->B);
  hipLaunchKernelGGL(photo_ident_kernel, grid, dim3(256), 0, st, *a);
  return fs_launch_status();
}

extern "C" int fs_photo_pose_grad(const float* geo, const float* dP, float* dT0, float* dT1, int B, int S, int tiles,
                                  void* stream) {
  if (!geo || !dP || !dT0 || !dT1 || B < 1 || S < 1 || tiles < 1) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(photo_pose_grad_kernel, dim3(2 * B), dim3(256), 0, st, geo, dP, dT0, dT1, B, S, tiles);
  return fs_launch_status();
}
