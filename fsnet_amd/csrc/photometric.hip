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

// ---------------------------------------------------------------------------------------------
// warp: pred[s][f][b] = grid_sample(src_f, project(depth_s)), overlap mask
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void photo_warp_kernel(const FsPhotoArgs p) {
  const int b = blockIdx.y, s = blockIdx.z >> 1, f = blockIdx.z & 1;
  const long HW = (long)p.H * p.W;
  const float* ge = p.geo + (long)b * GEO_STRIDE;
  const float* src = p.img_src[f] + (long)b * 3 * HW;
  float* pred = p.pred + (((long)s * 2 + f) * p.B + b) * 3 * HW;
  uint8_t* ov = p.ov + (((long)s * 2 + f) * p.B + b) * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    int y = (int)(i / p.W), x = (int)(i % p.W);
    Geo g;
    project_pixel(p, p.depth[s], b, y, x, p.H, p.W, p.dh[s], p.dw[s], ge, f, g);
    Taps t;
    bilinear_taps(g.ixu, g.iyu, p.H, p.W, t);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* sc = src + c * HW;
      float v00 = sc[(long)t.y0 * p.W + t.x0], v01 = sc[(long)t.y0 * p.W + t.x1];
      float v10 = sc[(long)t.y1 * p.W + t.x0], v11 = sc[(long)t.y1 * p.W + t.x1];
      pred[c * HW + i] = (1.f - t.wy) * ((1.f - t.wx) * v00 + t.wx * v01) + t.wy * ((1.f - t.wx) * v10 + t.wx * v11);
    }
    // nearest sample of patched_mask with zeros padding (round half to even, like ATen's nearbyint)
    float xn = nearbyintf(g.ixu), yn = nearbyintf(g.iyu);
    bool inb = xn >= 0.f && xn <= (float)(p.W - 1) && yn >= 0.f && yn <= (float)(p.H - 1);
    float mv = 0.f;
    if (inb) {
      const long o = (long)b * HW + (long)yn * p.W + (long)xn;
      // fisheye: patched_mask x ray-table mask as one float plane (monodepth2_decoder.py:409)
      mv = p.warp_mask ? p.warp_mask[o] : (p.patched_mask ? (float)p.patched_mask[o] : 1.f);
    }
    ov[i] = (p.no_overlap_mask || mv == 1.f) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// loss forward: per-pixel min over {identity_+, identity_-, reproj_+, reproj_-}, masked sum
// ---------------------------------------------------------------------------------------------
constexpr int TW = 32, TH = 8;
constexpr int R2W = TW + 4, R2H = TH + 4;   // pred / target region of the backward
constexpr int R1W = TW + 2, R1H = TH + 2;   // coefficient region (= stencil region of the forward)

// SSIM + L1 of one pixel against both warped frames from LDS planes [3][R1H][R1W]; (ly, lx) = position inside the
// region.  The nine target taps of a channel are read once and serve both frames (sums in the same order as the
// one-frame form: results are bit-identical); a frame whose sample fell outside the image (want[f] false) is skipped.
__device__ __forceinline__ void reproj_lds2(const float (*xs)[3][R1H][R1W], const float (*ts)[R1H][R1W], int ly, int lx,
                                            const bool (&want)[2], float (&out)[2]) {
  float ssim_sum[2] = {0.f, 0.f}, l1[2] = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float tv[9];
    float sy = 0, syy = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        float v = ts[c][ly - 1 + a][lx - 1 + bb];
        tv[a * 3 + bb] = v; sy += v; syy += v * v;
      }
    const float k = 1.f / 9.f;
    const float muy = sy * k;
    const float sgy = syy * k - muy * muy;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (!want[f]) continue;
      float sx = 0, sxx = 0, sxy = 0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
          float xv = xs[f][c][ly - 1 + a][lx - 1 + bb];
          sx += xv; sxx += xv * xv; sxy += xv * tv[a * 3 + bb];
        }
      float mux = sx * k;
      float sgx = sxx * k - mux * mux, sgxy = sxy * k - mux * muy;
      float n = (2.f * mux * muy + C1) * (2.f * sgxy + C2);
      float d = (mux * mux + muy * muy + C1) * (sgx + sgy + C2);
      ssim_sum[f] += fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
      l1[f] += fabsf(tv[4] - xs[f][c][ly][lx]);
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) out[f] = 0.85f * (ssim_sum[f] / 3.f) + 0.15f * (l1[f] / 3.f);
}

// LDS-tiled: a block owns 32x8 pixels of one (scale, batch) plane; target and both warped images of the
// tile (+1 halo, reflection resolved while loading) are staged once and every SSIM window reads LDS.
__global__ __launch_bounds__(256) void photo_loss_fwd_kernel(const FsPhotoArgs p) {
  __shared__ float s_t[3][R1H][R1W];
  __shared__ float s_x[2][3][R1H][R1W];
  __shared__ double sh[4];
  const int s = blockIdx.z / p.B, b = blockIdx.z % p.B;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  const int H = p.H, W = p.W, tid = threadIdx.x;
  const long HW = (long)H * W;
  const float* timg = p.img0 + (long)b * 3 * HW;
  const float* pr0 = p.pred + (((long)s * 2 + 0) * p.B + b) * 3 * HW;
  const float* pr1 = p.pred + (((long)s * 2 + 1) * p.B + b) * 3 * HW;
  for (int i = tid; i < R1H * R1W; i += 256) {
    int ry = i / R1W, rx = i - ry * R1W;
    int y = min(max(refl(ty0 - 1 + ry, H), 0), H - 1), x = min(max(refl(tx0 - 1 + rx, W), 0), W - 1);
    long o = (long)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_t[c][ry][rx] = timg[c * HW + o];
      s_x[0][c][ry][rx] = pr0[c * HW + o];
      s_x[1][c][ry][rx] = pr1[c * HW + o];
    }
  }
  __syncthreads();
  const int lx = tid % TW, ly = tid / TW;
  const int qx = tx0 + lx, qy = ty0 + ly;
  // device-resident seed (bumped once per step by fs_counter_incr) keeps the launch replayable from a hipGraph
  const int seed = p.noise_seed_ptr ? (*p.noise_seed_ptr & 0x3fffffff) : p.noise_seed;
  double acc = 0.0;
  if (qx < W && qy < H) {
    const long i = (long)qy * W + qx;
    float best = 0.f; int bi = 0;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      uint32_t key = (uint32_t)((((long)s * 2 + f) * p.B + b) * HW + i);
      float v = p.ident[((long)b * 2 + f) * HW + i] + tie_noise(seed, key);
      if (f == 0 || v < best) { best = v; bi = f; }
    }
    bool want[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) want[f] = p.ov[(((long)s * 2 + f) * p.B + b) * HW + i] != 0;
    float rv[2];
    reproj_lds2(s_x, s_t, ly + 1, lx + 1, want, rv);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float v = want[f] ? rv[f] : 100.f;
      if (v < best) { best = v; bi = 2 + f; }
    }
    p.sel[((long)s * p.B + b) * HW + i] = (uint8_t)bi;
    double pm = p.patched_mask ? p.patched_mask[(long)b * HW + i] : 1.0;
    acc = (double)best * pm;
  }
  acc = block_sum_d(acc, sh);
  if (tid == 0) atomicAdd(p.loss_sums + s * p.B + b, acc);
}

// ---------------------------------------------------------------------------------------------
// loss backward, LDS tiled: tile 32x8 pixels q, one thread per pixel; coefficients of the window centres p in
// tile(+)1 from pred/target in tile(+)2.  A window centre contributes to exactly one source frame — the one its
// per-pixel minimum selected (sel == 2 + f) — so its nine coefficients are computed and stored once, with the frame
// index beside them, and the gather at q sorts them into the two frames' accumulators.  (The first version ran the
// two frames on separate thread halves with a coefficient plane each: twice the SSIM-derivative arithmetic and
// twice the LDS gather traffic, half of it zeros by construction.)  Sums are formed in the same order as before.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int refl_mult(int pc, int qc, int n) {
  // how many taps delta in {-1,0,1} of window centre pc land (after reflection) on pixel qc
  int m = 0;
#pragma unroll
  for (int d = -1; d <= 1; ++d) m += (refl(pc + d, n) == qc) ? 1 : 0;
  return m;
}

__global__ __launch_bounds__(256) void photo_loss_bwd_kernel(const FsPhotoArgs p) {
  __shared__ float s_t[3][R2H][R2W];
  __shared__ float s_x[2][3][R2H][R2W];
  __shared__ float s_coef[9][R1H][R1W];      // c*3 + {A,B,C} of the frame selected at p
  __shared__ int s_fr[R1H][R1W];             // that frame (0 / 1), -1: identity selected, masked out or outside
  __shared__ float s_red[2][12][4];
  __shared__ float s_dd[(TH + 2) * (TW + 2)];
  const int s = blockIdx.z / p.B, b = blockIdx.z % p.B;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  const int H = p.H, W = p.W;
  const long HW = (long)H * W;
  const int tid = threadIdx.x;
  const float* timg = p.img0 + (long)b * 3 * HW;
  const uint8_t* sel = p.sel + ((long)s * p.B + b) * HW;
  const float* ge = p.geo + (long)b * GEO_STRIDE;
  const double gout = p.gout ? *p.gout : 1.0;
  double msum = 0.0;
  for (int k = 0; k < p.B; ++k) msum += p.mask_sum[k];
  const float gscale = (float)(gout / ((double)p.S * (msum + 1e-6)));
  const float* pred0 = p.pred + (((long)s * 2 + 0) * p.B + b) * 3 * HW;
  const float* pred1 = p.pred + (((long)s * 2 + 1) * p.B + b) * 3 * HW;
  // tiles whose 2-pixel halo lies inside the image need no reflection bookkeeping (the common case)
  const bool interior = ty0 >= 2 && tx0 >= 2 && ty0 + TH + 2 <= H && tx0 + TW + 2 <= W;

  for (int i = tid; i < R2H * R2W; i += 256) {
    int ry = i / R2W, rx = i - ry * R2W;
    int y = ty0 - 2 + ry, x = tx0 - 2 + rx;
    bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    long o = (long)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_t[c][ry][rx] = in ? timg[c * HW + o] : 0.f;
      s_x[0][c][ry][rx] = in ? pred0[c * HW + o] : 0.f;
      s_x[1][c][ry][rx] = in ? pred1[c * HW + o] : 0.f;
    }
  }
  for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) s_dd[i] = 0.f;
  __syncthreads();

  // ---- coefficients at p in tile(+)1, for the frame p selected ----
  for (int i = tid; i < R1H * R1W; i += 256) {
    int ry = i / R1W, rx = i - ry * R1W;
    int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
    float wgt = 0.f;
    int f = -1;
    bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    if (in) {
      int sv = sel[(long)y * W + x];
      if (sv >= 2) {
        float pm = p.patched_mask ? (float)p.patched_mask[(long)b * HW + (long)y * W + x] : 1.f;
        wgt = pm * gscale * (0.85f / 3.f);
        f = sv - 2;
      }
    }
    if (wgt == 0.f) { s_fr[ry][rx] = -1; continue; }
    s_fr[ry][rx] = f;
    int ys[3] = {ry, ry + 1, ry + 2}, xs[3] = {rx, rx + 1, rx + 2};   // interior: window = R2 rows ry..ry+2
    if (!interior) {
      ys[0] = refl(y - 1, H) - (ty0 - 2); ys[2] = refl(y + 1, H) - (ty0 - 2);
      xs[0] = refl(x - 1, W) - (tx0 - 2); xs[2] = refl(x + 1, W) - (tx0 - 2);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
      if (interior) {
        const float* xb = &s_x[f][c][ry][rx];
        const float* tb = &s_t[c][ry][rx];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) {
            float xv = xb[a * R2W + bb], tv = tb[a * R2W + bb];
            sx += xv; sy += tv; sxx += xv * xv; syy += tv * tv; sxy += xv * tv;
          }
      } else {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) {
            float xv = s_x[f][c][ys[a]][xs[bb]], tv = s_t[c][ys[a]][xs[bb]];
            sx += xv; sy += tv; sxx += xv * xv; syy += tv * tv; sxy += xv * tv;
          }
      }
      const float k9 = 1.f / 9.f;
      float mux = sx * k9, muy = sy * k9;
      float sgx = sxx * k9 - mux * mux, sgy = syy * k9 - muy * muy, sgxy = sxy * k9 - mux * muy;
      float n1 = 2.f * mux * muy + C1, n2 = 2.f * sgxy + C2;
      float d1 = mux * mux + muy * muy + C1, d2 = sgx + sgy + C2;
      float n = n1 * n2, d = d1 * d2;
      float sv = (1.f - n / d) * 0.5f;
      float A = 0.f, Bc = 0.f, Cc = 0.f;
      if (sv >= 0.f && sv <= 1.f) {
        // d n / d x(q) = a1 + a2 (t(q) - muy),  d d / d x(q) = b1 + b2 (x(q) - mux)   (each tap weight 1/9)
        float a1 = 2.f * muy * n2 * k9, a2 = 2.f * n1 * k9;
        float b1 = 2.f * mux * d2 * k9, b2 = 2.f * d1 * k9;
        float h = -0.5f / (d * d);
        A = h * ((a1 - a2 * muy) * d - n * (b1 - b2 * mux));
        Bc = -h * n * b2;
        Cc = h * a2 * d;
      }
      s_coef[c * 3 + 0][ry][rx] = wgt * A;
      s_coef[c * 3 + 1][ry][rx] = wgt * Bc;
      s_coef[c * 3 + 2][ry][rx] = wgt * Cc;
    }
  }
  __syncthreads();

  // ---- gather d loss / d pred_f(q) for both frames, chain through the sampler and the projection ----
  const int qx = tx0 + (tid % TW), qy = ty0 + (tid / TW);
  const bool qin = qx < W && qy < H;
  const int ly = tid / TW + 2, lx = tid % TW + 2;  // q inside the R2 arrays
  float dpred[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  if (qin) {
    if (interior) {
      // every window centre in [q-1, q+1]^2 contributes exactly once
      const int cy = tid / TW, cx = tid % TW;     // coefficient-region row/col of (qy-1, qx-1)
      float acc[2][9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { acc[0][k] = 0.f; acc[1][k] = 0.f; }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
          const int fr = s_fr[cy + a][cx + bb];
          if (fr < 0) continue;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const float cv = s_coef[k][cy + a][cx + bb];
            acc[0][k] += fr == 0 ? cv : 0.f;
            acc[1][k] += fr == 1 ? cv : 0.f;
          }
        }
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          dpred[f][c] = acc[f][c * 3] + acc[f][c * 3 + 1] * s_x[f][c][ly][lx] + acc[f][c * 3 + 2] * s_t[c][ly][lx];
    } else {
      for (int py = qy - 1; py <= qy + 1; ++py) {
        if ((unsigned)py >= (unsigned)H) continue;
        int my = refl_mult(py, qy, H);
        if (!my) continue;
        for (int px = qx - 1; px <= qx + 1; ++px) {
          if ((unsigned)px >= (unsigned)W) continue;
          int mx = refl_mult(px, qx, W);
          if (!mx) continue;
          float mult = (float)(my * mx);
          int ry = py - (ty0 - 1), rx = px - (tx0 - 1);
          const int fr = s_fr[ry][rx];
          if (fr < 0) continue;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float v = mult * (s_coef[c * 3][ry][rx] + s_coef[c * 3 + 1][ry][rx] * s_x[fr][c][ly][lx] +
                                    s_coef[c * 3 + 2][ry][rx] * s_t[c][ly][lx]);
            dpred[0][c] += fr == 0 ? v : 0.f;
            dpred[1][c] += fr == 1 ? v : 0.f;
          }
        }
      }
    }
    const int sq = sel[(long)qy * W + qx];
    if (sq >= 2) {
      const int f = sq - 2;
      float pm = p.patched_mask ? (float)p.patched_mask[(long)b * HW + (long)qy * W + qx] : 1.f;
      float wl1 = pm * gscale * (0.15f / 3.f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float df = s_x[f][c][ly][lx] - s_t[c][ly][lx];
        float g1 = wl1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
        dpred[0][c] += f == 0 ? g1 : 0.f;
        dpred[1][c] += f == 1 ? g1 : 0.f;
      }
    }
  }
  float dDf[2] = {0.f, 0.f};
  const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    float dP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) dP[k] = 0.f;
    if (qin && (dpred[f][0] != 0.f || dpred[f][1] != 0.f || dpred[f][2] != 0.f)) {
      Geo gq;
      project_pixel(p, p.depth[s], b, qy, qx, H, W, p.dh[s], p.dw[s], ge, f, gq);
      Taps t;
      bilinear_taps(gq.ixu, gq.iyu, H, W, t);
      const float* src = p.img_src[f] + (long)b * 3 * HW;
      float gix = 0.f, giy = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* sc = src + c * HW;
        float v00 = sc[(long)t.y0 * W + t.x0], v01 = sc[(long)t.y0 * W + t.x1];
        float v10 = sc[(long)t.y1 * W + t.x0], v11 = sc[(long)t.y1 * W + t.x1];
        gix += dpred[f][c] * ((v01 - v00) * (1.f - t.wy) + (v11 - v10) * t.wy);
        giy += dpred[f][c] * ((v10 - v00) * (1.f - t.wx) + (v11 - v01) * t.wx);
      }
      float du = gix * t.mx, dv = giy * t.my;   // (W-1)/2 of the sampler cancels 2/(W-1) of Project3D
      float dX, dY, dZ;
      if (p.lut_ptrs) {
        float dq[3];
        mei_cam2image_bwd(p.mei + (long)b * 8, gq.X, gq.Y, gq.Zp, du, dv, dq);
        dX = dq[0]; dY = dq[1]; dZ = dq[2];
      } else {
        float iz = 1.f / gq.Zp;
        dX = du * iz; dY = dv * iz;
        dZ = -(du * gq.X + dv * gq.Y) * iz * iz;
      }
      const float* P = ge + 18 + f * 12;
      float pr0 = P[0] * gq.r[0] + P[1] * gq.r[1] + P[2] * gq.r[2];
      float pr1 = P[4] * gq.r[0] + P[5] * gq.r[1] + P[6] * gq.r[2];
      float pr2 = P[8] * gq.r[0] + P[9] * gq.r[1] + P[10] * gq.r[2];
      dDf[f] = dX * pr0 + dY * pr1 + dZ * pr2;
      float cam[3] = {gq.D * gq.r[0], gq.D * gq.r[1], gq.D * gq.r[2]};
      float dxyz[3] = {dX, dY, dZ};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) dP[i * 4 + j] = dxyz[i] * cam[j];
        dP[i * 4 + 3] = dxyz[i];
      }
    }
    // the 12 projection-matrix partials of this frame: wave sums -> LDS
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      float v = wave_sum(dP[k]);
      if ((tid & 63) == 0) s_red[f][k][tid >> 6] = v;
    }
  }
  __syncthreads();
  if (tid < 24) {
    // per-block partial, reduced by photo_pose_grad_kernel: 23040 blocks adding into 24 cache lines cost ~90 us of
    // serialised L2 atomics at the bench shape (and made dT depend on the block schedule)
    const int f = tid / 12, k = tid % 12;
    p.dP[(blk * 2 + f) * 12 + k] = s_red[f][k][0] + s_red[f][k][1] + s_red[f][k][2] + s_red[f][k][3];
  }
  // ---- transpose of the bilinear depth upsample: accumulate the tile's contributions in LDS first, then
  //      one global atomic per touched low-res pixel (scale-3 maps receive 256 full-res pixels each) ----
  const int h = p.dh[s], w = p.dw[s];
  Geo g0;
  upsample_taps(ty0, tx0, H, W, h, w, g0);
  const int by = g0.y0, bx = g0.x0;
  {
    float dD_total = dDf[0] + dDf[1];
    if (qin && dD_total != 0.f) {
      Geo g;
      upsample_taps(qy, qx, H, W, h, w, g);
      float w00 = (1.f - g.ly) * (1.f - g.lx), w01 = (1.f - g.ly) * g.lx, w10 = g.ly * (1.f - g.lx), w11 = g.ly * g.lx;
      int ly0 = g.y0 - by, ly1 = g.y1 - by, lx0 = g.x0 - bx, lx1 = g.x1 - bx;
      if (w00 != 0.f) atomicAdd(&s_dd[ly0 * (TW + 2) + lx0], w00 * dD_total);
      if (w01 != 0.f) atomicAdd(&s_dd[ly0 * (TW + 2) + lx1], w01 * dD_total);
      if (w10 != 0.f) atomicAdd(&s_dd[ly1 * (TW + 2) + lx0], w10 * dD_total);
      if (w11 != 0.f) atomicAdd(&s_dd[ly1 * (TW + 2) + lx1], w11 * dD_total);
    }
  }
  __syncthreads();
  float* dd = p.d_depth[s] + (long)b * h * w;
  for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
    float v = s_dd[i];
    if (v != 0.f) {
      int yy = by + i / (TW + 2), xx = bx + i % (TW + 2);
      if (yy < h && xx < w) atomicAdd(dd + yy * w + xx, v);
    }
  }
}

// dT[f][b] (4x4, row 3 = 0) = K^T-contracted dP:  P = K3 * T[:3]  =>  dT[k][j] = sum_i K3[i][k] dP[i][j].
// One block per (b, f): sums the per-tile partials dP[s][b][tile][f][12] of photo_loss_bwd_kernel in a fixed order.
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

extern "C" int fs_photo_warp(const FsPhotoArgs* a, void* stream) {
  if (!valid(a) || !a->pred || !a->ov) return FS_EINVAL;
  for (int s = 0; s < a->S; ++s) if (!a->depth[s]) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long HW = (long)a->H * a->W;
  dim3 grid((unsigned)std::min<long>((HW + 255) / 256, 2048), a->B, a->S * 2);
  hipLaunchKernelGGL(photo_warp_kernel, grid, dim3(256), 0, st, *a);
  return fs_launch_status();
}

extern "C" int fs_photo_loss_fwd(const FsPhotoArgs* a, void* stream) {
  if (!valid(a) || !a->pred || !a->ov || !a->ident || !a->sel || !a->loss_sums) return FS_EINVAL;
  if (a->motion_mask) return FS_EINVAL;                 // only the fused kernels know the motion mask
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long HW = (long)a->H * a->W;
  dim3 grid((a->W + TW - 1) / TW, (a->H + TH - 1) / TH, a->S * a->B);
  hipLaunchKernelGGL(photo_loss_fwd_kernel, grid, dim3(256), 0, st, *a);
  return fs_launch_status();
}

extern "C" int fs_photo_loss_bwd(const FsPhotoArgs* a, void* stream) {
  if (!valid(a) || !a->pred || !a->sel || !a->dP || !a->mask_sum || a->motion_mask) return FS_EINVAL;
  for (int s = 0; s < a->S; ++s) if (!a->depth[s] || !a->d_depth[s]) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((a->W + TW - 1) / TW, (a->H + TH - 1) / TH, a->S * a->B);
  hipLaunchKernelGGL(photo_loss_bwd_kernel, grid, dim3(256), 0, st, *a);
  return fs_launch_status();
}

extern "C" int64_t fs_photo_bwd_tiles(int H, int W) {
  if (H < 2 || W < 2) return -1;
  return (int64_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
}

extern "C" int fs_photo_pose_grad(const float* geo, const float* dP, float* dT0, float* dT1, int B, int S, int tiles,
                                  void* stream) {
  if (!geo || !dP || !dT0 || !dT1 || B < 1 || S < 1 || tiles < 1) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(photo_pose_grad_kernel, dim3(2 * B), dim3(256), 0, st, geo, dP, dT0, dT1, B, S, tiles);
  return fs_launch_status();
}
