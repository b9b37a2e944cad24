// Depth-bin head (clamp -> softmax -> expectation over log-spaced bins -> depth, disp) and the
// pose tail (spatial mean x0.01 -> axis-angle/translation -> 4x4 transform), forward + backward.
// Replaces:
//   MultiChannelDepthDecoder._gather_activation / gather_output   depth_encoder.py:76-88,114-121
//   depth_to_disp                                                 monodepth_utils.py:19-24
//   PoseDecoder tail (mean(3).mean(2), 0.01*view, slicing)        pose_decoder.py:39-45
//   transformation_from_parameters / rot_from_axisangle           monodepth_utils.py:31-63,298-337
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

// ---------------------------------------------------------------------------------------------
// depth head: one lane per pixel, K logits (fp32) contiguous per pixel
// ---------------------------------------------------------------------------------------------
// sc: per-image depth scale P2[0,0] / base_fx (DepthDecoder._get_scale, depth_encoder.py:36-43) or 0 = none; with a
// scale the depth is the expectation x sc and the disparity is taken against (min x sc, max x sc) (gather_output
// :115-121)
template <int K>
__device__ __forceinline__ void head_fwd_row(const float* __restrict__ logits, const float* sb, float* __restrict__ depth,
                                             float* __restrict__ disp, long m, int Cl, float inv_rng, float max_d,
                                             float sc = 0.f, float min_d = 0.f) {
  float v[K];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; k += 4) {
    float4 t = *reinterpret_cast<const float4*>(logits + m * Cl + k);
    v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) { v[k] = fminf(fmaxf(v[k], -10.f), 10.f); mx = fmaxf(mx, v[k]); }
  float se = 0.f, sd = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { float e = expf(v[k] - mx); se += e; sd += e * sb[k]; }
  float d = sd / se;
  if (sc != 0.f) {
    d = d * sc;
    const float mn = min_d * sc, mx2 = max_d * sc;
    depth[m] = d;
    disp[m] = (1.f / d - 1.f / mx2) / (1.f / mn - 1.f / mx2);
    return;
  }
  depth[m] = d;
  disp[m] = (1.f / d - 1.f / max_d) * inv_rng;
}

template <int K, typename T>
__device__ __forceinline__ void head_bwd_row(const float* __restrict__ logits, const float* sb,
                                             const float* __restrict__ d_depth, const float* __restrict__ d_disp,
                                             T* __restrict__ dlogits, long m, int Cl, float inv_rng,
                                             float sc = 0.f, float min_d = 0.f, float max_d = 0.f) {
  float v[K], raw[K];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; k += 4) {
    float4 t = *reinterpret_cast<const float4*>(logits + m * Cl + k);
    raw[k] = t.x; raw[k + 1] = t.y; raw[k + 2] = t.z; raw[k + 3] = t.w;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) { v[k] = fminf(fmaxf(raw[k], -10.f), 10.f); mx = fmaxf(mx, v[k]); }
  float se = 0.f, sd = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { v[k] = expf(v[k] - mx); se += v[k]; sd += v[k] * sb[k]; }
  float d = sd / se;
  float g = (d_depth ? d_depth[m] : 0.f);
  if (sc != 0.f) {
    // depth = d sc, disp = (1/depth - 1/(max sc)) / (1/(min sc) - 1/(max sc))
    const float ds = d * sc;
    if (d_disp) g += d_disp[m] * (-1.f / (ds * ds)) / (1.f / (min_d * sc) - 1.f / (max_d * sc));
    g *= sc;
  } else if (d_disp) g += d_disp[m] * (-inv_rng / (d * d));
  float o[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float pk = v[k] / se;
    bool inside = raw[k] >= -10.f && raw[k] <= 10.f;  // clamp passes gradient on the closed interval
    o[k] = inside ? pk * (sb[k] - d) * g : 0.f;
  }
#pragma unroll
  for (int k = 0; k < K; k += 4) store4<T>(dlogits + m * Cl + k, &o[k]);
}

template <int K>
__global__ __launch_bounds__(256) void depth_head_fwd_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ bins, float* __restrict__ depth,
                                                             float* __restrict__ disp, long M, int Cl, float min_d,
                                                             float max_d) {
  __shared__ float sb[K];
  if (threadIdx.x < K) sb[threadIdx.x] = bins[threadIdx.x];
  __syncthreads();
  const float inv_rng = 1.f / (1.f / min_d - 1.f / max_d);
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256)
    head_fwd_row<K>(logits, sb, depth, disp, m, Cl, inv_rng, max_d);
}

template <int K, typename T>
__global__ __launch_bounds__(256) void depth_head_bwd_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ bins,
                                                             const float* __restrict__ d_depth,
                                                             const float* __restrict__ d_disp, T* __restrict__ dlogits,
                                                             long M, int Cl, float min_d, float max_d) {
  __shared__ float sb[K];
  if (threadIdx.x < K) sb[threadIdx.x] = bins[threadIdx.x];
  __syncthreads();
  const float inv_rng = 1.f / (1.f / min_d - 1.f / max_d);
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256)
    head_bwd_row<K, T>(logits, sb, d_depth, d_disp, dlogits, m, Cl, inv_rng);
}

// all scales of the decoder in ONE launch (the per-scale launches are 10-15 us latency-bound nodes on the serial
// part of the step: four after the decoder's forward, four at the head of its backward)
struct HeadBlocks { int start[FS_HEAD_MAX + 1]; };

template <int K>
__global__ __launch_bounds__(256) void depth_head_fwd_multi_kernel(const FsHeadBatch hb, const HeadBlocks blk,
                                                                   const float* __restrict__ bins, int Cl, float min_d,
                                                                   float max_d) {
  __shared__ float sb[K];
  if (threadIdx.x < K) sb[threadIdx.x] = bins[threadIdx.x];
  __syncthreads();
  const float inv_rng = 1.f / (1.f / min_d - 1.f / max_d);
  int s = 0;
  while (s + 1 < hb.n && (int)blockIdx.x >= blk.start[s + 1]) ++s;
  const long nb = blk.start[s + 1] - blk.start[s], lb = blockIdx.x - blk.start[s];
  const long rows_img = hb.nimg > 0 ? hb.M[s] / hb.nimg : 0;
  for (long m = lb * 256 + threadIdx.x; m < hb.M[s]; m += nb * 256) {
    const float sc = hb.P2 ? hb.P2[(m / rows_img) * 12] / hb.base_fx : 0.f;
    head_fwd_row<K>(hb.logits[s], sb, hb.depth[s], hb.disp[s], m, Cl, inv_rng, max_d, sc, min_d);
  }
}

template <int K, typename T>
__global__ __launch_bounds__(256) void depth_head_bwd_multi_kernel(const FsHeadBatch hb, const HeadBlocks blk,
                                                                   const float* __restrict__ bins, int Cl, float min_d,
                                                                   float max_d) {
  __shared__ float sb[K];
  if (threadIdx.x < K) sb[threadIdx.x] = bins[threadIdx.x];
  __syncthreads();
  const float inv_rng = 1.f / (1.f / min_d - 1.f / max_d);
  int s = 0;
  while (s + 1 < hb.n && (int)blockIdx.x >= blk.start[s + 1]) ++s;
  const long nb = blk.start[s + 1] - blk.start[s], lb = blockIdx.x - blk.start[s];
  const long rows_img = hb.nimg > 0 ? hb.M[s] / hb.nimg : 0;
  for (long m = lb * 256 + threadIdx.x; m < hb.M[s]; m += nb * 256) {
    const float sc = hb.P2 ? hb.P2[(m / rows_img) * 12] / hb.base_fx : 0.f;
    head_bwd_row<K, T>(hb.logits[s], sb, hb.d_depth[s], hb.d_disp[s], reinterpret_cast<T*>(hb.dlogits[s]), m, Cl, inv_rng,
                       sc, min_d, max_d);
  }
}

// ---------------------------------------------------------------------------------------------
// pose tail.  Forward-mode duals over the 6 inputs (axis-angle, translation) give the Jacobian of
// the 3x4 transform; the backward contracts it with dT.
// ---------------------------------------------------------------------------------------------
struct Dual {
  float v; float d[6];
};
__device__ inline Dual dconst(float c) { Dual r; r.v = c; for (int i = 0; i < 6; ++i) r.d[i] = 0.f; return r; }
__device__ inline Dual dvar(float c, int i) { Dual r = dconst(c); r.d[i] = 1.f; return r; }
__device__ inline Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ inline Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ inline Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < 6; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ inline Dual operator/(const Dual& a, const Dual& b) { Dual r; r.v = a.v / b.v; for (int i = 0; i < 6; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
__device__ inline Dual dneg(const Dual& a) { Dual r; r.v = -a.v; for (int i = 0; i < 6; ++i) r.d[i] = -a.d[i]; return r; }
__device__ inline Dual dsqrt(const Dual& a) { Dual r; r.v = sqrtf(a.v); float k = r.v > 0.f ? 0.5f / r.v : 0.f; for (int i = 0; i < 6; ++i) r.d[i] = k * a.d[i]; return r; }
__device__ inline Dual dsin(const Dual& a) { Dual r; r.v = sinf(a.v); float k = cosf(a.v); for (int i = 0; i < 6; ++i) r.d[i] = k * a.d[i]; return r; }
__device__ inline Dual dcos(const Dual& a) { Dual r; r.v = cosf(a.v); float k = -sinf(a.v); for (int i = 0; i < 6; ++i) r.d[i] = k * a.d[i]; return r; }

// M = T(t) R(v)  or, inverted, R(v)^T T(-t)   (rows 0..2 of the 4x4)
__device__ inline void pose_matrix(const float in[6], int invert, Dual M[3][4]) {
  Dual vx = dvar(in[0], 0), vy = dvar(in[1], 1), vz = dvar(in[2], 2);
  Dual tx = dvar(in[3], 3), ty = dvar(in[4], 4), tz = dvar(in[5], 5);
  Dual angle = dsqrt(vx * vx + vy * vy + vz * vz);
  Dual den = angle + dconst(1e-7f);
  Dual x = vx / den, y = vy / den, z = vz / den;
  Dual ca = dcos(angle), sa = dsin(angle);
  Dual Cc = dconst(1.f) - ca;
  Dual xs = x * sa, ys = y * sa, zs = z * sa;
  Dual xC = x * Cc, yC = y * Cc, zC = z * Cc;
  Dual xyC = x * yC, yzC = y * zC, zxC = z * xC;
  Dual R[3][3];
  R[0][0] = x * xC + ca; R[0][1] = xyC - zs;    R[0][2] = zxC + ys;
  R[1][0] = xyC + zs;    R[1][1] = y * yC + ca; R[1][2] = yzC - xs;
  R[2][0] = zxC - ys;    R[2][1] = yzC + xs;    R[2][2] = z * zC + ca;
  if (!invert) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i][j] = R[i][j];
    M[0][3] = tx; M[1][3] = ty; M[2][3] = tz;
  } else {
    Dual nt[3] = {dneg(tx), dneg(ty), dneg(tz)};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) M[i][j] = R[j][i];
      M[i][3] = R[0][i] * nt[0] + R[1][i] * nt[1] + R[2][i] * nt[2];
    }
  }
}

// one block (64 lanes) per batch element; x is the last pose conv output fp32 [B, hw, Cx]
__global__ __launch_bounds__(64) void pose_tail_fwd_kernel(const float* __restrict__ x, float* __restrict__ axisangle,
                                                           float* __restrict__ translation, float* __restrict__ Tm,
                                                           int hw, int Cx, int nframes, int invert, float scale) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nout = 6 * nframes;
  __shared__ float mean[64];
  for (int c = 0; c < nout; ++c) {
    float s = 0.f;
    for (int i = lane; i < hw; i += 64) s += x[((long)b * hw + i) * Cx + c];
    s = wave_sum(s);
    if (lane == 0) mean[c] = scale * (s / (float)hw);
  }
  __syncthreads();
  if (lane < nout) {
    int f = lane / 6, k = lane % 6;
    if (k < 3) axisangle[((long)b * nframes + f) * 3 + k] = mean[lane];
    else translation[((long)b * nframes + f) * 3 + (k - 3)] = mean[lane];
  }
  if (lane == 0) {
    float in[6] = {mean[0], mean[1], mean[2], mean[3], mean[4], mean[5]};
    Dual M[3][4];
    pose_matrix(in, invert, M);
    float* o = Tm + (long)b * 16;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) o[i * 4 + j] = M[i][j].v;
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
  }
}

// dT [B,4,4] (rows 0..2 used) -> dx [B, hw, Cx] (dtype T): every spatial position of output channel
// k<6 receives 0.01/hw * d(in_k); other channels zero.
template <typename T>
__global__ __launch_bounds__(64) void pose_tail_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dT,
                                                           T* __restrict__ dx, int hw, int Cx, int nframes,
                                                           int invert, float scale) {
  const int b = blockIdx.x, lane = threadIdx.x;
  __shared__ float mean[8];
  __shared__ float gin[8];
  for (int c = 0; c < 6; ++c) {
    float s = 0.f;
    for (int i = lane; i < hw; i += 64) s += x[((long)b * hw + i) * Cx + c];
    s = wave_sum(s);
    if (lane == 0) mean[c] = scale * (s / (float)hw);
  }
  __syncthreads();
  if (lane == 0) {
    float in[6] = {mean[0], mean[1], mean[2], mean[3], mean[4], mean[5]};
    Dual M[3][4];
    pose_matrix(in, invert, M);
    const float* g = dT + (long)b * 16;
    for (int k = 0; k < 6; ++k) {
      float a = 0.f;
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) a += g[i * 4 + j] * M[i][j].d[k];
      gin[k] = a * scale / (float)hw;
    }
  }
  __syncthreads();
  for (int i = lane; i < hw * Cx; i += 64) {
    int c = i % Cx;
    dx[(long)b * hw * Cx + i] = ElemTraits<T>::from_f(c < 6 ? gin[c] : 0.f);
  }
}

int grid_for(long items) {
  long b = (items + 255) / 256;
  return (int)std::max<long>(1, std::min<long>(b, 8192));
}

}  // namespace

extern "C" int fs_depth_head_fwd(const float* logits, const float* bins, float* depth, float* disp, int64_t M, int K,
                                 int Cl, float min_depth, float max_depth, void* stream) {
  if (!logits || !bins || !depth || !disp || Cl < K) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for(M));
  if (K == 16) hipLaunchKernelGGL(depth_head_fwd_kernel<16>, grid, dim3(256), 0, st, logits, bins, depth, disp, (long)M, Cl, min_depth, max_depth);
  else if (K == 32) hipLaunchKernelGGL(depth_head_fwd_kernel<32>, grid, dim3(256), 0, st, logits, bins, depth, disp, (long)M, Cl, min_depth, max_depth);
  else if (K == 64) hipLaunchKernelGGL(depth_head_fwd_kernel<64>, grid, dim3(256), 0, st, logits, bins, depth, disp, (long)M, Cl, min_depth, max_depth);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_depth_head_bwd(const float* logits, const float* bins, const float* d_depth, const float* d_disp,
                                 void* dlogits, int64_t M, int K, int Cl, float min_depth, float max_depth, int dtype,
                                 void* stream) {
  if (!logits || !bins || !dlogits || Cl < K) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for(M));
#define FS_DH(KK, TT) hipLaunchKernelGGL((depth_head_bwd_kernel<KK, TT>), grid, dim3(256), 0, st, logits, bins, d_depth, d_disp, (TT*)dlogits, (long)M, Cl, min_depth, max_depth)
  if (dtype == FS_DTYPE_BF16) {
    if (K == 16) FS_DH(16, bf16); else if (K == 32) FS_DH(32, bf16); else if (K == 64) FS_DH(64, bf16); else return FS_EINVAL;
  } else if (dtype == FS_DTYPE_F32) {
    if (K == 16) FS_DH(16, float); else if (K == 32) FS_DH(32, float); else if (K == 64) FS_DH(64, float); else return FS_EINVAL;
  } else return FS_EINVAL;
#undef FS_DH
  return fs_launch_status();
}

namespace {
bool head_blocks(const FsHeadBatch* hb, bool bwd, HeadBlocks& blk, int& total) {
  if (!hb || hb->n < 1 || hb->n > FS_HEAD_MAX) return false;
  if (hb->P2 && (hb->nimg < 1 || !(hb->base_fx > 0.f))) return false;
  total = 0;
  for (int s = 0; s < hb->n; ++s) {
    if (!hb->logits[s] || hb->M[s] <= 0) return false;
    if (hb->P2 && hb->M[s] % hb->nimg != 0) return false;
    if (bwd ? !hb->dlogits[s] : (!hb->depth[s] || !hb->disp[s])) return false;
    blk.start[s] = total;
    total += grid_for(hb->M[s]);
  }
  blk.start[hb->n] = total;
  return true;
}
}  // namespace

extern "C" int fs_depth_head_fwd_multi(const FsHeadBatch* hb, const float* bins, int K, int Cl, float min_depth,
                                       float max_depth, void* stream) {
  HeadBlocks blk; int total = 0;
  if (!bins || Cl < K || !head_blocks(hb, false, blk, total)) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(total);
  if (K == 16) hipLaunchKernelGGL(depth_head_fwd_multi_kernel<16>, grid, dim3(256), 0, st, *hb, blk, bins, Cl, min_depth, max_depth);
  else if (K == 32) hipLaunchKernelGGL(depth_head_fwd_multi_kernel<32>, grid, dim3(256), 0, st, *hb, blk, bins, Cl, min_depth, max_depth);
  else if (K == 64) hipLaunchKernelGGL(depth_head_fwd_multi_kernel<64>, grid, dim3(256), 0, st, *hb, blk, bins, Cl, min_depth, max_depth);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_depth_head_bwd_multi(const FsHeadBatch* hb, const float* bins, int K, int Cl, float min_depth,
                                       float max_depth, int dtype, void* stream) {
  HeadBlocks blk; int total = 0;
  if (!bins || Cl < K || !head_blocks(hb, true, blk, total)) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(total);
#define FS_DHM(KK, TT) hipLaunchKernelGGL((depth_head_bwd_multi_kernel<KK, TT>), grid, dim3(256), 0, st, *hb, blk, bins, Cl, min_depth, max_depth)
  if (dtype == FS_DTYPE_BF16) {
    if (K == 16) FS_DHM(16, bf16); else if (K == 32) FS_DHM(32, bf16); else if (K == 64) FS_DHM(64, bf16); else return FS_EINVAL;
  } else if (dtype == FS_DTYPE_F32) {
    if (K == 16) FS_DHM(16, float); else if (K == 32) FS_DHM(32, float); else if (K == 64) FS_DHM(64, float); else return FS_EINVAL;
  } else return FS_EINVAL;
#undef FS_DHM
  return fs_launch_status();
}

extern "C" int fs_pose_tail_fwd(const float* x, float* axisangle, float* translation, float* T, int B, int hw, int Cx,
                                int nframes, int invert, float scale, void* stream) {
  if (!x || !axisangle || !translation || !T || nframes < 1 || 6 * nframes > Cx || 6 * nframes > 64) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(pose_tail_fwd_kernel, dim3(B), dim3(64), 0, st, x, axisangle, translation, T, hw, Cx, nframes, invert, scale);
  return fs_launch_status();
}

extern "C" int fs_pose_tail_bwd(const float* x, const float* dT, void* dx, int B, int hw, int Cx, int nframes,
                                int invert, float scale, int dtype, void* stream) {
  if (!x || !dT || !dx || 6 * nframes > Cx) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(pose_tail_bwd_kernel<bf16>, dim3(B), dim3(64), 0, st, x, dT, (bf16*)dx, hw, Cx, nframes, invert, scale);
  else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(pose_tail_bwd_kernel<float>, dim3(B), dim3(64), 0, st, x, dT, (float*)dx, hw, Cx, nframes, invert, scale);
  else return FS_EINVAL;
  return fs_launch_status();
}
