// Self-distillation stage: the uncertainty head of MultiChannelDepthDecoderUncertain and the
// distillation loss of MonoDepth2Decoder.  Replaces:
//   torch.sigmoid(uncertain_logz conv)                          monodepth/networks/models/heads/depth_encoder.py:186
//   compute_distill_loss (L1 to the teacher's depth, optionally  monodepth/networks/models/heads/monodepth2_decoder.py:185-203
//     divided by the predicted uncertainty + log-uncertainty) and its autograd backward
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

// u[m] = sigmoid(logits[m][0])   (logits: fp32 NHWC rows of Cp channels, channel 0 is the head's only output)
__global__ __launch_bounds__(256) void sigmoid_head_fwd_kernel(const float* __restrict__ logits, float* __restrict__ u,
                                                               long M, int Cp) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long)gridDim.x * 256)
    u[i] = 1.f / (1.f + __expf(-logits[i * Cp]));
}

// dl[m][0] = du[m] * u (1 - u), dl[m][1..Cp) = 0   (written in the compute dtype for the conv backward)
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_head_bwd_kernel(const float* __restrict__ u, const float* __restrict__ du,
                                                               T* __restrict__ dl, long M, int Cp) {
  const int groups = Cp / 4;
  const long total = M * groups;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long m = i / groups; int g = (int)(i % groups);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (g == 0) { float s = u[m]; v[0] = du[m] * s * (1.f - s); }
    store4<T>(dl + m * Cp + g * 4, v);
  }
}

// sum_i  |t - p| / u + log(u + 1e-5)   (u == nullptr: sum |t - p|)  -> one f64 atomic per block
__global__ __launch_bounds__(256) void distill_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                          const float* __restrict__ u, long n, double* __restrict__ out) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float e = fabsf(t[i] - p[i]);
    acc += u ? (double)(e / u[i] + logf(u[i] + 1e-5f)) : (double)e;
  }
  __shared__ double sh[4];
  acc = block_sum_d(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}

// d loss / d p = -sign(t - p) / u * k,  d loss / d u = (-|t - p| / u^2 + 1 / (u + 1e-5)) * k,  k = gout / n
__global__ __launch_bounds__(256) void distill_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                          const float* __restrict__ u, long n, const double* __restrict__ gout,
                                                          float* __restrict__ dp, float* __restrict__ du) {
  const float k = (float)((gout ? *gout : 1.0) / (double)n);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float d = t[i] - p[i];
    float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);           // torch.abs backward: sign, 0 at 0
    if (u) {
      float uu = u[i];
      dp[i] = -sg / uu * k;
      du[i] = (-fabsf(d) / (uu * uu) + 1.f / (uu + 1e-5f)) * k;
    } else {
      dp[i] = -sg * k;
    }
  }
}

unsigned grid_for(long n, long cap) { return (unsigned)std::max<long>(1, std::min<long>((n + 255) / 256, cap)); }

}  // namespace

extern "C" int fs_sigmoid_head_fwd(const float* logits, float* u, int64_t M, int Cp, void* stream) {
  if (!logits || !u || M <= 0 || Cp < 1) return FS_EINVAL;
  hipLaunchKernelGGL(sigmoid_head_fwd_kernel, dim3(grid_for(M, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     logits, u, (long)M, Cp);
  return fs_launch_status();
}

extern "C" int fs_sigmoid_head_bwd(const float* u, const float* du, void* dl, int64_t M, int Cp, int dtype, void* stream) {
  if (!u || !du || !dl || M <= 0 || Cp < 4 || Cp % 4) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned grid = grid_for(M * (Cp / 4), 8192);
  if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(sigmoid_head_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, u, du, (bf16*)dl, (long)M, Cp);
  else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(sigmoid_head_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, u, du, (float*)dl, (long)M, Cp);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_distill_fwd(const float* pred, const float* teacher, const float* uncertain, int64_t n, double* sum_out,
                              void* stream) {
  if (!pred || !teacher || !sum_out || n <= 0) return FS_EINVAL;
  hipLaunchKernelGGL(distill_fwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     pred, teacher, uncertain, (long)n, sum_out);
  return fs_launch_status();
}

extern "C" int fs_distill_bwd(const float* pred, const float* teacher, const float* uncertain, int64_t n, const double* gout,
                              float* d_pred, float* d_uncertain, void* stream) {
  if (!pred || !teacher || !d_pred || n <= 0 || (uncertain && !d_uncertain)) return FS_EINVAL;
  hipLaunchKernelGGL(distill_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pred,
                     teacher, uncertain, (long)n, gout, d_pred, d_uncertain);
  return fs_launch_status();
}
