// Layout / packing kernels: OIHW fp32 master weights -> K-contiguous packed MFMA operands,
// NCHW fp32 images -> NHWC (channel-padded) activations.
#include "common.h"
#include "fsnet_hip_internal.h"

namespace {

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ dst, int Co, int Ci, int R,
                                    int S, int rows, int cs, int cs_p, long ktot_p, long total, int transpose) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long row = i / ktot_p; int k = (int)(i % ktot_p);
  int tap = k / cs_p, c = k % cs_p;
  float v = 0.f;
  if (row < rows && c < cs && tap < R * S) {
    int r = tap / S, s = tap % S;
    int co = transpose ? c : (int)row;
    int ci = transpose ? (int)row : c;
    v = w[(((long)co * Ci + ci) * R + r) * S + s];
  }
  dst[i] = ElemTraits<T>::from_f(v);
}

// images: up to two NCHW fp32 tensors concatenated along C, written as NHWC with Cp channels
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ a, const float* __restrict__ b, T* __restrict__ dst,
                                    int N, int Ca, int Cb, int H, int W, int Cp) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * H * W;
  if (i >= total) return;
  long hw = (long)H * W;
  long n = i / hw, p = i % hw;
  T* o = dst + i * Cp;
  for (int c = 0; c < Cp; ++c) {
    float v = 0.f;
    if (c < Ca) v = a[(n * Ca + c) * hw + p];
    else if (c < Ca + Cb) v = b[(n * Cb + (c - Ca)) * hw + p];
    o[c] = ElemTraits<T>::from_f(v);
  }
}

// all convolutions of a model in ONE launch: a block finds its descriptor by binary search over block_start
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const FsPackDesc* __restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  const long bid = blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block_start <= bid) lo = mid; else hi = mid - 1;
  }
  const FsPackDesc d = descs[lo];
  const long i = (bid - d.block_start) * 256 + threadIdx.x;
  const long nf = (long)d.rows_f * d.k_f, nd = (long)d.rows_d * d.k_d;
  if (i >= nf + nd) return;
  const bool tr = i >= nf;
  const long j = tr ? i - nf : i;
  const long kp = tr ? d.k_d : d.k_f;
  const int csp = tr ? d.cs_d : d.cs_f;
  const long row = j / kp; const int k = (int)(j % kp);
  const int tap = k / csp, c = k % csp;
  const int rows = tr ? d.Ci : d.Co, cs = tr ? d.Co : d.Ci;
  float v = 0.f;
  if (row < rows && c < cs && tap < d.R * d.S) {
    int r = tap / d.S, s = tap % d.S;
    int co = tr ? c : (int)row, ci = tr ? (int)row : c;
    v = d.w[(((long)co * d.Ci + ci) * d.R + r) * d.S + s];
  }
  T* dst = reinterpret_cast<T*>(tr ? d.dst_d : d.dst_f);
  dst[j] = ElemTraits<T>::from_f(v);
}

}  // namespace

extern "C" int fs_pack_weights_multi(const FsPackDesc* descs_dev, int n, int64_t total_blocks, int dtype, void* stream) {
  if (!descs_dev || n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL(pack_weights_multi_kernel<bf16>, dim3((unsigned)total_blocks), dim3(256), 0, st, descs_dev, n);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL(pack_weights_multi_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, st, descs_dev, n);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_pack_weights(const float* w_oihw, void* dst, int Co, int Ci, int R, int S, int rows_p,
                               int cs_p, int64_t ktot_p, int transpose, int dtype, void* stream) {
  if (!w_oihw || !dst || rows_p <= 0 || cs_p <= 0 || ktot_p < (int64_t)R * S * cs_p) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long total = (long)rows_p * ktot_p;
  int rows = transpose ? Ci : Co, cs = transpose ? Co : Ci;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL(pack_weights_kernel<bf16>, grid, dim3(256), 0, st, w_oihw, (bf16*)dst, Co, Ci, R, S, rows, cs,
                       cs_p, (long)ktot_p, total, transpose);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL(pack_weights_kernel<float>, grid, dim3(256), 0, st, w_oihw, (float*)dst, Co, Ci, R, S, rows,
                       cs, cs_p, (long)ktot_p, total, transpose);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_nchw_to_nhwc(const float* a, const float* b, void* dst, int N, int Ca, int Cb, int H, int W,
                               int Cp, int dtype, void* stream) {
  if (!a || !dst || Cp < Ca + Cb) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  long total = (long)N * H * W;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == FS_DTYPE_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, grid, dim3(256), 0, st, a, b, (bf16*)dst, N, Ca, Cb, H, W, Cp);
  else if (dtype == FS_DTYPE_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, st, a, b, (float*)dst, N, Ca, Cb, H, W, Cp);
  else return FS_EINVAL;
  return fs_launch_status();
}
