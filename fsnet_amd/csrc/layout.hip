// Layout / packing kernels: OIHW fp32 master weights -> K-contiguous packed MFMA operands,
// NCHW fp32 images -> NHWC (channel-padded) activations.
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>
#include <cstdint>

namespace {

// K position -> tap of the OIHW tensor.  order 1 (3x3 only): the dgrad operand of a stride-2 convolution keeps the
// taps of each output-parity class (y%2, x%2) contiguous, classes (0,0) (0,1) (1,0) (1,1): [4] [3,5] [1,7] [0,2,6,8],
// so that each class is a plain K slice of the operand (conv.py: one stride-1 launch per class).
__host__ __device__ inline int tap_at(int order, int pos, int RS) {
  if (order == 1 && RS == 9) {
    const int perm[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
    return perm[pos];
  }
  return pos;
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ dst, int Co, int Ci, int R,
                                    int S, int rows, int cs, int cs_p, long ktot_p, long total, int transpose) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long row = i / ktot_p; int k = (int)(i % ktot_p);
  int tap = k / cs_p, c = k % cs_p;
  float v = 0.f;
  if (row < rows && c < cs && tap < R * S) {
    tap = tap_at(transpose == 2 ? 1 : 0, tap, R * S);
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
  constexpr int VN = VecN<T>::N;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * H * W;
  if (i >= total) return;
  long hw = (long)H * W;
  long n = i / hw, p = i % hw;
  T* o = dst + i * Cp;
  for (int c0 = 0; c0 < Cp; c0 += VN) {          // one 16-byte store per VN channels (Cp % VN == 0, checked by the host)
    float v[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) {
      const int c = c0 + k;
      v[k] = 0.f;
      if (c < Ca) v[k] = a[(n * Ca + c) * hw + p];
      else if (c < Ca + Cb) v[k] = b[(n * Cb + (c - Ca)) * hw + p];
    }
    storev<T>(o + c0, v);
  }
}

// All convolutions of a model in ONE launch.  A block owns a 32 (co) x IB (ci) x RS tile of one layer's OIHW master
// weights: it reads the tile with coalesced rows (IB*RS contiguous floats per co), keeps it in LDS and writes
// the two MFMA operands as contiguous runs — forward [co][tap][ci] (runs of IB elements) and dgrad
// [ci][tap][co] (runs of 32).  (The first version mapped one thread per *output* element: its reads strode the
// OIHW tensor by 9 or by 9*Ci floats and the step-start re-pack took 234 us for 215 MB; this one ~20 us.)
// Padding rows / channels / K of the operands are zero from allocation and never touched.
constexpr int PACK_CB = 32, PACK_ROW = 289;      // tile rows (co) and LDS row length (floats, odd: no conflicts)

__host__ __device__ inline int pack_ib(int RS, int Ci) {
  int ib = 288 / RS;                     // tile row fits PACK_ROW - 1 floats
  int p2 = 1;
  while (p2 * 2 <= ib && p2 < 128) p2 *= 2;
  (void)Ci;
  return p2;
}

// tile body; RS_C / NCI_C / NCO_C > 0 fix the tap count and the tile extent at compile time (the full 32 x 32 x 9
// tiles of the 3x3 layers are 97 % of the work: with run-time divisors the index arithmetic — three div/mod chains per
// element and pass — cost more than the 176 MB the kernel moves: 139 us against ~50)
template <typename T, int RS_C, int NCI_C, int NCO_C>
__device__ __forceinline__ void pack_tile(const FsPackDesc& d, float* tile, int co0, int ci0, int nco_r, int nci_r) {
  const int RS = RS_C > 0 ? RS_C : d.R * d.S;
  const int nci = NCI_C > 0 ? NCI_C : nci_r;
  const int nco = NCO_C > 0 ? NCO_C : nco_r;
  const int run = nci * RS;                         // contiguous floats per co in the master tensor
  if (RS_C > 0 && NCI_C > 0 && (NCI_C * RS_C) % 4 == 0 && (d.Ci * RS) % 4 == 0 && (ci0 * RS) % 4 == 0) {
    constexpr int RUN4 = NCI_C * RS_C / 4 > 0 ? NCI_C * RS_C / 4 : 1;       // 16-byte loads of the master rows
    for (int i = threadIdx.x; i < nco * RUN4; i += 256) {
      const int col = i / RUN4, r4 = i - col * RUN4;
      const float4 v = *reinterpret_cast<const float4*>(d.w + ((long)(co0 + col) * d.Ci + ci0) * RS + r4 * 4);
      float* o = tile + col * PACK_ROW + r4 * 4;
      o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
  } else {
    for (int i = threadIdx.x; i < nco * run; i += 256) {
      int col = i / run, rem = i - col * run;
      tile[col * PACK_ROW + rem] = d.w[((long)(co0 + col) * d.Ci + ci0) * RS + rem];
    }
  }
  __syncthreads();
  T* df = reinterpret_cast<T*>(d.dst_f);
  T* dd = reinterpret_cast<T*>(d.dst_d);
  if constexpr (RS_C > 0 && NCI_C % 8 == 0 && NCI_C > 0 && NCO_C % 8 == 0 && NCO_C > 0) {
    // full tile: one thread gathers 8 consecutive channels from LDS and stores them as 16-byte lanes (a 2-byte
    // store per element kept the kernel at the store-issue rate)
    constexpr int VN = VecN<T>::N;
    for (int i = threadIdx.x; i < NCO_C * RS_C * (NCI_C / 8); i += 256) {     // (co, tap, 8 ci)
      const int cig = i % (NCI_C / 8); const int q = i / (NCI_C / 8); const int tap = q % RS_C; const int col = q / RS_C;
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = tile[col * PACK_ROW + (cig * 8 + k) * RS_C + tap];
      T* o = df + (long)(co0 + col) * d.k_f + (long)tap * d.cs_f + ci0 + cig * 8;
#pragma unroll
      for (int h = 0; h < 8 / VN; ++h) storev<T>(o + h * VN, v + h * VN);
    }
    if (dd) {
      for (int i = threadIdx.x; i < NCI_C * RS_C * (NCO_C / 8); i += 256) {   // (ci, tap, 8 co)
        const int cog = i % (NCO_C / 8); const int q = i / (NCO_C / 8); const int tap = q % RS_C; const int cil = q / RS_C;
        const int st = tap_at(d.tap_order_d, tap, RS_C);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = tile[(cog * 8 + k) * PACK_ROW + cil * RS_C + st];
        T* o = dd + (long)(ci0 + cil) * d.k_d + (long)tap * d.cs_d + co0 + cog * 8;
#pragma unroll
        for (int h = 0; h < 8 / VN; ++h) storev<T>(o + h * VN, v + h * VN);
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < nco * run; i += 256) {          // (co, tap, ci): ci fastest
    int cil = i % nci; int q = i / nci; int tap = q % RS; int col = q / RS;
    df[(long)(co0 + col) * d.k_f + (long)tap * d.cs_f + ci0 + cil] = ElemTraits<T>::from_f(tile[col * PACK_ROW + cil * RS + tap]);
  }
  if (dd) {
    for (int i = threadIdx.x; i < nco * run; i += 256) {        // (ci, tap, co): co fastest
      int col = i % nco; int q = i / nco; int tap = q % RS; int cil = q / RS;
      dd[(long)(ci0 + cil) * d.k_d + (long)tap * d.cs_d + co0 + col] =
          ElemTraits<T>::from_f(tile[col * PACK_ROW + cil * RS + tap_at(d.tap_order_d, tap, RS)]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const FsPackDesc* __restrict__ descs, int n) {
  __shared__ float tile[PACK_CB * PACK_ROW];
  int lo = 0, hi = n - 1;
  const long bid = blockIdx.x;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (descs[mid].block_start <= bid) lo = mid; else hi = mid - 1;
  }
  const FsPackDesc d = descs[lo];
  const int RS = d.R * d.S;
  const int IB = pack_ib(RS, d.Ci);
  const int nci_t = (d.Ci + IB - 1) / IB;
  const int local = (int)(bid - d.block_start);
  const int co0 = (local / nci_t) * PACK_CB, ci0 = (local % nci_t) * IB;
  const int nco = min(PACK_CB, d.Co - co0), nci = min(IB, d.Ci - ci0);
  if (nco <= 0 || nci <= 0) return;
  if (RS == 9 && nci == 32 && nco == PACK_CB) pack_tile<T, 9, 32, PACK_CB>(d, tile, co0, ci0, nco, nci);
  else if (RS == 9) pack_tile<T, 9, 0, 0>(d, tile, co0, ci0, nco, nci);
  else if (RS == 1 && nci == IB && nco == PACK_CB && IB == 128) pack_tile<T, 1, 128, PACK_CB>(d, tile, co0, ci0, nco, nci);
  else pack_tile<T, 0, 0, 0>(d, tile, co0, ci0, nco, nci);
}

}  // namespace

namespace {
struct CopyBatch { const char* src[FS_COPY_MAX]; char* dst[FS_COPY_MAX]; long bytes[FS_COPY_MAX]; long start[FS_COPY_MAX + 1]; int n; };

// up to FS_COPY_MAX device-to-device copies in ONE launch (the batch -> static hipGraph input buffers):
// blocks are dealt over the copies in proportion to their size; 16-byte lanes, byte tail by the last threads
__global__ __launch_bounds__(256) void copy_multi_kernel(const CopyBatch c) {
  int k = 0;
  while (k + 1 < c.n && (long)blockIdx.x >= c.start[k + 1]) ++k;
  const long nblk = c.start[k + 1] - c.start[k], lb = blockIdx.x - c.start[k];
  const long n16 = c.bytes[k] >> 4;
  const uint4* s = reinterpret_cast<const uint4*>(c.src[k]);
  uint4* d = reinterpret_cast<uint4*>(c.dst[k]);
  // four 16-byte loads in flight per thread (one per iteration left the copy latency-bound at ~1.1 TB/s)
  const long step = nblk * 256;
  long i = lb * 256 + threadIdx.x;
  for (; i + 3 * step < n16; i += 4 * step) {
    const uint4 a = s[i], b = s[i + step], e = s[i + 2 * step], f = s[i + 3 * step];
    d[i] = a; d[i + step] = b; d[i + 2 * step] = e; d[i + 3 * step] = f;
  }
  for (; i < n16; i += step) d[i] = s[i];
  if (lb == 0) {
    const long tail = c.bytes[k] & 15;
    if ((long)threadIdx.x < tail) c.dst[k][n16 * 16 + threadIdx.x] = c.src[k][n16 * 16 + threadIdx.x];
  }
}
}  // namespace

extern "C" int fs_copy_multi(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream) {
  if (!src || !dst || !bytes || n <= 0 || n > FS_COPY_MAX) return FS_EINVAL;
  CopyBatch c;
  long blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!src[i] || !dst[i] || bytes[i] <= 0) return FS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src[i]) | reinterpret_cast<uintptr_t>(dst[i])) & 15) return FS_EINVAL;
    c.src[i] = static_cast<const char*>(src[i]); c.dst[i] = static_cast<char*>(dst[i]); c.bytes[i] = bytes[i];
    c.start[i] = blocks;
    blocks += std::max<long>(1, std::min<long>((bytes[i] / 16 + 1023) / 1024, 512));   // ~4 lanes per thread
  }
  c.start[n] = blocks; c.n = n;
  hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), c);
  return fs_launch_status();
}

namespace {
// up to FS_COPY_MAX buffers zeroed in ONE launch (the step's scratch: BatchNorm statistics pools, loss accumulators, depth
// gradient maps, the gradient-norm scalar — each was a fill launch of its own somewhere along the step's chains)
__global__ __launch_bounds__(256) void zero_multi_kernel(const CopyBatch c) {
  int k = 0;
  while (k + 1 < c.n && (long)blockIdx.x >= c.start[k + 1]) ++k;
  const long nblk = c.start[k + 1] - c.start[k], lb = blockIdx.x - c.start[k];
  const long n16 = c.bytes[k] >> 4;
  uint4* d = reinterpret_cast<uint4*>(c.dst[k]);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (long i = lb * 256 + threadIdx.x; i < n16; i += nblk * 256) d[i] = z;
  if (lb == 0) {
    const long tail = c.bytes[k] & 15;
    if ((long)threadIdx.x < tail) c.dst[k][n16 * 16 + threadIdx.x] = 0;
  }
}
}  // namespace

extern "C" int fs_zero_multi(void* const* dst, const int64_t* bytes, int n, void* stream) {
  if (!dst || !bytes || n <= 0 || n > FS_COPY_MAX) return FS_EINVAL;
  CopyBatch c;
  long blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!dst[i] || bytes[i] <= 0 || (reinterpret_cast<uintptr_t>(dst[i]) & 15)) return FS_EINVAL;
    c.src[i] = nullptr; c.dst[i] = static_cast<char*>(dst[i]); c.bytes[i] = bytes[i];
    c.start[i] = blocks;
    blocks += std::max<long>(1, std::min<long>((bytes[i] / 16 + 1023) / 1024, 512));
  }
  c.start[n] = blocks; c.n = n;
  hipLaunchKernelGGL(zero_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), c);
  return fs_launch_status();
}

extern "C" int64_t fs_pack_tile_blocks(int Co, int Ci, int R, int S) {
  if (Co <= 0 || Ci <= 0 || R <= 0 || S <= 0 || R * S > 288) return -1;
  const int IB = pack_ib(R * S, Ci);
  return (int64_t)((Co + PACK_CB - 1) / PACK_CB) * ((Ci + IB - 1) / IB);
}

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
  if (!a || !dst || Cp < Ca + Cb || Cp % (dtype == FS_DTYPE_BF16 ? 8 : 4) != 0) return FS_EINVAL;
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
