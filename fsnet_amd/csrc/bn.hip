// Train-mode BatchNorm (+ residual add, + ReLU) forward and backward, HBM-bound elementwise /
// reduction kernels.  Replaces nn.BatchNorm2d (train mode) + F.relu + residual add at
//   vision_base/networks/models/backbone/resnet.py:33-50,70-89,201-203 (bn, relu, out += residual)
//   vision_base/networks/blocks/blocks.py:44-52 (ConvBnReLU)
// and their autograd backward.  Batch statistics arrive as per-channel f64 (sum, sumsq) produced by
// the conv epilogue (conv_igemm.hip) — under data parallelism the host all-reduces that small
// buffer between the conv and this kernel (SyncBatchNorm semantics, scripts/train.py:101).
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

// Index decoding of the element-wise passes.  These kernels move 16 bytes per thread, so the address arithmetic is
// most of their instruction count: a 64-bit division by a run-time value is >100 VALU instructions on CDNA and the
// first version did six per vector (vector -> (row, channel group), row -> (n, h, w) twice).  Channel-group counts
// are powers of two for every ResNet / decoder width (shift + mask), rows fit 32 bits, and dense tensors need no
// (n, h, w) at all.
__device__ __forceinline__ int pow2_shift(int v) { return (v & (v - 1)) == 0 ? __ffs(v) - 1 : -1; }
__device__ __forceinline__ void split_vec(long i, int CG, int sh, int& cg, long& m) {
  if (sh >= 0) { cg = (int)(i & (CG - 1)); m = i >> sh; }
  else if (i < (1L << 31)) { const unsigned u = (unsigned)i, q = u / (unsigned)CG; cg = (int)(u - q * CG); m = q; }
  else { cg = (int)(i % CG); m = i / CG; }
}
__device__ __forceinline__ void split_row(long m, int H, int W, long& n, int& h, int& w) {
  if (m < (1L << 31)) {
    const unsigned u = (unsigned)m, q = u / (unsigned)W, nn = q / (unsigned)H;
    w = (int)(u - q * W); h = (int)(q - nn * H); n = nn;
  } else {
    w = (int)(m % W); const long q = m / W; h = (int)(q % H); n = q / H;
  }
}


constexpr int MAXC = 2048;

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
__device__ inline void bn_channel_coeffs(const double* stats, const float* rmean, const float* rvar, int C, int c,
                                         double count, float eps, float gamma, float beta, float& mean,
                                         float& invstd, float& var_b, float& scale, float& shift) {
  double m, v;
  if (stats) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < FS_STAT_SLOTS; ++k) { s1 += stats[(long)k * 2 * C + c]; s2 += stats[(long)k * 2 * C + C + c]; }
    m = s1 / count; v = s2 / count - m * m;
  }
  else { m = rmean[c]; v = rvar[c]; }  // eval mode: running statistics
  if (v < 0) v = 0;
  mean = (float)m;
  var_b = (float)v;
  invstd = (float)(1.0 / sqrt(v + (double)eps));
  scale = gamma * invstd;
  shift = beta - mean * scale;
}

// BatchNorm statistics -> per-channel coefficients, without touching the activation: the normalisation itself is
// applied by the convolution that consumes the tensor (FsConvArgs.pro_mode = 1, FsWgradArgs.pro_a) — "BatchNorm
// folded into the operand staging".  Same arithmetic and running-statistics update as bn_apply_kernel's preamble.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const FsBnApplyArgs p, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  const int C = p.C;
  const int G = p.groups > 1 ? p.groups : 1;
  const int z = blockIdx.y;
  const long gstat = (long)FS_STAT_SLOTS * 2 * C;
  const double* stats_z = p.stats ? p.stats + z * gstat : nullptr;
  const bool train = p.stats != nullptr;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    float mean, invstd, varb, sc, sh;
    bn_channel_coeffs(stats_z, p.running_mean, p.running_var, C, c, p.count, p.eps, p.gamma[c], p.beta[c], mean, invstd, varb, sc, sh);
    p.save_mean[z * C + c] = mean; p.save_invstd[z * C + c] = invstd;
    scale[z * C + c] = sc; shift[z * C + c] = sh;
    if (z == 0 && train && p.running_mean) {
      float rm = p.running_mean[c], rv = p.running_var[c];
      for (int g = 0; g < G; ++g) {
        float mg = mean, vg = varb, t0, t1, t2;
        if (g > 0) bn_channel_coeffs(p.stats + g * gstat, nullptr, nullptr, C, c, p.count, p.eps, 1.f, 0.f, mg, t0, vg, t1, t2);
        double unb = p.count > 1.0 ? (double)vg * p.count / (p.count - 1.0) : (double)vg;
        rm = (1.f - p.momentum) * rm + p.momentum * mg;
        rv = (1.f - p.momentum) * rv + p.momentum * (float)unb;
      }
      p.running_mean[c] = rm; p.running_var[c] = rv;
    }
  }
  if (z == 0 && blockIdx.x == 0 && threadIdx.x == 0 && train && p.num_batches_tracked) *p.num_batches_tracked += G;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const FsDual<FsBnApplyArgs, FsNoGeom> d, const int fast) {
  // two problems per launch (fsnet_hip_internal.h, FsDual): the statistics groups of both stack along blockIdx.z
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsBnApplyArgs& p = d.a[prob];
  // per-channel coefficients in dynamic LDS (4*C floats): a fixed MAXC-sized array would cap the occupancy of
  // these bandwidth-bound kernels at 4-5 blocks per CU
  extern __shared__ float bn_smem[];
  const int C = p.C;
  constexpr int V = VecN<T>::N;
  // channel slabs: with more than 32 16-byte channel groups per row (C > 256 at bf16) blockIdx.y picks a slab of 32 groups
  // and the block only derives THAT slab's coefficients — the preamble re-reads 128 bytes of f64 sums per channel and
  // block: at C = 2048 that was 262 KB per block, 134-268 MB per launch for a 21 MB tensor (ResNet-50 layer 4: 1.2 TB/s)
  const int CG = C / V, cg_sh = pow2_shift(CG);
  const int SCG = gridDim.y > 1 ? 32 : CG, scg_sh = gridDim.y > 1 ? 5 : cg_sh;
  const int SC = SCG * V, c0 = (int)blockIdx.y * SC;
  float* s_scale = bn_smem; float* s_shift = bn_smem + SC;
  float* s_scale2 = bn_smem + 2 * SC; float* s_shift2 = bn_smem + 3 * SC;
  const bool has2 = p.gamma2 != nullptr;
  const bool train = p.stats != nullptr;
  // statistics groups: blockIdx.z owns the rows [z*Mg, (z+1)*Mg) and the z-th statistics / saved-state slice
  const int G = p.groups > 1 ? p.groups : 1;
  const int z = (int)blockIdx.z - (prob ? d.nb0 : 0);
  const int Mg = p.M / G;
  const long gstat = (long)FS_STAT_SLOTS * 2 * C;
  const double* stats_z = p.stats ? p.stats + z * gstat : nullptr;
  const double* stats2_z = p.stats2 ? p.stats2 + z * gstat : nullptr;
  const bool first = blockIdx.x == 0;
  // the thread's first vector is requested BEFORE the coefficient preamble (a dependent chain of L2 loads and f64
  // arithmetic that every block repeats): on the small encoder layers the preamble was half of the kernel's time and the
  // operand loads only started behind it
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
  const long total = (long)Mg * SCG;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  float pv[V], pr[V];
  if (i0 < total) {
    int cg; long m;
    split_vec(i0, SCG, scg_sh, cg, m);
    m += (long)z * Mg;
    loadv<T>(x + m * C + c0 + cg * V, pv);
    if (res) loadv<T>(res + m * C + c0 + cg * V, pr);
  }
  for (int cl = threadIdx.x; cl < SC; cl += 256) {
    const int c = c0 + cl;
    float mean, invstd, varb, sc, sh;
    bn_channel_coeffs(stats_z, p.running_mean, p.running_var, C, c, p.count, p.eps, p.gamma[c], p.beta[c], mean, invstd, varb, sc, sh);
    s_scale[cl] = sc; s_shift[cl] = sh;
    if (first) { p.save_mean[z * C + c] = mean; p.save_invstd[z * C + c] = invstd; }
    if (first && z == 0 && train && p.running_mean) {
      // one momentum update per group, in group order (= the order the reference calls the module in)
      float rm = p.running_mean[c], rv = p.running_var[c];
      for (int g = 0; g < G; ++g) {
        float mg = mean, vg = varb, t0, t1, t2;
        if (g > 0) bn_channel_coeffs(p.stats + g * gstat, nullptr, nullptr, C, c, p.count, p.eps, 1.f, 0.f, mg, t0, vg, t1, t2);
        double unb = p.count > 1.0 ? (double)vg * p.count / (p.count - 1.0) : (double)vg;
        rm = (1.f - p.momentum) * rm + p.momentum * mg;
        rv = (1.f - p.momentum) * rv + p.momentum * (float)unb;
      }
      p.running_mean[c] = rm; p.running_var[c] = rv;
    }
    if (has2) {
      bn_channel_coeffs(stats2_z, p.running_mean2, p.running_var2, C, c, p.count, p.eps, p.gamma2[c], p.beta2[c], mean, invstd, varb, sc, sh);
      s_scale2[cl] = sc; s_shift2[cl] = sh;
      if (first) { p.save_mean2[z * C + c] = mean; p.save_invstd2[z * C + c] = invstd; }
      if (first && z == 0 && train && p.running_mean2) {
        float rm = p.running_mean2[c], rv = p.running_var2[c];
        for (int g = 0; g < G; ++g) {
          float mg = mean, vg = varb, t0, t1, t2;
          if (g > 0) bn_channel_coeffs(p.stats2 + g * gstat, nullptr, nullptr, C, c, p.count, p.eps, 1.f, 0.f, mg, t0, vg, t1, t2);
          double unb = p.count > 1.0 ? (double)vg * p.count / (p.count - 1.0) : (double)vg;
          rm = (1.f - p.momentum) * rm + p.momentum * mg;
          rv = (1.f - p.momentum) * rv + p.momentum * (float)unb;
        }
        p.running_mean2[c] = rm; p.running_var2[c] = rv;
      }
    }
  }
  if (first && z == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (train && p.num_batches_tracked) *p.num_batches_tracked += G;
    if (train && p.num_batches_tracked2) *p.num_batches_tracked2 += G;
  }
  __syncthreads();

  T* __restrict__ y = reinterpret_cast<T*>(p.y);
  const bool dense_y = !p.pad_out && p.yW == C && p.yH == (long)p.W * C && p.yN == (long)p.H * p.W * C;
  const long stride = (long)gridDim.x * 256;
  if (fast && dense_y && scg_sh >= 0) {
    // dense output (every encoder layer): fixed channel group per thread, coefficients in registers, the raw operands of
    // two later rows in flight (see bn_bwd_apply_kernel)
    if (i0 >= total) return;
    const int cl = (int)(i0 & (SCG - 1)) * V;
    const long step = (stride >> scg_sh) * C;
    long off = ((long)z * Mg + (i0 >> scg_sh)) * C + c0 + cl;
    long left = (total - i0 + stride - 1) / stride;
    float ks[V], kh[V], ks2[V], kh2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      ks[j] = s_scale[cl + j]; kh[j] = s_shift[cl + j];
      ks2[j] = has2 ? s_scale2[cl + j] : 1.f; kh2[j] = has2 ? s_shift2[cl + j] : 0.f;
    }
    const bool relu = p.relu != 0;
    uint4 x1, r1, x2, r2;
    x1 = r1 = x2 = r2 = make_uint4(0, 0, 0, 0);
    if (left > 1) { x1 = *reinterpret_cast<const uint4*>(x + off + step); if (res) r1 = *reinterpret_cast<const uint4*>(res + off + step); }
    if (left > 2) { x2 = *reinterpret_cast<const uint4*>(x + off + 2 * step); if (res) r2 = *reinterpret_cast<const uint4*>(res + off + 2 * step); }
    float v[V], r[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { v[j] = pv[j]; r[j] = pr[j]; }
    for (;;) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = v[j] * ks[j] + kh[j];
      if (res) {
        if (has2) {
#pragma unroll
          for (int j = 0; j < V; ++j) r[j] = r[j] * ks2[j] + kh2[j];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] += r[j];
      }
      if (relu) {
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      storev<T>(y + off, v);
      if (--left == 0) break;
      off += step;
      Unit<T>::unpack(x1, v);
      if (res) Unit<T>::unpack(r1, r);
      x1 = x2; r1 = r2;
      if (left > 2) { x2 = *reinterpret_cast<const uint4*>(x + off + 2 * step); if (res) r2 = *reinterpret_cast<const uint4*>(res + off + 2 * step); }
    }
    return;
  }
  for (long i = i0; i < total; i += stride) {
    int cg; long m;
    split_vec(i, SCG, scg_sh, cg, m);
    m += (long)z * Mg;
    const int cl = cg * V, c = c0 + cl;
    float v[V], r[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { v[j] = pv[j]; r[j] = pr[j]; }
    // the next vector of this thread is in flight while this one is normalised and stored (a block lives for several
    // vectors: the grid is held to ~4 blocks per CU so that the coefficient preamble is paid that many times, not per 256
    // vectors)
    if (i + stride < total) {
      int cg2; long m2;
      split_vec(i + stride, SCG, scg_sh, cg2, m2);
      m2 += (long)z * Mg;
      loadv<T>(x + m2 * C + c0 + cg2 * V, pv);
      if (res) loadv<T>(res + m2 * C + c0 + cg2 * V, pr);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = v[j] * s_scale[cl + j] + s_shift[cl + j];
    if (res) {
      if (has2) {
#pragma unroll
        for (int j = 0; j < V; ++j) r[j] = r[j] * s_scale2[cl + j] + s_shift2[cl + j];
      }
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] += r[j];
    }
    if (p.relu) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (dense_y) { storev<T>(y + m * C + c, v); continue; }
    int w, h; long n;
    split_row(m, p.H, p.W, n, h, w);
    if (!p.pad_out) {
      storev<T>(y + n * p.yN + (long)h * p.yH + (long)w * p.yW + c, v);
    } else {
      // output buffer is [N, H+2, W+2, C] with a replicated border (consumer uses replicate padding)
      // (H, W >= 2: decoder maps are never a single row/column)
      int hs[2] = {h + 1, 0}, ws[2] = {w + 1, 0};
      int nh = 1, nw = 1;
      if (h == 0) { hs[1] = 0; nh = 2; } else if (h == p.H - 1) { hs[1] = p.H + 1; nh = 2; }
      if (w == 0) { ws[1] = 0; nw = 2; } else if (w == p.W - 1) { ws[1] = p.W + 1; nw = 2; }
      for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b)
          storev<T>(y + n * p.yN + (long)hs[a] * p.yH + (long)ws[b] * p.yW + c, v);
    }
  }
}


// BatchNorm + ReLU + MaxPool2d(3, 2, 1) of the encoder stem in one pass (resnet.py:201-206: x = relu(bn1(conv1(x))),
// features[0] = x, x = maxpool(x)): a thread owns one pooled pixel x 16 bytes of channels, normalises the nine window
// inputs from the raw convolution output, rounds them to the storage type (the value the separate pass would have
// stored: same maxima, same argmax codes) and — when the activation itself is wanted (FsBnApplyArgs.y: the depth
// encoder's features[0]; NULL for the pose encoder, whose decoder only reads the last feature) — stores the 2 x 2
// pixels (2ho + {0,1}, 2wo + {0,1}) that only this window owns.  The activation of 36 images at 96 x 320 x 64 is 141 MB:
// the separate passes wrote it, read it back for the pooling, and wrote and read it again in the backward.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const FsDual<FsBnApplyArgs, FsNoGeom> d) {
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsBnApplyArgs& p = d.a[prob];
  extern __shared__ float bn_smem[];
  const int C = p.C;
  float* s_scale = bn_smem; float* s_shift = bn_smem + C;
  const int G = p.groups > 1 ? p.groups : 1;
  const int z = (int)blockIdx.z - (prob ? d.nb0 : 0);
  const long gstat = (long)FS_STAT_SLOTS * 2 * C;
  const double* stats_z = p.stats + z * gstat;
  const bool first = blockIdx.x == 0;
  for (int c = threadIdx.x; c < C; c += 256) {
    float mean, invstd, varb, sc, sh;
    bn_channel_coeffs(stats_z, p.running_mean, p.running_var, C, c, p.count, p.eps, p.gamma[c], p.beta[c], mean, invstd, varb, sc, sh);
    s_scale[c] = sc; s_shift[c] = sh;
    if (first) { p.save_mean[z * C + c] = mean; p.save_invstd[z * C + c] = invstd; }
    if (first && z == 0 && p.running_mean) {
      float rm = p.running_mean[c], rv = p.running_var[c];
      for (int g = 0; g < G; ++g) {
        float mg = mean, vg = varb, t0, t1, t2;
        if (g > 0) bn_channel_coeffs(p.stats + g * gstat, nullptr, nullptr, C, c, p.count, p.eps, 1.f, 0.f, mg, t0, vg, t1, t2);
        double unb = p.count > 1.0 ? (double)vg * p.count / (p.count - 1.0) : (double)vg;
        rm = (1.f - p.momentum) * rm + p.momentum * mg;
        rv = (1.f - p.momentum) * rv + p.momentum * (float)unb;
      }
      p.running_mean[c] = rm; p.running_var[c] = rv;
    }
  }
  if (first && z == 0 && threadIdx.x == 0 && p.num_batches_tracked) *p.num_batches_tracked += G;
  __syncthreads();

  constexpr int V = VecN<T>::N;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  T* __restrict__ y = reinterpret_cast<T*>(p.y);
  T* __restrict__ py = reinterpret_cast<T*>(p.pool_y);
  const int H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
  const int CG = C / V, cg_sh = pow2_shift(CG);
  const int Ng = (p.M / (H * W)) / G;                    // images per statistics group
  const unsigned total = (unsigned)Ng * Ho * Wo * CG;    // (< 2^31: checked by the host)
  const unsigned stride = gridDim.x * 256u;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += stride) {
    unsigned cg, mo;
    if (cg_sh >= 0) { cg = i & (CG - 1); mo = i >> cg_sh; } else { mo = i / CG; cg = i - mo * CG; }
    const unsigned q = mo / Wo, wo = mo - q * Wo;
    const unsigned nn = q / Ho, ho = q - nn * Ho;
    const long n = (long)z * Ng + nn;
    const int c = cg * V;
    float sc[V], sh[V], best[V];
    int bi[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = s_scale[c + j]; sh[j] = s_shift[c + j]; best[j] = -INFINITY; bi[j] = 0; }
    bool fst = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int h = (int)ho * 2 - 1 + r;
      if (h < 0) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int w = (int)wo * 2 - 1 + s;
        if (w < 0) continue;
        const long off = ((n * H + h) * W + w) * C + c;
        float v[V];
        loadv<T>(x + off, v);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          v[j] = v[j] * sc[j] + sh[j];
          if (p.relu) v[j] = fmaxf(v[j], 0.f);
        }
        const uint4 u = Unit<T>::pack(v);         // rounded to the storage type: what the separate pass stores
        Unit<T>::unpack(u, v);
        if (y && r >= 1 && s >= 1) *reinterpret_cast<uint4*>(y + off) = u;
#pragma unroll
        for (int j = 0; j < V; ++j)
          if (fst || v[j] > best[j]) { best[j] = v[j]; bi[j] = r * 3 + s; }      // first max wins (ATen)
        fst = false;
      }
    }
    const long mg = ((n * Ho + ho) * Wo + wo) * C + c;
    storev<T>(py + mg, best);
#pragma unroll
    for (int k = 0; k < V / 4; ++k) {
      const uint32_t packed = (uint32_t)bi[4 * k] | ((uint32_t)bi[4 * k + 1] << 8) | ((uint32_t)bi[4 * k + 2] << 16) |
                              ((uint32_t)bi[4 * k + 3] << 24);
      reinterpret_cast<uint32_t*>(p.pool_idx + mg)[k] = packed;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// gradient w.r.t. the block output at interior pixel (n,h,w), channels c..c+V-1.  `fold`: dout is a
// replicate-padded buffer [N,H+2,W+2,C]; the border copies fold back onto the edge pixel.
template <typename T>
__device__ inline void load_dout(const FsBnBwdArgs& p, const T* dout, long n, int h, int w, int c, float* g) {
  constexpr int V = VecN<T>::N;
  if (!p.fold) {
    loadv<T>(dout + n * p.gN + (long)h * p.gH + (long)w * p.gW + c, g);
    return;
  }
#pragma unroll
  for (int j = 0; j < V; ++j) g[j] = 0.f;
  int hs[2] = {h + 1, 0}, ws[2] = {w + 1, 0};
  int nh = 1, nw = 1;
  if (h == 0) { hs[1] = 0; nh = 2; } else if (h == p.H - 1) { hs[1] = p.H + 1; nh = 2; }
  if (w == 0) { ws[1] = 0; nw = 2; } else if (w == p.W - 1) { ws[1] = p.W + 1; nw = 2; }
  for (int a = 0; a < nh; ++a)
    for (int b = 0; b < nw; ++b) {
      float t[V];
      loadv<T>(dout + n * p.gN + (long)hs[a] * p.gH + (long)ws[b] * p.gW + c, t);
#pragma unroll
      for (int j = 0; j < V; ++j) g[j] += t[j];
    }
}

template <typename T>
__device__ inline void masked_grad(const FsBnBwdArgs& p, const T* dout, const T* yv, long m, int c, float* g) {
  constexpr int V = VecN<T>::N;
  const long hw = (long)p.H * p.W;
  if (!p.fold && p.gW == p.C && p.gH == (long)p.W * p.C && p.gN == hw * p.C &&
      (!p.relu || (p.yW == p.C && p.yH == (long)p.W * p.C && p.yN == hw * p.C))) {      // dense: no (n, h, w)
    loadv<T>(dout + m * p.C + c, g);
    if (p.relu) {
      float yy[V];
      loadv<T>(yv + m * p.C + c, yy);
#pragma unroll
      for (int j = 0; j < V; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
    }
    return;
  }
  int w, h; long n;
  split_row(m, p.H, p.W, n, h, w);
  load_dout<T>(p, dout, n, h, w, c, g);
  if (p.relu) {
    float yy[V];
    loadv<T>(yv + n * p.yN + (long)h * p.yH + (long)w * p.yW + c, yy);
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
  }
}


// POOL mode (FsBnBwdArgs.pool_dy): the gradient w.r.t. relu(bn(x)) at pixel m = (n, h, w) is not a tensor — it is the
// max-pool backward of the pooled gradient (the <= 4 windows that contain the pixel and whose argmax code names it;
// resnet.py:206, MaxPool2d(3, 2, 1) on an even H x W) plus the dense gradient `dout` of the un-pooled feature (NULL:
// none), gathered here instead of being written out by fs_maxpool_bwd and read back twice.
template <typename T>
__device__ inline void pool_grad(const FsBnBwdArgs& p, long m, int c, float* g) {
  constexpr int V = VecN<T>::N;
  const int C = p.C, H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
  const unsigned u = (unsigned)m, q = u / (unsigned)W, nn = q / (unsigned)H;
  const int w = (int)(u - q * W), h = (int)(q - nn * H);
  // h = 2 ho - 1 + r: the window row ho = h >> 1 always holds the pixel (r = 1 for even h, 2 for odd h); for odd h the
  // row below does too (ho + 1, r = 0) unless it is past the last one.  Columns alike.  All eight loads are issued
  // unconditionally (an absent window re-reads the first one and is not counted): no branch sits between them.
  const int h2 = h >> 1, w2 = w >> 1;
  const int r0 = (h & 1) ? 2 : 1, s0 = (w & 1) ? 2 : 1;
  const bool vh = (h & 1) && h2 + 1 < Ho, vw = (w & 1) && w2 + 1 < Wo;
  const long base = (((long)nn * Ho + h2) * Wo + w2) * C + c;
  const long dh = vh ? (long)Wo * C : 0, dw = vw ? C : 0;
  const long off[4] = {base, base + dw, base + dh, base + dh + dw};
  const int code[4] = {r0 * 3 + s0, r0 * 3, s0, 0};
  const bool ok[4] = {true, vw, vh, vh && vw};
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.pool_dy);
  uint32_t packed[4][V / 4];
  float t[4][V];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int e = 0; e < V / 4; ++e) packed[k][e] = reinterpret_cast<const uint32_t*>(p.pool_idx + off[k])[e];
    loadv<T>(dy + off[k], t[k]);
  }
  if (p.dout) loadv<T>(reinterpret_cast<const T*>(p.dout) + m * C + c, g);
  else {
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j)
      if (ok[k] && (int)((packed[k][j >> 2] >> (8 * (j & 3))) & 0xff) == code[k]) g[j] += t[k][j];
}
// ... and its ReLU mask: from the stored activation when there is one, else the sign of the forward's own expression
// scale * x + shift on the raw convolution output x (sc = gamma * invstd, sh = beta - mean * sc: bn_channel_coeffs)
template <typename T>
__device__ inline void pool_mask(const FsBnBwdArgs& p, const T* yv, long m, int c, const float* xr, const float* sc,
                                 const float* sh, float* g) {
  constexpr int V = VecN<T>::N;
  if (!p.relu) return;
  if (yv) {
    float yy[V];
    loadv<T>(yv + m * p.C + c, yy);
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) g[j] = (xr[j] * sc[j] + sh[j]) > 0.f ? g[j] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// POOL mode by 2 x 2 pixel quads.  The per-pixel gather above (pool_grad) decodes four windows per pixel — two integer
// divisions, eight gather loads and 128 compare / select operations per 16-byte vector: the stem's two backward passes ran
// at 1.3-2.5 TB/s, bound by vector-ALU issue (ResNet-18, 36 images: 245 us of the step).  The quad (2a, 2b) .. (2a+1, 2b+1)
// is reached by exactly the windows (a, b), (a, b+1), (a+1, b), (a+1, b+1): four window loads serve four pixels (nine
// window-pixel matches instead of sixteen), the index arithmetic is per quad, and a quad IS a pooled position, so the
// thread index decodes with the pooled sizes.  Same additions in the same order per pixel as pool_grad.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pool_quad(const FsBnBwdArgs& p, long nn, int a, int b, int c, float g[4][VecN<T>::N]) {
  constexpr int V = VecN<T>::N;
  const int C = p.C, H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
  const bool v01 = b + 1 < Wo, v10 = a + 1 < Ho;
  const long base = ((nn * Ho + a) * Wo + b) * C + c;
  const long o01 = v01 ? C : 0, o10 = v10 ? (long)Wo * C : 0;
  const long off[4] = {base, base + o01, base + o10, base + o10 + o01};
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.pool_dy);
  uint32_t pk[4][V / 4];
  float t[4][V];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int e = 0; e < V / 4; ++e) pk[k][e] = reinterpret_cast<const uint32_t*>(p.pool_idx + off[k])[e];
    loadv<T>(dy + off[k], t[k]);
  }
  if (p.dout) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      loadv<T>(reinterpret_cast<const T*>(p.dout) + ((nn * H + 2 * a + (k >> 1)) * W + 2 * b + (k & 1)) * C + c, g[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) g[k][j] = 0.f;
  }
  // pixel (dh, dw) at window (a + i, b + j) sits at row 2 dh' .. : code = r * 3 + s with r = dh + 1 - 2 i, s = dw + 1 - 2 j
  auto add = [&](int pix, int win, int code, bool ok) {
#pragma unroll
    for (int j = 0; j < V; ++j)
      if (ok && (int)((pk[win][j >> 2] >> (8 * (j & 3))) & 0xff) == code) g[pix][j] += t[win][j];
  };
  add(0, 0, 4, true);
  add(1, 0, 5, true); add(1, 1, 3, v01);
  add(2, 0, 7, true); add(2, 2, 1, v10);
  add(3, 0, 8, true); add(3, 1, 6, v01); add(3, 2, 2, v10); add(3, 3, 0, v01 && v10);
}

// pass 1 of POOL mode: a thread owns one 16-byte channel lane of one quad per iteration
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const FsDual<FsBnBwdArgs, FsNoGeom> d) {
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsBnBwdArgs& p = d.a[prob];
  constexpr int V = VecN<T>::N;
  __shared__ float red[2][V][256];
  const int C = p.C, CG = C / V, PL = 256 / CG;          // CG: a power of two <= 32 (host)
  const int cgl = threadIdx.x % CG, pl = threadIdx.x / CG, c = cgl * V;
  const T* __restrict__ yv = reinterpret_cast<const T*>(p.y);
  const T* __restrict__ xv = reinterpret_cast<const T*>(p.x);
  const int G = p.groups > 1 ? p.groups : 1;
  const int z = (int)blockIdx.z - (prob ? d.nb0 : 0);
  const int H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
  const unsigned Qg = (unsigned)(p.M / G) / 4;           // quads of a statistics group (whole images)
  float mean[V], istd[V], sc[V], sh[V], s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    mean[j] = p.save_mean[z * C + c + j]; istd[j] = p.save_invstd[z * C + c + j];
    sc[j] = p.gamma[c + j] * istd[j];
    sh[j] = (p.beta ? p.beta[c + j] : 0.f) - mean[j] * sc[j];
    s1[j] = 0.f; s2[j] = 0.f;
  }
  for (unsigned q = blockIdx.x * PL + pl; q < Qg; q += gridDim.x * PL) {
    const unsigned qq = (unsigned)z * Qg + q, r = qq / (unsigned)Wo, nn = r / (unsigned)Ho;
    const int b = (int)(qq - r * Wo), a = (int)(r - nn * Ho);
    float g[4][V], x[4][V];
#pragma unroll
    for (int k = 0; k < 4; ++k) loadv<T>(xv + (((long)nn * H + 2 * a + (k >> 1)) * W + 2 * b + (k & 1)) * C + c, x[k]);
    pool_quad<T>(p, nn, a, b, c, g);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pool_mask<T>(p, yv, ((long)nn * H + 2 * a + (k >> 1)) * W + 2 * b + (k & 1), c, x[k], sc, sh, g[k]);
#pragma unroll
      for (int j = 0; j < V; ++j) { s1[j] += g[k][j]; s2[j] += g[k][j] * (x[k][j] - mean[j]) * istd[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { red[0][j][threadIdx.x] = s1[j]; red[1][j][threadIdx.x] = s2[j]; }
  __syncthreads();
  if (pl < 2 * V && pl < PL) {
    for (int kj = pl; kj < 2 * V; kj += PL) {
      const int kind = kj / V, j = kj % V;
      float acc = 0.f;
      for (int k = 0; k < PL; ++k) acc += red[kind][j][k * CG + cgl];
      double* sl = p.sums + ((long)z * FS_STAT_SLOTS + blockIdx.x % FS_STAT_SLOTS) * 2 * C;
      atomicAdd(sl + kind * C + c + j, (double)acc);
    }
  }
}

// pass 2 of POOL mode
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const FsDual<FsBnBwdArgs, FsNoGeom> d) {
  extern __shared__ float bn_smem[];
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsBnBwdArgs& p = d.a[prob];
  constexpr int V = VecN<T>::N;
  const int C = p.C, CG = C / V, PL = 256 / CG;
  float* s_a = bn_smem; float* s_b = bn_smem + C; float* s_k = bn_smem + 2 * C;
  float* s_mean = bn_smem + 3 * C; float* s_istd = bn_smem + 4 * C; float* s_sh = bn_smem + 5 * C;
  const int G = p.groups > 1 ? p.groups : 1;
  const int z = (int)blockIdx.z - (prob ? d.nb0 : 0);
  const double* sums = p.sums + (long)z * FS_STAT_SLOTS * 2 * C;
  const double* sums_local = p.sums_local ? p.sums_local + (long)z * FS_STAT_SLOTS * 2 * C : nullptr;
  for (int c = threadIdx.x; c < C; c += 256) {
    double sg = 0.0, sgx = 0.0, lg = 0.0, lgx = 0.0;
#pragma unroll
    for (int k = 0; k < FS_STAT_SLOTS; ++k) {
      sg += sums[(long)k * 2 * C + c]; sgx += sums[(long)k * 2 * C + C + c];
      if (sums_local) { lg += sums_local[(long)k * 2 * C + c]; lgx += sums_local[(long)k * 2 * C + C + c]; }
    }
    const float istd = p.save_invstd[z * C + c];
    s_mean[c] = p.save_mean[z * C + c]; s_istd[c] = istd;
    s_k[c] = p.gamma[c] * istd;
    s_sh[c] = (p.beta ? p.beta[c] : 0.f) - s_mean[c] * s_k[c];     // the forward's shift
    s_a[c] = (float)(sg / p.count);
    s_b[c] = (float)(sgx / p.count);
    if (blockIdx.x == 0) {
      if (G > 1) {
        if (p.dgamma) atomicAdd(p.dgamma + c, (float)(sums_local ? lgx : sgx));
        if (p.dbeta) atomicAdd(p.dbeta + c, (float)(sums_local ? lg : sg));
      } else {
        if (p.dgamma) p.dgamma[c] += (float)(sums_local ? lgx : sgx);
        if (p.dbeta) p.dbeta[c] += (float)(sums_local ? lg : sg);
      }
    }
  }
  __syncthreads();
  const int cgl = threadIdx.x % CG, pl = threadIdx.x / CG, c = cgl * V;
  const T* __restrict__ yv = reinterpret_cast<const T*>(p.y);
  const T* __restrict__ xv = reinterpret_cast<const T*>(p.x);
  T* __restrict__ dx = reinterpret_cast<T*>(p.dx);
  const int H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1;
  const unsigned Qg = (unsigned)(p.M / G) / 4;
  for (unsigned q = blockIdx.x * PL + pl; q < Qg; q += gridDim.x * PL) {
    const unsigned qq = (unsigned)z * Qg + q, r = qq / (unsigned)Wo, nn = r / (unsigned)Ho;
    const int b = (int)(qq - r * Wo), a = (int)(r - nn * Ho);
    float g[4][V], x[4][V];
#pragma unroll
    for (int k = 0; k < 4; ++k) loadv<T>(xv + (((long)nn * H + 2 * a + (k >> 1)) * W + 2 * b + (k & 1)) * C + c, x[k]);
    pool_quad<T>(p, nn, a, b, c, g);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long m = ((long)nn * H + 2 * a + (k >> 1)) * W + 2 * b + (k & 1);
      pool_mask<T>(p, yv, m, c, x[k], s_k + c, s_sh + c, g[k]);
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (x[k][j] - s_mean[c + j]) * s_istd[c + j];
        o[j] = s_k[c + j] * (g[k][j] - s_a[c + j] - xh * s_b[c + j]);
      }
      storev<T>(dx + m * C + c, o);
    }
  }
}

// pass 1: per-channel sum(g), sum(g * xhat) with g = dout * (y > 0).  A block covers CGB channel groups
// (16-byte lanes) x PL pixel lanes; two rows per iteration keep more loads in flight.
template <typename T, int CGB, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const FsDual<FsBnBwdArgs, FsNoGeom> d, const int fast) {
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsBnBwdArgs& p = d.a[prob];
  constexpr int V = VecN<T>::N;
  constexpr int PL = 256 / CGB;
  __shared__ float red[2][V][256];
  const int C = p.C, CG = C / V;
  const int cgl = threadIdx.x % CGB, pl = threadIdx.x / CGB;
  const int cg = blockIdx.y * CGB + cgl;
  const bool act = cg < CG;
  const int c = cg * V;
  const T* __restrict__ dout = reinterpret_cast<const T*>(p.dout);
  const T* __restrict__ yv = reinterpret_cast<const T*>(p.y);
  const T* __restrict__ xv = reinterpret_cast<const T*>(p.x);
  const int G = p.groups > 1 ? p.groups : 1;
  const int z = (int)blockIdx.z - (prob ? d.nb0 : 0);
  const long Mg = p.M / G, m_end = (z + 1) * Mg;
  float mean[V], istd[V], s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { mean[j] = act ? p.save_mean[z * C + c + j] : 0.f; istd[j] = act ? p.save_invstd[z * C + c + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; }
  if constexpr (POOL) {
    if (act) {
      float sc[V], sh[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        sc[j] = p.gamma[c + j] * istd[j];
        sh[j] = (p.beta ? p.beta[c + j] : 0.f) - mean[j] * sc[j];
      }
      const long stride = (long)gridDim.x * PL;
      long m = z * Mg + (long)blockIdx.x * PL + pl;
      for (; m + stride < m_end; m += 2 * stride) {          // two pixels' loads in flight
        float g0[V], x0[V], g1[V], x1[V];
        loadv<T>(xv + m * C + c, x0);
        loadv<T>(xv + (m + stride) * C + c, x1);
        pool_grad<T>(p, m, c, g0);
        pool_grad<T>(p, m + stride, c, g1);
        pool_mask<T>(p, yv, m, c, x0, sc, sh, g0);
        pool_mask<T>(p, yv, m + stride, c, x1, sc, sh, g1);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          s1[j] += g0[j] + g1[j];
          s2[j] += g0[j] * (x0[j] - mean[j]) * istd[j] + g1[j] * (x1[j] - mean[j]) * istd[j];
        }
      }
      for (; m < m_end; m += stride) {
        float g0[V], x0[V];
        loadv<T>(xv + m * C + c, x0);
        pool_grad<T>(p, m, c, g0);
        pool_mask<T>(p, yv, m, c, x0, sc, sh, g0);
#pragma unroll
        for (int j = 0; j < V; ++j) { s1[j] += g0[j]; s2[j] += g0[j] * (x0[j] - mean[j]) * istd[j]; }
      }
    }
  } else
  if (act) {
    const long stride = (long)gridDim.x * PL;
    long m = z * Mg + (long)blockIdx.x * PL + pl;
    const long hw = (long)p.H * p.W;
    if (fast && !p.fold && p.gW == C && p.gH == (long)p.W * C && p.gN == hw * C &&
        (!p.relu || (p.yW == C && p.yH == (long)p.W * C && p.yN == hw * C))) {
      // dense tensors: raw operands of two later rows in flight while the current row is summed (bn_bwd_apply_kernel)
      const bool relu = p.relu != 0;
      long left = m < m_end ? (m_end - m + stride - 1) / stride : 0;
      long off = m * C + c;
      const long step = stride * C;
      uint4 d0, x0, y0, d1, x1, y1, d2, x2, y2;
      d0 = x0 = y0 = d1 = x1 = y1 = d2 = x2 = y2 = make_uint4(0, 0, 0, 0);
      auto req = [&](long o, uint4& dd, uint4& xx, uint4& yy) {
        dd = *reinterpret_cast<const uint4*>(dout + o); xx = *reinterpret_cast<const uint4*>(xv + o);
        if (relu) yy = *reinterpret_cast<const uint4*>(yv + o);
      };
      if constexpr (sizeof(T) == 2) {
        if (left > 0) req(off, d0, x0, y0);
        if (left > 1) req(off + step, d1, x1, y1);
        if (left > 2) req(off + 2 * step, d2, x2, y2);
        while (left > 0) {
          float g[V], xr[V];
          Unit<T>::unpack(d0, g); Unit<T>::unpack(x0, xr);
          if (relu) {
            float yy[V];
            Unit<T>::unpack(y0, yy);
#pragma unroll
            for (int j = 0; j < V; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
          }
          d0 = d1; x0 = x1; y0 = y1; d1 = d2; x1 = x2; y1 = y2;
          --left; off += step;
          if (left > 2) req(off + 2 * step, d2, x2, y2);
#pragma unroll
          for (int j = 0; j < V; ++j) { s1[j] += g[j]; s2[j] += g[j] * (xr[j] - mean[j]) * istd[j]; }
        }
        m = m_end;       // (nothing left for the generic loops below)
      }
    }
    for (; m + stride < m_end; m += 2 * stride) {
      float g0[V], g1[V], x0[V], x1[V];
      masked_grad<T>(p, dout, yv, m, c, g0);
      masked_grad<T>(p, dout, yv, m + stride, c, g1);
      loadv<T>(xv + m * C + c, x0);
      loadv<T>(xv + (m + stride) * C + c, x1);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s1[j] += g0[j] + g1[j];
        s2[j] += g0[j] * (x0[j] - mean[j]) * istd[j] + g1[j] * (x1[j] - mean[j]) * istd[j];
      }
    }
    for (; m < m_end; m += stride) {
      float g0[V], x0[V];
      masked_grad<T>(p, dout, yv, m, c, g0);
      loadv<T>(xv + m * C + c, x0);
#pragma unroll
      for (int j = 0; j < V; ++j) { s1[j] += g0[j]; s2[j] += g0[j] * (x0[j] - mean[j]) * istd[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { red[0][j][threadIdx.x] = s1[j]; red[1][j][threadIdx.x] = s2[j]; }
  __syncthreads();
  // threads (pl < 2*V) of each channel group finish one (kind, j) pair each
  if (act && pl < 2 * V && pl < PL) {
    for (int kj = pl; kj < 2 * V; kj += PL) {
      int kind = kj / V, j = kj % V;
      float a = 0.f;
      for (int k = 0; k < PL; ++k) a += red[kind][j][k * CGB + cgl];
      double* sl = p.sums + ((long)z * FS_STAT_SLOTS + blockIdx.x % FS_STAT_SLOTS) * 2 * C;
      atomicAdd(sl + kind * C + c + j, (double)a);
    }
  }
}

// pass 2: dx = gamma*invstd * (g - sum_g/count - xhat * sum_gx/count); optional g output; param grads
template <typename T, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const FsDual<FsBnBwdArgs, FsNoGeom> d, const int fast) {
  extern __shared__ float bn_smem[];
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsBnBwdArgs& p = d.a[prob];
  const int C = p.C;
  constexpr int V = VecN<T>::N;
  // channel slabs as in bn_apply_kernel (blockIdx.y: 32 channel groups of a row wider than that)
  const int CG = C / V, cg_sh = pow2_shift(CG);
  const int SCG = gridDim.y > 1 ? 32 : CG, scg_sh = gridDim.y > 1 ? 5 : cg_sh;
  const int SC = SCG * V, c0 = (int)blockIdx.y * SC;
  float* s_a = bn_smem; float* s_b = bn_smem + SC; float* s_k = bn_smem + 2 * SC;
  float* s_mean = bn_smem + 3 * SC; float* s_istd = bn_smem + 4 * SC;
  const int G = p.groups > 1 ? p.groups : 1;
  const int z = (int)blockIdx.z - (prob ? d.nb0 : 0);
  const int Mg = p.M / G;
  const double* sums = p.sums + (long)z * FS_STAT_SLOTS * 2 * C;
  const double* sums_local = p.sums_local ? p.sums_local + (long)z * FS_STAT_SLOTS * 2 * C : nullptr;
  // first vector requested before the coefficient preamble (see bn_apply_kernel)
  const T* __restrict__ dout = reinterpret_cast<const T*>(p.dout);
  const T* __restrict__ yv = reinterpret_cast<const T*>(p.y);
  const T* __restrict__ xv = reinterpret_cast<const T*>(p.x);
  const long total = (long)Mg * SCG;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  float pg[V], px[V];
  if (i0 < total) {
    int cg; long m;
    split_vec(i0, SCG, scg_sh, cg, m);
    m += (long)z * Mg;
    if constexpr (POOL) pool_grad<T>(p, m, c0 + cg * V, pg);   // (masked below, once the coefficients are there)
    else masked_grad<T>(p, dout, yv, m, c0 + cg * V, pg);
    loadv<T>(xv + m * C + c0 + cg * V, px);
  }
  for (int cl = threadIdx.x; cl < SC; cl += 256) {
    const int c = c0 + cl;
    double sg = 0.0, sgx = 0.0, lg = 0.0, lgx = 0.0;
#pragma unroll
    for (int k = 0; k < FS_STAT_SLOTS; ++k) {
      sg += sums[(long)k * 2 * C + c]; sgx += sums[(long)k * 2 * C + C + c];
      if (sums_local) { lg += sums_local[(long)k * 2 * C + c]; lgx += sums_local[(long)k * 2 * C + C + c]; }
    }
    float istd = p.save_invstd[z * C + c];
    s_mean[cl] = p.save_mean[z * C + c]; s_istd[cl] = istd;
    s_k[cl] = p.gamma[c] * istd;
    if constexpr (POOL) bn_smem[5 * SC + cl] = (p.beta ? p.beta[c] : 0.f) - s_mean[cl] * s_k[cl];     // the forward's shift
    s_a[cl] = (float)(sg / p.count);
    s_b[cl] = (float)(sgx / p.count);
    if (blockIdx.x == 0) {
      // dgamma / dbeta of the local shard (the data-parallel all-reduce of gradients averages them later)
      if (G > 1) {       // the groups' blocks add concurrently
        if (p.dgamma) atomicAdd(p.dgamma + c, (float)(sums_local ? lgx : sgx));
        if (p.dbeta) atomicAdd(p.dbeta + c, (float)(sums_local ? lg : sg));
      } else {
        if (p.dgamma) p.dgamma[c] += (float)(sums_local ? lgx : sgx);
        if (p.dbeta) p.dbeta[c] += (float)(sums_local ? lg : sg);
      }
    }
  }
  __syncthreads();
  T* __restrict__ dx = reinterpret_cast<T*>(p.dx);
  T* __restrict__ gout = reinterpret_cast<T*>(p.g_out);
  const long stride = (long)gridDim.x * 256;
  if constexpr (!POOL) {
    // Dense tensors (every ResNet layer): the thread's channel group is fixed, its rows advance by a constant — raw 16-byte
    // operands of TWO later rows stay in flight while the current one is processed (one row ahead left a wave with one
    // load per operand in flight: 2.5 TB/s on the 168 MB tensors of ResNet-50 layer 1 where a copy runs at 5.1)
    const long hw = (long)p.H * p.W;
    const bool plain = fast && !p.fold && scg_sh >= 0 && p.gW == C && p.gH == (long)p.W * C && p.gN == hw * C &&
                       (!p.relu || (p.yW == C && p.yH == (long)p.W * C && p.yN == hw * C));
    if (plain) {
      if (i0 >= total) return;
      const int cl = (int)(i0 & (SCG - 1)) * V;
      const long rows = stride >> scg_sh, step = rows * C;          // stride is a multiple of the (power-of-two) group count
      long off = ((long)z * Mg + (i0 >> scg_sh)) * C + c0 + cl;
      long left = (total - i0 + stride - 1) / stride;               // vectors of this thread
      const bool relu = p.relu != 0;
      float ka[V], kb[V], kk[V], km[V], ki[V];
#pragma unroll
      for (int j = 0; j < V; ++j) { ka[j] = s_a[cl + j]; kb[j] = s_b[cl + j]; kk[j] = s_k[cl + j]; km[j] = s_mean[cl + j]; ki[j] = s_istd[cl + j]; }
      uint4 d1, x1, y1, d2, x2, y2;
      d1 = x1 = y1 = d2 = x2 = y2 = make_uint4(0, 0, 0, 0);
      if (left > 1) {
        d1 = *reinterpret_cast<const uint4*>(dout + off + step); x1 = *reinterpret_cast<const uint4*>(xv + off + step);
        if (relu) y1 = *reinterpret_cast<const uint4*>(yv + off + step);
      }
      if (left > 2) {
        d2 = *reinterpret_cast<const uint4*>(dout + off + 2 * step); x2 = *reinterpret_cast<const uint4*>(xv + off + 2 * step);
        if (relu) y2 = *reinterpret_cast<const uint4*>(yv + off + 2 * step);
      }
      // (the first vector was requested before the preamble as floats: pg is already masked)
      float g[V], xr[V], o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) { g[j] = pg[j]; xr[j] = px[j]; }
      for (;;) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float xh = (xr[j] - km[j]) * ki[j];
          o[j] = kk[j] * (g[j] - ka[j] - xh * kb[j]);
        }
        storev<T>(dx + off, o);
        if (gout) storev<T>(gout + off, g);
        if (--left == 0) break;
        off += step;
        Unit<T>::unpack(d1, g); Unit<T>::unpack(x1, xr);
        if (relu) {
          float yy[V];
          Unit<T>::unpack(y1, yy);
#pragma unroll
          for (int j = 0; j < V; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
        }
        d1 = d2; x1 = x2; y1 = y2;
        if (left > 2) {
          d2 = *reinterpret_cast<const uint4*>(dout + off + 2 * step); x2 = *reinterpret_cast<const uint4*>(xv + off + 2 * step);
          if (relu) y2 = *reinterpret_cast<const uint4*>(yv + off + 2 * step);
        }
      }
      return;
    }
  }
  for (long i = i0; i < total; i += stride) {
    int cg; long m;
    split_vec(i, SCG, scg_sh, cg, m);
    m += (long)z * Mg;
    const int cl = cg * V, c = c0 + cl;
    float g[V], xr[V], o[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { g[j] = pg[j]; xr[j] = px[j]; }
    if (i + stride < total) {            // next vector in flight (see bn_apply_kernel)
      int cg2; long m2;
      split_vec(i + stride, SCG, scg_sh, cg2, m2);
      m2 += (long)z * Mg;
      if constexpr (POOL) pool_grad<T>(p, m2, c0 + cg2 * V, pg);
      else masked_grad<T>(p, dout, yv, m2, c0 + cg2 * V, pg);
      loadv<T>(xv + m2 * C + c0 + cg2 * V, px);
    }
    if constexpr (POOL) pool_mask<T>(p, yv, m, c, xr, s_k + cl, bn_smem + 5 * SC + cl, g);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float xh = (xr[j] - s_mean[cl + j]) * s_istd[cl + j];
      o[j] = s_k[cl + j] * (g[j] - s_a[cl + j] - xh * s_b[cl + j]);
    }
    storev<T>(dx + m * C + c, o);
    if (gout) storev<T>(gout + m * C + c, g);
  }
}

// channel slabs of the element-wise passes: 32 16-byte groups each where a row has more (a power of two of them)
int bn_slabs(int CG) { return (CG > 32 && (CG & (CG - 1)) == 0) ? CG / 32 : 1; }

// POOL mode by quads: power-of-two lane counts up to 32 (the stems: 8 at bf16); other widths gather per pixel
bool bn_pool_quads(int CG) { return CG >= 1 && CG <= 32 && (CG & (CG - 1)) == 0; }

int grid_for(long items) {
  long b = (items + 255) / 256;
  // ~2 blocks per CU, each a walk over many rows: the per-block coefficient preamble (8-64 KB of f64 sums re-read by EVERY
  // block) is then paid 512 times per launch instead of once per 256 vectors — with one vector per thread it moved more
  // bytes than the tensor on the deep layers.  Measured on the step with the generic loop (cap 4096 / 1536 / 1024 / 768 /
  // 512): ResNet-18 B=12 5.75 / 5.71 / 5.65 / 5.66 / 5.67 ms, ResNet-50 @320x1024 27.7 / 26.1 / 25.8 / 25.8 / 25.8; again with
  // the dense fast path, whose threads keep two rows in flight (cap 2048 / 1024 / 768 / 512 / 384 / 256, same box each):
  // ResNet-18 5.67 / 5.62 / - / 5.54 and 5.76 / 5.72 / 5.74 / 5.77, ResNet-50 20.65 / 20.09 / - / 19.89 and
  // 20.44 / 20.41 / 20.43 / 20.58.
  const long cap = 512;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int fs_bn_finalize(const FsBnApplyArgs* a, float* scale, float* shift, void* stream) {
  if (!a || !a->gamma || !a->beta || !a->save_mean || !a->save_invstd || !scale || !shift) return FS_EINVAL;
  if (!a->stats && (!a->running_mean || !a->running_var)) return FS_EINVAL;
  if (a->C <= 0 || a->C > MAXC || a->gamma2) return FS_EINVAL;
  const int G = a->groups > 1 ? a->groups : 1;
  if (G > 1 && !a->stats) return FS_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((a->C + 255) / 256, G), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     *a, scale, shift);
  return fs_launch_status();
}

namespace {
int bn_apply_check(const FsBnApplyArgs* a) {
  if (!a || !a->x || (!a->y && !a->pool_y) || !a->gamma || !a->beta || !a->save_mean || !a->save_invstd) return FS_EINVAL;
  if (a->pool_y) {
    // fused max-pool: train mode, dense output, even H x W, no residual; the element index is 32-bit
    if (!a->pool_idx || !a->stats || a->res || a->gamma2 || a->pad_out || a->H <= 0 || a->W <= 0 || (a->H & 1) || (a->W & 1))
      return FS_EINVAL;
    if (a->M % (a->H * a->W) != 0 || (long)a->M * (a->C / 4) >= 0x7fffffffL) return FS_EINVAL;
    if (a->y && !(a->yW == a->C && a->yH == (long)a->W * a->C && a->yN == (long)a->H * a->W * a->C)) return FS_EINVAL;
  }
  if (!a->stats && (!a->running_mean || !a->running_var)) return FS_EINVAL;
  if (a->C % 8 != 0 || a->C > MAXC || a->M <= 0) return FS_EINVAL;
  if (a->gamma2 && (!a->res || !a->beta2 || !a->save_mean2 || !a->save_invstd2)) return FS_EINVAL;
  if (a->gamma2 && !a->stats2 && (!a->running_mean2 || !a->running_var2)) return FS_EINVAL;
  const int G = a->groups > 1 ? a->groups : 1;
  if (a->M % G != 0 || (G > 1 && !a->stats)) return FS_EINVAL;
  return FS_OK;
}
int bn_bwd_check(const FsBnBwdArgs* a, bool apply) {
  if (!a || (!a->dout && !a->pool_dy) || !a->x || !a->sums || !a->save_mean || !a->save_invstd) return FS_EINVAL;
  // common to both paths: channel tiling, the LDS table's extent, statistics groups, what the second pass writes
  if (apply && (!a->dx || !a->gamma)) return FS_EINVAL;
  if (a->C % 8 != 0 || a->C > MAXC || a->M <= 0) return FS_EINVAL;
  const int G = a->groups > 1 ? a->groups : 1;
  if (a->M % G != 0) return FS_EINVAL;
  if (a->pool_dy) {
    if (!a->pool_idx || !a->gamma || a->fold || a->g_out || a->H <= 0 || a->W <= 0 || (a->H & 1) || (a->W & 1)) return FS_EINVAL;
    if (a->relu && !a->y && !a->beta) return FS_EINVAL;
    const long hw = (long)a->H * a->W;
    // (a statistics group is a whole number of images: the quad mode walks 2x2 windows of one image)
    if (a->M % hw != 0 || (a->M / G) % hw != 0 || (long)a->M >= 0x7fffffffL) return FS_EINVAL;
    if (a->dout && !(a->gW == a->C && a->gH == (long)a->W * a->C && a->gN == hw * a->C)) return FS_EINVAL;
    if (a->y && !(a->yW == a->C && a->yH == (long)a->W * a->C && a->yN == hw * a->C)) return FS_EINVAL;
    return FS_OK;
  }
  if (a->relu && !a->y) return FS_EINVAL;
  return FS_OK;
}
template <typename A>
FsDual<A, FsNoGeom> bn_dual(const A* a, const A* b) {
  FsDual<A, FsNoGeom> d;
  d.a[0] = *a; d.a[1] = b ? *b : *a;
  d.g[0].unused = d.g[1].unused = 0;
  d.nb0 = a->groups > 1 ? a->groups : 1;
  d.nprob = b ? 2 : 1;
  return d;
}
}  // namespace

// b != NULL: a second BatchNorm of the same width in the same launch (its statistics groups follow a's along blockIdx.z)
extern "C" int fs_bn_apply2(const FsBnApplyArgs* a, const FsBnApplyArgs* b, int dtype, void* stream) {
  int r = bn_apply_check(a);
  if (r != FS_OK) return r;
  if (b) {
    r = bn_apply_check(b);
    if (r != FS_OK) return r;
    if (b->C != a->C || (a->pool_y != nullptr) != (b->pool_y != nullptr) || (a->pool_y && (a->H != b->H || a->W != b->W))) {
      r = fs_bn_apply2(a, nullptr, dtype, stream);
      return r != FS_OK ? r : fs_bn_apply2(b, nullptr, dtype, stream);
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype == FS_DTYPE_BF16 ? 8 : 4;
  const int G = a->groups > 1 ? a->groups : 1, G1 = b ? (b->groups > 1 ? b->groups : 1) : 0;
  long items = (long)(a->M / G) * (a->C / vec);
  if (b) items = std::max(items, (long)(b->M / G1) * (b->C / vec));
  const FsDual<FsBnApplyArgs, FsNoGeom> d = bn_dual(a, b);
  if (a->pool_y) {
    // one thread per pooled pixel and 16-byte channel lane (nine window loads each): twice the element-wise passes' cap
    dim3 pgrid(std::min(2 * grid_for(items / 4), (int)((items / 4 + 255) / 256)), 1, G + G1);
    if (pgrid.x < 1) pgrid.x = 1;
    const unsigned plds = 2u * a->C * sizeof(float);
    if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(bn_apply_pool_kernel<bf16>, pgrid, dim3(256), plds, st, d);
    else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(bn_apply_pool_kernel<float>, pgrid, dim3(256), plds, st, d);
    else return FS_EINVAL;
    return fs_launch_status();
  }
  const int nslab = bn_slabs(a->C / vec);
  dim3 grid(std::min(grid_for(items / nslab), std::max(32, grid_for(1L << 40) / nslab)), nslab, G + G1);   // (the cap holds per launch)
  const unsigned lds = ((a->gamma2 || (b && b->gamma2)) ? 4u : 2u) * (a->C / nslab) * sizeof(float);
  const int fast = 1;       // dense tensors take the kernels' fast path (the generic loop serves strided / padded ones)
  if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16>, grid, dim3(256), lds, st, d, fast);
  else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), lds, st, d, fast);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_bn_apply(const FsBnApplyArgs* a, int dtype, void* stream) { return fs_bn_apply2(a, nullptr, dtype, stream); }

extern "C" int fs_bn_bwd_reduce2(const FsBnBwdArgs* a, const FsBnBwdArgs* b, int dtype, void* stream) {
  int r = bn_bwd_check(a, false);
  if (r != FS_OK) return r;
  if (b) {
    r = bn_bwd_check(b, false);
    if (r != FS_OK) return r;
    if (b->C != a->C || (a->pool_dy != nullptr) != (b->pool_dy != nullptr) || (a->pool_dy && (a->H != b->H || a->W != b->W))) {
      r = fs_bn_bwd_reduce2(a, nullptr, dtype, stream);
      return r != FS_OK ? r : fs_bn_bwd_reduce2(b, nullptr, dtype, stream);
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int CG = a->C / (dtype == FS_DTYPE_BF16 ? 8 : 4);
  const int G = a->groups > 1 ? a->groups : 1, G1 = b ? (b->groups > 1 ? b->groups : 1) : 0;
  long Mg = a->M / G;
  if (b) Mg = std::max<long>(Mg, b->M / G1);
  const FsDual<FsBnBwdArgs, FsNoGeom> d = bn_dual(a, b);
  if (a->pool_dy && bn_pool_quads(CG)) {
    long Qg = (a->M / G) / 4;
    if (b) Qg = std::max<long>(Qg, (b->M / G1) / 4);
    dim3 grid((unsigned)std::min<long>((Qg + 256 / CG - 1) / (256 / CG), 1024), 1, G + G1);
    if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<bf16>, grid, dim3(256), 0, st, d);
    else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<float>, grid, dim3(256), 0, st, d);
    else return FS_EINVAL;
    return fs_launch_status();
  }
  // channel groups (16-byte lanes) per block: 32 for wide layers (8 pixel lanes), 8, or — for the 16 / 32-channel
  // decoder layers, whose 2 / 4 lanes would leave three quarters of an 8-lane block idle — 4 and 2
#define LAUNCH_REDUCE(CGB, ROWS_PER_BLOCK, MAXB)                                                               \
  {                                                                                                             \
    dim3 grid((unsigned)std::min<long>((Mg + (ROWS_PER_BLOCK) - 1) / (ROWS_PER_BLOCK), MAXB), (CG + CGB - 1) / CGB, G + G1); \
    if (a->pool_dy) {                                                                                           \
      if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16, CGB, true>), grid, dim3(256), 0, st, d, 0);  \
      else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, CGB, true>), grid, dim3(256), 0, st, d, 0); \
      else return FS_EINVAL;                                                                                    \
    } else                                                                                                      \
    if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16, CGB>), grid, dim3(256), 0, st, d, rfast);  \
    else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, CGB>), grid, dim3(256), 0, st, d, rfast); \
    else return FS_EINVAL;                                                                                      \
  }
  const int rfast = 1;
  // (wide rows: every block ends in 2 x 256 f64 atomics and a block-wide fold — hold the grid near 1024 blocks)
  const long wide_cap = std::max<long>(32, std::min<long>(512, 1024 / ((long)((CG + 31) / 32) * (G + G1))));
  if (CG >= 32) LAUNCH_REDUCE(32, 16, wide_cap)
  else if (CG >= 8) LAUNCH_REDUCE(8, 64, 1024)
  else if (CG >= 4) LAUNCH_REDUCE(4, 128, 1024)
  else LAUNCH_REDUCE(2, 256, 1024)
#undef LAUNCH_REDUCE
  return fs_launch_status();
}

extern "C" int fs_bn_bwd_reduce(const FsBnBwdArgs* a, int dtype, void* stream) { return fs_bn_bwd_reduce2(a, nullptr, dtype, stream); }

extern "C" int fs_bn_bwd_apply2(const FsBnBwdArgs* a, const FsBnBwdArgs* b, int dtype, void* stream) {
  int r = bn_bwd_check(a, true);
  if (r != FS_OK) return r;
  if (b) {
    r = bn_bwd_check(b, true);
    if (r != FS_OK) return r;
    if (b->C != a->C || (a->pool_dy != nullptr) != (b->pool_dy != nullptr) || (a->pool_dy && (a->H != b->H || a->W != b->W))) {
      r = fs_bn_bwd_apply2(a, nullptr, dtype, stream);
      return r != FS_OK ? r : fs_bn_bwd_apply2(b, nullptr, dtype, stream);
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int vec = dtype == FS_DTYPE_BF16 ? 8 : 4;
  const int G = a->groups > 1 ? a->groups : 1, G1 = b ? (b->groups > 1 ? b->groups : 1) : 0;
  long items = (long)(a->M / G) * (a->C / vec);
  if (b) items = std::max(items, (long)(b->M / G1) * (b->C / vec));
  const int nslab = a->pool_dy ? 1 : bn_slabs(a->C / vec);
  dim3 grid(std::min(grid_for(items / nslab), std::max(32, grid_for(1L << 40) / nslab)), nslab, G + G1);
  const unsigned lds = (a->pool_dy ? 6u : 5u) * (a->C / nslab) * sizeof(float);
  const FsDual<FsBnBwdArgs, FsNoGeom> d = bn_dual(a, b);
  const int fast = 1;       // dense tensors take the kernels' fast path (the generic loop serves strided / padded ones)
  if (a->pool_dy && bn_pool_quads(a->C / vec)) {
    const int CG = a->C / vec;
    long Qg = (a->M / G) / 4;
    if (b) Qg = std::max<long>(Qg, (b->M / G1) / 4);
    dim3 qgrid((unsigned)std::min<long>((Qg + 256 / CG - 1) / (256 / CG), 2048), 1, G + G1);
    if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(bn_bwd_apply_pool_kernel<bf16>, qgrid, dim3(256), lds, st, d);
    else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(bn_bwd_apply_pool_kernel<float>, qgrid, dim3(256), lds, st, d);
    else return FS_EINVAL;
    return fs_launch_status();
  }
  if (a->pool_dy) {
    if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, true>), grid, dim3(256), lds, st, d, 0);
    else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), grid, dim3(256), lds, st, d, 0);
    else return FS_EINVAL;
    return fs_launch_status();
  }
  if (dtype == FS_DTYPE_BF16) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, grid, dim3(256), lds, st, d, fast);
  else if (dtype == FS_DTYPE_F32) hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), lds, st, d, fast);
  else return FS_EINVAL;
  return fs_launch_status();
}

extern "C" int fs_bn_bwd_apply(const FsBnBwdArgs* a, int dtype, void* stream) { return fs_bn_bwd_apply2(a, nullptr, dtype, stream); }
