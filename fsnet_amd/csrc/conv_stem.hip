// 7x7 / stride-2 / pad-3 stem convolution (ResNet conv1) over 8-channel bf16 pixels, forward only.
//
// Replaces nn.Conv2d(3 | 6, 64, 7, stride=2, padding=3) of the two encoders (reference
// vision_base/networks/models/backbone/resnet.py:118-121, 204; images padded to 8 channels by fs_nchw_to_nhwc).
// The generic implicit-GEMM kernel gathers every 16-byte pixel once per tap: 49 x 16 B per output pixel, 330 MB into
// the CUs for the batch-12 depth stem — the layer ran at the CUs' fill rate (72 / 134 us), not at the 8 us its MFMAs
// need, and walking four taps per address computation changed nothing (rejected experiments).  Here
//   * a block keeps ALL weights in LDS (64 co x 49 taps x 16 B = 49 KB, loaded once) and is persistent over
//     8 x 16-pixel output tiles;
//   * per tile the 21 x 37 input patch (12 KB) is fetched once and the im2col happens in LDS: an MFMA K step is four
//     horizontally adjacent taps (r, 4j .. 4j+3) x 8 channels, lane (pixel li, k-group lg) reads patch pixel
//     (2y + r, 2x + 4j + lg); the non-existent tap s = 7 gets a zero weight fragment;
//   * BatchNorm statistics accumulate in registers over the block's tiles and leave as one f64 atomic per channel.
// Input bytes per output pixel drop from 784 to ~97.
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

constexpr int TY = 8, TX = 16;                 // output tile: 8 rows x 16 pixels
constexpr int PH = 2 * TY + 5, PW = 2 * TX + 6; // input patch rows / row pitch (one spare pixel: tap s = 7 of x = 15)
constexpr int CO = 64, TAPS = 49;

__device__ __forceinline__ uint4 stem_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

struct StemFwdGeom { int tiles_x, tiles_y, ntiles, tiles_per_block; FsDiv dTX, dTY, dIPG; };

__global__ __launch_bounds__(256) void conv_stem_kernel(const FsDual<FsConvArgs, StemFwdGeom> d) {
  // two problems per launch (fsnet_hip_internal.h, FsDual): the depth encoder's stem (3 real input channels) and the
  // stacked pose encoder's (6) are the same kernel on the same 8-channel pixels with different weights
  const int prob = (int)blockIdx.x >= d.nb0 ? 1 : 0;
  const FsConvArgs& p = d.a[prob];
  const int bid = (int)blockIdx.x - (prob ? d.nb0 : 0);
  const int tiles_x = d.g[prob].tiles_x, tiles_y = d.g[prob].tiles_y, ntiles = d.g[prob].ntiles;
  const int tiles_per_block = d.g[prob].tiles_per_block;
  const FsDiv dTX = d.g[prob].dTX, dTY = d.g[prob].dTY, dIPG = d.g[prob].dIPG;
  typedef bf16 T;
  __shared__ uint4 lds_w[CO * TAPS];
  __shared__ uint4 lds_x[PH * PW];
  constexpr int OOB = 0x7fffffff;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, lg = lane >> 4;

  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (int)p.wgt_bytes, 0x00020000);
  const int wrow_bytes = p.nchunks * p.kg * 16;          // packed forward operand: [co][tap][8 ch], K padded
  for (int i = t; i < CO * TAPS; i += 256) {
    const int co = i / TAPS, tap = i - co * TAPS;
    lds_w[i] = stem_load16(rs_wgt, co * wrow_bytes + tap * 16);
  }

  float s1[4][4], s2[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[c][j] = 0.f; s2[c][j] = 0.f; }
  int cur_group = -1;
  const int tile0 = bid * tiles_per_block;
  const int tile1 = min(tile0 + tiles_per_block, ntiles);

  auto flush_stats = [&](int group) {
    // [wave][co][2] through the patch buffer (free between tiles), then one f64 atomic per channel and moment
    __syncthreads();
    float* red = reinterpret_cast<float*>(&lds_x[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = row16_sum(s1[c][j]), w = row16_sum(s2[c][j]);
        if (li == 0) {
          const int co = c * 16 + lg * 4 + j;
          red[(wave * CO + co) * 2] = u; red[(wave * CO + co) * 2 + 1] = w;
        }
        s1[c][j] = 0.f; s2[c][j] = 0.f;
      }
    __syncthreads();
    if (t < CO && t < p.Co) {
      float u = 0.f, w = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { u += red[(k * CO + t) * 2]; w += red[(k * CO + t) * 2 + 1]; }
      double* sl = p.stats + ((long)group * FS_STAT_SLOTS + bid % FS_STAT_SLOTS) * 2 * p.Co;
      atomicAdd(sl + t, (double)u);
      atomicAdd(sl + p.Co + t, (double)w);
    }
    __syncthreads();
  };

  for (int tile = tile0; tile < tile1; ++tile) {
    const int q = fs_div(tile, dTX); const int tx_i = tile - q * tiles_x;
    const int n = fs_div(q, dTY); const int ty_i = q - n * tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    if (p.stats) {
      const int group = p.stat_group_rows > 0 ? fs_div(n, dIPG) : 0;
      if (group != cur_group) {
        if (cur_group >= 0) flush_stats(cur_group);
        cur_group = group;
      }
    }
    __syncthreads();                     // previous tile fully multiplied (and the weights landed, first time)
    for (int i = t; i < PH * PW; i += 256) {
      const int py = i / PW, px = i - py * PW;
      const int sy = 2 * y0 - 3 + py, sx = 2 * x0 - 3 + px;
      const bool ok = (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      lds_x[i] = stem_load16(rs_src, ok ? (int)(((long)n * p.sN + (long)sy * p.sH + (long)sx * p.sW) * 2) : OOB);
    }
    __syncthreads();

    f32x4 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[c][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int s = 4 * j + lg;
        uint4 fa[4], fb[2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint4 w = lds_w[(c * 16 + li) * TAPS + r * 7 + min(s, 6)];
          fa[c] = s < 7 ? w : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = lds_x[(2 * (2 * wave + b) + r) * PW + 2 * li + s];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[c][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[c]),
                                                                __builtin_bit_cast(bf16x8, fb[b]), acc[c][b], 0, 0, 0);
      }

    // D rows = co (lg*4 + j), cols = pixel li of tile row 2*wave + b
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int y = y0 + 2 * wave + b, x = x0 + li;
      if (y < p.Hd && x < p.Wd) {
        T* d = reinterpret_cast<T*>(p.dst) + (long)n * p.dN + (long)y * p.dH + (long)x * p.dW;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int co = c * 16 + lg * 4;
          if (co < p.Co) {
            float v[4] = {acc[c][b][0], acc[c][b][1], acc[c][b][2], acc[c][b][3]};
            store4<T>(d + co, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[c][j] += v[j]; s2[c][j] += v[j] * v[j]; }
          }
        }
      }
    }
  }
  if (p.stats && cur_group >= 0) flush_stats(cur_group);
}

}  // namespace

namespace {
int stem_check(const FsConvArgs* a, int dtype) {
  if (!a || !a->src || !a->wgt || !a->dst) return FS_EINVAL;
  if (dtype != FS_DTYPE_BF16 || a->Cs != 8 || a->Co != 64 || a->Co_p != 64) return FS_EINVAL;
  if (a->bias || a->addend || a->mask || a->bnb_x || a->relu || a->out_f32 || a->grp_imgs || a->ncls > 1) return FS_EINVAL;
  if (a->Hd != (a->Hs + 6 - 7) / 2 + 1 || a->Wd != (a->Ws + 6 - 7) / 2 + 1 || a->N <= 0) return FS_EINVAL;
  if (a->nchunks * a->kg * 8 < TAPS * 8) return FS_EINVAL;
  if (a->src_bytes <= 0 || a->src_bytes > 0x7fffffffLL || a->wgt_bytes <= 0 || a->wgt_bytes > 0x7fffffffLL)
    return FS_EINVAL;
  if ((long)a->N * ((a->Wd + TX - 1) / TX) * ((a->Hd + TY - 1) / TY) > 0x7fffffffL) return FS_EINVAL;
  if (a->stat_group_rows > 0 && a->stat_group_rows % ((long)a->Hd * a->Wd) != 0) return FS_EINVAL;   // whole images
  return FS_OK;
}
// tiles per persistent block when the launch holds `total` tiles: two blocks per CU (62 KB of LDS each), each walking a
// contiguous run of tiles with the weights resident
int stem_geom(const FsConvArgs& a, long total, StemFwdGeom& g) {
  g.tiles_x = (a.Wd + TX - 1) / TX; g.tiles_y = (a.Hd + TY - 1) / TY;
  g.ntiles = (int)((long)a.N * g.tiles_x * g.tiles_y);
  g.tiles_per_block = (int)std::max<long>(1, (total + 511) / 512);
  g.dTX = fs_make_div(g.tiles_x); g.dTY = fs_make_div(g.tiles_y);
  g.dIPG = a.stat_group_rows > 0 ? fs_make_div((int)(a.stat_group_rows / ((long)a.Hd * a.Wd))) : fs_make_div(1);
  return (g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
}
}  // namespace

// a1 != NULL: a second stem (its own weights, images and statistics) in the same launch
extern "C" int fs_conv_stem2(const FsConvArgs* a, const FsConvArgs* a1, int dtype, void* stream) {
  int r = stem_check(a, dtype);
  if (r != FS_OK) return r;
  if (a1 && (r = stem_check(a1, dtype)) != FS_OK) return r;
  FsDual<FsConvArgs, StemFwdGeom> d;
  d.a[0] = *a; d.a[1] = a1 ? *a1 : *a;
  d.nprob = a1 ? 2 : 1;
  auto tiles = [](const FsConvArgs& q) { return (long)q.N * ((q.Wd + TX - 1) / TX) * ((q.Hd + TY - 1) / TY); };
  const long total = tiles(*a) + (a1 ? tiles(*a1) : 0);
  int blocks = stem_geom(*a, total, d.g[0]);
  d.g[1] = d.g[0];
  d.nb0 = blocks;
  if (a1) blocks += stem_geom(*a1, total, d.g[1]);
  hipLaunchKernelGGL(conv_stem_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d);
  return fs_launch_status();
}

extern "C" int fs_conv_stem(const FsConvArgs* a, int dtype, void* stream) { return fs_conv_stem2(a, nullptr, dtype, stream); }
