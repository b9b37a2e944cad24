// 3x3 / stride-1 convolution (forward and data gradient) for layers whose source channels fit ONE 64-byte chunk and whose
// output is at most 32 channels — the depth decoder's 16- and 32-channel layers at 192x640 and 96x320
// (monodepth/networks/models/heads/depth_encoder.py:45-63,123-139: upconv(0,*), upconv(1,0)'s data gradient, the dispconvs).
// They are memory-bound (K = 9 x 16 or 32) and run on the decoder's serial chain; on the LDS-halo kernel a block staged
// 9-18 KB of weights and re-derived its tile indices for every 11 KB halo it multiplied, 5 760 blocks of them at 192x640.
// Here the weights are staged ONCE per block and the block walks pixel tiles b, b + grid, ... with the next tile's halo in
// flight while the current one is multiplied and stored (the same lesson as the BatchNorm passes, bn.hip: what every block
// repeats must be small against what it moves).  Same arguments, packed weights and epilogue as conv3x3_halo.hip; no operand
// prologue.  Entered through fs_conv3x3_halo.
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

__device__ __forceinline__ int p1_swz64(int row) { return ((row >> 3) & 1) << 1; }
__device__ __forceinline__ uint4 p1_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}
__device__ __forceinline__ void p1_barrier() {       // LDS-only workgroup barrier (global loads stay in flight)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct P1Geom {
  int TH, TW, tiles_x, tiles_y, ntiles;
  unsigned mTW, mHW;
  FsDiv dTX, dTY, dIPG;
};

// UQ = 16-byte units of a source pixel that hold channels (2: a 16-channel bf16 layer fills half a chunk — the other half of
// every halo row is zeroed once and never staged again; 4: a whole chunk)
template <typename T, int CO, int UQ>
__global__ __launch_bounds__(256, CO == 16 ? 3 : 2) void conv3x3_p1_kernel(const FsConvArgs p, const P1Geom g) {
  constexpr int WPIX = 64, TP = 4, TC = CO / 16;          // 4 waves x 64 pixels; a wave owns all CO channels of its pixels
  constexpr int HMAX = 360, HS = 6;                       // halo pixels / 16-byte units per halo row (conv3x3_halo.hip)
  constexpr int LH = (HMAX * UQ + 255) / 256, LW = (9 * CO * 4 + 255) / 256;
  constexpr int OOB = 0x7fffffff;
  __shared__ uint4 lds_h[HMAX * HS];
  __shared__ uint4 lds_w[9 * CO * 4];
  __shared__ float red[4 * CO * 2];

  const int t = threadIdx.x, lane = t & 63, wp = t >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int HW = g.TW + 2, nhalo = (g.TH + 2) * HW, ntile = g.TH * g.TW;
  const int fwd = p.sgn > 0;
  const int es = (int)sizeof(T);
  const int row_bytes = p.Cs * es;
  if ((int)blockIdx.x >= g.ntiles) return;

  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (int)p.wgt_bytes, 0x00020000);

  // ---- the nine taps' weights: once per block ----
  {
    const int wrow_bytes = p.nchunks * p.kg * 16;
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int idx = t + i * 256, q = idx & 3, rt = idx >> 2;
      const int tap = rt / CO, row = rt - tap * CO;
      const int off = (tap < 9 && q * 16 < row_bytes) ? row * wrow_bytes + tap * row_bytes + q * 16 : OOB;
      const uint4 w = p1_load16(rs_wgt, off);
      if (rt < 9 * CO) lds_w[rt * 4 + (q ^ p1_swz64(rt))] = w;
    }
  }

  // ---- what does not change from tile to tile: a thread's halo pixels relative to the tile origin ----
  int hrel[LH], hyx[LH];
#pragma unroll
  for (int i = 0; i < LH; ++i) {
    const int idx = t + i * 256, hp = idx / UQ, q = idx % UQ;
    const int hy = fs_fastdiv(min(hp, 4095), g.mHW), hx = hp - hy * HW;
    const bool ok = hp < nhalo && q * 16 < row_bytes;
    hyx[i] = ok ? ((hy << 16) | hx) : -1;
    hrel[i] = (int)(((long)hy * p.sH + (long)hx * p.sW) * es) + q * 16;
  }
  int hbase[TP];
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    int pi = wp * WPIX + b * 16 + li;
    if (pi >= ntile) pi = 0;                       // padding lanes read a valid halo row; results are discarded
    const int ty = fs_fastdiv(pi, g.mTW), tx = pi - ty * g.TW;
    hbase[b] = (ty * HW + tx) * HS + lg;
  }
  int prow[TP];                                    // tile-relative (row, column) of the lane's output pixels, -1 outside
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    const int pi = wp * WPIX + b * 16 + li;
    const int ty = fs_fastdiv(pi, g.mTW), tx = pi - ty * g.TW;
    prow[b] = pi < ntile ? ((ty << 16) | tx) : -1;
  }

  uint4 rh[LH];
  auto tile_origin = [&](int tile, int& n, int& y0, int& x0) {
    const int tq = fs_div(tile, g.dTX); const int tx_i = tile - tq * g.tiles_x;
    n = fs_div(tq, g.dTY); const int ty_i = tq - n * g.tiles_y;
    y0 = ty_i * g.TH; x0 = tx_i * g.TW;
  };
  auto load_tile = [&](int tile) {
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
    // halo origin in the source image: forward rows y0 - pad ..; data gradient rows y0 + pad - 2 ..
    const int oy = y0 + p.hb_add + (fwd ? 0 : -2), ox = x0 + p.hb_add + (fwd ? 0 : -2);
    const long base = ((long)n * p.sN + (long)oy * p.sH + (long)ox * p.sW) * es;
#pragma unroll
    for (int i = 0; i < LH; ++i) {
      const int hy = hyx[i] >> 16, hx = hyx[i] & 0xffff;
      const bool ok = hyx[i] >= 0 && (unsigned)(oy + hy) < (unsigned)p.Hs && (unsigned)(ox + hx) < (unsigned)p.Ws;
      rh[i] = p1_load16(rs_src, ok ? (int)(base + hrel[i]) : OOB);
    }
  };

  const bool has_add = p.addend != nullptr, has_mask = p.mask != nullptr;
  const bool has_bnb = p.bnb_x != nullptr, has_mbn = has_bnb && p.bnb_scale != nullptr;

  // Order of a step (gfx9 has ONE counter for vector loads and stores, and with both kinds pending a wait for the loads is a
  // wait for everything): the next tile's halo goes to LDS and the tile after that is requested BEFORE this tile's results
  // are stored, so the wait in front of the LDS stores only ever sees requests that are a whole multiplication phase old.
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < LH; ++i) {
      const int idx = t + i * 256, hp = idx / UQ, q = idx % UQ;
      if (hp < HMAX) lds_h[hp * HS + q] = rh[i];
    }
  };
  if constexpr (UQ < 4) {          // the units no load ever fills: zero, once
    for (int idx = t; idx < HMAX * (4 - UQ); idx += 256) lds_h[(idx / (4 - UQ)) * HS + UQ + idx % (4 - UQ)] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int gstep = (int)gridDim.x;
  load_tile(blockIdx.x);
  store_halo();
  if ((int)blockIdx.x + gstep < g.ntiles) load_tile(blockIdx.x + gstep);
  p1_barrier();                    // weights and the first halo are in LDS
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gstep) {
    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (one kernel row at a time: fully unrolled, hipcc keeps all nine taps' fragments live — 240 registers, two blocks per CU)
#pragma unroll 1
    for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = r * 3 + s;
      const int hoff = (fwd ? (r * HW + s) : ((2 - r) * HW + (2 - s))) * HS;
      uint4 fa[TC], fb[TP];
#pragma unroll
      for (int a = 0; a < TC; ++a) {
        const int row = tap * CO + a * 16 + li;
        fa[a] = lds_w[row * 4 + (lg ^ p1_swz64(row))];
      }
#pragma unroll
      for (int b = 0; b < TP; ++b) fb[b] = lds_h[hbase[b] + hoff];
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) {
          if constexpr (sizeof(T) == 2) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
          } else {
            const f32x4 va = __builtin_bit_cast(f32x4, fa[a]), vb = __builtin_bit_cast(f32x4, fb[b]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[j], vb[j], acc[a][b], 0, 0, 0);
          }
        }
    }

    p1_barrier();                  // every wave has read its fragments: the halo buffer is free
    if (tile + gstep < g.ntiles) {
      store_halo();                                                     // (requested one multiplication phase ago)
      if (tile + 2 * gstep < g.ntiles) load_tile(tile + 2 * gstep);
    }

    // ---- epilogue of this tile (conv3x3_halo.hip's, per tile) ----
    int n, y0, x0;
    tile_origin(tile, n, y0, x0);
    const int sgoff = p.stat_group_rows > 0 ? fs_div(n, g.dIPG) * p.Co : 0;
    float s1[TC][4], s2[TC][4];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }
    int doff[TP], aoff[TP], moff[TP];
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      const int y = y0 + (prow[b] >> 16), x = x0 + (prow[b] & 0xffff);
      const bool mok = prow[b] >= 0 && y < p.Hd && x < p.Wd;
      doff[b] = mok ? n * (int)p.dN + y * (int)p.dH + x * (int)p.dW : -1;
      aoff[b] = n * (int)p.aN + y * (int)p.aH + x * (int)p.aW;
      moff[b] = n * (int)p.mN + y * (int)p.mH + x * (int)p.mW;
    }
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      const int co = a * 16 + lg * 4;
      if (co >= p.Co) continue;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), mu = bv, is = bv, msc = bv, msh = bv;
      if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + co);
      if (has_bnb) {
        mu = *reinterpret_cast<const float4*>(p.bnb_mean + sgoff + co);
        is = *reinterpret_cast<const float4*>(p.bnb_invstd + sgoff + co);
        if (has_mbn) {
          msc = *reinterpret_cast<const float4*>(p.bnb_scale + sgoff + co);
          msh = *reinterpret_cast<const float4*>(p.bnb_shift + sgoff + co);
        }
      }
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        if (doff[b] < 0) continue;
        float v[4] = {acc[a][b][0] + bv.x, acc[a][b][1] + bv.y, acc[a][b][2] + bv.z, acc[a][b][3] + bv.w};
        if (has_add) {
          float av[4];
          load4<T>(reinterpret_cast<const T*>(p.addend) + aoff[b] + co, av);
          v[0] += av[0]; v[1] += av[1]; v[2] += av[2]; v[3] += av[3];
        }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (has_mask) {
          float mv[4];
          load4<T>(reinterpret_cast<const T*>(p.mask) + moff[b] + co, mv);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
        }
        if (has_bnb) {
          float cv[4];
          load4<T>(reinterpret_cast<const T*>(p.bnb_x) + doff[b] + co, cv);
          if (has_mbn) {
            v[0] = (cv[0] * msc.x + msh.x) > 0.f ? v[0] : 0.f; v[1] = (cv[1] * msc.y + msh.y) > 0.f ? v[1] : 0.f;
            v[2] = (cv[2] * msc.z + msh.z) > 0.f ? v[2] : 0.f; v[3] = (cv[3] * msc.w + msh.w) > 0.f ? v[3] : 0.f;
          }
          s1[a][0] += v[0]; s1[a][1] += v[1]; s1[a][2] += v[2]; s1[a][3] += v[3];
          s2[a][0] += v[0] * (cv[0] - mu.x) * is.x; s2[a][1] += v[1] * (cv[1] - mu.y) * is.y;
          s2[a][2] += v[2] * (cv[2] - mu.z) * is.z; s2[a][3] += v[3] * (cv[3] - mu.w) * is.w;
        } else if (p.stats) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { s1[a][j] += v[j]; s2[a][j] += v[j] * v[j]; }
        }
        if (p.out_f32) store4<float>(reinterpret_cast<float*>(p.dst) + doff[b] + co, v);
        else store4<T>(reinterpret_cast<T*>(p.dst) + doff[b] + co, v);
      }
    }
    if (p.stats) {
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = row16_sum(s1[a][j]), w = row16_sum(s2[a][j]);     // over the 16 pixel lanes of this channel (DPP)
          if (li == 0) {
            const int cl = a * 16 + lg * 4 + j;
            red[(wp * CO + cl) * 2] = u; red[(wp * CO + cl) * 2 + 1] = w;
          }
        }
      p1_barrier();
      if (t < CO) {
        float u = 0.f, w = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { u += red[(k * CO + t) * 2]; w += red[(k * CO + t) * 2 + 1]; }
        if (t < p.Co) {
          const long sg = p.stat_group_rows > 0 ? fs_div(n, g.dIPG) : 0;
          double* sl = p.stats + (sg * FS_STAT_SLOTS + tile % FS_STAT_SLOTS) * 2 * p.Co;
          atomicAdd(sl + t, (double)u);
          atomicAdd(sl + p.Co + t, (double)w);
        }
      }
    }
    p1_barrier();                  // next halo complete in LDS; the statistics hand-off buffer is free again
  }
}

P1Geom p1_pick_geom(int Hd, int Wd) {
  P1Geom best{};
  double best_cost = 1e30;
  for (int tw = std::min(4, Wd); tw <= std::min(Wd, 64); ++tw) {
    const int th = std::min(256 / tw, Hd);
    if (th < 1 || (th + 2) * (tw + 2) > 360) continue;
    const int tx = (Wd + tw - 1) / tw, ty = (Hd + th - 1) / th;
    const double waste = (double)tx * ty * 256 / ((double)Hd * Wd);
    const double halo = (double)(th + 2) * (tw + 2) / ((double)th * tw);
    const double cost = waste * (1.0 + 0.15 * halo);
    if (cost < best_cost - 1e-9) { best_cost = cost; best.TH = th; best.TW = tw; best.tiles_x = tx; best.tiles_y = ty; }
  }
  if (best.TW > 0) {
    best.mTW = fs_div_magic(best.TW); best.mHW = fs_div_magic(best.TW + 2);
    best.dTX = fs_make_div(best.tiles_x); best.dTY = fs_make_div(best.tiles_y);
  }
  return best;
}

template <typename K>
int p1_resident(K kernel) {
  int dev = 0, cus = 256, per_cu = 1;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return cus * per_cu;
}

template <typename T, int CO, int UQ>
int p1_launch(const FsConvArgs& a, hipStream_t st) {
  P1Geom g = p1_pick_geom(a.Hd, a.Wd);
  if (g.TH == 0) return FS_EINVAL;
  g.dIPG = FsDiv{0u, 0u};
  if (a.stat_group_rows > 0) {
    const long hw = (long)a.Hd * a.Wd;
    if (a.stat_group_rows % hw != 0) return FS_EINVAL;
    g.dIPG = fs_make_div((int)(a.stat_group_rows / hw));
  }
  const long ntiles = (long)a.N * g.tiles_x * g.tiles_y;
  // worth it from two tiles per resident block on (below that the one-tile-per-block kernel has as little to repeat)
  // (FsConvArgs.force_impl = 5: a test forces small launches onto this kernel)
  const long min_rounds_x10 = a.force_impl == 5 ? 0 : 20;
  static const int slots = p1_resident(conv3x3_p1_kernel<T, CO, UQ>);
  if (ntiles * 10 < (long)slots * min_rounds_x10 || ntiles > 0x7fffffffL) return FS_EINVAL;
  g.ntiles = (int)ntiles;
  const int blocks = (int)std::min<long>(ntiles, slots);
  if (fs_conv3x3_plan_slot) {
    fs_conv3x3_plan_slot[0] = 2; fs_conv3x3_plan_slot[1] = blocks; fs_conv3x3_plan_slot[2] = 256; fs_conv3x3_plan_slot[3] = CO;
    return FS_OK;
  }
  hipLaunchKernelGGL((conv3x3_p1_kernel<T, CO, UQ>), dim3(blocks), dim3(256), 0, st, a, g);
  return fs_launch_status();
}

}  // namespace

// internal entry: FS_EINVAL = "not mine" (fs_conv3x3_halo goes on to its other kernels)
int fs_conv3x3_p1(const FsConvArgs& a, int dtype, hipStream_t st) {
  const int es = dtype == FS_DTYPE_BF16 ? 2 : 4;
  if (a.pro_mode != 0 || a.hb_mul != 1 || a.Cs * es > 64 || (a.Co_p != 16 && a.Co_p != 32) || a.Co % 4 != 0) return FS_EINVAL;
  if (a.src_bytes >= 0x7ffff000LL) return FS_EINVAL;
  const bool half = a.Cs * es <= 32;        // half-filled chunk: two 16-byte units per pixel
  if (dtype == FS_DTYPE_BF16) {
    if (a.Co_p == 32) return half ? p1_launch<bf16, 32, 2>(a, st) : p1_launch<bf16, 32, 4>(a, st);
    return half ? p1_launch<bf16, 16, 2>(a, st) : p1_launch<bf16, 16, 4>(a, st);
  }
  if (dtype == FS_DTYPE_F32) {
    if (a.Co_p == 32) return half ? p1_launch<float, 32, 2>(a, st) : p1_launch<float, 32, 4>(a, st);
    return half ? p1_launch<float, 16, 2>(a, st) : p1_launch<float, 16, 4>(a, st);
  }
  return FS_EINVAL;
}
