// 3x3 / stride-1 convolution (forward and data-gradient) with an LDS-resident input halo tile.
//
// Same call sites as conv_igemm.hip (the 3x3 stride-1 family is ~85 % of the conv FLOPs of the monodepth
// step: every BasicBlock conv, all decoder convs, the pose convs).  The generic implicit-GEMM kernel
// re-fetches every input pixel once per tap (9x) and synchronises every 8 MFMAs; measured, it is bound by
// operand delivery (texture-addresser busy 74 %, ~8 TB/s L2->LDS) at the 32 FLOP/B of its 64x64 tile.
// Here one block owns a TH x TW pixel tile x CO channels and walks the input channels in chunks of 64 bytes
// (32 bf16 / 16 f32):
//   * the (TH+2) x (TW+2) input halo of the chunk is fetched ONCE (raw buffer loads, OOB = zero padding)
//     and reused by all 9 taps straight from LDS (MFMA B fragments are read at shifted halo positions);
//   * the 9 taps' weights of the chunk ([tap][co][64 B]) are staged together, so a wave issues
//     9 * TP * TC MFMAs (36 for the 128x32 tile the dispatcher prefers, 72 for 128x64) between two barriers;
//   * the next chunk's halo + weights are prefetched into registers while the current one is multiplied.
// Arithmetic intensity: 98 FLOP per fetched byte on a 128x64 tile, 67 on 128x32 (2-3x the generic kernel) — but see
// dispatch(): occupancy, not reuse, decides the tile on the monodepth shapes.
// Dgrad = same kernel on dY with Wt[ci][r][s][co] and the taps mirrored (sgn = -1).
// The epilogue (bias / addend / ReLU / ReLU-mask / fp32 out / fused BN statistics) matches conv_igemm.hip.
// Round 4: the operand prologue of conv3x3_t32.hip (conv_pro.h: BatchNorm + ReLU in front of the convolution, coefficients
// derived in the kernel) and the derived ReLU mask (bnb_scale) exist here too, so that every BasicBlock of every stage folds, not only the launches large
// enough for the 32x32-tile kernel.
#include "common.h"
#include "fsnet_hip_internal.h"
#include "conv_pro.h"
#include <algorithm>
#include <cstdlib>

namespace {

__device__ __forceinline__ int swz64(int row) { return ((row >> 3) & 1) << 1; }

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

// workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope fence, which on
// gfx9 is `s_waitcnt vmcnt(0)`: every barrier then waits for the block's outstanding global loads AND stores (one
// L2 round trip per barrier — PMC on the layer-1 shapes: waves parked 58 % of their cycles).  LDS hand-offs between
// the waves of a block only need the DS counter drained.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct HaloGeom {
  int TH, TW;        // pixel tile
  int tiles_x, tiles_y;
  unsigned mTW, mHW; // fs_div_magic(TW), fs_div_magic(TW + 2)
  FsDiv dTX, dTY;    // tiles_x, tiles_y
  FsDiv dIPG;        // images per BatchNorm statistics group (stat_group_rows / (Hd*Wd)); unused when 0 groups
  FsDiv dPRG;        // images per prologue coefficient group
  int pix_major;     // block -> tile mapping keeps a pixel tile's channel tiles on one XCD (else: a channel tile's)
};

// waves per SIMD the register allocation is held to: the forward prologue must not cost the 128x32 tile its fourth block
// per CU (132 registers without the bound: measured +6 us on a 22 us launch, more than the BatchNorm pass it replaces saves)
constexpr int halo_minwaves(int PIX, int CO, int PRO) { return (PRO == 1 && PIX == 128) ? 4 : 1; }

// STR = 2 (round 4; forward only): the stride-2 3x3 convolutions at the ResNet stage entries (resnet.py:140-160,
// conv1 of layer2-4 block 0) on the same machinery instead of the generic implicit GEMM (22-42 us for 3.4 GFLOP).  The halo of
// a TH x TW output tile is (2TH+1) x (2TW+1) source pixels; it is stored with the columns de-interleaved by parity —
// slot = row * 2(TW+1) + (x & 1) * (TW+1) + (x >> 1) — so that the 16 pixels of an MFMA tile, which sit 2 source columns
// apart, read consecutive slots (the conflict-free pattern of the stride-1 case) and a tap is still one immediate offset.
constexpr int halo_hmax(int PIX, int STR) { return STR == 2 ? (PIX == 128 ? 620 : 340) : (PIX == 256 ? 360 : (PIX == 128 ? 208 : 120)); }

template <typename T, int PIX, int CO, int WP, int PRO, int STR = 1>
__global__ __launch_bounds__(256, halo_minwaves(PIX, CO, PRO)) void conv3x3_halo_kernel(const FsDual<FsConvArgs, HaloGeom> d) {
  // two problems per launch (fsnet_hip_internal.h, FsDual): blocks [0, nb0) take the first argument set
  const int prob = (int)blockIdx.x >= d.nb0 ? 1 : 0;
  const FsConvArgs& p = d.a[prob];
  const HaloGeom& g = d.g[prob];
  const int bid = (int)blockIdx.x - (prob ? d.nb0 : 0);
  static_assert(STR == 1 || PRO == 0, "the stride-2 variant has no operand prologue");
  constexpr int WC = 4 / WP;
  constexpr int WPIX = PIX / WP, WCO = CO / WC;
  constexpr int TP = WPIX / 16, TC = WCO / 16;
  constexpr int HMAX = halo_hmax(PIX, STR);          // halo slots that fit the LDS budget
  constexpr int LH = (HMAX * 4 + 255) / 256;         // halo 16-byte units per thread
  constexpr int LW = (9 * CO * 4 + 255) / 256;       // weight 16-byte units per thread
  constexpr int OOB = 0x7fffffff;

  // halo rows padded to HS = 6 x 16 B = 96 bytes.  ds_read_b128 is serviced in four groups of 16 lanes,
  // {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): 8 pixels with k-group lg and the
  // other 8 with lg^1.  With consecutive pixels a row stride of 2 mod 4 slots puts those 16 lanes on 16 distinct
  // 16-byte slots at ANY tap shift (the shift stays an immediate LDS offset).  The first version used 5 slots
  // ("odd => conflict-free" holds for 16 CONSECUTIVE lanes, not for this grouping): PMC showed
  // SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE, every B-fragment read was 2-way conflicted.
  constexpr int HS = 6;
  __shared__ uint4 lds_h[HMAX * HS];
  __shared__ uint4 lds_w[9 * CO * 4];
  extern __shared__ float halo_pro_tab[];            // [pro_ncoef<PRO>()][Cs] (PRO != 0 launches only)
  constexpr int UN = 16 / (int)sizeof(T);            // elements per 16-byte unit

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wp = wave % WP, wc = wave / WP;
  const int li = lane & 15, lg = lane >> 4;
  const int HW = STR * g.TW + 3 - STR, HH = STR * g.TH + 3 - STR;      // source halo, pixels
  const int HWs = STR == 2 ? 2 * (g.TW + 1) : HW;                      // LDS slots per halo row
  const int nhalo = HH * HW;
  const int ntile = g.TH * g.TW;

  // ---- tile mapping: XCD-aware over the channel tiles (see conv_igemm.hip) ----
  const int npix = p.N * g.tiles_y * g.tiles_x, nco = p.Co_p / CO;
  int px, cy;
  {
    const int id = bid, xcd = id & 7, slot = id >> 3;
    if (g.pix_major) {
      // activations outweigh the weights (layer 1 / 2, the decoder): the channel tiles of ONE pixel tile run on the
      // same XCD in consecutive slots, so the halo is fetched across the fabric once and re-read from that L2
      cy = slot % nco; px = (slot / nco) * 8 + xcd;
    }
    else if (nco % 8 == 0) { const int q = nco >> 3; cy = xcd + 8 * (slot % q); px = slot / q; }
    else if (8 % nco == 0) { const int q = 8 / nco; cy = xcd % nco; px = slot * q + xcd / nco; }
    else { cy = id % nco; px = id / nco; }
    if (px >= npix) return;
  }
  const int tq = fs_div(px, g.dTX); const int tx_i = px - tq * g.tiles_x;
  const int n = fs_div(tq, g.dTY); const int ty_i = tq - n * g.tiles_y;
  const int y0 = ty_i * g.TH, x0 = tx_i * g.TW;
  const int co0 = cy * CO;
  const int fwd = p.sgn > 0;
  // halo origin in the source image: forward rows y0 - pad ..; dgrad rows y0 + pad - 2 ..
  const int oy = STR * y0 + p.hb_add + (fwd ? 0 : -2), ox = STR * x0 + p.hb_add + (fwd ? 0 : -2);

  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (int)p.wgt_bytes, 0x00020000);

  // the prologue's coefficient table first: its f64 sums are the oldest loads in flight when the halo arrives, and
  // none of the walk's scalar state is live yet (the table is complete at the loop's first barrier)
  if constexpr (PRO != 0) pro_build_table<PRO>(p, halo_pro_tab, (p.pro_group_imgs > 0) ? fs_div(n, g.dPRG) : 0, t, 256);

  // ---- per-thread load units (fixed over the channel walk) ----
  const int row_bytes = p.Cs * (int)sizeof(T);     // real bytes per pixel / per tap
  int hvoff[LH], wvoff[LW];
#pragma unroll
  for (int i = 0; i < LH; ++i) {
    int idx = t + i * 256;
    int hp = idx >> 2, q = idx & 3;
    int hy = fs_fastdiv(hp, g.mHW), hx = hp - hy * HW;
    int sy = oy + hy, sx = ox + hx;
    // (a 16-channel bf16 layer fills half a 64-byte chunk: the upper units stay zero, as do their weights)
    bool ok = hp < nhalo && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws && q * 16 < row_bytes;
    hvoff[i] = ok ? (int)(((long)n * p.sN + (long)sy * p.sH + (long)sx * p.sW) * (long)sizeof(T)) + q * 16 : OOB;
  }

  const int wrow_bytes = p.nchunks * p.kg * 16;    // packed weight row stride (as packed for conv_igemm)
  const int tap_bytes = p.Cs * (int)sizeof(T);
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    int idx = t + i * 256;
    int q = idx & 3, rt = idx >> 2;
    int tap = rt / CO, row = rt - tap * CO;
    wvoff[i] = (tap < 9 && q * 16 < row_bytes) ? (co0 + row) * wrow_bytes + tap * tap_bytes + q * 16 : OOB;
  }

  uint4 rh[LH], rw[LW];
  auto load_regs = [&](int cc) {
    const int coff = cc * 64;
#pragma unroll
    for (int i = 0; i < LH; ++i) rh[i] = buf_load16(rs_src, hvoff[i] == OOB ? OOB : hvoff[i] + coff);
#pragma unroll
    for (int i = 0; i < LW; ++i) rw[i] = buf_load16(rs_wgt, wvoff[i] == OOB ? OOB : wvoff[i] + coff);
  };
  auto store_lds = [&](int cc) {
    float ka[PRO != 0 ? UN : 1], kb[PRO != 0 ? UN : 1];
    if constexpr (PRO != 0) {
      // this thread's channels of the chunk: 256 % 4 == 0, so slot q = t & 3 of every pixel it stages
      const int c0r = cc * (64 / (int)sizeof(T)) + (t & 3) * UN;
      const int c0 = c0r < p.Cs ? c0r : 0;
#pragma unroll
      for (int j = 0; j < UN; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(halo_pro_tab + c0 + j);
        const float4 b = *reinterpret_cast<const float4*>(halo_pro_tab + p.Cs + c0 + j);
        ka[j] = a.x; ka[j + 1] = a.y; ka[j + 2] = a.z; ka[j + 3] = a.w;
        kb[j] = b.x; kb[j + 1] = b.y; kb[j + 2] = b.z; kb[j + 3] = b.w;
      }
    }
#pragma unroll
    for (int i = 0; i < LH; ++i) {
      int idx = t + i * 256;
      int hp = idx >> 2, q = idx & 3;
      uint4 u = rh[i];
      if constexpr (PRO == 1) {
        float v[UN];
        Unit<T>::unpack(u, v);
#pragma unroll
        for (int j = 0; j < UN; ++j) {
          v[j] = v[j] * ka[j] + kb[j];
          if (p.pro_relu) v[j] = fmaxf(v[j], 0.f);
        }
        u = Unit<T>::pack(v);
        if (hvoff[i] == OOB) u = make_uint4(0u, 0u, 0u, 0u);     // padding applies to the transformed tensor
      }
      if constexpr (STR == 2) {
        const int hy = fs_fastdiv(hp, g.mHW), hx = hp - hy * HW;
        if (hp < nhalo) lds_h[(hy * HWs + (hx & 1) * (g.TW + 1) + (hx >> 1)) * HS + q] = u;
      } else {
        if (hp < HMAX) lds_h[hp * HS + q] = u;
      }
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      int idx = t + i * 256;
      int q = idx & 3, rt = idx >> 2;
      if (rt < 9 * CO) lds_w[rt * 4 + (q ^ swz64(rt))] = rw[i];   // CO % 16 == 0 -> swizzle follows the co row
    }
  };

  // ---- per-lane halo base rows of the TP pixel tiles this wave multiplies ----
  int hbase[TP];
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    int pi = wp * WPIX + b * 16 + li;
    if (pi >= ntile) pi = 0;                       // padding lanes read a valid halo row; results are discarded
    int ty = fs_fastdiv(pi, g.mTW), tx = pi - ty * g.TW;
    hbase[b] = (STR * ty * HWs + tx) * HS + lg;
  }

  f32x4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int s2_row = HWs * HS, s2_par = (g.TW + 1) * HS;     // stride 2: slot distance of a halo row / of the odd columns

  const int nchunk = (p.Cs * (int)sizeof(T) + 63) / 64;
  load_regs(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    lds_barrier();                 // previous chunk fully multiplied (first pass: the coefficient table is complete)
    store_lds(cc);
    lds_barrier();
    if (cc + 1 < nchunk) load_regs(cc + 1);
    // (the compiler's own schedule of this tap walk — every fragment read next to its use — is as fast as a hand-
    // pipelined one with the reads one or two taps ahead: measured)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s = tap - r * 3;
      const int hoff = STR == 2 ? r * s2_row + (s & 1) * s2_par + (s >> 1) * HS
                                : (fwd ? (r * HW + s) : ((2 - r) * HW + (2 - s))) * HS;
      uint4 fa[TC], fb[TP];
#pragma unroll
      for (int a = 0; a < TC; ++a) {
        int row = tap * CO + wc * WCO + a * 16 + li;
        fa[a] = lds_w[row * 4 + (lg ^ swz64(row))];
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
            f32x4 va = __builtin_bit_cast(f32x4, fa[a]);
            f32x4 vb = __builtin_bit_cast(f32x4, fb[b]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[j], vb[j], acc[a][b], 0, 0, 0);
          }
        }
    }
  }

  // ---- epilogue ----
  // (the data-gradient options do not exist in the stride-2 forward variant: their arguments are then never loaded)
  const bool has_add = STR == 1 && p.addend != nullptr, has_mask = STR == 1 && p.mask != nullptr;
  const bool has_bnb = STR == 1 && p.bnb_x != nullptr, has_mbn = has_bnb && p.bnb_scale != nullptr;
  float s1[TC][4], s2[TC][4];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }
  const int sgoff = p.stat_group_rows > 0 ? fs_div(n, g.dIPG) * p.Co : 0;
  // per pixel tile b: element offsets into dst / addend / mask / bnb_x (32-bit: every activation tensor on this
  // path is far below 2^31 elements), -1 = lane holds no pixel.  Channel-only terms are hoisted per tile a.
  int doff[TP], aoff[TP], moff[TP];
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    int pi = wp * WPIX + b * 16 + li;
    int ty = fs_fastdiv(pi, g.mTW), tx = pi - ty * g.TW;
    int y = y0 + ty, x = x0 + tx;
    bool mok = pi < ntile && y < p.Hd && x < p.Wd;
    doff[b] = mok ? n * (int)p.dN + y * (int)p.dH + x * (int)p.dW : -1;
    aoff[b] = n * (int)p.aN + y * (int)p.aH + x * (int)p.aW;
    moff[b] = n * (int)p.mN + y * (int)p.mH + x * (int)p.mW;
  }
#pragma unroll
  for (int a = 0; a < TC; ++a) {
    const int co = co0 + wc * WCO + a * 16 + lg * 4;
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
      if (has_bnb) {   // BatchNorm-backward sums of the layer this gradient flows into: (sum g, sum g*xhat)
        float cv[4];
        load4<T>(reinterpret_cast<const T*>(p.bnb_x) + doff[b] + co, cv);   // same layout as dst
        if (has_mbn) {
          // ReLU mask of a folded BatchNorm: the sign of the forward prologue's own expression
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
    lds_barrier();
    float* red = reinterpret_cast<float*>(&lds_w[0]);   // [WP][CO][2]
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = s1[a][j], w = s2[a][j];
        u = row16_sum(u); w = row16_sum(w);     // over the 16 pixel lanes of this channel (DPP, no LDS)
        if (li == 0) {
          int cl = wc * WCO + a * 16 + lg * 4 + j;
          red[(wp * CO + cl) * 2] = u; red[(wp * CO + cl) * 2 + 1] = w;
        }
      }
    lds_barrier();
    if (t < CO) {
      float u = 0.f, w = 0.f;
#pragma unroll
      for (int k = 0; k < WP; ++k) { u += red[(k * CO + t) * 2]; w += red[(k * CO + t) * 2 + 1]; }
      int co = co0 + t;
      if (co < p.Co) {
        const long sg = p.stat_group_rows > 0 ? fs_div(n, g.dIPG) : 0;   // whole images
        double* sl = p.stats + (sg * FS_STAT_SLOTS + px % FS_STAT_SLOTS) * 2 * p.Co;
        atomicAdd(sl + co, (double)u);
        atomicAdd(sl + p.Co + co, (double)w);
      }
    }
  }
  // (last, so that its arguments are not live across the walk: saved statistics / running statistics / dgamma, dbeta)
  if constexpr (PRO != 0) { if (bid == 0) pro_block0<PRO>(p, t, 256); }
}

// pick the pixel tile (TH x TW <= PIX, halo <= hmax) that wastes the fewest lanes, preferring wide tiles
HaloGeom pick_geom(int Hd, int Wd, int PIX, int hmax) {
  HaloGeom best{0, 0, 0, 0, 0u, 0u, FsDiv{0u, 0u}, FsDiv{0u, 0u}, FsDiv{0u, 0u}, FsDiv{0u, 0u}, 0};
  double best_cost = 1e30;
  for (int tw = std::min(4, Wd); tw <= std::min(Wd, 64); ++tw) {
    int th = std::min(PIX / tw, Hd);
    if (th < 1 || (th + 2) * (tw + 2) > hmax) continue;
    int tx = (Wd + tw - 1) / tw, ty = (Hd + th - 1) / th;
    double waste = (double)tx * ty * PIX / ((double)Hd * Wd);          // MFMA lanes spent per useful pixel
    double halo = (double)(th + 2) * (tw + 2) / ((double)th * tw);     // fetch overhead
    double cost = waste * (1.0 + 0.15 * halo);
    if (cost < best_cost - 1e-9) { best_cost = cost; best.TH = th; best.TW = tw; best.tiles_x = tx; best.tiles_y = ty; }
  }
  if (best.TW > 0) {
    best.mTW = fs_div_magic(best.TW); best.mHW = fs_div_magic(best.TW + 2);
    best.dTX = fs_make_div(best.tiles_x); best.dTY = fs_make_div(best.tiles_y);
  }
  return best;
}

// stride 2: TH x TW output pixels whose (2TH+1) x (2TW+2) halo slots fit
HaloGeom pick_geom_s2(int Hd, int Wd, int PIX, int hmax) {
  HaloGeom best{0, 0, 0, 0, 0u, 0u, FsDiv{0u, 0u}, FsDiv{0u, 0u}, FsDiv{0u, 0u}, FsDiv{0u, 0u}, 0};
  double best_cost = 1e30;
  for (int tw = std::min(4, Wd); tw <= std::min(Wd, 64); ++tw) {
    int th = std::min(PIX / tw, Hd);
    while (th >= 1 && (2 * th + 1) * (2 * tw + 2) > hmax) --th;
    if (th < 1) continue;
    int tx = (Wd + tw - 1) / tw, ty = (Hd + th - 1) / th;
    double waste = (double)tx * ty * PIX / ((double)Hd * Wd);
    double halo = (double)(2 * th + 1) * (2 * tw + 1) / (4.0 * th * tw);
    double cost = waste * (1.0 + 0.15 * halo);
    if (cost < best_cost - 1e-9) { best_cost = cost; best.TH = th; best.TW = tw; best.tiles_x = tx; best.tiles_y = ty; }
  }
  if (best.TW > 0) {
    best.mTW = fs_div_magic(best.TW); best.mHW = fs_div_magic(2 * best.TW + 1);
    best.dTX = fs_make_div(best.tiles_x); best.dTY = fs_make_div(best.tiles_y);
  }
  return best;
}

// geometry and grid of one problem; blocks = 0: the arguments do not fit this tiling
template <int PIX, int CO, int STR>
int halo_problem(const FsConvArgs& a, HaloGeom& g) {
  g = STR == 2 ? pick_geom_s2(a.Hd, a.Wd, PIX, halo_hmax(PIX, 2)) : pick_geom(a.Hd, a.Wd, PIX, halo_hmax(PIX, 1));
  if (g.TH == 0) return 0;
  if (a.stat_group_rows > 0) {
    const long hw = (long)a.Hd * a.Wd;
    if (a.stat_group_rows % hw != 0) return 0;                  // statistics groups are whole images
    g.dIPG = fs_make_div((int)(a.stat_group_rows / hw));
  }
  if (a.pro_group_imgs > 0) g.dPRG = fs_make_div(a.pro_group_imgs);
  const int npix = a.N * g.tiles_x * g.tiles_y, nco = a.Co_p / CO;
  int blocks = npix * nco;
  // which operand is worth keeping XCD-local: the input activation (fetched once per channel tile) or the weights
  g.pix_major = (nco > 1 && a.src_bytes > 2 * a.wgt_bytes) ? 1 : 0;
  if (g.pix_major) blocks = 8 * ((npix + 7) / 8) * nco;
  else if (nco % 8 != 0 && 8 % nco == 0) { const int q = 8 / nco; blocks = 8 * ((npix + q - 1) / q); }
  return blocks;
}

// b != nullptr: a second problem of the same shape class (conv3x3_pairable) in the same launch
template <typename T, int PIX, int CO, int WP, int PRO, int STR = 1>
int launch_halo_pro(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  FsDual<FsConvArgs, HaloGeom> d;
  d.a[0] = a; d.a[1] = b ? *b : a;
  d.nprob = b ? 2 : 1;
  int blocks = halo_problem<PIX, CO, STR>(a, d.g[0]);
  if (blocks == 0) return FS_EINVAL;
  d.g[1] = d.g[0];
  d.nb0 = blocks;
  if (b) {
    const int b1 = halo_problem<PIX, CO, STR>(*b, d.g[1]);
    if (b1 == 0) return FS_EINVAL;
    d.nb0 = fs_xcd_round(blocks);
    blocks = d.nb0 + b1;
  }
  if (fs_conv3x3_plan_slot) {
    fs_conv3x3_plan_slot[0] = 0; fs_conv3x3_plan_slot[1] = blocks; fs_conv3x3_plan_slot[2] = PIX; fs_conv3x3_plan_slot[3] = CO;
    return FS_OK;
  }
  hipLaunchKernelGGL((conv3x3_halo_kernel<T, PIX, CO, WP, PRO, STR>), dim3(blocks), dim3(256), pro_lds_bytes<PRO>(a), st, d);
  return fs_launch_status();
}

template <typename T, int PIX, int CO, int WP>
int launch_halo(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  switch (a.pro_mode) {
    case 0: return launch_halo_pro<T, PIX, CO, WP, 0>(a, b, st);
    case 1: return launch_halo_pro<T, PIX, CO, WP, 1>(a, b, st);
    default: return FS_EINVAL;
  }
}

// stride-2 forward: 128-pixel tiles x 32 output channels (two blocks per CU).  Measured alone against the implicit GEMM
// and against 16-channel tiles (tools/probes/s2_time.py, us, bf16, B = 12 / 24): 64->128 @48x160 21.8 / 31.4 (igemm 24.6 /
// 31.4; 16-channel tiles 26.7 / 42.6), 128->256 @24x80 19.4 / 27.0 (22.7 / 30.0; 22.5 / 34.3), 256->512 @12x40 20.6 / 25.9
// (31.0 / 37.2; 21.2 / 32.0): the halo is four times the output tile, so the second channel tile's re-fetch costs more
// than the extra blocks buy
template <typename T>
int dispatch_s2(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  HaloGeom g = pick_geom_s2(a.Hd, a.Wd, 128, halo_hmax(128, 2));
  if (g.TH == 0 || a.Co_p % 16 != 0) return FS_EINVAL;
  if (a.Co_p % 32 == 0) return launch_halo_pro<T, 128, 32, 4, 0, 2>(a, b, st);
  return launch_halo_pro<T, 128, 16, 4, 0, 2>(a, b, st);
}

template <typename T>
int dispatch(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  const int cop = a.Co_p;
  const int nimg = a.N + (b ? b->N : 0);           // the tile choice looks at the whole launch
  auto blocks_for = [&](int PIX, int CO) {
    HaloGeom g = pick_geom(a.Hd, a.Wd, PIX, PIX == 256 ? 360 : (PIX == 128 ? 208 : 120));
    return g.TH == 0 ? 0L : (long)nimg * g.tiles_x * g.tiles_y * (cop / CO);
  };
  if (cop % 32 == 0) {
    // Occupancy decides here, not operand reuse: these launches are latency-bound (one wave of blocks, each a chain
    // of load -> LDS -> MFMA -> store phases).  A 64-channel tile stages 36.8 KB of weights per chunk and only two
    // blocks fit a CU; the 32-channel tile (18.4 KB, four blocks per CU, twice the blocks) is 5-15 % faster on every
    // ResNet stage although each halo is then fetched twice, and the 16-channel tile wins when even that leaves the
    // chip short of blocks (layer 4 at batch 12: 192 -> 384 blocks, 33 -> 25 us).  Measured the other way too:
    // 256-pixel tiles, which halve the weight fill per pixel, are 20-30 % slower.
    if (blocks_for(128, 32) < 256) return launch_halo<T, 128, 16, 4>(a, b, st);
    // (measured and removed: fragment reads pipelined one / two taps ahead of the MFMAs — within 5 % either way —,
    // 128x64 tiles with 2x4 / 4x2 MFMA tiles per wave at every prefetch depth: 10-20 % slower at two blocks per CU, also
    // on ResNet-50's 164 k-pixel launches — DESIGN section 7)
    return launch_halo<T, 128, 32, 4>(a, b, st);
  }
  if (cop % 16 == 0) {   // 16-channel decoder layers at 96x320 / 192x640: memory-bound, large pixel tiles
    if (blocks_for(256, 16) >= 1024) return launch_halo<T, 256, 16, 4>(a, b, st);
    return launch_halo<T, 128, 16, 4>(a, b, st);
  }
  return FS_EINVAL;
}

}  // namespace

int fs_conv3x3_t32(const FsConvArgs& a, const FsConvArgs* b, int dtype, hipStream_t st);      // conv3x3_t32.hip
int fs_conv3x3_p1(const FsConvArgs& a, int dtype, hipStream_t st);       // conv3x3_p1.hip

namespace {
int conv3x3_check(const FsConvArgs* args, int dtype) {
  if (!args || !args->src || !args->wgt || !args->dst) return FS_EINVAL;
  const int es = dtype == FS_DTYPE_BF16 ? 2 : 4;
  if (args->Cs <= 0 || (args->Cs * es) % 32 != 0 || ((args->Cs * es) % 64 != 0 && args->Cs * es != 32) || args->dshift != 0)
    return FS_EINVAL;
  // stride 2: forward launches with pad 1 and whole 64-byte channel chunks, no prologue / no derived mask, and none of the
  // data-gradient epilogue options (the stride-2 instantiation compiles them out: ADVICE r04)
  const bool s2 = args->hb_mul == 2;
  if (args->hb_mul != 1 && !(s2 && args->sgn > 0 && args->hb_add == -1 && (args->Cs * es) % 64 == 0 && args->pro_mode == 0 &&
                             !args->bnb_x && !args->addend && !args->mask &&
                             args->Hd == (args->Hs - 1) / 2 + 1 && args->Wd == (args->Ws - 1) / 2 + 1))
    return FS_EINVAL;
  if (args->Co % 4 != 0 || args->Co_p % 16 != 0 || args->N <= 0) return FS_EINVAL;
  if (args->src_bytes <= 0 || args->src_bytes > 0x7fffffffLL || args->wgt_bytes <= 0 ||
      args->wgt_bytes > 0x7fffffffLL)
    return FS_EINVAL;
  if (!pro_args_ok(*args)) return FS_EINVAL;
  if (args->pro_mode != 0 && (args->Cs * es) % 64 != 0) return FS_EINVAL;   // whole 64-byte chunks with a prologue
  // every kernel of this family addresses the destination, the addend and the mask with 32-bit element offsets
  {
    auto span_ok = [&](const void* ptr, int64_t sn, int64_t sh, int64_t sw) {
      if (!ptr) return true;
      if (sn < 0 || sh < 0 || sw < 0) return false;
      return (int64_t)(args->N - 1) * sn + (int64_t)(args->Hd - 1) * sh + (int64_t)(args->Wd - 1) * sw + args->Co_p < 0x7fffffffLL;
    };
    if (!span_ok(args->dst, args->dN, args->dH, args->dW) || !span_ok(args->addend, args->aN, args->aH, args->aW) ||
        !span_ok(args->mask, args->mN, args->mH, args->mW))
      return FS_EINVAL;
  }
  if (args->bnb_scale && (!args->bnb_x || !args->bnb_shift || args->mask)) return FS_EINVAL;
  if (args->bnb_x && (!args->stats || !args->bnb_mean || !args->bnb_invstd)) return FS_EINVAL;
  return FS_OK;
}

// may the two problems share one launch?  Everything that selects a kernel instantiation or a tile geometry must agree;
// pointers, strides, batch sizes and statistics groups are per problem
bool conv3x3_pairable(const FsConvArgs& a, const FsConvArgs& b) {
  auto nn = [](const void* x, const void* y) { return (x == nullptr) == (y == nullptr); };
  return a.Hs == b.Hs && a.Ws == b.Ws && a.Hd == b.Hd && a.Wd == b.Wd && a.Cs == b.Cs && a.Co == b.Co && a.Co_p == b.Co_p &&
         a.nchunks == b.nchunks && a.kg == b.kg && a.hb_mul == b.hb_mul && a.hb_add == b.hb_add && a.sgn == b.sgn &&
         a.relu == b.relu && a.out_f32 == b.out_f32 && a.pro_mode == b.pro_mode && a.pro_relu == b.pro_relu &&
         a.wgt_bytes == b.wgt_bytes && nn(a.bias, b.bias) && nn(a.addend, b.addend) && nn(a.mask, b.mask) &&
         nn(a.stats, b.stats) && nn(a.bnb_x, b.bnb_x) && nn(a.bnb_scale, b.bnb_scale) && nn(a.pro_stats, b.pro_stats);
}

int conv3x3_entry(const FsConvArgs* args, const FsConvArgs* b, int dtype, hipStream_t st) {
  int r = conv3x3_check(args, dtype);
  if (r != FS_OK) return r;
  if (b) {
    r = conv3x3_check(b, dtype);
    if (r != FS_OK) return r;
    if (!conv3x3_pairable(*args, *b)) {
      // not one launch's worth of agreement: two launches, same results
      r = conv3x3_entry(args, nullptr, dtype, st);
      return r != FS_OK ? r : conv3x3_entry(b, nullptr, dtype, st);
    }
  }
  // 32x32-tile kernel for the launches it wants (whole 64-byte channel chunks, >= 32 output channels, enough 256-pixel
  // tiles); the 16x16-tile kernel below takes everything else, with the same prologues and epilogues
  if (args->hb_mul == 2) {
    if (dtype == FS_DTYPE_BF16) return dispatch_s2<bf16>(*args, b, st);
    if (dtype == FS_DTYPE_F32) return dispatch_s2<float>(*args, b, st);
    return FS_EINVAL;
  }
  // FsConvArgs.force_impl (tests): 0 = the choice below; 1 = the 16x16-tile kernel; 2-4 = the 32x32-tile kernel in tile
  // configuration 1-3 (FS_EINVAL where it cannot take the launch); 5 = the persistent one-chunk kernel whatever the number
  // of tiles (shapes it does not cover go on to the usual choice).
  const int force = args->force_impl;
  if (force < 0 || force > 5 || (b && b->force_impl != force)) return FS_EINVAL;
  if (!b && (force == 0 || force == 5)) {
    // one-chunk layers with <= 32 output channels and many pixel tiles (the decoder's 192x640 / 96x320 layers): the
    // persistent kernel with resident weights
    r = fs_conv3x3_p1(*args, dtype, st);
    if (r != FS_EINVAL) return r;
  }
  if (force != 1) {
    r = fs_conv3x3_t32(*args, b, dtype, st);
    if (r != FS_EINVAL || (force >= 2 && force <= 4)) return r;
  }
  if (dtype == FS_DTYPE_BF16) return dispatch<bf16>(*args, b, st);
  if (dtype == FS_DTYPE_F32) return dispatch<float>(*args, b, st);
  return FS_EINVAL;
}
}  // namespace

extern "C" int fs_conv3x3_halo(const FsConvArgs* args, int dtype, void* stream) {
  return conv3x3_entry(args, nullptr, dtype, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int fs_conv3x3_halo2(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, void* stream) {
  return conv3x3_entry(a0, a1, dtype, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int fs_conv3x3_halo_plan(const FsConvArgs* args, int dtype, int32_t* plan) {
  if (!plan) return FS_EINVAL;
  plan[0] = -1; plan[1] = plan[2] = plan[3] = 0;
  fs_conv3x3_plan_slot = plan;
  const int r = conv3x3_entry(args, nullptr, dtype, nullptr);
  fs_conv3x3_plan_slot = nullptr;
  return r;
}

extern "C" int fs_conv3x3_halo2_plan(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, int32_t* plan) {
  if (!plan || !a0) return FS_EINVAL;
  plan[0] = -1; plan[1] = plan[2] = plan[3] = 0;
  if (a1 && !(conv3x3_check(a0, dtype) == FS_OK && conv3x3_check(a1, dtype) == FS_OK && conv3x3_pairable(*a0, *a1)))
    return FS_EINVAL;                       // (two launches: ask for each plan separately)
  fs_conv3x3_plan_slot = plan;
  const int r = conv3x3_entry(a0, a1, dtype, nullptr);
  fs_conv3x3_plan_slot = nullptr;
  return r;
}
