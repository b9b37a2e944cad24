// Implicit-GEMM convolution (forward and data-gradient) on CDNA4 MFMA.
//
// Replaces, on the monodepth hot path, every nn.Conv2d forward and
// convolution_backward(input) that the reference issues through ATen/cuDNN:
//   vision_base/networks/models/backbone/resnet.py:6-9,119,148-160 (encoder convs)
//   vision_base/networks/blocks/blocks.py:41-46 (ConvBnReLU conv)
//   monodepth/networks/models/heads/depth_encoder.py:45-63 (decoder / dispconv)
//   monodepth/networks/models/heads/pose_decoder.py:17-21 (pose convs)
//
// Layout: activations NHWC with arbitrary (n,h,w) element strides and the channel
// axis contiguous; weights packed [co_p][ktot_p] with K = (r, s, c) contiguous.
// One block computes a PIX x CO output tile.  The GEMM is issued "swapped":
// MFMA A = weights (rows = output channels), MFMA B = gathered pixels, so that
// each lane ends up with 4 consecutive channels of one pixel (one 8/16-byte store).
// K is walked in 64- or 128-byte stages (kg = 4 / 8 sixteen-byte groups); both operand
// tiles are staged through double-buffered, XOR-swizzled LDS with register prefetch.
//
// The gather is the VALU-critical part (first version: 16 VALU per MFMA, MFMA pipe 15 % busy):
//   * operands are fetched with raw buffer loads: an out-of-range offset returns zeros in hardware,
//     so zero padding / ragged tiles need no exec-mask branches;
//   * the host supplies, per K unit, the byte delta of the tap (r*sH + s*sW + c) and the signed
//     (r, s) pair, so a lane adds one table value to its per-pixel base offset (32-bit);
//   * with kg = 8 one address computation feeds four 16-byte loads (a 64-byte unit of one tap).
// The same kernel computes dgrad: source = dY, packed weights Wt[ci][r][s][co], taps walked with
// sgn = -1, plus a parity test and halved strides for stride 2.
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

template <int KG> __device__ __forceinline__ int lds_swz(int row) {
  // XOR swizzle of the 16-byte group index that makes the ds_read_b128 lane groups conflict-free
  // (64-byte rows: 2 bits from row bit 3; 128-byte rows: 3 bits from row bits 1..3)
  return KG == 4 ? (((row >> 3) & 1) << 1) : ((row >> 1) & 7);
}

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

struct IgGeom { FsDiv dW, dH; };

template <typename T, int PIX, int CO, int WP, int KG>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const FsDual<FsConvArgs, IgGeom> d) {
  // two problems per launch (fsnet_hip_internal.h, FsDual): blocks [0, nb0) of blockIdx.x take the first argument set
  const int prob = (int)blockIdx.x >= d.nb0 ? 1 : 0;
  const FsConvArgs& p = d.a[prob];
  const FsDiv dW = d.g[prob].dW, dH = d.g[prob].dH;
  const int bid = (int)blockIdx.x - (prob ? d.nb0 : 0);
  // (blockIdx.z spans the larger group count of the two problems)
  if ((int)blockIdx.z >= (p.grp_imgs > 0 ? p.N / p.grp_imgs : 1)) return;
  using TR = ElemTraits<T>;
  constexpr int WC = 4 / WP;
  constexpr int WPIX = PIX / WP, WCO = CO / WC;
  constexpr int TP = WPIX / 16, TC = WCO / 16;
  constexpr int UG = KG == 8 ? 4 : 1;          // 16-byte groups fetched per address computation
  constexpr int UPR = KG / UG;                 // units per tile row per stage
  constexpr int LPU = (PIX * UPR + 255) / 256;
  constexpr int LCU = (CO * UPR + 255) / 256;
  constexpr int UPRS = UPR == 1 ? 0 : (UPR == 2 ? 1 : 2);
  static_assert(WPIX % 16 == 0 && WCO % 16 == 0, "tile");

  __shared__ uint4 lds_p[2][PIX * KG];
  __shared__ uint4 lds_c[2][CO * KG];
  __shared__ int2 s_ktab[1024];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wp = wave % WP, wc = wave / WP;
  const int li = lane & 15, lg = lane >> 4;
  // blockIdx.y = output-parity class of a stride-2 data gradient (conv.py): its own K slice of the operand, its own
  // unit table, and the (y%2, x%2) = (cls>>1, cls&1) sub-lattice of dst / addend / mask / bnb_x (whose strides are
  // the doubled ones of the sub-lattice)
  const int cls = p.ncls > 1 ? blockIdx.y : 0;
  // blockIdx.z = BatchNorm statistics group when its rows are not a multiple of the pixel tile (grp_imgs images per
  // group, M = rows of ONE group): a tile then never straddles two groups
  const int n0 = p.grp_imgs > 0 ? (int)blockIdx.z * p.grp_imgs : 0;
  const int nch = p.ncls > 1 ? p.cls_nch[cls] : p.nchunks;

  // XCD-aware tile mapping (block b runs on XCD b % 8, each XCD has a private 4 MB L2): all pixel tiles of one
  // channel tile go to the same XCD(s), so a weight tile is fetched from HBM/MALL once per XCD and then hit in L2.
  int px, cy;
  {
    const int npix = (p.M + PIX - 1) / PIX, nco = p.Co_p / CO;
    const int id = bid, xcd = id & 7, slot = id >> 3;
    if (nco % 8 == 0) { const int g = nco >> 3; cy = xcd + 8 * (slot % g); px = slot / g; }
    else if (8 % nco == 0) { const int g = 8 / nco; cy = xcd % nco; px = slot * g + xcd / nco; }
    else { cy = id % nco; px = id / nco; }
    if (px >= npix) return;
  }

  {
    const int2* kt = reinterpret_cast<const int2*>(p.ktab) + (p.ncls > 1 ? p.cls_ktab_off[cls] : 0);
    for (int i = t; i < nch * UPR; i += 256) s_ktab[i] = kt[i];
  }

  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.wgt)) +
                                            (p.ncls > 1 ? p.cls_wgt_off[cls] : 0), 0, (int)p.wgt_bytes, 0x00020000);

  // ---- per-thread gather units (fixed over the K walk) ----
  int pvoff[LPU], phb[LPU], pwb[LPU];
  const int pix0 = px * PIX;
  const int sHe = (int)(p.sH >> p.dshift), sWe = (int)(p.sW >> p.dshift);
#pragma unroll
  for (int i = 0; i < LPU; ++i) {
    int idx = t + i * 256;
    int row = idx >> UPRS;
    int m = pix0 + row;
    if (row < PIX && m < p.M) {
      int q = fs_div(m, dW); int x = m - q * p.Wd; int n = fs_div(q, dH); int y = q - n * p.Hd; n += n0;
      phb[i] = y * p.hb_mul + p.hb_add;
      pwb[i] = x * p.hb_mul + p.hb_add;
      pvoff[i] = (int)((n * p.sN + (long)phb[i] * sHe + (long)pwb[i] * sWe) * (long)sizeof(T));
    } else {
      pvoff[i] = 0; phb[i] = -(1 << 28); pwb[i] = -(1 << 28);
    }
  }
  const int stage_bytes = KG * 16;
  const int wrow_bytes = p.wgt_row_bytes ? (int)p.wgt_row_bytes : nch * stage_bytes;
  const int co0 = cy * CO;
  const int OOB = 0x7fffffff;
  int wvoff[LCU];
#pragma unroll
  for (int i = 0; i < LCU; ++i) {
    int idx = t + i * 256;
    int row = idx >> UPRS, qu = idx & (UPR - 1);
    wvoff[i] = row < CO ? ((co0 + row) * wrow_bytes + qu * UG * 16) : OOB;
  }
  const int dmask = (1 << p.dshift) - 1;

  // two register sets: up to two K stages are in flight behind the one being multiplied
  uint4 rpA[LPU][UG], rcA[LCU][UG], rpB[LPU][UG], rcB[LCU][UG];
  auto load_regs = [&](int kc, uint4 (&rp)[LPU][UG], uint4 (&rc)[LCU][UG]) {
#pragma unroll
    for (int i = 0; i < LPU; ++i) {
      int idx = t + i * 256;
      int qu = idx & (UPR - 1);
      int2 e = s_ktab[kc * UPR + qu];
      int r = (int)(short)(e.y & 0xffff), s = e.y >> 16;
      int h = phb[i] + r, w = pwb[i] + s;
      bool ok = (e.x != (int)0x80000000) && (((h | w) & dmask) == 0) &&
                ((unsigned)(h >> p.dshift) < (unsigned)p.Hs) && ((unsigned)(w >> p.dshift) < (unsigned)p.Ws);
      int voff = ok ? pvoff[i] + e.x : OOB;
#pragma unroll
      for (int j = 0; j < UG; ++j) rp[i][j] = buf_load16(rs_src, voff + j * 16);
    }
#pragma unroll
    for (int i = 0; i < LCU; ++i) {
      int voff = wvoff[i] == OOB ? OOB : wvoff[i] + kc * stage_bytes;
#pragma unroll
      for (int j = 0; j < UG; ++j) rc[i][j] = buf_load16(rs_wgt, voff + j * 16);
    }
  };
  auto store_lds = [&](int buf, const uint4 (&rp)[LPU][UG], const uint4 (&rc)[LCU][UG]) {
#pragma unroll
    for (int i = 0; i < LPU; ++i) {
      int idx = t + i * 256;
      int row = idx >> UPRS, qu = idx & (UPR - 1);
      if (row < PIX) {
#pragma unroll
        for (int j = 0; j < UG; ++j) lds_p[buf][row * KG + ((qu * UG + j) ^ lds_swz<KG>(row))] = rp[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < LCU; ++i) {
      int idx = t + i * 256;
      int row = idx >> UPRS, qu = idx & (UPR - 1);
      if (row < CO) {
#pragma unroll
        for (int j = 0; j < UG; ++j) lds_c[buf][row * KG + ((qu * UG + j) ^ lds_swz<KG>(row))] = rc[i][j];
      }
    }
  };

  f32x4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < KG / 4; ++kk) {
      uint4 fa[TC], fb[TP];
#pragma unroll
      for (int a = 0; a < TC; ++a) {
        int row = wc * WCO + a * 16 + li;
        fa[a] = lds_c[buf][row * KG + ((kk * 4 + lg) ^ lds_swz<KG>(row))];
      }
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        int row = wp * WPIX + b * 16 + li;
        fb[b] = lds_p[buf][row * KG + ((kk * 4 + lg) ^ lds_swz<KG>(row))];
      }
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
  };

  __syncthreads();  // s_ktab visible
  load_regs(0, rpA, rcA);
  if (nch > 1) load_regs(1, rpB, rcB);
  store_lds(0, rpA, rcA);
  __syncthreads();

  // stage kc lives in LDS[kc & 1]; set A carries even stages, set B odd stages.  While stage kc is
  // multiplied, stage kc+1 is landing in its register set and stage kc+2 is being requested.
  for (int kc = 0; kc < nch; kc += 2) {
    if (kc + 2 < nch) load_regs(kc + 2, rpA, rcA);
    compute(0);
    if (kc + 1 < nch) store_lds(1, rpB, rcB);
    __syncthreads();
    if (kc + 1 >= nch) break;
    if (kc + 3 < nch) load_regs(kc + 3, rpB, rcB);
    compute(1);
    if (kc + 2 < nch) store_lds(0, rpA, rcA);
    __syncthreads();
  }

  // ---- epilogue: lane holds, for pixel (tile b, li), channels lg*4..lg*4+3 of channel tile a ----
  float s1[TC][4], s2[TC][4];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }
  const long sgoff = p.grp_imgs > 0 ? (long)blockIdx.z * p.Co
                                    : (p.stat_group_rows > 0 ? (long)(pix0 / p.stat_group_rows) * p.Co : 0);

#pragma unroll
  for (int b = 0; b < TP; ++b) {
    int m = pix0 + wp * WPIX + b * 16 + li;
    bool mok = m < p.M;
    int x = 0, y = 0, n = 0;
    if (mok) { int q = fs_div(m, dW); x = m - q * p.Wd; n = fs_div(q, dH); y = q - n * p.Hd; n += n0; }
    long doff = (long)n * p.dN + (long)y * p.dH + (long)x * p.dW;
    long aoff = (long)n * p.aN + (long)y * p.aH + (long)x * p.aW;
    long moff = (long)n * p.mN + (long)y * p.mH + (long)x * p.mW;
    if (cls) {
      const int py = cls >> 1, px2 = cls & 1;
      doff += py * (p.dH >> 1) + px2 * (p.dW >> 1);
      aoff += py * (p.aH >> 1) + px2 * (p.aW >> 1);
      moff += py * (p.mH >> 1) + px2 * (p.mW >> 1);
    }
#pragma unroll
    for (int a = 0; a < TC; ++a) {
      int co = co0 + wc * WCO + a * 16 + lg * 4;
      if (!mok || co >= p.Co) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (p.bias) {
        float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (p.addend) {
        float av[4];
        load4<T>(reinterpret_cast<const T*>(p.addend) + aoff + co, av);
        v[0] += av[0]; v[1] += av[1]; v[2] += av[2]; v[3] += av[3];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (p.mask) {  // ReLU backward of the producing layer: pass the gradient where its output was > 0
        float mv[4];
        load4<T>(reinterpret_cast<const T*>(p.mask) + moff + co, mv);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
      }
      if (p.bnb_x) {   // BatchNorm-backward sums of the layer this gradient flows into: (sum g, sum g*xhat)
        float cv[4];
        load4<T>(reinterpret_cast<const T*>(p.bnb_x) + doff + co, cv);
        const float4 mu = *reinterpret_cast<const float4*>(p.bnb_mean + sgoff + co);
        const float4 is = *reinterpret_cast<const float4*>(p.bnb_invstd + sgoff + co);
        s1[a][0] += v[0]; s1[a][1] += v[1]; s1[a][2] += v[2]; s1[a][3] += v[3];
        s2[a][0] += v[0] * (cv[0] - mu.x) * is.x; s2[a][1] += v[1] * (cv[1] - mu.y) * is.y;
        s2[a][2] += v[2] * (cv[2] - mu.z) * is.z; s2[a][3] += v[3] * (cv[3] - mu.w) * is.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[a][j] += v[j]; s2[a][j] += v[j] * v[j]; }
      }
      if (p.out_f32) store4<float>(reinterpret_cast<float*>(p.dst) + doff + co, v);
      else store4<T>(reinterpret_cast<T*>(p.dst) + doff + co, v);
    }
  }

  if (p.stats) {
    // per-channel (sum, sumsq): 16 pixel lanes by shuffle, the WP pixel-waves through LDS, then ONE f64
    // atomic per channel per block into one of FS_STAT_SLOTS address slots (same-address atomics cost
    // ~12 ns each on MI355X; slots + block reduce keep the chain per address short).
    __syncthreads();                     // all waves are done reading the operand tiles
    float* red = reinterpret_cast<float*>(&lds_p[0][0]);   // [WP][CO][2]
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
    __syncthreads();
    if (t < CO) {
      float u = 0.f, w = 0.f;
#pragma unroll
      for (int k = 0; k < WP; ++k) { u += red[(k * CO + t) * 2]; w += red[(k * CO + t) * 2 + 1]; }
      int co = co0 + t;
      if (co < p.Co) {
        // statistics group of this tile (groups are multiples of the tile height: fs_conv_igemm checks)
        const long sg = p.grp_imgs > 0 ? (long)blockIdx.z : (p.stat_group_rows > 0 ? pix0 / p.stat_group_rows : 0);
        double* sl = p.stats + (sg * FS_STAT_SLOTS + px % FS_STAT_SLOTS) * 2 * p.Co;
        atomicAdd(sl + co, (double)u);
        atomicAdd(sl + p.Co + co, (double)w);
      }
    }
  }
}

template <int PIX, int CO>
int igemm_blocks(const FsConvArgs& a) {
  const int npix = (a.M + PIX - 1) / PIX, nco = a.Co_p / CO;
  int blocks = npix * nco;
  if (nco % 8 != 0 && 8 % nco == 0) { const int g = 8 / nco; blocks = 8 * ((npix + g - 1) / g); }
  return blocks;
}

// b != nullptr: a second problem of the same shape class in the same launch
template <typename T, int PIX, int CO, int WP, int KG>
int launch_tile(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  FsDual<FsConvArgs, IgGeom> d;
  d.a[0] = a; d.a[1] = b ? *b : a;
  d.g[0] = IgGeom{fs_make_div(a.Wd), fs_make_div(a.Hd)};
  d.g[1] = b ? IgGeom{fs_make_div(b->Wd), fs_make_div(b->Hd)} : d.g[0];
  d.nprob = b ? 2 : 1;
  int blocks = igemm_blocks<PIX, CO>(a);
  d.nb0 = blocks;
  int nz = a.grp_imgs > 0 ? a.N / a.grp_imgs : 1;
  if (b) {
    d.nb0 = fs_xcd_round(blocks);
    blocks = d.nb0 + igemm_blocks<PIX, CO>(*b);
    nz = std::max(nz, b->grp_imgs > 0 ? b->N / b->grp_imgs : 1);
  }
  hipLaunchKernelGGL((conv_igemm_kernel<T, PIX, CO, WP, KG>), dim3(blocks, a.ncls > 1 ? a.ncls : 1, nz), dim3(256), 0, st, d);
  return fs_launch_status();
}

// tile choice: channel tile = largest of {128,64,32,16} dividing Co_p; shrink the pixel tile when the
// grid would not fill 256 CUs.  kg (16-byte K groups per stage, 4 or 8) is fixed by the caller's packing.
template <typename T>
int launch_conv(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  const int cop = a.Co_p;
  const bool k8 = a.kg == 8;
  // rows of the whole launch (all statistics groups / parity classes / both problems) decide the pixel tile
  auto rows = [](const FsConvArgs& q) { return (long)q.M * (q.grp_imgs > 0 ? q.N / q.grp_imgs : 1); };
  const long mtot = rows(a) + (b ? rows(*b) : 0);
  const long m128 = b ? (mtot + 127) / 128 : (long)((a.M + 127) / 128);
  if (cop % 128 == 0 && !k8) {
    long blocks = m128 * (cop / 128);
    if (blocks >= 512) return launch_tile<T, 128, 128, 2, 4>(a, b, st);
    return launch_tile<T, 64, 64, 2, 4>(a, b, st);
  }
  if (cop % 64 == 0) {
    long blocks = m128 * (cop / 64);
    if (blocks >= 512) return k8 ? launch_tile<T, 128, 64, 2, 8>(a, b, st) : launch_tile<T, 128, 64, 2, 4>(a, b, st);
    return k8 ? launch_tile<T, 64, 64, 2, 8>(a, b, st) : launch_tile<T, 64, 64, 2, 4>(a, b, st);
  }
  if (cop % 32 == 0) return k8 ? launch_tile<T, 128, 32, 4, 8>(a, b, st) : launch_tile<T, 128, 32, 4, 4>(a, b, st);
  if (cop % 16 == 0 && !k8) return launch_tile<T, 256, 16, 4, 4>(a, b, st);
  return FS_EINVAL;
}

}  // namespace

namespace {
int igemm_check(const FsConvArgs* args) {
  if (!args || !args->src || !args->wgt || !args->dst || !args->ktab) return FS_EINVAL;
  if (args->kg != 4 && args->kg != 8) return FS_EINVAL;
  const int units = args->nchunks * (args->kg == 8 ? 2 : 4);
  if (args->nchunks <= 0 || units > 1024) return FS_EINVAL;
  if (args->Co % 4 != 0 || args->Co_p % 16 != 0) return FS_EINVAL;
  if (args->src_bytes <= 0 || args->src_bytes > 0x7fffffffLL || args->wgt_bytes <= 0 ||
      args->wgt_bytes > 0x7fffffffLL)
    return FS_EINVAL;
  if (args->dshift && ((args->sH | args->sW) & 1)) return FS_EINVAL;
  if (args->stats && args->stat_group_rows > 0 && args->stat_group_rows % 256 != 0) return FS_EINVAL;
  if (args->grp_imgs > 0 && (args->N <= 0 || args->N % args->grp_imgs != 0)) return FS_EINVAL;
  return FS_OK;
}
// everything that selects an instantiation or the grid's y extent must agree
bool igemm_pairable(const FsConvArgs& a, const FsConvArgs& b) {
  return a.Co_p == b.Co_p && a.kg == b.kg && (a.ncls > 1 ? a.ncls : 1) == (b.ncls > 1 ? b.ncls : 1);
}
}  // namespace

// a1 != NULL: a second convolution in the same launch when the two agree on channel tile, K stage depth and parity-class
// count (otherwise two launches, same results)
extern "C" int fs_conv_igemm2(const FsConvArgs* args, const FsConvArgs* a1, int dtype, void* stream) {
  int r = igemm_check(args);
  if (r != FS_OK) return r;
  if (a1) {
    r = igemm_check(a1);
    if (r != FS_OK) return r;
    if (!igemm_pairable(*args, *a1)) {
      r = fs_conv_igemm2(args, nullptr, dtype, stream);
      return r != FS_OK ? r : fs_conv_igemm2(a1, nullptr, dtype, stream);
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FS_DTYPE_BF16) return launch_conv<bf16>(*args, a1, st);
  if (dtype == FS_DTYPE_F32) return launch_conv<float>(*args, a1, st);
  return FS_EINVAL;
}

extern "C" int fs_conv_igemm(const FsConvArgs* args, int dtype, void* stream) {
  return fs_conv_igemm2(args, nullptr, dtype, stream);
}
