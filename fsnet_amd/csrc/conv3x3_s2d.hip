// Data gradient of a 3x3 / stride-2 / pad-1 convolution (the ResNet stage entries, resnet.py:33-50 with stride 2:
// convolution_backward(input) of layer2/3/4[0].conv1) — all four output-parity classes from ONE staged dY halo.
//
// dx[2a+py, 2b+px] only receives the taps with r = py+1 (mod 2), s = px+1 (mod 2):
//   (even, even): W[1][1] dy[a][b]
//   (even, odd ): W[1][0] dy[a][b+1] + W[1][2] dy[a][b]
//   (odd , even): W[0][1] dy[a+1][b] + W[2][1] dy[a][b]
//   (odd , odd ): W[0][0] dy[a+1][b+1] + W[0][2] dy[a+1][b] + W[2][0] dy[a][b+1] + W[2][2] dy[a][b]
// The implicit GEMM (conv_igemm.hip) runs the classes as four problems (blockIdx.y) with K walks of 1 / 2 / 2 / 4 taps:
// blocks that live for two to eight K stages and fetch every dY pixel once per class and tap.  Here a block stages the
// (TH+1) x (TW+1) halo of a TH x TW tile of dY positions and the nine taps' weights per 64-byte channel chunk — the
// staging of conv3x3_t32.hip — and a wave keeps FOUR 32 x 32 accumulator tiles, one per class: nine MFMAs per k step
// from four pixel fragments (the four (da, db) shifts) and nine weight fragments, 2a x 2b x 32 channels written per
// position.  Same arguments as the class launch of fs_conv_igemm (FsConvArgs.ncls = 4: dst / addend / mask / bnb_x are
// the sub-lattice of class (0,0), weights in class order — layout.hip tap_at), same epilogue options as the stride-1
// data gradients (residual addend, ReLU mask, BatchNorm-backward sums per statistics group).
// Reference call sites: vision_base/networks/models/backbone/resnet.py:33-50, 199-213 (autograd of conv1 with stride 2).
#include "t32_common.h"

namespace {

struct S2dGeom {
  int TH, TW;
  int tiles_x, tiles_y;
  unsigned mTW, mHW;
  FsDiv dTX, dTY;
  FsDiv dIPG;      // images per BatchNorm statistics group
  int pix_major;
};

constexpr int S2D_PIX = 128;      // dY positions per block (32 per wave)
constexpr int S2D_CO = 32;        // dx channels per block
constexpr int S2D_HMAX = 192;     // halo positions per stage: (TH+1) * (TW+1)

// class-ordered operand (layout.hip tap_at; conv.py ConvOp._S2_CLASSES): position -> parity class 2 py + px
constexpr int s2d_cls(int pos) { return pos == 0 ? 0 : (pos <= 2 ? 1 : (pos <= 4 ? 2 : 3)); }
// position -> dY shift (da, db) as 2 da + db
constexpr int s2d_off(int pos) { return pos == 1 ? 1 : (pos == 3 ? 2 : (pos == 5 ? 3 : (pos == 6 ? 2 : (pos == 7 ? 1 : 0)))); }

// DS: the data gradient of the block's 1x1 / stride-2 downsample projection (resnet.py:152-160, the other consumer of the
// block input) rides along: it only reaches the (even, even) class — dx[2a][2b] += Wd dcd[a][b] — i.e. one more K segment
// of that class's accumulator, from a second dY-shaped tensor (FsConvArgs.ds_src) and the projection's packed operand
// (ds_wgt): no launch of its own, and its result is never written out and read back as this launch's addend.
template <typename T, int EP, bool DS>
__global__ __launch_bounds__(256, 3) void conv3x3_s2d_kernel(const FsDual<FsConvArgs, S2dGeom> d) {
  const int prob = (int)blockIdx.x >= d.nb0 ? 1 : 0;
  const FsConvArgs& p = d.a[prob];
  const S2dGeom& g = d.g[prob];
  const int bid = (int)blockIdx.x - (prob ? d.nb0 : 0);
  constexpr int PIX = S2D_PIX, CO = S2D_CO, HMAX = S2D_HMAX;
  constexpr int HS = 5;                         // 16-byte units per halo pixel: 4 used + 1 pad (conv3x3_t32.hip)
  constexpr int LH = (HMAX * 4 + 255) / 256;
  constexpr int WU = 9 * CO * 4;
  constexpr int LW = (WU + 255) / 256;
  constexpr int BUFU = HMAX * HS + WU;
  constexpr int DSU = DS ? HMAX * HS + CO * 4 : 1;      // the projection's dY tile (same halo layout) + its weight rows
  constexpr int EPS = 9;
  constexpr int OOB = 0x7ffff000;
  static_assert(BUFU >= PIX * EPS + (4 * CO * 2 + 3) / 4, "epilogue overlay must fit the stage buffer");

  __shared__ uint4 lds[BUFU];
  uint4* const lds_w = lds + HMAX * HS;
  __shared__ uint4 lds_ds[DSU];
  uint4* const lds_dw = lds_ds + (DS ? HMAX * HS : 0);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hk = lane >> 5;
  const int HW = g.TW + 1;
  const int nhalo = (g.TH + 1) * HW;
  const int ntile = g.TH * g.TW;

  const int npix = p.N * g.tiles_y * g.tiles_x, nco = p.Co_p / CO;
  int px, cy;
  {
    const int id = bid, xcd = id & 7, slot = id >> 3;
    if (g.pix_major) { cy = slot % nco; px = (slot / nco) * 8 + xcd; }
    else if (nco % 8 == 0) { const int q = nco >> 3; cy = xcd + 8 * (slot % q); px = slot / q; }
    else if (8 % nco == 0) { const int q = 8 / nco; cy = xcd % nco; px = slot * q + xcd / nco; }
    else { cy = id % nco; px = id / nco; }
    if (px >= npix) return;
  }
  const int tq = fs_div(px, g.dTX); const int tx_i = px - tq * g.tiles_x;
  const int n = fs_div(tq, g.dTY); const int ty_i = tq - n * g.tiles_y;
  const int y0 = ty_i * g.TH, x0 = tx_i * g.TW;
  const int co0 = cy * CO;

  const int row_bytes = p.Cs * (int)sizeof(T);
  const int wrow_bytes = p.wgt_row_bytes ? (int)p.wgt_row_bytes : p.nchunks * p.kg * 16;
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, p.Co_p * wrow_bytes, 0x00020000);

  const int ds_wrow = DS ? (int)p.ds_wgt_row_bytes : 0;
  const __amdgpu_buffer_rsrc_t rs_ds =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(DS ? p.ds_src : p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(DS ? p.ds_wgt : p.wgt), 0, DS ? p.Co_p * ds_wrow : 16, 0x00020000);

  const int q4 = t & 3;
  int hvoff[LH], wvoff[LW];
  // (the projection's weights: 32 rows x one 64-byte chunk = 128 units, threads 0..127)
  const int dwvoff = (DS && t < CO * 4) ? (co0 + (t >> 2)) * ds_wrow + q4 * 16 : OOB;
#pragma unroll
  for (int i = 0; i < LH; ++i) {
    const int hp = (t >> 2) + i * 64;
    const int hy = fs_fastdiv(hp, g.mHW), hx = hp - hy * HW;
    const int sy = y0 + hy, sx = x0 + hx;
    const bool ok = hp < nhalo && sy < p.Hs && sx < p.Ws;
    hvoff[i] = ok ? (int)(((long)n * p.sN + (long)sy * p.sH + (long)sx * p.sW) * (long)sizeof(T)) + q4 * 16 : OOB;
  }
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    const int rt = (t >> 2) + i * 64;
    const int pos = rt / CO, row = rt - pos * CO;
    wvoff[i] = rt < 9 * CO ? (co0 + row) * wrow_bytes + pos * row_bytes + q4 * 16 : OOB;
  }

  uint4 rh[LH], rw[LW];
  uint4 rh2[DS ? LH : 1], rw2;
  auto load_regs = [&](int cc) {
    const int coff = cc * 64;
#pragma unroll
    for (int i = 0; i < LH; ++i)
      rh[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, hvoff[i], coff, 0));
    if constexpr (DS) {
#pragma unroll
      for (int i = 0; i < LH; ++i)
        rh2[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_ds, hvoff[i], coff, 0));
      rw2 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_dw, dwvoff, coff, 0));
    }
#pragma unroll
    for (int i = 0; i < LW; ++i)
      rw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_wgt, wvoff[i], coff, 0));
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < LH; ++i) {
      const int hp = (t >> 2) + i * 64;
      if (64 * (i + 1) <= HMAX || hp < HMAX) {
        lds[hp * HS + q4] = rh[i];
        if constexpr (DS) lds_ds[hp * HS + q4] = rh2[i];
      }
    }
    if constexpr (DS) {
      if (t < CO * 4) lds_dw[(t >> 2) * 4 + (q4 ^ (((t >> 2) >> 2) & 3))] = rw2;
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int rt = (t >> 2) + i * 64;
      if (64 * (i + 1) <= 9 * CO || rt < 9 * CO) lds_w[rt * 4 + (q4 ^ ((rt >> 2) & 3))] = rw[i];
    }
  };

  // per-lane fragment bases: lane (position l31 of the wave's 32, k half hk)
  int hbase;
  {
    int pi = wave * 32 + l31;
    if (pi >= ntile) pi = 0;
    const int ty = fs_fastdiv(pi, g.mTW), tx = pi - ty * g.TW;
    hbase = (ty * HW + tx) * HS + hk;
  }
  const int we = hk ^ ((l31 >> 2) & 3);
  const int wa0 = l31 * 4 + we, wa1 = l31 * 4 + (we ^ 2);

  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;

  const int nchunk = (row_bytes + 63) / 64;
  load_regs(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    t32_barrier();
    store_lds();
    t32_barrier();
    if (cc + 1 < nchunk) load_regs(cc + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fb[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) fb[o] = lds[hbase + ((o >> 1) * HW + (o & 1)) * HS + 2 * ks];
#pragma unroll
      for (int pos = 0; pos < 9; ++pos) {
        const uint4 fa = lds_w[(ks ? wa1 : wa0) + pos * CO * 4];
        Mma32<T>::run(acc[s2d_cls(pos)], fa, fb[s2d_off(pos)]);
      }
      if constexpr (DS) {
        const uint4 fa2 = lds_dw[ks ? wa1 : wa0];
        const uint4 fb2 = lds_ds[hbase + 2 * ks];
        Mma32<T>::run(acc[0], fa2, fb2);
      }
    }
  }

  // ---- epilogue (conv3x3_t32.hip: accumulators through the wave's LDS region, 8 channels of one pixel per lane) ----
  const bool has_add = T32_FLAG(EP_ADDEND, p.addend != nullptr);
  const bool has_mask = T32_FLAG(EP_MASK, p.mask != nullptr);
  const bool has_bnb = T32_FLAG(EP_BNB, p.bnb_x != nullptr);

  t32_barrier();
  float* ep = reinterpret_cast<float*>(lds + wave * (32 * EPS));
  const int l15 = lane & 15, cs = lane >> 4;
  int doff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pi = wave * 32 + i * 16 + l15;
    const int ty = fs_fastdiv(pi, g.mTW), tx = pi - ty * g.TW;
    const int y = y0 + ty, x = x0 + tx;
    const bool mok = pi < ntile && y < p.Hd && x < p.Wd;
    doff[i] = mok ? ((y << 16) | x) : -1;
  }
  float* red = reinterpret_cast<float*>(lds + PIX * EPS);       // [4 waves][CO][2]
  const int co = co0 + cs * 8;
  const bool cok = co < p.Co;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int py = a >> 1, pxx = a & 1;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 v = make_float4(acc[a][4 * gq], acc[a][4 * gq + 1], acc[a][4 * gq + 2], acc[a][4 * gq + 3]);
      *reinterpret_cast<float4*>(ep + (l31 * EPS + 2 * gq + hk) * 4) = v;
    }
    t32_wave_sync();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* rp = ep + ((i * 16 + l15) * EPS + 2 * cs) * 4;
      const float4 v0 = reinterpret_cast<const float4*>(rp)[0], v1 = reinterpret_cast<const float4*>(rp)[1];
      if (doff[i] < 0 || !cok) continue;
      const int y = doff[i] >> 16, x = doff[i] & 0xffff;
      const int dof = n * (int)p.dN + y * (int)p.dH + x * (int)p.dW + py * (int)(p.dH >> 1) + pxx * (int)(p.dW >> 1);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (has_add) {
        float av[8];
        load8<T>(reinterpret_cast<const T*>(p.addend) + n * (int)p.aN + y * (int)p.aH + x * (int)p.aW +
                     py * (int)(p.aH >> 1) + pxx * (int)(p.aW >> 1) + co, av);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += av[j];
      }
      if (has_mask) {
        float mv[8];
        load8<T>(reinterpret_cast<const T*>(p.mask) + n * (int)p.mN + y * (int)p.mH + x * (int)p.mW +
                     py * (int)(p.mH >> 1) + pxx * (int)(p.mW >> 1) + co, mv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
      }
      if (has_bnb) {
        float cv[8];
        load8<T>(reinterpret_cast<const T*>(p.bnb_x) + dof + co, cv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * cv[j]; }
      }
      store8<T>(reinterpret_cast<T*>(p.dst) + dof + co, v);
    }
    t32_wave_sync();
  }
  if (has_bnb) {
    float sv[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sv[j] = s1[j]; sv[8 + j] = s2[j]; }
    const float tot = t32_reduce16_row(sv, lane);
    const int j = ((lane >> 1) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1);
    red[(wave * CO + cs * 8 + j) * 2 + (lane & 1)] = tot;
    t32_barrier();
    if (t < CO) {
      float u = 0.f, w = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { u += red[(k * CO + t) * 2]; w += red[(k * CO + t) * 2 + 1]; }
      const int c = co0 + t;
      if (c < p.Co) {
        const long sg = p.stat_group_rows > 0 ? fs_div(n, g.dIPG) : 0;
        double* sl = p.stats + (sg * FS_STAT_SLOTS + px % FS_STAT_SLOTS) * 2 * p.Co;
        const double wd = ((double)w - (double)p.bnb_mean[sg * p.Co + c] * (double)u) * (double)p.bnb_invstd[sg * p.Co + c];
        atomicAdd(sl + c, (double)u);
        atomicAdd(sl + p.Co + c, wd);
      }
    }
  }
}

// tile of dY positions (TH x TW <= 128, halo (TH+1) x (TW+1) <= S2D_HMAX) that wastes the fewest MFMA lanes
inline S2dGeom s2d_pick_geom(int Hd, int Wd) {
  S2dGeom best{};
  double best_cost = 1e30;
  for (int tw = std::min(4, Wd); tw <= std::min(Wd, 64); ++tw) {
    const int th = std::min(S2D_PIX / tw, Hd);
    if (th < 1 || (th + 1) * (tw + 1) > S2D_HMAX) continue;
    const int tx = (Wd + tw - 1) / tw, ty = (Hd + th - 1) / th;
    const double waste = (double)tx * ty * S2D_PIX / ((double)Hd * Wd);
    const double halo = (double)(th + 1) * (tw + 1) / ((double)th * tw);
    double cost = waste * (1.0 + 0.15 * halo);
    if (tw % 32 != 0 && tw != Wd) cost *= 1.02;
    if (cost < best_cost - 1e-9) { best_cost = cost; best.TH = th; best.TW = tw; best.tiles_x = tx; best.tiles_y = ty; }
  }
  if (best.TW > 0) {
    best.mTW = fs_div_magic(best.TW); best.mHW = fs_div_magic(best.TW + 1);
    best.dTX = fs_make_div(best.tiles_x); best.dTY = fs_make_div(best.tiles_y);
  }
  return best;
}

int s2d_problem(const FsConvArgs& a, S2dGeom& g) {
  g = s2d_pick_geom(a.Hd, a.Wd);
  if (g.TH == 0) return 0;
  g.dIPG = FsDiv{0u, 0u};
  if (a.stat_group_rows > 0) {
    const long hw = (long)a.Hd * a.Wd;
    if (a.stat_group_rows % hw != 0) return 0;
    g.dIPG = fs_make_div((int)(a.stat_group_rows / hw));
  }
  const int npix = a.N * g.tiles_x * g.tiles_y, nco = a.Co_p / S2D_CO;
  int blocks = npix * nco;
  g.pix_major = (nco > 1 && a.src_bytes > 2 * a.wgt_bytes) ? 1 : 0;
  if (g.pix_major) blocks = 8 * ((npix + 7) / 8) * nco;
  else if (nco % 8 != 0 && 8 % nco == 0) { const int q = 8 / nco; blocks = 8 * ((npix + q - 1) / q); }
  return blocks;
}

template <typename T, int EP, bool DS>
int s2d_launch(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  FsDual<FsConvArgs, S2dGeom> d;
  d.a[0] = a; d.a[1] = b ? *b : a;
  d.nprob = b ? 2 : 1;
  int blocks = s2d_problem(a, d.g[0]);
  if (blocks == 0) return FS_EINVAL;
  d.g[1] = d.g[0];
  d.nb0 = blocks;
  if (b) {
    const int b1 = s2d_problem(*b, d.g[1]);
    if (b1 == 0) return FS_EINVAL;
    d.nb0 = fs_xcd_round(blocks);
    blocks = d.nb0 + b1;
  }
  hipLaunchKernelGGL((conv3x3_s2d_kernel<T, EP, DS>), dim3(blocks), dim3(256), 0, st, d);
  return fs_launch_status();
}

inline int s2d_ep_mask(const FsConvArgs& a) {
  return (a.addend ? EP_ADDEND : 0) | (a.mask ? EP_MASK : 0) | (a.bnb_x ? EP_BNB : 0);
}

// (ep: the epilogue options of the launch — of both problems when they agree, else -1 = tested at run time per problem)
template <typename T, bool DS>
int s2d_dispatch_ep(const FsConvArgs& a, const FsConvArgs* b, int ep, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    switch (ep) {
      case EP_ADDEND: return s2d_launch<T, EP_ADDEND, DS>(a, b, st);
      case EP_ADDEND | EP_MASK | EP_BNB: return s2d_launch<T, EP_ADDEND | EP_MASK | EP_BNB, DS>(a, b, st);
      case EP_MASK | EP_BNB: return s2d_launch<T, EP_MASK | EP_BNB, DS>(a, b, st);
      default: break;
    }
  }
  return s2d_launch<T, -1, DS>(a, b, st);
}

template <typename T>
int s2d_dispatch(const FsConvArgs& a, const FsConvArgs* b, hipStream_t st) {
  const int ep = (b && s2d_ep_mask(*b) != s2d_ep_mask(a)) ? -1 : s2d_ep_mask(a);
  return a.ds_src ? s2d_dispatch_ep<T, true>(a, b, ep, st) : s2d_dispatch_ep<T, false>(a, b, ep, st);
}

bool s2d_takes(const FsConvArgs& a, int dtype) {
  const int es = dtype == FS_DTYPE_BF16 ? 2 : 4;
  if (!a.src || !a.wgt || !a.dst) return false;
  if (a.ncls != 4 || a.hb_mul != 1 || a.hb_add != 0 || a.sgn != 1 || a.dshift != 0) return false;
  if (a.Hs != a.Hd || a.Ws != a.Wd || a.N <= 0) return false;
  if ((a.Cs * es) % 64 != 0 || a.Co_p % S2D_CO != 0 || a.Co % 8 != 0) return false;
  if (a.bias || a.relu || a.out_f32 || a.pro_mode || a.bnb_scale || a.grp_imgs) return false;
  if (a.stats && !a.bnb_x) return false;
  if ((a.ds_src != nullptr) != (a.ds_wgt != nullptr)) return false;
  if (a.ds_src && (a.ds_wgt_row_bytes < (int64_t)a.Cs * es || a.ds_wgt_row_bytes * a.Co_p >= 0x7ffff000LL ||
                   ((uintptr_t)a.ds_src & 15u) || ((uintptr_t)a.ds_wgt & 15u)))
    return false;
  if (a.bnb_x && (!a.stats || !a.bnb_mean || !a.bnb_invstd)) return false;
  if (a.src_bytes <= 0 || a.src_bytes >= 0x7ffff000LL) return false;
  const int64_t wrow = a.wgt_row_bytes ? a.wgt_row_bytes : (int64_t)a.nchunks * a.kg * 16;
  if (wrow < (int64_t)9 * a.Cs * es || (int64_t)a.Co_p * wrow >= 0x7ffff000LL) return false;
  // 16 bytes per lane with 32-bit element offsets; the class shift is half the (sub-lattice) row / pixel stride
  auto al16 = [&](const void* ptr, int64_t sn, int64_t sh, int64_t sw) {
    return ptr == nullptr || (((uintptr_t)ptr & 15u) == 0 && (sn * es) % 16 == 0 && (sh % 2) == 0 && (sw % 2) == 0 &&
                              ((sh / 2) * es) % 16 == 0 && ((sw / 2) * es) % 16 == 0);
  };
  if (!al16(a.dst, a.dN, a.dH, a.dW) || !al16(a.addend, a.aN, a.aH, a.aW) || !al16(a.mask, a.mN, a.mH, a.mW)) return false;
  auto span_ok = [&](const void* ptr, int64_t sn, int64_t sh, int64_t sw) {
    if (!ptr) return true;
    if (sn < 0 || sh < 0 || sw < 0) return false;
    return (int64_t)(a.N - 1) * sn + (int64_t)a.Hd * sh + (int64_t)a.Wd * sw + a.Co < 0x7fffffffLL;
  };
  if (!span_ok(a.dst, a.dN, a.dH, a.dW) || !span_ok(a.addend, a.aN, a.aH, a.aW) || !span_ok(a.mask, a.mN, a.mH, a.mW)) return false;
  if (a.Hd >= 0x7fff || a.Wd >= 0xffff) return false;
  S2dGeom g;
  return s2d_problem(a, g) > 0;      // (a tile exists and the statistics groups are whole images: the launch cannot decline)
}

bool s2d_pairable(const FsConvArgs& a, const FsConvArgs& b) {
  return a.Co_p == b.Co_p && a.Cs == b.Cs && (a.ds_src != nullptr) == (b.ds_src != nullptr);
}

int s2d_entry(const FsConvArgs* a, const FsConvArgs* b, int dtype, hipStream_t st) {
  if (!a || !s2d_takes(*a, dtype) || (b && !s2d_takes(*b, dtype))) return FS_EINVAL;
  if (b && !s2d_pairable(*a, *b)) {
    const int r = s2d_entry(a, nullptr, dtype, st);
    return r != FS_OK ? r : s2d_entry(b, nullptr, dtype, st);
  }
  if (dtype == FS_DTYPE_BF16) return s2d_dispatch<bf16>(*a, b, st);
  if (dtype == FS_DTYPE_F32) return s2d_dispatch<float>(*a, b, st);
  return FS_EINVAL;
}

}  // namespace

// FS_EINVAL = "not mine" (the caller's launch chain continues with fs_conv_igemm, which takes every class launch)
extern "C" int fs_conv3x3_s2d(const FsConvArgs* args, int dtype, void* stream) {
  return s2d_entry(args, nullptr, dtype, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int fs_conv3x3_s2d2(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, void* stream) {
  return s2d_entry(a0, a1, dtype, reinterpret_cast<hipStream_t>(stream));
}
