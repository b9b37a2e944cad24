// 1x1 convolution (forward, strided forward, stride-1 data gradient) as an LDS-staged bf16 GEMM on CDNA4.
//
// The Bottleneck's 1x1 convolutions and downsample projections (vision_base/networks/models/backbone/resnet.py:52-89,
// 119) at ResNet-50 / 320x1024 are GEMMs with M = N*H*W = 2.5 k .. 328 k rows and K, Co in 64 .. 2048: 148 launches
// and a third of that step.  The row-streaming kernel (conv1x1.hip) reads its pixel operand in MFMA-fragment shape
// (16 rows x 64 bytes per load instruction: half-used cache lines, every pixel row re-read by each 64-channel tile) and
// multiplies 64 x 32 wave tiles whose weight fragments alone saturate the LDS; it ran at 50-200 TFLOP/s.  Here:
//   * block tile PIX x CO = 128 pixels x 128 (64) channels (256-pixel tiles: a probe configuration, measured no better),
//     four waves splitting the pixels; a wave multiplies its 32 pixels x all channels with v_mfma_f32_32x32x16_bf16;
//   * both operands go global -> LDS by `buffer_load_dwordx4 ... lds` (lds_dma.h: no staging registers, whole 64-byte row
//     pieces per four lanes, out-of-range rows read as zero), a ring of two (K = 2048: three) 32-channel K stages, one
//     barrier per stage; the LDS image is lane-linear, so the 16-byte units of a row are XOR-permuted on the SOURCE side
//     and un-permuted by the fragment read ((row >> 2) & 3: the four 16-lane groups of a ds_read_b128 then hit 64
//     distinct banks); 35 KB of LDS and <= 128 registers: four blocks per CU, whose load / multiply / store phases overlap;
//   * the epilogue transposes the accumulators through LDS (fp32, per wave, 64 channels per pass) so that every global
//     access of the epilogue — addend, ReLU-backward mask, BatchNorm-backward operand, the store — is a 16-byte piece of
//     a contiguous pixel row (the fragment layout gives 8-byte pieces of 32 different rows); it is compiled per mode
//     (EPI: forward / data gradient / general), the data gradient's requesting the next row's operands a row ahead;
//   * a block walks several pixel tiles where BatchNorm sums are carried and adds them with one round of f64 atomics;
//   * channel tiles of one pixel tile run back to back on one XCD, so the pixel rows come from HBM once.
// Epilogue semantics are conv1x1.hip's / conv_igemm.hip's (bias, addend, ReLU, mask, BatchNorm statistics or
// backward sums with statistics groups, fp32 output).  Measurements: DESIGN section 18.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "fsnet_hip_internal.h"
#include "lds_dma.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int CO, int PIX, int NST>
struct GemmCfg {
  static constexpr int WPIX = PIX / 4;                 // pixels of a wave
  static constexpr int TP = WPIX / 32, TC = CO / 32;   // 32x32 accumulator tiles of a wave: TC x TP
  static constexpr int STAGE = (PIX + CO) * 64;        // one K stage: [PIX pixel rows][CO weight rows] x 64 bytes
  static constexpr int HC = 64;                        // channels per epilogue pass
  static constexpr int OROW = HC * 4 + 16;             // fp32 staging row of the epilogue (padded)
  static constexpr int OSTG = 32 * OROW;               // 32 pixel rows per wave and pass
  static constexpr int WORK = NST * STAGE > 4 * OSTG ? NST * STAGE : 4 * OSTG;   // stages, aliased by the staging rows
  static constexpr int LDS = WORK + 4 * CO * 2 * 4;    // + the waves' statistics sums [4][CO][2] fp32
};

// NST: K stages resident in LDS (NST - 1 in flight while one is multiplied).  EPI: 0 the forward's epilogue (bias, addend, ReLU,
// output statistics), 1 the data gradient's (addend, ReLU-backward mask, BatchNorm-backward sums: no bias, no ReLU — the
// operands of the NEXT pixel row are requested before the current row is processed), 2 every option (the first version).
template <int CO, int PIX, int NST, int EPI>
__global__ __launch_bounds__(256, NST == 2 && PIX == 128 ? (EPI == 1 ? 3 : 4) : 2) void conv1x1_gemm_kernel(const FsConvArgs p, const FsDiv dW, const FsDiv dH,
                                                               const int nco, const int pt) {
  using G = GemmCfg<CO, PIX, NST>;
  constexpr bool FWD = EPI == 0, DGR = EPI == 1;
  constexpr int WPIX = G::WPIX, TP = G::TP, TC = G::TC, STAGE = G::STAGE, OROW = G::OROW, OSTG = G::OSTG, HC = G::HC;
  constexpr int NLP = WPIX / 16, NLW = CO / 64;        // load instructions per wave and stage: pixels, weights
  constexpr int LPS = NLP + NLW;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = p.grp_imgs > 0 ? (int)blockIdx.z * p.grp_imgs : 0;
  const int npix = (p.M + PIX - 1) / PIX;
  // XCD-aware mapping: block b runs on XCD b % 8; consecutive slots of an XCD walk the channel tiles of ONE group of pt
  // consecutive pixel tiles.  A block multiplies its pt pixel tiles one after the other and adds the statistics of all
  // of them with one round of atomics (f64 atomics run at ~65 G/s device-wide whatever their scope — the compiler emits
  // the same instruction for workgroup and agent scope: at one 128-pixel tile per block they were a third of the
  // 64 -> 256 forward).
  const int slot = (int)blockIdx.x >> 3;
  const int spx = ((int)blockIdx.x & 7) + 8 * (slot / nco), cy = slot % nco;
  if (spx * pt >= npix) return;
  const int co0 = cy * CO;
  const int OOB = 0x7fffffff;
  const i32x4 rs_src = make_rsrc(p.src, p.src_bytes), rs_wgt = make_rsrc(p.wgt, p.wgt_bytes);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;

  // ---- loader: lane l of a load instruction fills LDS bytes [16 l, 16 l + 16) of a 16-row x 64-byte piece: row l >> 2,
  // slot l & 3, which holds source unit (l & 3) ^ ((l >> 4) & 3) ----
  const int lrow = lane >> 2, sunit = (lane & 3) ^ ((lane >> 4) & 3);
  int poff[NLP], woff[NLW];
  const int wrow_bytes = p.wgt_row_bytes ? (int)p.wgt_row_bytes : p.nchunks * p.kg * 16;
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    const int row = co0 + wave * (CO / 4) + i * 16 + lrow;
    woff[i] = row < p.Co_p ? row * wrow_bytes + sunit * 16 : OOB;
  }
  auto issue = [&](int kt, int buf) {
    const int kb = kt * 64;
    const unsigned base = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NLP; ++i)
      glds16(rs_src, poff[i] == OOB ? OOB : poff[i] + kb, base + (wave * WPIX + i * 16) * 64);
#pragma unroll
    for (int i = 0; i < NLW; ++i)
      glds16(rs_wgt, woff[i] == OOB ? OOB : woff[i] + kb, base + (PIX + wave * (CO / 4) + i * 16) * 64);
  };

  // ---- fragment reads: row lane & 31 of a 32-row tile, K unit kk*2 + (lane >> 5), un-permuted ----
  int foff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) foff[kk] = (lane & 31) * 64 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 2) & 3)) * 16);

  f32x16 acc[TC][TP];

  auto multiply = [&](const unsigned char* sb) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 fa[TC], fb[TP];
#pragma unroll
      for (int b = 0; b < TP; ++b)
        fb[b] = *reinterpret_cast<const uint4*>(sb + (wave * WPIX + b * 32) * 64 + foff[kk]);
#pragma unroll
      for (int a = 0; a < TC; ++a)
        fa[a] = *reinterpret_cast<const uint4*>(sb + (PIX + a * 32) * 64 + foff[kk]);
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]),
                                                              __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
    }
  };
  // Ring of NST stages, one barrier per stage: behind it stage kt has landed for every wave and stage kt-1 is fully
  // multiplied, so its buffer takes stage kt+NST-1; NST-2 later stages stay in flight across the barrier (the loads
  // retire in order: vmcnt counts the ones issued after stage kt).
  const int nkt = p.Cs / 32;
  constexpr int NH = CO / HC;                  // channel passes of the epilogue
  const int u = lane & 7, pr = lane >> 3;
  float* ssum = reinterpret_cast<float*>(lds + G::WORK) + wave * CO * 2;   // this wave's [CO][2], summed over its pixel tiles
  for (int i = lane; i < CO * 2; i += 64) ssum[i] = 0.f;

  for (int tile = 0; tile < pt; ++tile) {
  const int px = spx * pt + tile;
  if (px >= npix) break;
  const int pix0 = px * PIX;
  if (tile > 0) __syncthreads();              // the staging rows of the previous tile alias the operand stages
#pragma unroll
  for (int i = 0; i < NLP; ++i) {
    const int m = pix0 + wave * WPIX + i * 16 + lrow;
    if (m < p.M) {
      int q = fs_div(m, dW); int x = m - q * p.Wd; int n = fs_div(q, dH); int y = q - n * p.Hd; n += n0;
      poff[i] = (int)((n * p.sN + (long)(y * p.hb_mul) * p.sH + (long)(x * p.hb_mul) * p.sW) * 2) + sunit * 16;
    } else {
      poff[i] = OOB;
    }
  }
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) issue(s, s);
  int cb = 0, ib = NST - 1;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + NST - 2 < nkt) fs_wait_vm<(NST - 2) * LPS>();
    else fs_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nkt) issue(kt + NST - 1, ib);
    multiply(lds + cb * STAGE);
    cb = cb + 1 == NST ? 0 : cb + 1;
    ib = ib + 1 == NST ? 0 : ib + 1;
  }
  __syncthreads();              // every wave is done with the operand stages: the staging rows below alias them

  // ---- epilogue.  Accumulator tile (a, b): lane holds pixel b*32 + (lane & 31), channels a*32 + 8 q + 4 (lane >> 5)
  // + 0..3 in registers 4 q .. 4 q + 3.  Through the wave's staging rows (32 pixels x 64 channels per pass) it becomes:
  // lane = (pixel row lane / 8 + 8 i, 8-channel unit lane % 8) — 16-byte pieces of contiguous rows ----
  unsigned char* stg = lds + wave * OSTG;
  const long sgoff = p.grp_imgs > 0 ? (long)blockIdx.z * p.Co
                                    : (p.stat_group_rows > 0 ? (long)(pix0 / p.stat_group_rows) * p.Co : 0);
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int co = co0 + h * HC + u * 8;
    const bool cok = co < p.Co;
    float bias[DGR ? 1 : 8], mu[FWD ? 1 : 8], is[FWD ? 1 : 8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < (DGR ? 1 : 8); ++j) bias[j] = 0.f;
#pragma unroll
    for (int j = 0; j < (FWD ? 1 : 8); ++j) { mu[j] = 0.f; is[j] = 0.f; }
    if constexpr (!DGR) if (cok && p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
      bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    if constexpr (!FWD) if (cok && p.bnb_x) {
      const float4 m0 = *reinterpret_cast<const float4*>(p.bnb_mean + sgoff + co), m1 = *reinterpret_cast<const float4*>(p.bnb_mean + sgoff + co + 4);
      const float4 i0 = *reinterpret_cast<const float4*>(p.bnb_invstd + sgoff + co), i1 = *reinterpret_cast<const float4*>(p.bnb_invstd + sgoff + co + 4);
      mu[0] = m0.x; mu[1] = m0.y; mu[2] = m0.z; mu[3] = m0.w; mu[4] = m1.x; mu[5] = m1.y; mu[6] = m1.z; mu[7] = m1.w;
      is[0] = i0.x; is[1] = i0.y; is[2] = i0.z; is[3] = i0.w; is[4] = i1.x; is[5] = i1.y; is[6] = i1.z; is[7] = i1.w;
    }
#pragma unroll
    for (int b = 0; b < TP; ++b) {
      if (h + b > 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
      for (int a2 = 0; a2 < HC / 32; ++a2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x16& c = acc[h * (HC / 32) + a2][b];
          *reinterpret_cast<float4*>(stg + (lane & 31) * OROW + (a2 * 32 + 8 * q + 4 * (lane >> 5)) * 4) =
              make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if constexpr (DGR) {
        // one row ahead: the (up to three) 16-byte operands of row i + 1 are in flight while row i is processed.  A row
        // outside the problem reads element 0 and neither stores nor counts.
        uint4 ea = make_uint4(0, 0, 0, 0), em = ea, ex = ea;
        long doff = 0;
        bool ok = false;
        auto request = [&](int i, uint4& qa, uint4& qm, uint4& qx, long& qoff, bool& qok) {
          const int m = pix0 + wave * WPIX + b * 32 + i * 8 + pr;
          qok = m < p.M && cok;
          const int mm = qok ? m : 0, cc = qok ? co : 0;
          int qd = fs_div(mm, dW); int x = mm - qd * p.Wd; int n = fs_div(qd, dH); int y = qd - n * p.Hd; n += n0;
          qoff = (long)n * p.dN + (long)y * p.dH + (long)x * p.dW + cc;
          if (p.addend) qa = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.addend) + (long)n * p.aN + (long)y * p.aH + (long)x * p.aW + cc);
          if (p.mask) qm = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.mask) + (long)n * p.mN + (long)y * p.mH + (long)x * p.mW + cc);
          if (p.bnb_x) qx = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.bnb_x) + qoff);
        };
        request(0, ea, em, ex, doff, ok);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint4 na = make_uint4(0, 0, 0, 0), nm = na, nx = na;
          long noff = 0;
          bool nok = false;
          if (i + 1 < 4) request(i + 1, na, nm, nx, noff, nok);
          const int prow = i * 8 + pr;
          const float4 v0 = *reinterpret_cast<const float4*>(stg + prow * OROW + u * 32);
          const float4 v1 = *reinterpret_cast<const float4*>(stg + prow * OROW + u * 32 + 16);
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if (p.addend) {
            float av[8];
            Unit<bf16>::unpack(ea, av);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += av[j];
          }
          if (p.mask) {
            float mv[8];
            Unit<bf16>::unpack(em, mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : 0.f;
          if (p.bnb_x) {
            float cv[8];
            Unit<bf16>::unpack(ex, cv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * (cv[j] - mu[j]) * is[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
          }
          if (ok) {
            if (p.out_f32) {
              float* dst = reinterpret_cast<float*>(p.dst) + doff;
              *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
              *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
              *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.dst) + doff) = Unit<bf16>::pack(v);
            }
          }
          ea = na; em = nm; ex = nx; doff = noff; ok = nok;
        }
      } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int prow = i * 8 + pr;
        const int m = pix0 + wave * WPIX + b * 32 + prow;
        if (m >= p.M || !cok) continue;
        int qd = fs_div(m, dW); int x = m - qd * p.Wd; int n = fs_div(qd, dH); int y = qd - n * p.Hd; n += n0;
        const long doff = (long)n * p.dN + (long)y * p.dH + (long)x * p.dW + co;
        const float4 v0 = *reinterpret_cast<const float4*>(stg + prow * OROW + u * 32);
        const float4 v1 = *reinterpret_cast<const float4*>(stg + prow * OROW + u * 32 + 16);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bias[j];
        if (p.addend) {
          float av[8];
          loadv<bf16>(reinterpret_cast<const bf16*>(p.addend) + (long)n * p.aN + (long)y * p.aH + (long)x * p.aW + co, av);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += av[j];
        }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if constexpr (!FWD) {
          if (p.mask) {
            float mv[8];
            loadv<bf16>(reinterpret_cast<const bf16*>(p.mask) + (long)n * p.mN + (long)y * p.mH + (long)x * p.mW + co, mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
          }
        }
        bool bnb = false;
        if constexpr (!FWD) bnb = p.bnb_x != nullptr;
        if (bnb) {
          if constexpr (!FWD) {
            float cv[8];
            loadv<bf16>(reinterpret_cast<const bf16*>(p.bnb_x) + doff, cv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * (cv[j] - mu[j]) * is[j]; }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        }
        if (p.out_f32) {
          float* dst = reinterpret_cast<float*>(p.dst) + doff;
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          storev<bf16>(reinterpret_cast<bf16*>(p.dst) + doff, v);
        }
      }
      }
    }
    if (p.stats) {
      // lanes u, u + 8, ... hold the same eight channels: lane ^ 8 by DPP (row_ror:8), the four 16-lane rows by two
      // permutes; lanes 0..7 then own the wave's sums (plain read-modify-write: LDS float atomics run lane by lane)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s1[j] += dpp_mov<0x128>(s1[j]); s2[j] += dpp_mov<0x128>(s2[j]);
        s1[j] += __shfl_xor(s1[j], 16); s2[j] += __shfl_xor(s2[j], 16);
        s1[j] += __shfl_xor(s1[j], 32); s2[j] += __shfl_xor(s2[j], 32);
      }
      if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ssum[(h * HC + u * 8 + j) * 2] += s1[j];
          ssum[(h * HC + u * 8 + j) * 2 + 1] += s2[j];
        }
      }
    }
  }
  }   // pixel tiles of the block

  if (p.stats) {
    // ONE f64 atomic per channel and block into one of FS_STAT_SLOTS address slots (as conv1x1.hip)
    const int pix0 = spx * pt * PIX;
    __syncthreads();
    if (t < CO) {
      const int c = co0 + t;
      if (c < p.Co) {
        const long sg = p.grp_imgs > 0 ? (long)blockIdx.z : (p.stat_group_rows > 0 ? pix0 / p.stat_group_rows : 0);
        double* sl = p.stats + (sg * FS_STAT_SLOTS + spx % FS_STAT_SLOTS) * 2 * p.Co;
        const float* sw = reinterpret_cast<const float*>(lds + G::WORK);
        atomicAdd(sl + c, (double)((sw[2 * t] + sw[CO * 2 + 2 * t]) + (sw[CO * 4 + 2 * t] + sw[CO * 6 + 2 * t])));
        atomicAdd(sl + p.Co + c, (double)((sw[2 * t + 1] + sw[CO * 2 + 2 * t + 1]) + (sw[CO * 4 + 2 * t + 1] + sw[CO * 6 + 2 * t + 1])));
      }
    }
  }
}

template <int CO, int PIX, int NST, int EPI>
int launch_gemm_epi(const FsConvArgs& a, hipStream_t st) {
  const int npix = (a.M + PIX - 1) / PIX, nco = (a.Co_p + CO - 1) / CO;
  const int z = a.grp_imgs > 0 ? a.N / a.grp_imgs : 1;
  // pixel tiles per block: only where statistics are summed, while >= 1024 blocks (one round of four per CU) remain,
  // whole statistics groups
  int pt = 1;
  if (a.stats) {
    pt = (int)std::min<long>(8, std::max<long>(1, (long)npix * nco * z / 1024));
    while (pt > 1 && a.stat_group_rows > 0 && a.stat_group_rows % (pt * PIX) != 0) --pt;
  }
  const int nsp = (npix + pt - 1) / pt;
  const int blocks = 8 * ((nsp + 7) / 8) * nco;
  hipLaunchKernelGGL((conv1x1_gemm_kernel<CO, PIX, NST, EPI>), dim3(blocks, 1, z), dim3(256), 0, st, a, fs_make_div(a.Wd),
                     fs_make_div(a.Hd), nco, pt);
  return fs_launch_status();
}

// epilogue specialisation: forward (0), data gradient (1), anything else (2)
template <int CO, int PIX, int NST>
int launch_gemm(const FsConvArgs& a, hipStream_t st) {
  const bool dgr = (a.mask || a.bnb_x) && !a.bias && !a.relu, fwd = !a.mask && !a.bnb_x;
  if (dgr) return launch_gemm_epi<CO, PIX, NST, 1>(a, st);
  if (fwd) return launch_gemm_epi<CO, PIX, NST, 0>(a, st);
  return launch_gemm_epi<CO, PIX, NST, 2>(a, st);
}

// Measured per ResNet-50 shape at 320x1024 (tools/probes/conv1x1_shapes.py): 128-pixel tiles with two stages (35 KB of
// LDS, four blocks per CU whose load / multiply / store phases interleave) beat 256-pixel tiles and deeper rings on every
// shape but K = 2048, which wants three stages (the 256-pixel and four-stage instantiations were removed with their probes).
template <int CO>
int launch_co(const FsConvArgs& a, hipStream_t st) {
  return a.Cs >= 2048 ? launch_gemm<CO, 128, 3>(a, st) : launch_gemm<CO, 128, 2>(a, st);
}

}  // namespace

// FS_EINVAL = "not a case for this kernel" (fs_conv1x1 then runs the row-streaming kernel)
int fs_conv1x1_gemm(const FsConvArgs& a, hipStream_t st) {
  if (a.Cs % 32 != 0 || a.Co % 8 != 0 || a.Co_p % 16 != 0 || a.M < 128) return FS_EINVAL;
  if (a.bnb_scale || a.pro_mode != 0) return FS_EINVAL;
  const long wrow = a.wgt_row_bytes ? a.wgt_row_bytes : (long)a.nchunks * a.kg * 16;
  if (wrow % 16 != 0 || wrow < (long)a.Cs * 2) return FS_EINVAL;
  if (a.sN % 8 != 0 || a.sH % 8 != 0 || a.sW % 8 != 0 || a.dN % 8 != 0 || a.dH % 8 != 0 || a.dW % 8 != 0) return FS_EINVAL;
  if (a.addend && (a.aN % 8 != 0 || a.aH % 8 != 0 || a.aW % 8 != 0)) return FS_EINVAL;
  if (a.mask && (a.mN % 8 != 0 || a.mH % 8 != 0 || a.mW % 8 != 0)) return FS_EINVAL;
  if (((uintptr_t)a.src | (uintptr_t)a.wgt | (uintptr_t)a.dst | (uintptr_t)a.addend | (uintptr_t)a.mask | (uintptr_t)a.bnb_x) % 16 != 0)
    return FS_EINVAL;
  const bool co64 = a.Co_p % 128 != 0 && a.Co_p <= 64;
  // statistics groups must be whole pixel tiles
  if (a.stats && a.stat_group_rows > 0 && a.stat_group_rows % 128 != 0) return FS_EINVAL;
  return co64 ? launch_co<64>(a, st) : launch_co<128>(a, st);
}
