// 1x1 convolution (forward, strided forward, stride-1 data gradient) as an LDS-staged bf16 GEMM on CDNA4.
//
// The Bottleneck's 1x1 convolutions and downsample projections (vision_base/networks/models/backbone/resnet.py:52-89,
// 119) at ResNet-50 / 320x1024 are GEMMs with M = N*H*W = 2.5 k .. 328 k rows and K, Co in 64 .. 2048: 148 launches
// and a third of that step.  The row-streaming kernel (conv1x1.hip) reads its pixel operand in MFMA-fragment shape
// (16 rows x 64 bytes per load instruction: half-used cache lines, every pixel row re-read by each 64-channel tile) and
// multiplies 64 x 32 wave tiles whose weight fragments alone saturate the LDS; it ran at 50-200 TFLOP/s.  Here:
//   * block tile PIX x CO = 256 (128) pixels x 128 (64) channels, four waves splitting the pixels; a wave multiplies
//     64 x 128 with v_mfma_f32_32x32x16_bf16 (six 16-byte fragment reads per eight MFMAs);
//   * both operands go global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, whole 64-byte row pieces
//     per four lanes, out-of-range rows read as zero), two 32-channel K stages in flight, one barrier per stage; the
//     LDS image is lane-linear, so the 16-byte units of a row are XOR-permuted on the SOURCE side and un-permuted by
//     the fragment read ((row >> 2) & 3: the four 16-lane groups of a ds_read_b128 then hit 64 distinct banks);
//   * the epilogue transposes the accumulators through LDS (fp32, per wave) so that every global access of the
//     epilogue — addend, ReLU-backward mask, BatchNorm-backward operand, the store — is a 16-byte piece of a
//     contiguous pixel row (the fragment layout gives 8-byte pieces of 32 different rows);
//   * channel tiles of one pixel tile run back to back on one XCD, so the pixel rows come from HBM once.
// Epilogue semantics are conv1x1.hip's / conv_igemm.hip's (bias, addend, ReLU, mask, BatchNorm statistics or
// backward sums with statistics groups, fp32 output).
#include "common.h"
#include "fsnet_hip_internal.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// 16 bytes per lane global -> LDS without a register stop: LDS byte lds_addr + 16 * lane receives the 16 bytes at buffer
// offset voff (zero when out of range).  Inline assembly because hipcc orders every later ds_read behind an LDS-DMA it
// knows about with `s_waitcnt vmcnt(0)` — which would land the next stage before the current one is multiplied; the
// kernel counts these loads itself (vmcnt(0) in front of the stage barrier).
__device__ __forceinline__ void glds16(const i32x4 rsrc, int voff, unsigned lds_addr) {
  int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, long bytes) {
  const unsigned long long b = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

template <int CO, int PIX>
struct GemmCfg {
  static constexpr int WPIX = PIX / 4;                 // pixels of a wave
  static constexpr int TP = WPIX / 32, TC = CO / 32;   // 32x32 accumulator tiles of a wave: TC x TP
  static constexpr int STAGE = (PIX + CO) * 64;        // one K stage: [PIX pixel rows][CO weight rows] x 64 bytes
  static constexpr int OROW = CO * 4 + 16;             // fp32 staging row of the epilogue (padded: conflict-free)
  static constexpr int OSTG = 32 * OROW;               // 32 pixel rows per wave and pass
  static constexpr int LDS = 2 * STAGE > 4 * OSTG ? 2 * STAGE : 4 * OSTG;
};

template <int CO, int PIX>
__global__ __launch_bounds__(256, 2) void conv1x1_gemm_kernel(const FsConvArgs p, const FsDiv dW, const FsDiv dH,
                                                               const int nco) {
  using G = GemmCfg<CO, PIX>;
  constexpr int WPIX = G::WPIX, TP = G::TP, TC = G::TC, STAGE = G::STAGE, OROW = G::OROW, OSTG = G::OSTG;
  constexpr int NLP = WPIX / 16, NLW = CO / 64;        // load instructions per wave and stage: pixels, weights
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = p.grp_imgs > 0 ? (int)blockIdx.z * p.grp_imgs : 0;
  const int npix = (p.M + PIX - 1) / PIX;
  // XCD-aware mapping: block b runs on XCD b % 8; consecutive slots of an XCD walk the channel tiles of ONE pixel tile
  const int slot = (int)blockIdx.x >> 3;
  const int px = ((int)blockIdx.x & 7) + 8 * (slot / nco), cy = slot % nco;
  if (px >= npix) return;
  const int pix0 = px * PIX, co0 = cy * CO;
  const int OOB = 0x7fffffff;
  const i32x4 rs_src = make_rsrc(p.src, p.src_bytes), rs_wgt = make_rsrc(p.wgt, p.wgt_bytes);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;

  // ---- loader: lane l of a load instruction fills LDS bytes [16 l, 16 l + 16) of a 16-row x 64-byte piece: row l >> 2,
  // slot l & 3, which holds source unit (l & 3) ^ ((l >> 4) & 3) ----
  const int lrow = lane >> 2, sunit = (lane & 3) ^ ((lane >> 4) & 3);
  int poff[NLP], woff[NLW];
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
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

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
  // one barrier per stage: stage kt landed for every wave and stage kt-1 is fully multiplied (its buffer is free for
  // stage kt+1, which then flies during the multiplication of stage kt)
  const int nkt = p.Cs / 32;
  issue(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
    multiply(lds + (kt & 1) * STAGE);
  }
  __syncthreads();              // every wave is done with the operand stages: the staging rows below alias them

  // ---- epilogue.  Accumulator tile (a, b): lane holds pixel b*32 + (lane & 31), channels a*32 + 8 q + 4 (lane >> 5)
  // + 0..3 in registers 4 q .. 4 q + 3.  Through the wave's staging rows it becomes: lane = (pixel row pr = lane /
  // UPP + 4 i', channel unit u = lane % UPP) with 8 consecutive channels ----
  constexpr int UPP = CO / 8;                  // 8-channel units per pixel row
  constexpr int RPI = 64 / UPP;                // pixel rows per pass of the wave
  unsigned char* stg = lds + wave * OSTG;
  const int u = lane % UPP, pr = lane / UPP;
  const int co = co0 + u * 8;
  const bool cok = co < p.Co;
  const long sgoff = p.grp_imgs > 0 ? (long)blockIdx.z * p.Co
                                    : (p.stat_group_rows > 0 ? (long)(pix0 / p.stat_group_rows) * p.Co : 0);
  float bias[8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias[j] = 0.f; mu[j] = 0.f; is[j] = 0.f; }
  if (cok && p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
    bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
  }
  if (cok && p.bnb_x) {
    const float4 m0 = *reinterpret_cast<const float4*>(p.bnb_mean + sgoff + co), m1 = *reinterpret_cast<const float4*>(p.bnb_mean + sgoff + co + 4);
    const float4 i0 = *reinterpret_cast<const float4*>(p.bnb_invstd + sgoff + co), i1 = *reinterpret_cast<const float4*>(p.bnb_invstd + sgoff + co + 4);
    mu[0] = m0.x; mu[1] = m0.y; mu[2] = m0.z; mu[3] = m0.w; mu[4] = m1.x; mu[5] = m1.y; mu[6] = m1.z; mu[7] = m1.w;
    is[0] = i0.x; is[1] = i0.y; is[2] = i0.z; is[3] = i0.w; is[4] = i1.x; is[5] = i1.y; is[6] = i1.z; is[7] = i1.w;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

#pragma unroll
  for (int b = 0; b < TP; ++b) {
    if (b > 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        *reinterpret_cast<float4*>(stg + (lane & 31) * OROW + (a * 32 + 8 * q + 4 * (lane >> 5)) * 4) = v;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int prow = i * RPI + pr;
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
      if (p.mask) {
        float mv[8];
        loadv<bf16>(reinterpret_cast<const bf16*>(p.mask) + (long)n * p.mN + (long)y * p.mH + (long)x * p.mW + co, mv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
      }
      if (p.bnb_x) {
        float cv[8];
        loadv<bf16>(reinterpret_cast<const bf16*>(p.bnb_x) + doff, cv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * (cv[j] - mu[j]) * is[j]; }
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

  if (p.stats) {
    // lanes u, u + UPP, ... hold the same channels: fold them, then the four waves through LDS, then ONE f64 atomic per
    // channel and block into one of FS_STAT_SLOTS address slots (as conv1x1.hip)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int d = UPP; d < 64; d <<= 1) { s1[j] += __shfl_xor(s1[j], d); s2[j] += __shfl_xor(s2[j], d); }
    }
    __syncthreads();                     // all waves are done with their staging rows
    float* red = reinterpret_cast<float*>(lds);        // [4][CO][2]
    if (lane < UPP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * CO + u * 8 + j) * 2] = s1[j];
        red[(wave * CO + u * 8 + j) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (t < CO) {
      float a = 0.f, w = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { a += red[(k * CO + t) * 2]; w += red[(k * CO + t) * 2 + 1]; }
      const int c = co0 + t;
      if (c < p.Co) {
        const long sg = p.grp_imgs > 0 ? (long)blockIdx.z : (p.stat_group_rows > 0 ? pix0 / p.stat_group_rows : 0);
        double* sl = p.stats + (sg * FS_STAT_SLOTS + px % FS_STAT_SLOTS) * 2 * p.Co;
        atomicAdd(sl + c, (double)a);
        atomicAdd(sl + p.Co + c, (double)w);
      }
    }
  }
}

template <int CO, int PIX>
int launch_gemm(const FsConvArgs& a, hipStream_t st) {
  const int npix = (a.M + PIX - 1) / PIX, nco = (a.Co_p + CO - 1) / CO;
  const int blocks = 8 * ((npix + 7) / 8) * nco;
  hipLaunchKernelGGL((conv1x1_gemm_kernel<CO, PIX>), dim3(blocks, 1, a.grp_imgs > 0 ? a.N / a.grp_imgs : 1), dim3(256), 0,
                     st, a, fs_make_div(a.Wd), fs_make_div(a.Hd), nco);
  return fs_launch_status();
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
  // 256-pixel tiles while they still give every CU two blocks; statistics groups must be whole tiles
  const int co_tiles = co64 ? 1 : (a.Co_p + 127) / 128;
  const long z = a.grp_imgs > 0 ? a.N / a.grp_imgs : 1;
  bool big = (long)((a.M + 255) / 256) * co_tiles * z >= 512;
  if (a.stats && a.stat_group_rows > 0) {
    if (a.stat_group_rows % 128 != 0) return FS_EINVAL;
    if (a.stat_group_rows % 256 != 0) big = false;
  }
  if (co64) return big ? launch_gemm<64, 256>(a, st) : launch_gemm<64, 128>(a, st);
  return big ? launch_gemm<128, 256>(a, st) : launch_gemm<128, 128>(a, st);
}
