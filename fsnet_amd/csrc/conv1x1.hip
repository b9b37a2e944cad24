// 1x1 convolution (forward and stride-1 data gradient) as a row-streaming GEMM on CDNA4 MFMA.
//
// Replaces the Bottleneck's 1x1 convolutions and the downsample projections
// (vision_base/networks/models/backbone/resnet.py:52-89, 119) for which the implicit-GEMM kernel (conv_igemm.hip)
// pays a K-unit table, a gather and a double-buffered LDS pixel tile that a 1x1 kernel does not need: at ResNet-50 /
// 320x1024 its 160 launches per step ran at 109 TFLOP/s and were half of the step (profiles/r02c_*).  These GEMMs are
// HBM-bound — K = Ci is 64..2048 while M = N*H*W is up to 164 k rows: 64 -> 256 at 80x256 moves 105 MB for 5.4 GFLOP —
// so the kernel is built to stream: the weights of the block's channel tile sit in LDS (XOR-swizzled 16-byte units,
// conflict-free for ds_read_b128's lane groups), the pixel rows never touch LDS — NHWC rows are K-contiguous, so a
// lane's MFMA operand (8 consecutive channels of one pixel) is ONE 16-byte global load — and all pixel loads of a
// K chunk are in flight while the weights are staged.  MFMA roles as in conv_igemm (A = weights, B = pixels), so the
// epilogue (bias, addend, ReLU, ReLU-backward mask, BatchNorm statistics / backward sums, fp32 output, statistics
// groups) is the same code.
#include <cstdlib>
#include "common.h"
#include "fsnet_hip_internal.h"

namespace {

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

// KC: channels of the K walk staged per chunk (64 / 128 / 256); CO: channel tile; the four waves split the 128-pixel
// tile (32 pixels = two MFMA column tiles each) and every wave multiplies the whole channel tile.
template <typename T, int CO, int KC>
__global__ __launch_bounds__(256) void conv1x1_kernel(const FsConvArgs p, const FsDiv dW, const FsDiv dH) {
  constexpr int PIX = 128, WP = 4, WC = 1;
  constexpr int WPIX = PIX / WP, WCO = CO / WC;
  constexpr int TP = WPIX / 16, TC = WCO / 16;
  constexpr int UR = KC / 8;                   // 16-byte units per weight row per chunk
  constexpr int KS = KC / 32;                  // MFMA K steps per chunk
  constexpr int LCU = (CO * UR + 255) / 256;
  static_assert(sizeof(T) == 2, "bf16 only");
  __shared__ uint4 lds_c[CO * UR < 128 ? 128 : CO * UR];
  // output rows of a wave (32 pixels x CO channels, bf16) pass through LDS so that a store instruction writes whole
  // contiguous pixel rows (16 bytes per lane) instead of 8-byte pieces of 16 different rows; row stride padded by 16
  // bytes (16-byte aligned reads; the 16 pixel lanes of an 8-byte write are then at most 2-way conflicted)
  constexpr int OROW = CO * 2 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds_o[4][WPIX * OROW];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wp = wave % WP, wc = wave / WP;
  const int li = lane & 15, lg = lane >> 4;
  const int n0 = p.grp_imgs > 0 ? (int)blockIdx.z * p.grp_imgs : 0;

  // XCD-aware tile mapping (see conv_igemm.hip): pixel tiles of one channel tile share an XCD's L2 copy of its weights
  int px, cy;
  {
    const int npix = (p.M + PIX - 1) / PIX, nco = p.Co_p / CO;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    if (nco % 8 == 0) { const int g = nco >> 3; cy = xcd + 8 * (slot % g); px = slot / g; }
    else if (8 % nco == 0) { const int g = 8 / nco; cy = xcd % nco; px = slot * g + xcd / nco; }
    else { cy = id % nco; px = id / nco; }
    if (px >= npix) return;
  }
  const int pix0 = px * PIX, co0 = cy * CO;
  const int OOB = 0x7fffffff;
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (int)p.wgt_bytes, 0x00020000);

  // byte offset of this lane's pixel rows (hb_mul = stride of a strided 1x1: the downsample projection)
  int pvoff[TP];
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    const int m = pix0 + wp * WPIX + b * 16 + li;
    if (m < p.M) {
      int q = fs_div(m, dW); int x = m - q * p.Wd; int n = fs_div(q, dH); int y = q - n * p.Hd; n += n0;
      pvoff[b] = (int)((n * p.sN + (long)(y * p.hb_mul) * p.sH + (long)(x * p.hb_mul) * p.sW) * 2) + lg * 16;
    } else {
      pvoff[b] = OOB;
    }
  }
  const int wrow_bytes = p.wgt_row_bytes ? (int)p.wgt_row_bytes : p.nchunks * p.kg * 16;
  const int kbytes = p.Cs * 2;                 // real K extent of a row (channels beyond it are zero in both operands)

  f32x4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkc = (p.Cs + KC - 1) / KC;
  for (int kc = 0; kc < nkc; ++kc) {
    const int kb = kc * KC * 2;                // byte offset of the chunk inside a row
    // pixel operands of the whole chunk: requested first, they land while the weights are staged
    uint4 fb[KS][TP];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int b = 0; b < TP; ++b) {
        const int ko = kb + kk * 64;
        fb[kk][b] = buf_load16(rs_src, (pvoff[b] == OOB || ko + lg * 16 >= kbytes) ? OOB : pvoff[b] + ko);
      }
    uint4 rc[LCU];
#pragma unroll
    for (int i = 0; i < LCU; ++i) {
      const int idx = t + i * 256;
      const int row = idx / UR, u = idx - row * UR;
      rc[i] = buf_load16(rs_wgt, (row < CO && kb + u * 16 < kbytes) ? (co0 + row) * wrow_bytes + kb + u * 16 : OOB);
    }
    if (kc > 0) __syncthreads();               // previous chunk's weights fully multiplied
#pragma unroll
    for (int i = 0; i < LCU; ++i) {
      const int idx = t + i * 256;
      const int row = idx / UR, u = idx - row * UR;
      if (row < CO) lds_c[row * UR + (u ^ (row & 15 & (UR - 1)))] = rc[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      uint4 fa[TC];
#pragma unroll
      for (int a = 0; a < TC; ++a) {
        const int row = wc * WCO + a * 16 + li;
        fa[a] = lds_c[row * UR + ((kk * 4 + lg) ^ (row & 15 & (UR - 1)))];
      }
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[a]),
                                                              __builtin_bit_cast(bf16x8, fb[kk][b]), acc[a][b], 0, 0, 0);
    }
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
      else {
        uint2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(&lds_o[wave][(b * 16 + li) * OROW + (a * 16 + lg * 4) * 2]) = pk;
      }
    }
  }
  if (!p.out_f32) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int UPP = CO / 8;                 // 16-byte units per pixel row of the tile
    constexpr int PPI = 64 / UPP;               // pixel rows per store instruction
    const int u = lane % UPP, pr = lane / UPP;
#pragma unroll
    for (int i = 0; i < WPIX / PPI; ++i) {
      const int prow = i * PPI + pr;
      const int m = pix0 + wp * WPIX + prow;
      const int co = co0 + u * 8;
      if (m < p.M && co < p.Co) {
        int q = fs_div(m, dW); int x = m - q * p.Wd; int n = fs_div(q, dH); int y = q - n * p.Hd; n += n0;
        const long doff = (long)n * p.dN + (long)y * p.dH + (long)x * p.dW;
        const uint4 val = *reinterpret_cast<const uint4*>(&lds_o[wave][prow * OROW + u * 16]);
        T* dst = reinterpret_cast<T*>(p.dst) + doff + co;
        if (co + 8 <= p.Co) *reinterpret_cast<uint4*>(dst) = val;
        else *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);       // Co % 8 == 4: the last half unit
      }
    }
  }

  if (p.stats) {
    // per-channel (sum, sumsq): 16 pixel lanes by shuffle, the WP pixel-waves through LDS, then ONE f64
    // atomic per channel per block into one of FS_STAT_SLOTS address slots (same-address atomics cost
    // ~12 ns each on MI355X; slots + block reduce keep the chain per address short).
    __syncthreads();                     // all waves are done reading the operand tiles
    float* red = reinterpret_cast<float*>(&lds_c[0]);   // [WP][CO][2]
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


template <typename T, int CO, int KC>
int launch_1x1(const FsConvArgs& a, hipStream_t st) {
  constexpr int PIX = 128;
  const int npix = (a.M + PIX - 1) / PIX, nco = a.Co_p / CO;
  int blocks = npix * nco;
  if (nco % 8 != 0 && 8 % nco == 0) { const int g = 8 / nco; blocks = 8 * ((npix + g - 1) / g); }
  hipLaunchKernelGGL((conv1x1_kernel<T, CO, KC>), dim3(blocks, 1, a.grp_imgs > 0 ? a.N / a.grp_imgs : 1), dim3(256), 0, st,
                     a, fs_make_div(a.Wd), fs_make_div(a.Hd));
  return fs_launch_status();
}

template <typename T, int CO>
int launch_co(const FsConvArgs& a, hipStream_t st) {
  if (a.Cs <= 64) return launch_1x1<T, CO, 64>(a, st);
  if (a.Cs <= 128) return launch_1x1<T, CO, 128>(a, st);
  return launch_1x1<T, CO, 256>(a, st);
}

}  // namespace

// FS_EINVAL = "not a case for this kernel" (the caller falls back to fs_conv_igemm): fp32, a padded / dilated walk, a
// parity-class data gradient, a K extent that is not whole 64-byte steps.
extern "C" int fs_conv1x1(const FsConvArgs* args, int dtype, void* stream) {
  if (!args || !args->src || !args->wgt || !args->dst) return FS_EINVAL;
  if (dtype != FS_DTYPE_BF16 || args->dshift != 0 || args->ncls > 1 || args->hb_add != 0 || args->hb_mul < 1) return FS_EINVAL;
  if (args->Cs <= 0 || (args->Cs * 2) % 64 != 0 || args->Co % 4 != 0 || args->Co_p % 16 != 0 || args->M <= 0) return FS_EINVAL;
  if (args->src_bytes <= 0 || args->src_bytes > 0x7fffffffLL || args->wgt_bytes <= 0 || args->wgt_bytes > 0x7fffffffLL)
    return FS_EINVAL;
  if (args->stats && args->stat_group_rows > 0 && args->stat_group_rows % 128 != 0) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    // LDS-DMA GEMM kernel first; what it declines (< 128 rows, channel counts that are not multiples of 8) streams rows
    const int r = fs_conv1x1_gemm(*args, st);
    if (r != FS_EINVAL) return r;
  }
  const int cop = args->Co_p;
  if (cop % 64 == 0) return launch_co<bf16, 64>(*args, st);
  if (cop % 32 == 0) return launch_co<bf16, 32>(*args, st);
  return launch_co<bf16, 16>(*args, st);
}
