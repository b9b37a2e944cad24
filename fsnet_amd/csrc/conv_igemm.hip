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
// K is walked in 64- or 128-byte stages (kg = 4 / 8 sixteen-byte groups: 32 / 64 bf16 channels); both operand
// tiles are staged through double-buffered LDS with register prefetch of the next chunk.
// The same kernel computes dgrad: the gather source is dY, the packed weights are
// Wt[ci][r][s][co] and the tap walk runs with sgn=-1 (plus a parity test for stride 2).
#include "common.h"
#include "fsnet_hip_internal.h"

namespace {

template <int KG> __device__ __forceinline__ int lds_swz(int row) {
  // XOR swizzle of the 16-byte group index that makes the ds_read_b128 lane groups conflict-free
  // (64-byte rows: 2 bits from row bit 3; 128-byte rows: 3 bits from row bits 1..3)
  return KG == 4 ? (((row >> 3) & 1) << 1) : ((row >> 1) & 7);
}

template <typename T, int PIX, int CO, int WP, int KG>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const FsConvArgs p) {
  using TR = ElemTraits<T>;
  constexpr int EG = TR::EG;
  constexpr int WC = 4 / WP;
  constexpr int WPIX = PIX / WP, WCO = CO / WC;
  constexpr int TP = WPIX / 16, TC = WCO / 16;
  constexpr int LP = (PIX * KG + 255) / 256;
  constexpr int LC = (CO * KG + 255) / 256;
  constexpr int KGS = KG == 4 ? 2 : 3;   // log2(KG)
  static_assert(WPIX % 16 == 0 && WCO % 16 == 0, "tile");

  __shared__ uint4 lds_p[2][PIX * KG];
  __shared__ uint4 lds_c[2][CO * KG];
  __shared__ int s_ktab[2048];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wp = wave % WP, wc = wave / WP;
  const int li = lane & 15, lg = lane >> 4;
  const int nch = p.nchunks;
  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);

  for (int i = t; i < nch * KG; i += 256) s_ktab[i] = p.ktab[i];

  // ---- per-thread gather rows (fixed over the K walk) ----
  long pbase[LP]; int phb[LP], pwb[LP];
  const int pix0 = blockIdx.x * PIX;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    int idx = t + i * 256;
    int row = idx >> KGS;
    int m = pix0 + row;
    if (row < PIX && m < p.M) {
      int x = m % p.Wd; int q = m / p.Wd; int y = q % p.Hd; int n = q / p.Hd;
      pbase[i] = (long)n * p.sN;
      phb[i] = y * p.hb_mul + p.hb_add;
      pwb[i] = x * p.hb_mul + p.hb_add;
    } else {
      pbase[i] = 0; phb[i] = -(1 << 28); pwb[i] = -(1 << 28);
    }
  }
  const long wrow_stride = (long)nch * KG * EG;  // elements per packed weight row
  const int co0 = blockIdx.y * CO;
  const int dmask = (1 << p.dshift) - 1;

  uint4 rp[LP], rc[LC];
  auto load_regs = [&](int kc) {
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      int idx = t + i * 256;
      int q = idx & (KG - 1);
      int e = s_ktab[kc * KG + q];
      int c = e & 0xffff, r = (e >> 16) & 0xff, s = (e >> 24) & 0x7f;
      int h = phb[i] + p.sgn * r, w = pwb[i] + p.sgn * s;
      bool ok = (e >= 0) && (((h | w) & dmask) == 0);
      h >>= p.dshift; w >>= p.dshift;
      ok = ok && ((unsigned)h < (unsigned)p.Hs) && ((unsigned)w < (unsigned)p.Ws);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = *reinterpret_cast<const uint4*>(src + pbase[i] + (long)h * p.sH + (long)w * p.sW + c);
      rp[i] = v;
    }
#pragma unroll
    for (int i = 0; i < LC; ++i) {
      int idx = t + i * 256;
      int row = idx >> KGS, q = idx & (KG - 1);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < CO) v = *reinterpret_cast<const uint4*>(wgt + (long)(co0 + row) * wrow_stride + (long)(kc * KG + q) * EG);
      rc[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      int idx = t + i * 256;
      int row = idx >> KGS, q = idx & (KG - 1);
      if (row < PIX) lds_p[buf][row * KG + (q ^ lds_swz<KG>(row))] = rp[i];
    }
#pragma unroll
    for (int i = 0; i < LC; ++i) {
      int idx = t + i * 256;
      int row = idx >> KGS, q = idx & (KG - 1);
      if (row < CO) lds_c[buf][row * KG + (q ^ lds_swz<KG>(row))] = rc[i];
    }
  };

  f32x4 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();  // s_ktab visible
  load_regs(0);
  store_lds(0);
  __syncthreads();

  int buf = 0;
  for (int kc = 0; kc < nch; ++kc) {
    if (kc + 1 < nch) load_regs(kc + 1);
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
    if (kc + 1 < nch) store_lds(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: lane holds, for pixel (tile b, li), channels lg*4..lg*4+3 of channel tile a ----
  float s1[TC][4], s2[TC][4];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }

#pragma unroll
  for (int b = 0; b < TP; ++b) {
    int m = pix0 + wp * WPIX + b * 16 + li;
    bool mok = m < p.M;
    int x = 0, y = 0, n = 0;
    if (mok) { x = m % p.Wd; int q = m / p.Wd; y = q % p.Hd; n = q / p.Hd; }
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
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[a][j] += v[j]; s2[a][j] += v[j] * v[j]; }
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
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { u += __shfl_xor(u, o, 64); w += __shfl_xor(w, o, 64); }
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
        double* sl = p.stats + (long)(blockIdx.x % FS_STAT_SLOTS) * 2 * p.Co;
        atomicAdd(sl + co, (double)u);
        atomicAdd(sl + p.Co + co, (double)w);
      }
    }
  }
}

template <typename T, int PIX, int CO, int WP, int KG>
int launch_tile(const FsConvArgs& a, hipStream_t st) {
  dim3 grid((a.M + PIX - 1) / PIX, (a.Co_p + CO - 1) / CO);
  hipLaunchKernelGGL((conv_igemm_kernel<T, PIX, CO, WP, KG>), grid, dim3(256), 0, st, a);
  return fs_launch_status();
}

// tile choice: channel tile = largest of {128,64,32,16} dividing Co_p; shrink the pixel tile when the
// grid would not fill 256 CUs.  kg (16-byte K groups per stage, 4 or 8) is fixed by the caller's packing.
template <typename T>
int launch_conv(const FsConvArgs& a, hipStream_t st) {
  const int cop = a.Co_p;
  const bool k8 = a.kg == 8;
  if (cop % 128 == 0 && !k8) {
    long blocks = (long)((a.M + 127) / 128) * (cop / 128);
    if (blocks >= 512) return launch_tile<T, 128, 128, 2, 4>(a, st);
    return launch_tile<T, 64, 64, 2, 4>(a, st);
  }
  if (cop % 64 == 0) {
    long blocks = (long)((a.M + 127) / 128) * (cop / 64);
    if (blocks >= 512) return k8 ? launch_tile<T, 128, 64, 2, 8>(a, st) : launch_tile<T, 128, 64, 2, 4>(a, st);
    return k8 ? launch_tile<T, 64, 64, 2, 8>(a, st) : launch_tile<T, 64, 64, 2, 4>(a, st);
  }
  if (cop % 32 == 0) return k8 ? launch_tile<T, 128, 32, 4, 8>(a, st) : launch_tile<T, 128, 32, 4, 4>(a, st);
  if (cop % 16 == 0 && !k8) return launch_tile<T, 256, 16, 4, 4>(a, st);
  return FS_EINVAL;
}

}  // namespace

extern "C" int fs_conv_igemm(const FsConvArgs* args, int dtype, void* stream) {
  if (!args || !args->src || !args->wgt || !args->dst || !args->ktab) return FS_EINVAL;
  if (args->kg != 4 && args->kg != 8) return FS_EINVAL;
  if (args->nchunks <= 0 || args->nchunks * args->kg > 2048) return FS_EINVAL;
  if (args->Co % 4 != 0 || args->Co_p % 16 != 0) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FS_DTYPE_BF16) return launch_conv<bf16>(*args, st);
  if (dtype == FS_DTYPE_F32) return launch_conv<float>(*args, st);
  return FS_EINVAL;
}
