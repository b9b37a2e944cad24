// Convolution weight-gradient as an MFMA GEMM with the pixel axis as K.
//
// Replaces convolution_backward(weight) for every conv on the monodepth hot path
// (same call sites as conv_igemm.hip).  dW[co][ci][r][s] = sum_pixels dY[pix][co] * X[tap(pix,r,s)][ci].
// Both operands are channel-contiguous in HBM while the reduction runs over pixels, so the
// tiles are staged in LDS in their natural [pixel][channel] layout (coalesced 16-byte loads)
// and the MFMA fragments are fetched transposed: ds_read_b64_tr_b16 for bf16 (gfx950 LDS
// transpose-read), plain ds_read_b32 for f32 (one scalar per lane per MFMA).  The pixel range
// is split over blockIdx.z; an unsplit tile accumulates straight into the OIHW gradient the optimizer
// consumes, split tiles write dense partial slabs that a second kernel reduces (no atomics: same-address
// atomics serialise at ~12 ns each on MI355X and dominated the first version of this kernel).
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T, int COT, int CLT, int WR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const FsWgradArgs p) {
  using TR = ElemTraits<T>;
  constexpr int EG = TR::EG;
  constexpr int WCn = 4 / WR;                 // waves along columns
  constexpr int WROW = COT / WR, WCOL = CLT / WCn;
  constexpr int TA = WROW / 16, TB = WCOL / 16;
  constexpr int GA = COT / EG, GB = CLT / EG;  // 16-byte groups per pixel row
  constexpr int LA = (32 * GA + 255) / 256, LB = (32 * GB + 255) / 256;
  constexpr int PADE = 16 / sizeof(T);         // row padding (elements) to spread LDS banks
  constexpr int SA = COT + PADE, SB = CLT + PADE;

  __shared__ __attribute__((aligned(16))) T lds_a[2][32 * SA];
  __shared__ __attribute__((aligned(16))) T lds_b[2][32 * SB];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave % WR, wcn = wave / WR;
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const T* __restrict__ xs = reinterpret_cast<const T*>(p.x);

  const int col0 = blockIdx.x * CLT;   // first gemm column (r,s,ci) of the tile
  const int co0 = blockIdx.y * COT;
  const long m_begin = (long)blockIdx.z * p.pix_per_split;
  long m_end = m_begin + p.pix_per_split; if (m_end > p.M) m_end = p.M;
  const int nch = (int)((m_end - m_begin + 31) / 32);

  // ---- B loader state: fixed column group, pixel advancing by 32 per chunk ----
  int bc[LB], br[LB], bs[LB]; bool bval[LB];
  int bx[LB], by[LB], bn[LB]; long bm[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    int idx = t + i * 256;
    int pix = idx / GB, g = idx % GB;
    int kg = col0 / EG + g;
    bval[i] = (pix < 32) && (kg < p.ncolgroups);
    int e = bval[i] ? p.ktab[kg] : -1;
    bval[i] = bval[i] && (e >= 0);
    bc[i] = e & 0xffff; br[i] = (e >> 16) & 0xff; bs[i] = (e >> 24) & 0x7f;
    long m = m_begin + pix;
    bm[i] = m;
    bx[i] = (int)(m % p.Wd); long q = m / p.Wd; by[i] = (int)(q % p.Hd); bn[i] = (int)(q / p.Hd);
  }

  uint4 ra[LA], rb[LB];
  auto load_regs = [&](int kc) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int idx = t + i * 256;
      int pix = idx / GA, g = idx % GA;
      long m = m_begin + (long)kc * 32 + pix;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (pix < 32 && m < m_end && (co0 + g * EG) < p.Cd)
        v = *reinterpret_cast<const uint4*>(dy + m * p.Cd + co0 + g * EG);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (bval[i] && bm[i] < m_end) {
        int h = by[i] * p.stride - p.pad + br[i];
        int w = bx[i] * p.stride - p.pad + bs[i];
        if ((unsigned)h < (unsigned)p.Hs && (unsigned)w < (unsigned)p.Ws)
          v = *reinterpret_cast<const uint4*>(xs + (long)bn[i] * p.sN + (long)h * p.sH + (long)w * p.sW + bc[i]);
      }
      rb[i] = v;
      // advance this loader's pixel by one chunk
      bm[i] += 32; bx[i] += 32;
      while (bx[i] >= p.Wd) { bx[i] -= p.Wd; by[i] += 1; }
      while (by[i] >= p.Hd) { by[i] -= p.Hd; bn[i] += 1; }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int idx = t + i * 256;
      int pix = idx / GA, g = idx % GA;
      if (pix < 32) *reinterpret_cast<uint4*>(&lds_a[buf][pix * SA + g * EG]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      int idx = t + i * 256;
      int pix = idx / GB, g = idx % GB;
      if (pix < 32) *reinterpret_cast<uint4*>(&lds_b[buf][pix * SB + g * EG]) = rb[i];
    }
  };

  f32x4 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nch > 0) {
    load_regs(0);
    store_lds(0);
  }
  __syncthreads();
  int buf = 0;
  for (int kc = 0; kc < nch; ++kc) {
    if (kc + 1 < nch) load_regs(kc + 1);
    if constexpr (sizeof(T) == 2) {
      // transposed fragments: lane (li, lg) supplies the 8-byte row segment
      // [pixel lg*8 + (li>>2) (+4)][channel tile0 + (li&3)*4 ..+3] and receives channel tile0+li
      // for pixels lg*8 + {0..3} (+4).
      bf16x8 fa[TA], fb[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        const T* base = &lds_a[buf][(lg * 8 + (li >> 2)) * SA + wr * WROW + a * 16 + (li & 3) * 4];
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * SA));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        fa[a] = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const T* base = &lds_b[buf][(lg * 8 + (li >> 2)) * SB + wcn * WCOL + b * 16 + (li & 3) * 4];
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * SB));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        fb[b] = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
      }
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float fa[TA], fb[TB];
#pragma unroll
        for (int a = 0; a < TA; ++a) fa[a] = lds_a[buf][(kk * 4 + lg) * SA + wr * WROW + a * 16 + li];
#pragma unroll
        for (int b = 0; b < TB; ++b) fb[b] = lds_b[buf][(kk * 4 + lg) * SB + wcn * WCOL + b * 16 + li];
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
          for (int b = 0; b < TB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
    }
    if (kc + 1 < nch) store_lds(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue: D rows = co (lg*4+j), cols = gemm column (li) ----
  if (p.nsplit > 1) {
    // split-K partial: dense slab [split][Cd_t][ncols_t] in the workspace (reduced by wgrad_reduce_kernel)
    float* ws = p.workspace + (long)blockIdx.z * p.ws_rows * p.ws_cols;
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      int col = col0 + wcn * WCOL + b * 16 + li;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int co = co0 + wr * WROW + a * 16 + lg * 4 + j;
          ws[(long)co * p.ws_cols + col] = acc[a][b][j];
        }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < TB; ++b) {
    int col = col0 + wcn * WCOL + b * 16 + li;
    int kg = col / EG;
    if (kg >= p.ncolgroups) continue;
    int e = p.ktab[kg];
    if (e < 0) continue;
    int ci = (e & 0xffff) + (col % EG);
    int r = (e >> 16) & 0xff, s = (e >> 24) & 0x7f;
    if (ci >= p.Ci) continue;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int co = co0 + wr * WROW + a * 16 + lg * 4 + j;
        if (co < p.Co) p.dw[(((long)co * p.Ci + ci) * p.R + r) * p.S + s] += acc[a][b][j];  // sole owner: plain RMW
      }
  }
}

// dw[co][ci][r][s] += sum_z workspace[z][co][col].  64 consecutive columns x 4 split-lanes per block: each
// lane strides the splits by 4 with independent (unrolled) loads, then the 4 lanes combine through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const FsWgradArgs p, int eg) {
  __shared__ float red[4][64];
  const int ncols = p.ncolgroups * eg;
  const int cblocks = (ncols + 63) / 64;
  const int co = blockIdx.x / cblocks, col = (blockIdx.x % cblocks) * 64 + (threadIdx.x & 63);
  const int zl = threadIdx.x >> 6;
  float acc = 0.f;
  if (col < ncols) {
    const float* ws = p.workspace + (long)co * p.ws_cols + col;
    const long slab = (long)p.ws_rows * p.ws_cols;
    int z = zl;
    for (; z + 12 < p.nsplit; z += 16) {
      float a0 = ws[(long)z * slab], a1 = ws[(long)(z + 4) * slab], a2 = ws[(long)(z + 8) * slab], a3 = ws[(long)(z + 12) * slab];
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; z < p.nsplit; z += 4) acc += ws[(long)z * slab];
  }
  red[zl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (zl == 0 && col < ncols) {
    float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    int e = p.ktab[col / eg];
    if (e >= 0) {
      int ci = (e & 0xffff) + (col % eg);
      if (ci < p.Ci) {
        int r = (e >> 16) & 0xff, s = (e >> 24) & 0x7f;
        p.dw[(((long)co * p.Ci + ci) * p.R + r) * p.S + s] += v;
      }
    }
  }
}

template <typename T, int COT, int CLT, int WR>
int launch_tile(const FsWgradArgs& a, hipStream_t st) {
  constexpr int EG = ElemTraits<T>::EG;
  FsWgradArgs b = a;
  const int ncols = a.ncolgroups * EG;
  const int ct = (ncols + CLT - 1) / CLT, rt = (a.Cd + COT - 1) / COT;
  const long tiles = (long)ct * rt;
  const long chunks = (a.M + 31) / 32;
  // split the pixel (K) range until ~768 blocks are in flight, keeping >= 8 chunks per block and the
  // partial slabs inside the caller's workspace
  long splits = (640 + tiles - 1) / tiles;
  splits = std::min<long>(splits, std::max<long>(1, chunks / 8));
  b.ws_rows = rt * COT; b.ws_cols = ct * CLT;
  const long slab = (long)b.ws_rows * b.ws_cols;
  if (!a.workspace) splits = 1;
  else splits = std::min<long>(splits, std::max<long>(1, a.workspace_elems / slab));
  long cps = (chunks + splits - 1) / splits;
  b.pix_per_split = (int)(cps * 32);
  b.nsplit = (int)((chunks + cps - 1) / cps);
  dim3 grid(ct, rt, b.nsplit);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, COT, CLT, WR>), grid, dim3(256), 0, st, b);
  if (b.nsplit > 1) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(a.Co * ((ncols + 63) / 64))), dim3(256), 0, st, b, EG);
  }
  return fs_launch_status();
}

template <typename T>
int launch_wgrad(const FsWgradArgs& a, hipStream_t st) {
  constexpr bool kBf16 = sizeof(T) == 2;  // f32 tiles are capped by the 64 KB static LDS limit
  if (a.Cd % 64 == 0) return launch_tile<T, 64, 64, 2>(a, st);
  if (a.Cd % 32 == 0) return launch_tile<T, 32, 128, 1>(a, st);
  if (a.Cd % 16 == 0) {
    if constexpr (kBf16) return launch_tile<T, 16, 256, 1>(a, st);
    else return launch_tile<T, 16, 128, 1>(a, st);
  }
  return FS_EINVAL;
}

}  // namespace

extern "C" int fs_conv_wgrad(const FsWgradArgs* args, int dtype, void* stream) {
  if (!args || !args->dy || !args->x || !args->dw || !args->ktab) return FS_EINVAL;
  if (args->M <= 0 || args->ncolgroups <= 0) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FS_DTYPE_BF16) return launch_wgrad<bf16>(*args, st);
  if (dtype == FS_DTYPE_F32) return launch_wgrad<float>(*args, st);
  return FS_EINVAL;
}
