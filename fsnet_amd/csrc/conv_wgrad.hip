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
#include "lds_dma.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

// fs_conv_wgrad_plan: the launch functions below report (kernel, blocks, threads, resident blocks) instead of launching
thread_local int32_t* g_plan = nullptr;
inline bool wg_plan(int kind, long blocks, int threads, int resident) {
  if (!g_plan) return false;
  g_plan[0] = kind; g_plan[1] = (int32_t)blocks; g_plan[2] = threads; g_plan[3] = resident;
  return true;
}


typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WgDiv { FsDiv dW, dH; };

// Two problems per launch (fsnet_hip_internal.h, FsDual) in every kernel of this file: the pixel splits of both stack
// along blockIdx.z (the persistent stem kernel: along blockIdx.x), [0, nb0) belong to the first argument set; each
// problem has its own slab region of the workspace and its own dW.
template <typename T, int COT, int CLT, int WR, int CH>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const FsDual<FsWgradArgs, WgDiv> d) {
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsWgradArgs& p = d.a[prob];
  const FsDiv dW = d.g[prob].dW, dH = d.g[prob].dH;
  const int zb = (int)blockIdx.z - (prob ? d.nb0 : 0);
  using TR = ElemTraits<T>;
  constexpr int EG = TR::EG;
  constexpr int WCn = 4 / WR;                 // waves along columns
  constexpr int WROW = COT / WR, WCOL = CLT / WCn;
  constexpr int TA = WROW / 16, TB = WCOL / 16;
  constexpr int GA = COT / EG, GB = CLT / EG;  // 16-byte groups per pixel row
  constexpr int LA = (CH * GA + 255) / 256, LB = (CH * GB + 255) / 256;   // CH pixels per pipeline stage
  constexpr int PADE = 16 / sizeof(T);         // row padding (elements) to spread LDS banks
  constexpr int SA = COT + PADE, SB = CLT + PADE;

  __shared__ __attribute__((aligned(16))) T lds_a[2][CH * SA];
  __shared__ __attribute__((aligned(16))) T lds_b[2][CH * SB];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave % WR, wcn = wave / WR;
  const int li = lane & 15, lg = lane >> 4;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const T* __restrict__ xs = reinterpret_cast<const T*>(p.x);

  const int col0 = blockIdx.x * CLT;   // first gemm column (r,s,ci) of the tile
  const int co0 = blockIdx.y * COT;
  const long m_begin = (long)zb * p.pix_per_split;
  long m_end = m_begin + p.pix_per_split; if (m_end > p.M) m_end = p.M;
  const int nch = (int)((m_end - m_begin + CH - 1) / CH);

  // ---- B loader state: fixed column group, pixel advancing by 32 per chunk ----
  int bc[LB], br[LB], bs[LB]; bool bval[LB];
  int bx[LB], by[LB], bn[LB]; long bm[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    int idx = t + i * 256;
    int pix = idx / GB, g = idx % GB;
    int kg = col0 / EG + g;
    bval[i] = (pix < CH) && (kg < p.ncolgroups);
    int e = bval[i] ? p.ktab[kg] : -1;
    bval[i] = bval[i] && (e >= 0);
    bc[i] = e & 0xffff; br[i] = (e >> 16) & 0xff; bs[i] = (e >> 24) & 0x7f;
    long m = m_begin + pix;
    bm[i] = m;
    const int q = fs_div((int)m, dW); bx[i] = (int)m - q * p.Wd; bn[i] = fs_div(q, dH); by[i] = q - bn[i] * p.Hd;
  }

  uint4 ra[LA], rb[LB];
  auto load_regs = [&](int kc) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int idx = t + i * 256;
      int pix = idx / GA, g = idx % GA;
      long m = m_begin + (long)kc * CH + pix;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (pix < CH && m < m_end && (co0 + g * EG) < p.Cd)
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
      bm[i] += CH; bx[i] += CH;
      while (bx[i] >= p.Wd) { bx[i] -= p.Wd; by[i] += 1; }
      while (by[i] >= p.Hd) { by[i] -= p.Hd; bn[i] += 1; }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int idx = t + i * 256;
      int pix = idx / GA, g = idx % GA;
      if (pix < CH) *reinterpret_cast<uint4*>(&lds_a[buf][pix * SA + g * EG]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      int idx = t + i * 256;
      int pix = idx / GB, g = idx % GB;
      if (pix < CH) *reinterpret_cast<uint4*>(&lds_b[buf][pix * SB + g * EG]) = rb[i];
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
#pragma unroll
     for (int ks = 0; ks < CH / 32; ++ks) {
      bf16x8 fa[TA], fb[TB];
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        const T* base = &lds_a[buf][(ks * 32 + lg * 8 + (li >> 2)) * SA + wr * WROW + a * 16 + (li & 3) * 4];
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * SA));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        fa[a] = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
      }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const T* base = &lds_b[buf][(ks * 32 + lg * 8 + (li >> 2)) * SB + wcn * WCOL + b * 16 + (li & 3) * 4];
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
     }
    } else {
#pragma unroll
      for (int kk = 0; kk < CH / 4; ++kk) {
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
    float* ws = p.workspace + (long)zb * p.ws_rows * p.ws_cols;
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

// Slab reductions take the two problems of a shared launch in one launch (blockIdx.z).  (Round 5 also had a batched form —
// the reductions of a whole hand-over batch of layers in two or three launches — which measured 0.8 % slower on the step:
// a reduction launched right behind its main kernel reads slabs that are still in L2 / Infinity Cache.  Removed in round 6.)
constexpr int WG_MULTI = 2;
struct WgMulti { FsWgradArgs a[WG_MULTI]; };

// dw[co][ci][r][s] += sum_z workspace[z][co][col].  A block owns 64 consecutive columns of one row: 16 column quads x
// 16 split lanes, every lane's 16-byte loads (slabs z, z + 16, ...) issued eight at a time — a 128-slab reduction is
// ONE round of loads per thread (the 4-byte / four-in-flight version spent eight round trips: 10.9 us for 18.9 MB) —
// then the 16 lanes combine through LDS.  (ncols and ws_cols are multiples of 4 by construction.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgMulti d, int eg) {
  const FsWgradArgs& p = d.a[blockIdx.z];
  __shared__ float4 red[16][16];
  const int ncols = p.ncolgroups * eg;
  const int cblocks = (ncols + 63) / 64;
  if ((int)blockIdx.x >= p.Co * cblocks) return;      // (the grid is the larger problem's)
  const int co = blockIdx.x / cblocks, cbase = (blockIdx.x % cblocks) * 64;
  const int q = threadIdx.x & 15, zl = threadIdx.x >> 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cbase + q * 4 < ncols) {
    const float* ws = p.workspace + (long)co * p.ws_cols + cbase + q * 4;
    const long slab = (long)p.ws_rows * p.ws_cols;
    int z = zl;
    for (; z + 112 < p.nsplit; z += 128) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(ws + (long)(z + 16 * k) * slab);
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        acc.x += v[k].x + v[k + 1].x; acc.y += v[k].y + v[k + 1].y; acc.z += v[k].z + v[k + 1].z; acc.w += v[k].w + v[k + 1].w;
      }
    }
    for (; z + 16 < p.nsplit; z += 32) {
      const float4 a0 = *reinterpret_cast<const float4*>(ws + (long)z * slab);
      const float4 a1 = *reinterpret_cast<const float4*>(ws + (long)(z + 16) * slab);
      acc.x += a0.x + a1.x; acc.y += a0.y + a1.y; acc.z += a0.z + a1.z; acc.w += a0.w + a1.w;
    }
    for (; z < p.nsplit; z += 16) {
      const float4 a0 = *reinterpret_cast<const float4*>(ws + (long)z * slab);
      acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
    }
  }
  red[zl][q] = acc;
  __syncthreads();
  const int col = cbase + threadIdx.x;
  if (threadIdx.x < 64 && col < ncols) {
    const float* r = reinterpret_cast<const float*>(&red[0][0]) + threadIdx.x;     // lane l: r[l * 64]
    float v = 0.f;
#pragma unroll
    for (int l = 0; l < 16; l += 4) v += (r[l * 64] + r[(l + 1) * 64]) + (r[(l + 2) * 64] + r[(l + 3) * 64]);
    int e = p.ktab[col / eg];
    if (e >= 0) {
      int ci = (e & 0xffff) + (col % eg);
      if (ci < p.Ci) {
        int r2 = (e >> 16) & 0xff, s2 = (e >> 24) & 0x7f;
        p.dw[(((long)co * p.Ci + ci) * p.R + r2) * p.S + s2] += v;
      }
    }
  }
}

// the same reduction for fewer than 32 slabs (the 16 split lanes of the kernel above would mostly idle: ResNet-50's 1x1
// layers, 9-31 slabs, measured 13.1 us on this kernel against 15.2 on that one in the step): 64 consecutive columns x 4
// split lanes per block, each lane strides the splits by 4 with independent (unrolled) loads, the 4 lanes combine through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const WgMulti d, int eg) {
  const FsWgradArgs& p = d.a[blockIdx.z];
  __shared__ float red[4][64];
  const int ncols = p.ncolgroups * eg;
  const int cblocks = (ncols + 63) / 64;
  if ((int)blockIdx.x >= p.Co * cblocks) return;
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

// few splits (deep stages: wide dW, 2-8 slabs): one thread per column, all slabs summed with independent loads —
// the 4-lane kernel above would run 4x the blocks with most lanes idle
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const WgMulti d, int eg) {
  const FsWgradArgs& p = d.a[blockIdx.z];
  const int ncols = p.ncolgroups * eg;
  const int cblocks = (ncols + 255) / 256;
  if ((int)blockIdx.x >= p.Co * cblocks) return;
  const int co = blockIdx.x / cblocks, col = (blockIdx.x % cblocks) * 256 + threadIdx.x;
  if (col >= ncols) return;
  const float* ws = p.workspace + (long)co * p.ws_cols + col;
  const long slab = (long)p.ws_rows * p.ws_cols;
  float v[8];
#pragma unroll
  for (int z = 0; z < 8; ++z) v[z] = z < p.nsplit ? ws[(long)z * slab] : 0.f;
  float acc = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  int e = p.ktab[col / eg];
  if (e >= 0) {
    int ci = (e & 0xffff) + (col % eg);
    if (ci < p.Ci) {
      int r = (e >> 16) & 0xff, s = (e >> 24) & 0x7f;
      p.dw[(((long)co * p.Ci + ci) * p.R + r) * p.S + s] += acc;
    }
  }
}

// 3x3 slabs in tap-major column order (col = tap*Cs + ci), few splits: a block owns one co x 32 ci x 9 taps.  It
// reads nine 128-byte column segments per slab, transposes through LDS and adds into dW[co][ci][3][3] as ONE
// contiguous 288-float run (the flat kernel's lanes hit dW with a 36-byte stride: 9x the lines per wave).
__global__ __launch_bounds__(320) void wgrad_reduce3x3_kernel(const WgMulti d, int eg) {
  const FsWgradArgs& p = d.a[blockIdx.z];
  __shared__ float tmp[288];
  const int Cs = p.ncolgroups * eg / 9;
  const int co = blockIdx.y, ci0 = blockIdx.x * 32;
  if (ci0 >= Cs || co >= p.Co) return;
  const int e = threadIdx.x;
  if (e < 288) {
    const int tap = e >> 5, j = e & 31;
    const float* ws = p.workspace + (long)co * p.ws_cols + tap * Cs + ci0 + j;
    const long slab = (long)p.ws_rows * p.ws_cols;
    float acc = 0.f;
    int z = 0;
    for (; z + 7 < p.nsplit; z += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ws[(long)(z + k) * slab];
      acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; z + 3 < p.nsplit; z += 4) {
      float a0 = ws[(long)z * slab], a1 = ws[(long)(z + 1) * slab], a2 = ws[(long)(z + 2) * slab], a3 = ws[(long)(z + 3) * slab];
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; z < p.nsplit; ++z) acc += ws[(long)z * slab];
    tmp[j * 9 + tap] = acc;
  }
  __syncthreads();
  if (e < 288) {
    const int ci = ci0 + e / 9;
    if (co < p.Co && ci < p.Ci) p.dw[((long)co * p.Ci + ci0) * 9 + e] += tmp[e];
  }
}

enum { RK_3X3 = 0, RK_FLAT = 1, RK_4 = 2, RK_16 = 3 };
struct WgPending { FsWgradArgs a; int kind, eg, gx, gy; };

void launch_reduce_group(const WgPending* it, int n, hipStream_t st) {
  WgMulti d;
  int gx = 1, gy = 1;
  for (int i = 0; i < WG_MULTI; ++i) d.a[i] = it[i < n ? i : 0].a;
  for (int i = 0; i < n; ++i) { gx = std::max(gx, it[i].gx); gy = std::max(gy, it[i].gy); }
  const int eg = it[0].eg;
  switch (it[0].kind) {
    case RK_3X3: hipLaunchKernelGGL(wgrad_reduce3x3_kernel, dim3(gx, gy, n), dim3(320), 0, st, d, eg); break;
    case RK_FLAT: hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3(gx, 1, n), dim3(256), 0, st, d, eg); break;
    case RK_4: hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(gx, 1, n), dim3(256), 0, st, d, eg); break;
    default: hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, 1, n), dim3(256), 0, st, d, eg); break;
  }
}

// b: the launch's (filled-in) arguments; b2 != nullptr: the second problem's, same dW shape — one reduce launch for both
// (blockIdx.z = problem).  A problem that was not split (nsplit == 1) accumulated into its dW directly.
void launch_reduce(const FsWgradArgs& b, const FsWgradArgs* b2, int Co, int ncols, int eg, hipStream_t st, bool always = false) {
  WgPending it[2];
  int n = 0, ns = 0;
  if (b.nsplit > 1 || always) { it[n++].a = b; ns = std::max(ns, b.nsplit); }
  if (b2 && (b2->nsplit > 1 || always)) { it[n++].a = *b2; ns = std::max(ns, b2->nsplit); }
  if (n == 0) return;
  const bool tapmajor3x3 = b.R == 3 && b.S == 3 && ncols % 9 == 0 && (ncols / 9) % 32 == 0;
  int kind, gx, gy = 1;
  if (tapmajor3x3 && ns <= 16 && ncols >= 9 * 64) { kind = RK_3X3; gx = ncols / 9 / 32; gy = Co; }
  else if (ns <= 8 && ncols >= 256) { kind = RK_FLAT; gx = Co * ((ncols + 255) / 256); }
  else if (ns < 32) { kind = RK_4; gx = Co * ((ncols + 63) / 64); }
  else { kind = RK_16; gx = Co * ((ncols + 63) / 64); }
  for (int i = 0; i < n; ++i) { it[i].kind = kind; it[i].eg = eg; it[i].gx = gx; it[i].gy = gy; }
  launch_reduce_group(it, n, st);
}

// pixel splits of a launch shared by two problems: `total` slots divided in proportion to their work, at least one each
inline void wg_share(long total, long work0, long work1, long cap0, long cap1, long& s0, long& s1) {
  if (total < 2) total = 2;
  s0 = (total * work0 + (work0 + work1) / 2) / (work0 + work1);
  s0 = std::max<long>(1, std::min(s0, total - 1));
  s1 = total - s0;
  s0 = std::max<long>(1, std::min(s0, cap0));
  s1 = std::max<long>(1, std::min(s1, cap1));
}

// resident blocks of a kernel on the whole device (occupancy x CUs).  A split-K grid a little larger than this runs
// as TWO rounds — 288 blocks of a one-block-per-CU kernel took twice the time of 256 (measured).
template <typename K>
int wg_resident_blocks(K kernel, int threads) {
  int dev = 0, cus = 256, per_cu = 1;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return cus * per_cu;
}

template <typename T, int COT, int CLT, int WR>
int launch_tile(const FsWgradArgs& a, const FsWgradArgs* a2, hipStream_t st) {
  constexpr int EG = ElemTraits<T>::EG;
  // pixels per pipeline stage: every stage waits one global-load latency, so bf16 tiles that fit the LDS budget
  // take 64 pixels (two MFMA K steps) per stage
  constexpr int CH = (sizeof(T) == 2 && (COT + CLT) <= 192) ? 64 : 32;
  FsDual<FsWgradArgs, WgDiv> d;
  FsWgradArgs& b = d.a[0];
  FsWgradArgs& b2 = d.a[1];
  b = a; b2 = a2 ? *a2 : a;
  d.g[0] = WgDiv{fs_make_div(a.Wd), fs_make_div(a.Hd)};
  d.g[1] = WgDiv{fs_make_div(b2.Wd), fs_make_div(b2.Hd)};
  d.nprob = a2 ? 2 : 1;
  const int ncols = a.ncolgroups * EG;
  const int ct = (ncols + CLT - 1) / CLT, rt = (a.Cd + COT - 1) / COT;
  const long tiles = (long)ct * rt;
  const long chunks = (a.M + CH - 1) / CH, chunks2 = a2 ? (a2->M + CH - 1) / CH : 0;
  // split the pixel (K) range until ~640 blocks are in flight, keeping >= 256 pixels per block and the partial
  // slabs inside the caller's workspace.  (Measured: fewer, longer blocks lose — every stage of the pixel loop
  // waits one global-load latency, so the kernel wants many short chains; the slab traffic is the smaller cost.)
  long splits = (640 + tiles - 1) / tiles, splits2 = 0;
  b.ws_rows = rt * COT; b.ws_cols = ct * CLT;
  b2.ws_rows = b.ws_rows; b2.ws_cols = b.ws_cols;
  const long slab = (long)b.ws_rows * b.ws_cols;
  if (a2) {
    const long cap = std::max<long>(1, chunks / (256 / CH)), cap2 = std::max<long>(1, chunks2 / (256 / CH));
    wg_share(splits, chunks, chunks2, cap, cap2, splits, splits2);
    // (the two problems share the first problem's workspace: both or neither split)
    const long room = a.workspace ? a.workspace_elems / slab : 0;
    if (room < 2) { splits = 1; splits2 = 1; }
    else if (splits + splits2 > room) { splits = std::max<long>(1, room * splits / (splits + splits2)); splits2 = std::max<long>(1, room - splits); }
  } else {
    splits = std::min<long>(splits, std::max<long>(1, chunks / (256 / CH)));
    if (!a.workspace) splits = 1;
    else splits = std::min<long>(splits, std::max<long>(1, a.workspace_elems / slab));
  }
  long cps = (chunks + splits - 1) / splits;
  b.pix_per_split = (int)(cps * CH);
  b.nsplit = (int)((chunks + cps - 1) / cps);
  d.nb0 = b.nsplit;
  int nz = b.nsplit;
  if (a2) {
    cps = (chunks2 + splits2 - 1) / splits2;
    b2.pix_per_split = (int)(cps * CH);
    b2.nsplit = (int)((chunks2 + cps - 1) / cps);
    b2.workspace = a.workspace ? a.workspace + (long)b.nsplit * slab : nullptr;
    nz += b2.nsplit;
  }
  dim3 grid(ct, rt, nz);
  if (g_plan) return wg_plan(0, (long)ct * rt * nz, 256, wg_resident_blocks(conv_wgrad_kernel<T, COT, CLT, WR, CH>, 256)) ? 0 : FS_EINVAL;
  hipLaunchKernelGGL((conv_wgrad_kernel<T, COT, CLT, WR, CH>), grid, dim3(256), 0, st, d);
  launch_reduce(b, a2 ? &b2 : nullptr, a.Co, ncols, EG, st);
  return fs_launch_status();
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride-1 weight gradient with an LDS-resident input halo (bf16).  One block owns a
// COT x (9 taps x CIT) slice of dW and walks pixel tiles (TH x TW = 128 pixels): the dY tile and the
// (TH+2) x (TW+2) input halo are fetched once and all nine taps multiply out of LDS (transposed fragment
// reads at shifted halo rows), i.e. 9x fewer input fetches and 72 MFMAs per wave between barriers; the
// generic kernel above re-gathers the input per tap and synchronises every 4 MFMAs.
// ---------------------------------------------------------------------------------------------
struct WGeom { int TH, TW, tiles_x, tiles_y, N, Cs, nsplit; unsigned mTW, mHW; FsDiv dTX, dTY; };

__device__ __forceinline__ uint4 wg_buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

// NG tap groups: NG x 4 waves share the staged tile, group g multiplies only its taps [tap0(g), tap0(g+1)) — the CU
// holds NG waves per SIMD on ONE stage buffer and ONE output slab (the nine taps are disjoint slices of it, nothing is
// added across groups), each wave's LDS -> MFMA chain is 9/NG taps long and the tile is staged by NG x 256 threads.
// (With one group a wave's chain per tile — 88 transposing reads, 72 MFMAs, 16 LDS stores, two barriers — ran alone on
// its SIMD: ds_read_b64_tr_b16 reaches its rate only from several waves per SIMD, MI355X_MICROARCH.md / LDS.)
template <int COT, int CIT, int NG>
__global__ __launch_bounds__(256 * NG) void wgrad3x3_halo_kernel(const FsDual<FsWgradArgs, WGeom> d) {
  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsWgradArgs& p = d.a[prob];
  const WGeom& g = d.g[prob];
  const int zb = (int)blockIdx.z - (prob ? d.nb0 : 0);
  typedef bf16 T;
  constexpr int NT = 256 * NG;
  constexpr int TPG = (9 + NG - 1) / NG;               // taps per group (the last groups may hold one fewer)
  constexpr int PIXT = 128, HMAX = 208, OOB = 0x7fffffff;
  constexpr int SA = COT + 24, SB = CIT + 24;   // LDS row pads (elements): 8 / 16 / 24 / 32 each measured, 24 + 24 is 3-6 % ahead of 8 + 8; a 128-byte x row costs 30-50 %
  constexpr int TA = COT / 32, TB = CIT / 32;          // per-wave 16x16 sub-tiles (2 x 2 waves)
  constexpr int UA = COT / 8, UB = CIT / 8;            // 16-byte units per pixel row
  constexpr int LA = (PIXT * UA + NT - 1) / NT, LB = (HMAX * UB + NT - 1) / NT;
  constexpr int LDS_BYTES = (PIXT * SA + HMAX * SB) * 2;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  T* lds_a = reinterpret_cast<T*>(lds_raw);
  T* lds_b = lds_a + PIXT * SA;

  const int t = threadIdx.x, lane = t & 63, wave = (t >> 6) & 3;
  const int grp = NG > 1 ? __builtin_amdgcn_readfirstlane(t >> 8) : 0;
  const int tap0 = (grp * 9 + NG - 1) / NG, ntap = ((grp + 1) * 9 + NG - 1) / NG - tap0;
  const int wr = wave & 1, wcn = wave >> 1;
  const int li = lane & 15, lg = lane >> 4;
  const int HW = g.TW + 2, nhalo = (g.TH + 2) * HW, ntile = g.TH * g.TW;
  const int ci0 = blockIdx.x * CIT, co0 = blockIdx.y * COT;
  const int npix = g.N * g.tiles_y * g.tiles_x;

  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.dy), 0, (int)((long)p.M * p.Cd * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);

  // Which pixel a thread stages.  A 16-byte ds_write is served 16 lanes at a time; with the padded row strides (44 / 28
  // dwords) the two (four) consecutive pixels a 16-lane group would write overlap on 12 of the 64 banks.  Within each run
  // of 16 pixel slots the group takes pixels whose rows tile the banks exactly instead: p and p + 8 for 8 units per pixel
  // (11 k = 8 mod 16 at k = 8), p, p + 4, p + 8, p + 12 for 4 (7 k = 12, 8, 4 mod 16 at k = 4, 8, 12).  The global loads
  // keep a pixel's units in consecutive lanes.
  auto stage_pix = [](int q, int U) {
    const int q16 = q & 15, base = q - q16;
    return U == 8 ? base + (q16 >> 1) + 8 * (q16 & 1) : (U == 4 ? base + (q16 >> 2) + 4 * (q16 & 3) : q);
  };
  // per-thread load units: tile-relative coordinates are fixed, only (n, y0, x0) changes per tile
  int aty[LA], atx[LA], bhy[LB], bhx[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    int pix = stage_pix((t + i * NT) / UA, UA);
    const int aq = fs_fastdiv(min(pix, 4095), g.mTW);
    aty[i] = pix < ntile ? aq : -1; atx[i] = pix < ntile ? pix - aq * g.TW : 0;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    int hp = stage_pix((t + i * NT) / UB, UB);
    const int bq = fs_fastdiv(min(hp, 4095), g.mHW);
    bhy[i] = hp < nhalo ? bq : -1; bhx[i] = hp < nhalo ? hp - bq * HW : 0;
  }
  // byte offsets of the units relative to the tile origin (fixed over the tiles)
  int arel[LA], brel[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) arel[i] = ((aty[i] * p.Wd + atx[i]) * p.Cd + co0 + ((t + i * NT) % UA) * 8) * 2;
#pragma unroll
  for (int i = 0; i < LB; ++i)
    brel[i] = (int)(((long)bhy[i] * p.sH + (long)bhx[i] * p.sW + ci0 + ((t + i * NT) % UB) * 8) * 2);
  uint4 ra[LA], rb[LB];
  // operand prologue (BatchNorm + ReLU folded into the staging of x): this thread's 16-byte units all hold the same 8
  // input channels (NT % UB == 0), coefficients per statistics group of the tile's image
  static_assert(NT % UB == 0, "a thread's x units must share their channel slot");
  const bool has_pro = p.pro_a != nullptr;
  float ka[8], kb[8];
  unsigned bok = 0u;                       // which of the thread's x units lie inside the image (padding stays zero)
  auto load_regs = [&](int pt) {
    int q = fs_div(pt, g.dTX); int tx_i = pt - q * g.tiles_x; int n = fs_div(q, g.dTY); int ty_i = q - n * g.tiles_y;
    if (has_pro) {
      const int pg = p.pro_group_imgs > 0 ? n / p.pro_group_imgs : 0;
      const float* pa = p.pro_a + (long)pg * g.Cs + ci0 + (t % UB) * 8;
      const float* pb = p.pro_b + (long)pg * g.Cs + ci0 + (t % UB) * 8;
      const float4 a0 = reinterpret_cast<const float4*>(pa)[0], a1 = reinterpret_cast<const float4*>(pa)[1];
      const float4 b0 = reinterpret_cast<const float4*>(pb)[0], b1 = reinterpret_cast<const float4*>(pb)[1];
      ka[0] = a0.x; ka[1] = a0.y; ka[2] = a0.z; ka[3] = a0.w; ka[4] = a1.x; ka[5] = a1.y; ka[6] = a1.z; ka[7] = a1.w;
      kb[0] = b0.x; kb[1] = b0.y; kb[2] = b0.z; kb[3] = b0.w; kb[4] = b1.x; kb[5] = b1.y; kb[6] = b1.z; kb[7] = b1.w;
      bok = 0u;
    }
    int y0 = ty_i * g.TH, x0 = tx_i * g.TW;
    // tile origin as 32-bit byte offsets (both tensors are below 2 GiB: the buffer descriptors require it)
    const int abase = (((n * p.Hd + y0) * p.Wd + x0) * p.Cd) * 2;
    const int bbase = (int)(((long)n * p.sN + (long)(y0 - p.pad) * p.sH + (long)(x0 - p.pad) * p.sW) * 2);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const bool ok = aty[i] >= 0 && y0 + aty[i] < p.Hd && x0 + atx[i] < p.Wd;
      ra[i] = wg_buf_load16(rs_dy, ok ? abase + arel[i] : OOB);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int sy = y0 - p.pad + bhy[i], sx = x0 - p.pad + bhx[i];
      const bool ok = bhy[i] >= 0 && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      rb[i] = wg_buf_load16(rs_x, ok ? bbase + brel[i] : OOB);
      if (has_pro && ok) bok |= 1u << i;
    }
  };
  auto pro_unit = [&](uint4 u) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = bf16_bits_to_f(w[j] & 0xffffu) * ka[2 * j] + kb[2 * j];
      float hi = __uint_as_float(w[j] & 0xffff0000u) * ka[2 * j + 1] + kb[2 * j + 1];
      if (p.pro_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
      o[j] = pack_bf16x2(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int idx = t + i * NT; int pix = stage_pix(idx / UA, UA), u = idx % UA;
      if (pix < PIXT) *reinterpret_cast<uint4*>(&lds_a[pix * SA + u * 8]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      int idx = t + i * NT; int hp = stage_pix(idx / UB, UB), u = idx % UB;
      uint4 v = rb[i];
      if (has_pro) v = ((bok >> i) & 1u) ? pro_unit(v) : make_uint4(0u, 0u, 0u, 0u);
      if (hp < HMAX) *reinterpret_cast<uint4*>(&lds_b[hp * SB + u * 8]) = v;
    }
  };

  // transposed-fragment row offsets.  A ds_read_b64_tr_b16 hands lane (li, lg) four consecutive pixels' values of
  // channel li from the four rows its 16-lane group addresses; which 8 of the K step's 32 pixels a lane group takes is
  // free as long as dY and x agree.  The 32 lanes served in one LDS cycle take same-parity pixels of a 16-pixel span:
  // with row strides of 44 (dY) and 28 (x) dwords — as with 36 and 20 — those eight 32-byte rows tile the 64 banks
  // (consecutive pixels collide two-way) — pixel k = ks*32 + (lg>>1)*16 + 2*((lg&1)*4 + (li>>2)) + half
  int arow[4][2], brow[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      int pk = ks * 32 + (lg >> 1) * 16 + 2 * ((lg & 1) * 4 + (li >> 2)) + hf;
      arow[ks][hf] = pk * SA + wr * (COT / 2) + (li & 3) * 4;
      int pv = pk < ntile ? pk : 0;
      const int pq = fs_fastdiv(pv, g.mTW);
      brow[ks][hf] = (pq * HW + (pv - pq * g.TW)) * SB + wcn * (CIT / 2) + (li & 3) * 4;
    }
  // the group's tap offsets (halo elements), wave-uniform
  int toff[TPG];
#pragma unroll
  for (int tp = 0; tp < TPG; ++tp) {
    const int tap = tap0 + (tp < ntap ? tp : 0);
    toff[tp] = ((tap / 3) * HW + (tap % 3)) * SB;
  }

  f32x4 acc[TPG][TA][TB];
#pragma unroll
  for (int tp = 0; tp < TPG; ++tp)
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) acc[tp][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the block's pixel tiles z, z + nsplit, ...
  const int step = g.nsplit;
  int pt = zb;
  if (pt < npix) load_regs(pt);
  for (; pt < npix; pt += step) {
    __syncthreads();
    store_lds();
    __syncthreads();
    if (pt + step < npix) load_regs(pt + step);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[TA];
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_a[arow[ks][0] + a * 16]));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_a[arow[ks][1] + a * 16]));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        fa[a] = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
      }
#pragma unroll
      for (int tp = 0; tp < TPG; ++tp) {
        if (NG > 1 && tp >= ntap) break;
#pragma unroll
        for (int b = 0; b < TB; ++b) {
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_b[brow[ks][0] + toff[tp] + b * 16]));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_b[brow[ks][1] + toff[tp] + b * 16]));
          uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          bf16x8 fb = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
#pragma unroll
          for (int a = 0; a < TA; ++a)
            acc[tp][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[a], acc[tp][a][b], 0, 0, 0);   // D[ci][co]
        }
      }
    }
  }

  // ---- epilogue: the MFMA ran with the input-channel fragment as its row operand, D rows = ci (lg*4 + j), cols = co
  // (li): a lane holds four consecutive input channels of one output channel, i.e. one 16-byte run of the slab row
  // [co][tap][ci] (the other orientation stored 288 single floats per lane: a sixth of the kernel) ----
#pragma unroll
  for (int tp = 0; tp < TPG; ++tp) {
    if (NG > 1 && tp >= ntap) break;
    const int tap = tap0 + tp;
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int ci = ci0 + wcn * (CIT / 2) + b * 16 + lg * 4;
#pragma unroll
      for (int a = 0; a < TA; ++a) {
        const int co = co0 + wr * (COT / 2) + a * 16 + li;
        const f32x4 v = acc[tp][a][b];
        if (g.nsplit > 1) {
          *reinterpret_cast<float4*>(&p.workspace[((long)zb * p.ws_rows + co) * p.ws_cols + tap * g.Cs + ci]) =
              make_float4(v[0], v[1], v[2], v[3]);
        } else if (co < p.Co) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ci + j < p.Ci) p.dw[(((long)co * p.Ci + ci + j) * 3 + tap / 3) * 3 + tap % 3] += v[j];
        }
      }
    }
  }
}

template <int COT, int CIT>
__global__ __launch_bounds__(256) void wgrad3x3_narrow_kernel(const FsWgradArgs p, const WGeom g) {
  typedef bf16 T;
  constexpr int NWR = COT / 16, NWK = 4 / NWR, KS = 4 / NWK;   // waves along co / along pixels, K steps per wave
  constexpr int PIXT = 128, HMAX = 208, OOB = 0x7fffffff;
  constexpr int SA = COT + 8, SB = CIT + 8;
  constexpr int TB = CIT / 16;
  constexpr int UA = COT / 8, UB = CIT / 8;
  constexpr int LA = (PIXT * UA + 255) / 256, LB = (HMAX * UB + 255) / 256;
  constexpr int NACC = 9 * TB * 4;                       // accumulator floats per lane
  // one arena: operand tiles during the walk, then (re-used) one wave's partial accumulators at a time
  constexpr int OPER_BYTES = (PIXT * SA + HMAX * SB) * 2, RED_BYTES = NWR * NACC * 64 * 4;
  __shared__ __attribute__((aligned(16))) char smem[OPER_BYTES > RED_BYTES ? OPER_BYTES : RED_BYTES];
  T* lds_a = reinterpret_cast<T*>(smem);
  T* lds_b = lds_a + PIXT * SA;
  float* lds_red = reinterpret_cast<float*>(smem);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave % NWR, wk = wave / NWR;
  const int li = lane & 15, lg = lane >> 4;
  const int HW = g.TW + 2, nhalo = (g.TH + 2) * HW, ntile = g.TH * g.TW;
  const int ci0 = blockIdx.x * CIT;
  const int npix = g.N * g.tiles_y * g.tiles_x;

  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.dy), 0, (int)((long)p.M * p.Cd * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);

  int aty[LA], atx[LA], bhy[LB], bhx[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    int pix = (t + i * 256) / UA;
    const int aq = fs_fastdiv(min(pix, 4095), g.mTW);
    aty[i] = pix < ntile ? aq : -1; atx[i] = pix < ntile ? pix - aq * g.TW : 0;
  }
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    int hp = (t + i * 256) / UB;
    const int bq = fs_fastdiv(min(hp, 4095), g.mHW);
    bhy[i] = hp < nhalo ? bq : -1; bhx[i] = hp < nhalo ? hp - bq * HW : 0;
  }
  // byte offsets of the units relative to the tile origin (fixed over the tiles)
  int arel[LA], brel[LB];
#pragma unroll
  for (int i = 0; i < LA; ++i) arel[i] = ((aty[i] * p.Wd + atx[i]) * p.Cd + ((t + i * 256) % UA) * 8) * 2;
#pragma unroll
  for (int i = 0; i < LB; ++i)
    brel[i] = (int)(((long)bhy[i] * p.sH + (long)bhx[i] * p.sW + ci0 + ((t + i * 256) % UB) * 8) * 2);
  uint4 ra[LA], rb[LB];
  // (no operand prologue here: launch_wgrad sends every launch that carries one to the halo kernel above)
  auto load_regs = [&](int pt) {
    int q = fs_div(pt, g.dTX); int tx_i = pt - q * g.tiles_x; int n = fs_div(q, g.dTY); int ty_i = q - n * g.tiles_y;
    int y0 = ty_i * g.TH, x0 = tx_i * g.TW;
    const int abase = (((n * p.Hd + y0) * p.Wd + x0) * p.Cd) * 2;
    const int bbase = (int)(((long)n * p.sN + (long)(y0 - p.pad) * p.sH + (long)(x0 - p.pad) * p.sW) * 2);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const bool ok = aty[i] >= 0 && y0 + aty[i] < p.Hd && x0 + atx[i] < p.Wd;
      ra[i] = wg_buf_load16(rs_dy, ok ? abase + arel[i] : OOB);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int sy = y0 - p.pad + bhy[i], sx = x0 - p.pad + bhx[i];
      const bool ok = bhy[i] >= 0 && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      rb[i] = wg_buf_load16(rs_x, ok ? bbase + brel[i] : OOB);
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      int idx = t + i * 256; int pix = idx / UA, u = idx % UA;
      if (pix < PIXT) *reinterpret_cast<uint4*>(&lds_a[pix * SA + u * 8]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      int idx = t + i * 256; int hp = idx / UB, u = idx % UB;
      if (hp < HMAX) *reinterpret_cast<uint4*>(&lds_b[hp * SB + u * 8]) = rb[i];
    }
  };

  // this wave's K step: pixels wave*32 .. wave*32+31; lane (li, lg) supplies pixel lg*8 + (li>>2) (+4)
  int arow[KS][2], brow[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      int pk = (wk * KS + ks) * 32 + lg * 8 + (li >> 2) + hf * 4;
      arow[ks][hf] = pk * SA + wr * 16 + (li & 3) * 4;
      int pv = pk < ntile ? pk : 0;          // (pixels past the tile hold zero dY rows)
      const int pq = fs_fastdiv(pv, g.mTW);
      brow[ks][hf] = (pq * HW + (pv - pq * g.TW)) * SB + (li & 3) * 4;
    }

  f32x4 acc[9][TB];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[tp][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int pt = blockIdx.z;
  if (pt < npix) load_regs(pt);
  for (; pt < npix; pt += g.nsplit) {
    __syncthreads();
    store_lds();
    __syncthreads();
    if (pt + g.nsplit < npix) load_regs(pt + g.nsplit);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_a[arow[ks][0]]));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_a[arow[ks][1]]));
      uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      const bf16x8 fa = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int toff = ((tp / 3) * HW + (tp % 3)) * SB;
#pragma unroll
        for (int b = 0; b < TB; ++b) {
          s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_b[brow[ks][0] + toff + b * 16]));
          s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_b[brow[ks][1] + toff + b * 16]));
          uint2 bl = __builtin_bit_cast(uint2, blo), bh = __builtin_bit_cast(uint2, bhi);
          bf16x8 fb = __builtin_bit_cast(bf16x8, make_uint4(bl.x, bl.y, bh.x, bh.y));
          acc[tp][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[tp][b], 0, 0, 0);
        }
      }
    }
  }

  // ---- the four K partials -> wave 0, one wave at a time through the (now idle) operand arena ----
  for (int w = 1; w < NWK; ++w) {
    __syncthreads();
    if (wk == w) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) lds_red[((wr * NACC + (tp * TB + b) * 4 + j)) * 64 + lane] = acc[tp][b][j];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[tp][b][j] += lds_red[((wr * NACC + (tp * TB + b) * 4 + j)) * 64 + lane];
    }
  }
  if (wk > 0) return;
  // ---- epilogue: D rows = co (lg*4 + j), cols = ci (li) ----
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      int ci = ci0 + b * 16 + li;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int co = wr * 16 + lg * 4 + j;
        float v = acc[tp][b][j];
        if (g.nsplit > 1) {
          p.workspace[((long)blockIdx.z * p.ws_rows + co) * p.ws_cols + tp * g.Cs + ci] = v;
        } else if (co < p.Co && ci < p.Ci) {
          p.dw[(((long)co * p.Ci + ci) * 3 + tp / 3) * 3 + tp % 3] += v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// 7x7 / stride-2 / pad-3 stem (8-channel bf16 input, 64 output channels): dW[co][(r,s,ci)] = sum over pixels of
// dY[pix][co] * X[2y+r-3][2x+s-3][ci].  The generic kernel gathers the im2col operand from global memory, 49 x 16 B
// per output pixel (578 MB for the stacked pose batch: 145 us at the tail of the step, where nothing overlaps it).
// Here a persistent block stages an 8 x 16-pixel tile of dY and the 21 x 37-pixel input patch in LDS once; an MFMA
// column tile is TWO horizontally adjacent taps x 8 channels = 32 contiguous bytes of the patch, so the transposing
// LDS read that serves the 3x3 kernel's B operand works unchanged (rows = pixels at patch stride 2).  Each wave owns
// 7 of the 28 column tiles (7 kernel rows x 4 tap pairs; the 8th tap of a row does not exist and is not written) for
// all 64 output channels and keeps them in accumulators over the block's tiles; one fp32 slab per block at the end.
// ---------------------------------------------------------------------------------------------
struct StemGeom { int tiles_x, tiles_y, ntiles, tiles_per_block; FsDiv dTX, dTY; };

__global__ __launch_bounds__(256) void wgrad_stem_kernel(const FsDual<FsWgradArgs, StemGeom> d) {
  const int prob = (int)blockIdx.x >= d.nb0 ? 1 : 0;
  const FsWgradArgs& p = d.a[prob];
  const int bid = (int)blockIdx.x - (prob ? d.nb0 : 0);
  const int tiles_x = d.g[prob].tiles_x, tiles_y = d.g[prob].tiles_y, ntiles = d.g[prob].ntiles;
  const int tiles_per_block = d.g[prob].tiles_per_block;
  const FsDiv dTX = d.g[prob].dTX, dTY = d.g[prob].dTY;
  typedef bf16 T;
  constexpr int TY = 8, TX = 16, PIXT = TY * TX, PH = 2 * TY + 5, PW = 2 * TX + 6;
  constexpr int SA = 64 + 8, OOB = 0x7fffffff;
  constexpr int LA = PIXT * 8 / 256, LB = (PH * PW + 255) / 256;
  __shared__ __attribute__((aligned(16))) T lds_a[PIXT * SA];
  __shared__ __attribute__((aligned(16))) T lds_b[PH * PW * 8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.dy), 0, (int)((long)p.M * p.Cd * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);

  uint4 ra[LA], rb[LB];
  auto load_regs = [&](int tile) {
    const int q = fs_div(tile, dTX); const int tx_i = tile - q * tiles_x; const int n = fs_div(q, dTY); const int ty_i = q - n * tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int idx = t + i * 256, pix = idx >> 3, u = idx & 7;
      const int y = y0 + (pix >> 4), x = x0 + (pix & 15);
      const bool ok = y < p.Hd && x < p.Wd;
      ra[i] = wg_buf_load16(rs_dy, ok ? (int)((((long)n * p.Hd + y) * p.Wd + x) * p.Cd + u * 8) * 2 : OOB);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int idx = t + i * 256, py = idx / PW, px = idx - py * PW;
      const int sy = 2 * y0 - 3 + py, sx = 2 * x0 - 3 + px;
      const bool ok = idx < PH * PW && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      rb[i] = wg_buf_load16(rs_x, ok ? (int)(((long)n * p.sN + (long)sy * p.sH + (long)sx * p.sW) * 2) : OOB);
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int idx = t + i * 256, pix = idx >> 3, u = idx & 7;
      *reinterpret_cast<uint4*>(&lds_a[pix * SA + u * 8]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int idx = t + i * 256;
      if (idx < PH * PW) *reinterpret_cast<uint4*>(&lds_b[idx * 8]) = rb[i];
    }
  };

  // transposed-fragment row offsets (see wgrad3x3_halo_kernel): lane (li, lg) supplies pixel
  // k = ks*32 + lg*8 + (li>>2) (+4), i.e. tile row ks*2 + (lg>>1), column (lg&1)*8 + (li>>2) (+4)
  int arow[4][2], brow[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int ty = ks * 2 + (lg >> 1), tx = (lg & 1) * 8 + (li >> 2) + hf * 4;
      arow[ks][hf] = (ty * TX + tx) * SA + (li & 3) * 4;
      brow[ks][hf] = ((2 * ty) * PW + 2 * tx) * 8 + (li & 3) * 4;
    }

  f32x4 acc[4][7];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 7; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tile0 = bid * tiles_per_block, tile1 = min(tile0 + tiles_per_block, ntiles);
  if (tile0 < tile1) load_regs(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    __syncthreads();
    store_lds();
    __syncthreads();
    if (tile + 1 < tile1) load_regs(tile + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_a[arow[ks][0] + a * 16]));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_a[arow[ks][1] + a * 16]));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        fa[a] = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
      }
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        const int ct = wave * 7 + c;                         // column tile: kernel row ct / 4, tap pair ct % 4
        const int toff = ((ct >> 2) * PW + 2 * (ct & 3)) * 8;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_b[brow[ks][0] + toff]));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(&lds_b[brow[ks][1] + toff]));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        const bf16x8 fb = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[a][c], 0, 0, 0);
      }
    }
  }

  // D rows = co (a*16 + lg*4 + j), cols = li: tap 2*pair + (li >> 3), channel li & 7
  float* ws = p.workspace + (long)bid * p.ws_rows * p.ws_cols;
#pragma unroll
  for (int c = 0; c < 7; ++c) {
    const int ct = wave * 7 + c, r = ct >> 2, s = 2 * (ct & 3) + (li >> 3);
    if (s >= 7) continue;
    const int col = (r * 7 + s) * 8 + (li & 7);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) ws[(long)(a * 16 + lg * 4 + j) * p.ws_cols + col] = acc[a][c][j];
  }
}

WGeom wgrad_pick_geom(int Hd, int Wd) {
  WGeom best{0, 0, 0, 0, 0, 0, 1, 0u, 0u, FsDiv{0u, 0u}, FsDiv{0u, 0u}};
  double best_cost = 1e30;
  for (int tw = std::min(4, Wd); tw <= std::min(Wd, 64); ++tw) {
    int th = std::min(128 / tw, Hd);
    if (th < 1 || (th + 2) * (tw + 2) > 208) continue;
    int tx = (Wd + tw - 1) / tw, ty = (Hd + th - 1) / th;
    double waste = (double)tx * ty * 128 / ((double)Hd * Wd);
    double halo = (double)(th + 2) * (tw + 2) / ((double)th * tw);
    double cost = waste * (1.0 + 0.15 * halo);
    if (cost < best_cost - 1e-9) { best_cost = cost; best.TH = th; best.TW = tw; best.tiles_x = tx; best.tiles_y = ty; }
  }
  if (best.TW > 0) {
    best.mTW = fs_div_magic(best.TW); best.mHW = fs_div_magic(best.TW + 2);
    best.dTX = fs_make_div(best.tiles_x); best.dTY = fs_make_div(best.tiles_y);
  }
  return best;
}

template <int COT, int CIT>
int launch_wgrad_halo_t(const FsWgradArgs& a, const FsWgradArgs* a2, hipStream_t st) {
  FsDual<FsWgradArgs, WGeom> d;
  FsWgradArgs& b = d.a[0];
  FsWgradArgs& b2 = d.a[1];
  b = a; b2 = a2 ? *a2 : a;
  d.nprob = a2 ? 2 : 1;
  const int Cs = a.ncolgroups * 8 / 9;
  WGeom& g = d.g[0];
  g = wgrad_pick_geom(a.Hd, a.Wd);
  if (g.TH == 0) return FS_EINVAL;
  g.N = a.M / (a.Hd * a.Wd); g.Cs = Cs;
  d.g[1] = g;
  d.g[1].N = b2.M / (b2.Hd * b2.Wd);
  const int out_tiles = (a.Cd / COT) * (Cs / CIT);
  const int npix = g.N * g.tiles_x * g.tiles_y, npix2 = a2 ? d.g[1].N * g.tiles_x * g.tiles_y : 0;
  b.ws_rows = a.Cd; b.ws_cols = 9 * Cs;
  b2.ws_rows = b.ws_rows; b2.ws_cols = b.ws_cols;
  const long slab = (long)b.ws_rows * b.ws_cols;
  // one round of resident blocks: every split adds a Cd x 9Cs fp32 slab to write and re-read, every block beyond the
  // resident ones waits for a whole block to finish
  static const int slots = wg_resident_blocks(wgrad3x3_halo_kernel<COT, CIT, 2>, 512);
  long splits, splits2 = 0;
  if (a2) {
    // both problems in the one round: the device's slots divided by their pixel tiles; one shared slab arena (a's)
    wg_share(std::max(2, slots / out_tiles), npix, npix2, std::max(1, npix / 2), std::max(1, npix2 / 2), splits, splits2);
    const long room = a.workspace ? a.workspace_elems / slab : 0;
    if (room < 2) { splits = 1; splits2 = 1; }
    else if (splits + splits2 > room) { splits = std::max<long>(1, room * splits / (splits + splits2)); splits2 = std::max<long>(1, room - splits); }
  } else {
    splits = std::max<long>(1, std::min<long>(npix / 2 > 0 ? npix / 2 : 1, std::max(1, slots / out_tiles)));
    if (!a.workspace) splits = 1;
    else splits = std::min<long>(splits, std::max<long>(1, a.workspace_elems / slab));
  }
  g.nsplit = (int)splits; b.nsplit = g.nsplit;
  d.nb0 = g.nsplit;
  int nz = g.nsplit;
  if (a2) {
    d.g[1].nsplit = (int)splits2; b2.nsplit = (int)splits2;
    b2.workspace = a.workspace ? a.workspace + (long)b.nsplit * slab : nullptr;
    nz += (int)splits2;
  }
  dim3 grid(Cs / CIT, a.Cd / COT, nz);
  // two tap groups: one, three and four measured — 35.0 / 30.3 / 30.3 / 30.1 us at 288 blocks, 25.0 / 22.9 / - / 22.5 us
  // at 256 (64 -> 64 @48x160 B=12, with the reduce); in the step four groups (1024-thread blocks) lose to two
  if (wg_plan(1, (long)grid.x * grid.y * grid.z, 512, slots)) return 0;
  hipLaunchKernelGGL((wgrad3x3_halo_kernel<COT, CIT, 2>), grid, dim3(512), 0, st, d);
  launch_reduce(b, a2 ? &b2 : nullptr, a.Co, 9 * Cs, 8, st);
  return fs_launch_status();
}

int launch_wgrad_halo(const FsWgradArgs& a, const FsWgradArgs* a2, hipStream_t st) {
  // (32 x 32 tiles — twice the tiles, half the pixel splits and slabs — measured 0.5-1 % slower end to end)
  return launch_wgrad_halo_t<64, 32>(a, a2, st);
}

int launch_wgrad_narrow(const FsWgradArgs& a, hipStream_t st) {
  FsWgradArgs b = a;
  const int Cs = a.ncolgroups * 8 / 9;
  WGeom g = wgrad_pick_geom(a.Hd, a.Wd);
  if (g.TH == 0) return FS_EINVAL;
  g.N = a.M / (a.Hd * a.Wd); g.Cs = Cs;
  const int CIT = Cs % 32 == 0 ? 32 : 16;
  const int out_tiles = Cs / CIT;
  const int COT = a.Cd;                      // 16 or 32: the block owns all output channels
  const int npix = g.N * g.tiles_x * g.tiles_y;
  b.ws_rows = COT; b.ws_cols = 9 * Cs;
  const long slab = (long)b.ws_rows * b.ws_cols;
  // every tile step waits one global-load latency: as many short chains as the device holds at once (2-4 blocks per
  // CU by the instantiation's registers) — one round, see wg_resident_blocks; the slabs are tiny (16 x 9Cs floats)
  static const int slots_32_32 = wg_resident_blocks(wgrad3x3_narrow_kernel<32, 32>, 256);
  static const int slots_16_32 = wg_resident_blocks(wgrad3x3_narrow_kernel<16, 32>, 256);
  static const int slots_16_16 = wg_resident_blocks(wgrad3x3_narrow_kernel<16, 16>, 256);
  const int slots = COT == 32 ? slots_32_32 : CIT == 32 ? slots_16_32 : slots_16_16;
  long splits = std::max<long>(1, std::min<long>(npix / 4 > 0 ? npix / 4 : 1, std::max(1, slots / out_tiles)));
  if (!a.workspace) splits = 1;
  else splits = std::min<long>(splits, std::max<long>(1, a.workspace_elems / slab));
  g.nsplit = (int)splits; b.nsplit = g.nsplit;
  dim3 grid(out_tiles, 1, g.nsplit);
  if (wg_plan(2, (long)grid.x * grid.z, 256, slots)) return (COT == 32 && CIT == 32) || (COT == 16 && (CIT == 32 || CIT == 16)) ? 0 : FS_EINVAL;
  if (COT == 32 && CIT == 32) hipLaunchKernelGGL((wgrad3x3_narrow_kernel<32, 32>), grid, dim3(256), 0, st, b, g);
  else if (COT == 16 && CIT == 32) hipLaunchKernelGGL((wgrad3x3_narrow_kernel<16, 32>), grid, dim3(256), 0, st, b, g);
  else if (COT == 16 && CIT == 16) hipLaunchKernelGGL((wgrad3x3_narrow_kernel<16, 16>), grid, dim3(256), 0, st, b, g);
  else return FS_EINVAL;
  launch_reduce(b, nullptr, a.Co, 9 * Cs, 8, st);
  return fs_launch_status();
}

int launch_wgrad_stem(const FsWgradArgs& a, const FsWgradArgs* a2, hipStream_t st) {
  FsDual<FsWgradArgs, StemGeom> d;
  FsWgradArgs& b = d.a[0];
  FsWgradArgs& b2 = d.a[1];
  b = a; b2 = a2 ? *a2 : a;
  d.nprob = a2 ? 2 : 1;
  const int tiles_x = (a.Wd + 15) / 16, tiles_y = (a.Hd + 7) / 8;
  const long ntiles = (long)(a.M / (a.Hd * a.Wd)) * tiles_x * tiles_y;
  const long ntiles2 = a2 ? (long)(a2->M / (a2->Hd * a2->Wd)) * tiles_x * tiles_y : 0;
  b.ws_rows = 64; b.ws_cols = 49 * 8;
  b2.ws_rows = 64; b2.ws_cols = 49 * 8;
  const long slab = (long)b.ws_rows * b.ws_cols;
  // one slab per persistent block (100 KB each, written and re-read), as many blocks as the device holds at once
  static const int slots = wg_resident_blocks(wgrad_stem_kernel, 256);   // (320 blocks of this one-block-per-CU kernel ran as two rounds)
  const long max_blocks = std::min<long>(slots, a.workspace_elems / slab);
  if (max_blocks < (a2 ? 2 : 1) || ntiles < 1 || (a2 && ntiles2 < 1)) return FS_EINVAL;
  long mb = max_blocks, mb2 = 0;
  if (a2) wg_share(max_blocks, ntiles, ntiles2, ntiles, ntiles2, mb, mb2);
  const int per = (int)((ntiles + mb - 1) / mb);
  const int blocks = (int)((ntiles + per - 1) / per);
  b.nsplit = blocks;
  d.g[0] = StemGeom{tiles_x, tiles_y, (int)ntiles, per, fs_make_div(tiles_x), fs_make_div(tiles_y)};
  d.g[1] = d.g[0];
  d.nb0 = blocks;
  int total = blocks;
  if (a2) {
    const int per2 = (int)((ntiles2 + mb2 - 1) / mb2);
    const int blocks2 = (int)((ntiles2 + per2 - 1) / per2);
    b2.nsplit = blocks2;
    b2.workspace = a.workspace + (long)blocks * slab;
    d.g[1].ntiles = (int)ntiles2; d.g[1].tiles_per_block = per2;
    total += blocks2;
  }
  if (wg_plan(3, total, 256, slots)) return 0;
  hipLaunchKernelGGL(wgrad_stem_kernel, dim3(total), dim3(256), 0, st, d);
  // (the persistent blocks always leave slabs, also a single one)
  launch_reduce(b, a2 ? &b2 : nullptr, a.Co, 49 * 8, 8, st, true);
  return fs_launch_status();
}

// ---------------------------------------------------------------------------------------------
// 1x1 weight gradient (bf16) with the operands brought in by LDS-DMA.  dW[co][ci] = sum over pixels of dY[m][co] x[m][ci]
// is a GEMM whose K axis is the pixel axis: at ResNet-50 / 320x1024 the Bottleneck's 1x1 layers (resnet.py:52-89) are 80
// of these per step with K = 2.5 k .. 328 k pixels, and the generic kernel above — one register-staged stage in flight,
// every 64-pixel stage waiting out a global-load latency — ran them at 150-250 TFLOP/s however its splits were chosen.
// Here a ring of NST 32-pixel stages sits in LDS, filled by `buffer_load ... lds` (lds_dma.h) with NST-1 stages in flight
// across one barrier per stage; the images are the natural [pixel][channel] rows (a 1 KB load instruction = 4 or 8 whole
// rows), the fragments are read transposed (ds_read_b64_tr_b16) as in the kernels above, and the 16-byte units of a row
// are XOR-permuted on the source side so that the 32 lanes a transposed read serves together (pixel rows 0-3 and 8-11
// of a k group pair) fall into 64 distinct banks.  Split-K, slabs and the reduce launch are the generic kernel's.
// ---------------------------------------------------------------------------------------------
template <int RB>
__device__ __forceinline__ int w1_key(int r) {     // XOR key (in 16-byte units) of pixel row r of an image with RB-byte rows
  if constexpr (RB == 256) return ((r & 3) | ((r >> 1) & 4)) << 1;
  else return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1;   // (128-byte rows: odd rows already sit in the other bank half)
}

template <int COT, int CIT, int NST>
__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(const FsDual<FsWgradArgs, WgDiv> d) {
  constexpr int RBA = COT * 2, RBB = CIT * 2;          // row bytes of the dY / x images
  constexpr int IMA = 32 * RBA, STAGE = 32 * (RBA + RBB);
  constexpr int NA = RBA / 32, NB = RBB / 32;           // 1 KB load instructions per stage and image
  constexpr int LPW = (NA + NB) / 4;                    // ... per wave
  constexpr int UA = RBA / 16, UB = RBB / 16;           // 16-byte units per row
  constexpr int WROW = COT / 2, WCOL = CIT / 2, TA = WROW / 16, TB = WCOL / 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE];

  const int prob = (int)blockIdx.z >= d.nb0 ? 1 : 0;
  const FsWgradArgs& p = d.a[prob];
  const FsDiv dW = d.g[prob].dW, dH = d.g[prob].dH;
  const int zb = (int)blockIdx.z - (prob ? d.nb0 : 0);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wave & 1, wcn = wave >> 1;
  const int li = lane & 15, lg = lane >> 4;
  const int ci0 = blockIdx.x * CIT, co0 = blockIdx.y * COT;
  const int Cs = p.ncolgroups * 8;
  const long m_begin = (long)zb * p.pix_per_split;
  long m_end = m_begin + p.pix_per_split; if (m_end > p.M) m_end = p.M;
  const int nkt = (int)((m_end - m_begin + 31) / 32);
  const int OOB = 0x7fffffff;
  // dY is dense [M][Cd]: rows from m_end on are beyond the descriptor's extent (they read as zero)
  const i32x4 rs_dy = make_rsrc(p.dy, m_end * p.Cd * 2), rs_x = make_rsrc(p.x, p.x_bytes);
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;

  // ---- loader state: instruction q = wave + 4 i of a stage; q < NA fills rows of the dY image, the rest the x image ----
  int aoff[LPW];            // dY instructions: byte offset of this lane's unit at stage 0 (OOB: channel tail)
  int brow[LPW], bcol[LPW]; // x instructions: this lane's pixel row of the stage and its channel offset in bytes (-1: tail)
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int q = wave + 4 * i;
    if (q < NA) {
      const int r = q * (64 / UA) + lane / UA, u = (lane % UA) ^ w1_key<RBA>(r);
      aoff[i] = (co0 + u * 8 < p.Cd) ? (int)(((m_begin + r) * p.Cd + co0 + u * 8) * 2) : OOB;
      brow[i] = 0; bcol[i] = 0;
    } else {
      const int r = (q - NA) * (64 / UB) + lane / UB, u = (lane % UB) ^ w1_key<RBB>(r);
      brow[i] = r; bcol[i] = (ci0 + u * 8 < Cs) ? (ci0 + u * 8) * 2 : -1;
      aoff[i] = 0;
    }
  }
  auto issue = [&](int kt, int buf) {
    const unsigned base = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int q = wave + 4 * i;
      if (q < NA) {
        glds16(rs_dy, aoff[i] == OOB ? OOB : aoff[i] + kt * (32 * p.Cd * 2), base + q * 1024);
      } else {
        const long m = m_begin + (long)kt * 32 + brow[i];
        int voff = OOB;
        if (m < m_end && bcol[i] >= 0) {
          const int mi = (int)m;
          int qd = fs_div(mi, dW); int x = mi - qd * p.Wd; int n = fs_div(qd, dH); int y = qd - n * p.Hd;
          voff = (int)((n * p.sN + (long)(y * p.stride) * p.sH + (long)(x * p.stride) * p.sW) * 2) + bcol[i];
        }
        glds16(rs_x, voff, base + IMA + (q - NA) * 1024);
      }
    }
  };

  // ---- transposed fragment reads (16x16x32): lane (li, lg) addresses the 8-byte piece [pixel lg*8 + (li >> 2) (+4)]
  // [channel tile*16 + (li & 3)*4 ..] and receives channel tile*16 + li for pixels lg*8 + 0..3 (+4) ----
  const int r_lo = lg * 8 + (li >> 2);
  const int sub = ((li & 3) & 1) * 8, uq = (li & 3) >> 1;
  const int keyA = w1_key<RBA>(r_lo), keyB = w1_key<RBB>(r_lo);      // (rows r and r + 4 share a key)
  int fa_off[TA], fb_off[TB];
#pragma unroll
  for (int a = 0; a < TA; ++a) fa_off[a] = r_lo * RBA + (((wr * (WROW / 8) + a * 2 + uq) ^ keyA) << 4) + sub;
#pragma unroll
  for (int b = 0; b < TB; ++b) fb_off[b] = IMA + r_lo * RBB + (((wcn * (WCOL / 8) + b * 2 + uq) ^ keyB) << 4) + sub;

  f32x4 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) issue(s, s);
  int cb = 0, ib = NST - 1;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + NST - 2 < nkt) fs_wait_vm<(NST - 2) * LPW>();
    else fs_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nkt) issue(kt + NST - 1, ib);
    const unsigned char* sb = lds + cb * STAGE;
    bf16x8 fa[TA], fb[TB];
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fa_off[a]));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fa_off[a] + 4 * RBA));
      uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      fa[a] = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
    }
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fb_off[b]));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sb + fb_off[b] + 4 * RBB));
      uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      fb[b] = __builtin_bit_cast(bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
    }
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    cb = cb + 1 == NST ? 0 : cb + 1;
    ib = ib + 1 == NST ? 0 : ib + 1;
  }

  // ---- epilogue: D rows = co (lg*4 + j), columns = ci (li) ----
  if (p.nsplit > 1) {
    float* ws = p.workspace + (long)zb * p.ws_rows * p.ws_cols;
#pragma unroll
    for (int b = 0; b < TB; ++b) {
      const int col = ci0 + wcn * WCOL + b * 16 + li;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co0 + wr * WROW + a * 16 + lg * 4 + j;
          ws[(long)co * p.ws_cols + col] = acc[a][b][j];
        }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < TB; ++b) {
    const int ci = ci0 + wcn * WCOL + b * 16 + li;
    if (ci >= p.Ci) continue;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + wr * WROW + a * 16 + lg * 4 + j;
        if (co < p.Co) p.dw[(long)co * p.Ci + ci] += acc[a][b][j];      // sole owner: plain RMW
      }
  }
}

template <int COT, int CIT>
int launch_w1(const FsWgradArgs& a, const FsWgradArgs* a2, hipStream_t st) {
  constexpr int NST = 4;
  FsDual<FsWgradArgs, WgDiv> d;
  FsWgradArgs& b = d.a[0];
  FsWgradArgs& b2 = d.a[1];
  b = a; b2 = a2 ? *a2 : a;
  d.g[0] = WgDiv{fs_make_div(a.Wd), fs_make_div(a.Hd)};
  d.g[1] = WgDiv{fs_make_div(b2.Wd), fs_make_div(b2.Hd)};
  d.nprob = a2 ? 2 : 1;
  const int ncols = a.ncolgroups * 8;
  const int ct = (ncols + CIT - 1) / CIT, rt = (a.Cd + COT - 1) / COT;
  const long tiles = (long)ct * rt;
  const long chunks = (a.M + 31) / 32, chunks2 = a2 ? (a2->M + 31) / 32 : 0;
  // one round of resident blocks (two per CU), >= 256 pixels per block, the slabs inside the caller's workspace
  static const int slots = wg_resident_blocks(wgrad1x1_kernel<COT, CIT, NST>, 256);
  long splits = std::max<long>(1, slots / tiles), splits2 = 0;
  b.ws_rows = rt * COT; b.ws_cols = ct * CIT;
  b2.ws_rows = b.ws_rows; b2.ws_cols = b.ws_cols;
  const long slab = (long)b.ws_rows * b.ws_cols;
  if (a2) {
    wg_share(std::max<long>(2, splits), chunks, chunks2, std::max<long>(1, chunks / 8), std::max<long>(1, chunks2 / 8), splits, splits2);
    const long room = a.workspace ? a.workspace_elems / slab : 0;
    if (room < 2) { splits = 1; splits2 = 1; }
    else if (splits + splits2 > room) { splits = std::max<long>(1, room * splits / (splits + splits2)); splits2 = std::max<long>(1, room - splits); }
  } else {
    splits = std::min<long>(splits, std::max<long>(1, chunks / 8));
    if (!a.workspace) splits = 1;
    else splits = std::min<long>(splits, std::max<long>(1, a.workspace_elems / slab));
  }
  long cps = (chunks + splits - 1) / splits;
  b.pix_per_split = (int)(cps * 32);
  b.nsplit = (int)((chunks + cps - 1) / cps);
  d.nb0 = b.nsplit;
  int nz = b.nsplit;
  if (a2) {
    cps = (chunks2 + splits2 - 1) / splits2;
    b2.pix_per_split = (int)(cps * 32);
    b2.nsplit = (int)((chunks2 + cps - 1) / cps);
    b2.workspace = a.workspace ? a.workspace + (long)b.nsplit * slab : nullptr;
    nz += b2.nsplit;
  }
  if (g_plan) return wg_plan(0, tiles * nz, 256, slots) ? 0 : FS_EINVAL;
  hipLaunchKernelGGL((wgrad1x1_kernel<COT, CIT, NST>), dim3(ct, rt, nz), dim3(256), 0, st, d);
  launch_reduce(b, a2 ? &b2 : nullptr, a.Co, ncols, 8, st);
  return fs_launch_status();
}

// 1x1 / pad 0 layers with whole 64-channel tiles on both sides and enough pixels to pipeline
inline bool w1_takes(const FsWgradArgs& a) {
  const int Cs = a.ncolgroups * 8;
  return a.R == 1 && a.S == 1 && a.pad == 0 && !a.pro_a && a.stride >= 1 && a.Cd % 64 == 0 && Cs % 64 == 0 && a.M >= 2048 &&
         a.x_bytes > 0 && a.x_bytes <= 0x7fffffffLL && (long)a.M * a.Cd * 2 <= 0x7fffffffLL &&
         a.sN % 8 == 0 && a.sH % 8 == 0 && a.sW % 8 == 0 && ((uintptr_t)a.dy | (uintptr_t)a.x) % 16 == 0;
}
inline int launch_wgrad_1x1(const FsWgradArgs& a, const FsWgradArgs* a2, hipStream_t st) {
  const int Cs = a.ncolgroups * 8;
  const bool co128 = a.Cd % 128 == 0, ci128 = Cs % 128 == 0;
  if (co128 && ci128) return launch_w1<128, 128>(a, a2, st);
  if (co128) return launch_w1<128, 64>(a, a2, st);
  if (ci128) return launch_w1<64, 128>(a, a2, st);
  return launch_w1<64, 64>(a, a2, st);
}

// which kernel a problem runs on: 1 stem, 2 3x3 LDS-halo, 3 narrow 3x3, 4 / 5 / 6 / 7 / 8 generic tiles, 9 1x1 LDS-DMA, -1 none
template <typename T>
int wgrad_path(const FsWgradArgs& a) {
  constexpr bool kBf16 = sizeof(T) == 2;  // f32 tiles are capped by the 64 KB static LDS limit
  if constexpr (kBf16) {
    const int Cs = a.ncolgroups * 8 / (a.R * a.S);
    if (a.use_halo && a.R == 7 && a.S == 7 && a.stride == 2 && a.pad == 3 && Cs == 8 && a.Cd == 64 && a.Co == 64 &&
        a.workspace && a.x_bytes > 0 && a.x_bytes <= 0x7fffffffLL && (long)a.M * a.Cd * 2 <= 0x7fffffffLL &&
        a.M >= 4096 && a.M % (a.Hd * a.Wd) == 0)
      return 1;
    if (a.use_halo && a.R == 3 && a.S == 3 && a.stride == 1 && a.Cd % 64 == 0 && Cs % 32 == 0 && a.x_bytes > 0 &&
        a.x_bytes <= 0x7fffffffLL && (long)a.M * a.Cd * 2 <= 0x7fffffffLL && (a.M >= 1024 || a.pro_a))
      return 2;                          // (tiny pixel counts: too few tiles to split, the generic kernel wins)
    if (a.pro_a) return -1;              // only the halo kernel stages x through the prologue
    if (w1_takes(a)) return 9;
    if (a.use_halo && a.R == 3 && a.S == 3 && a.stride == 1 && ((a.Cd == 16 && Cs % 16 == 0) || (a.Cd == 32 && Cs % 32 == 0)) &&
        a.x_bytes > 0 && a.x_bytes <= 0x7fffffffLL && (long)a.M * a.Cd * 2 <= 0x7fffffffLL && a.M >= 4096)
      return 3;
    // few pixels, wide dW (deep stages, pose decoder): 128x128 tiles halve the operand re-fetch per MFMA and
    // put 16 MFMAs per wave between barriers (the 64x64 tile: 4)
    const int ncols = a.ncolgroups * 8;
    if (a.Cd % 128 == 0 && ncols >= 1024 && (long)(a.Cd / 128) * ((ncols + 127) / 128) >= 96) return 4;
    if (a.Cd % 64 == 0 && ncols >= 256) return 5;
  }
  if (a.Cd % 64 == 0) return 6;
  if (a.Cd % 32 == 0) return 7;
  if (a.Cd % 16 == 0) return 8;
  return -1;
}

// everything that selects a kernel, its tiles or the shape of dW must agree; pixel counts, pointers and statistics
// groups of the prologue are per problem
inline bool wgrad_pairable(const FsWgradArgs& a, const FsWgradArgs& b) {
  return a.Hs == b.Hs && a.Ws == b.Ws && a.Hd == b.Hd && a.Wd == b.Wd && a.Cd == b.Cd && a.Co == b.Co && a.Ci == b.Ci &&
         a.R == b.R && a.S == b.S && a.stride == b.stride && a.pad == b.pad && a.ncolgroups == b.ncolgroups &&
         (a.pro_a == nullptr) == (b.pro_a == nullptr) && a.pro_relu == b.pro_relu && a.sH == b.sH && a.sW == b.sW;
}

template <typename T>
int launch_wgrad(const FsWgradArgs& a, const FsWgradArgs* a2, hipStream_t st) {
  constexpr bool kBf16 = sizeof(T) == 2;
  const int path = wgrad_path<T>(a);
  if (a2) {
    // (the stem pairs although the two encoders' Ci differ — 3 and 6 real channels of the same 8-channel pixels: dW is
    // written through each problem's own Ci)
    const bool stem_pair = path == 1 && wgrad_path<T>(*a2) == 1 && a.Hd == a2->Hd && a.Wd == a2->Wd && a.Hs == a2->Hs &&
                           a.Ws == a2->Ws && a.sH == a2->sH && a.sW == a2->sW;
    if (!(stem_pair || (path != 3 && path == wgrad_path<T>(*a2) && wgrad_pairable(a, *a2))) || !a.workspace) {
      const int r = launch_wgrad<T>(a, nullptr, st);
      return r != FS_OK ? r : launch_wgrad<T>(*a2, nullptr, st);
    }
  }
  if constexpr (kBf16) {
    if (path == 1) return launch_wgrad_stem(a, a2, st);
    if (path == 2) return launch_wgrad_halo(a, a2, st);
    if (path == 3) return launch_wgrad_narrow(a, st);
    if (path == 9) return launch_wgrad_1x1(a, a2, st);
    if (path == 4) return launch_tile<T, 128, 128, 2>(a, a2, st);
    if (path == 5) return launch_tile<T, 64, 128, 2>(a, a2, st);
  }
  if (path == 6) return launch_tile<T, 64, 64, 2>(a, a2, st);
  if (path == 7) return launch_tile<T, 32, 128, 1>(a, a2, st);
  if (path == 8) {
    if constexpr (kBf16) return launch_tile<T, 16, 256, 1>(a, a2, st);
    else return launch_tile<T, 16, 128, 1>(a, a2, st);
  }
  return FS_EINVAL;
}

int wgrad_check(const FsWgradArgs* args, int dtype) {
  if (!args || !args->dy || !args->x || !args->dw || !args->ktab) return FS_EINVAL;
  if (args->M <= 0 || args->ncolgroups <= 0) return FS_EINVAL;
  if (args->pro_a && (!args->pro_b || dtype != FS_DTYPE_BF16)) return FS_EINVAL;
  return FS_OK;
}

}  // namespace

// a1 != NULL: the weight gradient of a second convolution of the same shape in the same launch: the pixel splits of the
// two share the device's block slots and the FIRST problem's workspace (slab regions back to back), one reduce launch
// writes both dW.  Problems that do not agree on the kernel run as two launches.
extern "C" int fs_conv_wgrad2(const FsWgradArgs* args, const FsWgradArgs* a1, int dtype, void* stream) {
  int r = wgrad_check(args, dtype);
  if (r != FS_OK) return r;
  if (a1 && (r = wgrad_check(a1, dtype)) != FS_OK) return r;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == FS_DTYPE_BF16) return launch_wgrad<bf16>(*args, a1, st);
  if (dtype == FS_DTYPE_F32) return launch_wgrad<float>(*args, a1, st);
  return FS_EINVAL;
}

extern "C" int fs_conv_wgrad(const FsWgradArgs* args, int dtype, void* stream) {
  return fs_conv_wgrad2(args, nullptr, dtype, stream);
}

extern "C" int fs_conv_wgrad_plan(const FsWgradArgs* args, int dtype, int32_t* plan) {
  if (!args || !plan || args->M <= 0 || args->ncolgroups <= 0) return FS_EINVAL;
  if (args->pro_a && (!args->pro_b || dtype != FS_DTYPE_BF16)) return FS_EINVAL;
  g_plan = plan;
  int r = FS_EINVAL;
  if (dtype == FS_DTYPE_BF16) r = launch_wgrad<bf16>(*args, nullptr, nullptr);
  else if (dtype == FS_DTYPE_F32) r = launch_wgrad<float>(*args, nullptr, nullptr);
  g_plan = nullptr;
  return r;
}

extern "C" int fs_conv_wgrad2_plan(const FsWgradArgs* args, const FsWgradArgs* a1, int dtype, int32_t* plan) {
  if (!plan || wgrad_check(args, dtype) != FS_OK || (a1 && wgrad_check(a1, dtype) != FS_OK)) return FS_EINVAL;
  g_plan = plan;
  int r = FS_EINVAL;
  if (dtype == FS_DTYPE_BF16) r = launch_wgrad<bf16>(*args, a1, nullptr);
  else if (dtype == FS_DTYPE_F32) r = launch_wgrad<float>(*args, a1, nullptr);
  g_plan = nullptr;
  return r;
}
