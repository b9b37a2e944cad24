// Pieces of the 32x32-MFMA-tile 3x3 convolution kernel (conv3x3_t32.hip): operand units,
// 16-byte epilogue accessors, the v_permlane32_swap accumulator transposition and the halving statistics reduction.
#pragma once
#include "common.h"
#include "fsnet_hip_internal.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ uint4 t32_load16(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}

// workgroup barrier that orders LDS traffic only (no vmcnt drain: the next chunk's global loads stay in flight)
__device__ __forceinline__ void t32_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void t32_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct T32Geom {
  int TH, TW;
  int tiles_x, tiles_y;
  unsigned mTW, mHW;
  FsDiv dTX, dTY;
  FsDiv dIPG;      // images per BatchNorm statistics group
  FsDiv dPRG;      // images per prologue coefficient group
  int pix_major;
};

enum : int { EP_BIAS = 1, EP_ADDEND = 2, EP_RELU = 4, EP_MASK = 8, EP_STATS = 16, EP_BNB = 32, EP_F32 = 64,
             EP_MASKBN = 128 };

template <typename T> struct Mma32;
template <> struct Mma32<bf16> {
  static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma32<float> {
  static __device__ __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
    const f32x4 va = __builtin_bit_cast(f32x4, a), vb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j], vb[j], acc, 0, 0, 0);
  }
};

// 8 consecutive channels of one pixel <-> floats (epilogue side)
template <typename T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float* v) {
  Unit<bf16>::unpack(*reinterpret_cast<const uint4*>(p), v);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float* v) {
  *reinterpret_cast<uint4*>(p) = Unit<bf16>::pack(v);
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

#define T32_FLAG(bit, rt) (EP < 0 ? (rt) : ((EP & (bit)) != 0))

constexpr int t32_hmax(int PIX) { return PIX == 256 ? 360 : 208; }
constexpr int t32_lds_units(int PIX, int CO) { return t32_hmax(PIX) * 5 + 9 * CO * 4; }
// blocks per CU the LDS footprint admits (= waves per SIMD: a block is one wave per SIMD)
constexpr int t32_occupancy(int PIX, int CO) {
  return 163840 / (t32_lds_units(PIX, CO) * 16) > 4 ? 4 : 163840 / (t32_lds_units(PIX, CO) * 16);
}

// v_permlane32_swap: lanes 32-63 of `lo_dst` <-> lanes 0-31 of `hi_src` (checked on the device:
// tools/probes/permlane_probe.hip).  Inline asm because hipcc (ROCm 7.2) loses the instruction's second result when
// both come back into elements of an accumulator tuple — it re-uses the tied source register without copying it out
// (seen in the ISA; every second channel quad of the output was garbage).  hipcc pads nothing inside an asm
// statement: the s_nop before covers a VALU write of either operand right in front of it, the one after a read of
// the results right behind.
__device__ __forceinline__ void t32_swap32(float& lo_dst, float& hi_src) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo_dst), "+v"(hi_src));
}
// Sum of 16 per-lane values over the 32 lanes of a wave half, by a halving exchange: in each of four steps a lane
// keeps half of its values, hands the other half to a partner that keeps exactly those, and adds what it receives —
// 8 + 4 + 2 + 1 DPP adds instead of 16 x 5 for sixteen full butterflies (the statistics reduction was 390 of the 553
// vector instructions of an epilogue).  Partners: row_mirror (lane i <-> 15 - i), row_half_mirror (i <-> 7 - i),
// quad_perm xor 2, xor 1 — each flips the selecting lane bit and none of the bits used before it.  A last exchange
// (v_permlane16_swap) adds the two 16-lane rows of the half.  Every lane ends up with the total of ONE input index:
// idx = 8*(lane & 1) + 4*((lane >> 1) & 1) + 2*((lane >> 2) & 1) + ((lane >> 3) & 1).
template <int CTRL>
__device__ __forceinline__ float t32_xadd(float keep, float send) {
  return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTRL, 0xf, 0xf, true));
}
// (the four row-level steps alone: the sum over the 16 lanes of a DPP row)
__device__ __forceinline__ float t32_reduce16_row(const float* v, int lane) {
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
  float w8[8], w4[4], w2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) w8[k] = t32_xadd<0x140>(b3 ? v[2 * k + 1] : v[2 * k], b3 ? v[2 * k] : v[2 * k + 1]);
#pragma unroll
  for (int k = 0; k < 4; ++k) w4[k] = t32_xadd<0x141>(b2 ? w8[2 * k + 1] : w8[2 * k], b2 ? w8[2 * k] : w8[2 * k + 1]);
#pragma unroll
  for (int k = 0; k < 2; ++k) w2[k] = t32_xadd<0x4E>(b1 ? w4[2 * k + 1] : w4[2 * k], b1 ? w4[2 * k] : w4[2 * k + 1]);
  return t32_xadd<0xB1>(b0 ? w2[1] : w2[0], b0 ? w2[0] : w2[1]);
}
__device__ __forceinline__ float t32_reduce16(const float* v, int lane) {
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
  float w8[8], w4[4], w2[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) w8[k] = t32_xadd<0x140>(b3 ? v[2 * k + 1] : v[2 * k], b3 ? v[2 * k] : v[2 * k + 1]);
#pragma unroll
  for (int k = 0; k < 4; ++k) w4[k] = t32_xadd<0x141>(b2 ? w8[2 * k + 1] : w8[2 * k], b2 ? w8[2 * k] : w8[2 * k + 1]);
#pragma unroll
  for (int k = 0; k < 2; ++k) w2[k] = t32_xadd<0x4E>(b1 ? w4[2 * k + 1] : w4[2 * k], b1 ? w4[2 * k] : w4[2 * k + 1]);
  const float w1 = t32_xadd<0xB1>(b0 ? w2[1] : w2[0], b0 ? w2[0] : w2[1]);
  // rows 1, 3 of x <-> rows 0, 2 of y (inline asm for the same reason as t32_swap32)
  float x = w1, y = w1;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return w1 + ((lane & 16) ? x : y);
}

struct T32Item { int n, y0, x0, co0, px; };

// scheduling groups of one pipelined k-step: NR fragment reads spread evenly between its NM MFMAs
template <int NR, int NM, int K = 0>
__device__ __forceinline__ void t32_sched_step() {
  if constexpr (K < NR) {
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    constexpr int m = (NM * (K + 1)) / NR - (NM * K) / NR;
    if constexpr (m > 0) __builtin_amdgcn_sched_group_barrier(0x008, m, 0);
    t32_sched_step<NR, NM, K + 1>();
  }
}

// pixel tile (TH x TW <= PIX, halo <= hmax) that wastes the fewest MFMA lanes; wide tiles preferred (a 32-pixel
// MFMA tile that is one image row reads its halo conflict-free)
inline T32Geom t32_pick_geom(int Hd, int Wd, int PIX, int hmax) {
  T32Geom best{};
  double best_cost = 1e30;
  for (int tw = std::min(4, Wd); tw <= std::min(Wd, 64); ++tw) {
    const int th = std::min(PIX / tw, Hd);
    if (th < 1 || (th + 2) * (tw + 2) > hmax) continue;
    const int tx = (Wd + tw - 1) / tw, ty = (Hd + th - 1) / th;
    const double waste = (double)tx * ty * PIX / ((double)Hd * Wd);
    const double halo = (double)(th + 2) * (tw + 2) / ((double)th * tw);
    double cost = waste * (1.0 + 0.15 * halo);
    if (tw % 32 != 0 && tw != Wd) cost *= 1.02;
    if (cost < best_cost - 1e-9) { best_cost = cost; best.TH = th; best.TW = tw; best.tiles_x = tx; best.tiles_y = ty; }
  }
  if (best.TW > 0) {
    best.mTW = fs_div_magic(best.TW); best.mHW = fs_div_magic(best.TW + 2);
    best.dTX = fs_make_div(best.tiles_x); best.dTY = fs_make_div(best.tiles_y);
  }
  return best;
}

inline int t32_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
    }
    return cus;
  }();
  return n;
}

inline int t32_ep_mask(const FsConvArgs& a) {
  int m = 0;
  if (a.bias) m |= EP_BIAS;
  if (a.addend) m |= EP_ADDEND;
  if (a.relu) m |= EP_RELU;
  if (a.mask) m |= EP_MASK;
  if (a.bnb_x) m |= EP_BNB;
  else if (a.stats) m |= EP_STATS;
  if (a.out_f32) m |= EP_F32;
  if (a.bnb_x && a.bnb_scale) m |= EP_MASKBN;
  return m;
}

}  // namespace
