// Shared device helpers for the fsnet_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FS_WAVE 64

// status codes returned by every C-ABI entry point
#define FS_OK 0
#define FS_EINVAL 1
#define FS_ELAUNCH 2

static inline int fs_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FS_OK : FS_ELAUNCH;
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int EG = 4;  // elements per 16-byte group
  __device__ static inline float to_f(float v) { return v; }
  __device__ static inline float from_f(float v) { return v; }
};
template <> struct ElemTraits<bf16> {
  static constexpr int EG = 8;
  __device__ static inline float to_f(bf16 v) { return (float)v; }
  __device__ static inline bf16 from_f(float v) { return (bf16)v; }
};

__device__ static inline float bf16_bits_to_f(uint32_t b16) { return __uint_as_float(b16 << 16); }
// round-to-nearest-even fp32 -> bf16 bits (NaN not expected on this path)
__device__ static inline uint32_t f_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// two fp32 -> packed bf16 pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 (the shift/add/mask
// sequence above costs ~10 VALU per pair, and the conv epilogues convert 32 outputs per lane)
__device__ static inline uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// load / store `n<=4` consecutive channel values as float, for either element type
template <typename T> __device__ inline void load4(const T* p, float v[4]);
template <> __device__ inline void load4<float>(const float* p, float v[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ inline void load4<bf16>(const bf16* p, float v[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = bf16_bits_to_f(t.x & 0xffffu); v[1] = bf16_bits_to_f(t.x >> 16);
  v[2] = bf16_bits_to_f(t.y & 0xffffu); v[3] = bf16_bits_to_f(t.y >> 16);
}
template <typename T> __device__ inline void store4(T* p, const float v[4]);
template <> __device__ inline void store4<float>(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ inline void store4<bf16>(bf16* p, const float v[4]) {
  uint2 t; t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = t;
}

// x / d for 0 <= x < 4096, 1 <= d <= 255 with a host-computed magic: the tile kernels decode (row, column) of ~10
// tile-relative indices per thread by run-time tile widths, and an integer division is ~35 VALU instructions on CDNA
__host__ __device__ static inline unsigned fs_div_magic(int d) { return (1u << 20) / (unsigned)d + 1u; }
__device__ static inline int fs_fastdiv(int x, unsigned magic) { return (int)(((unsigned)x * magic) >> 20); }

// x / d for 0 <= x < 2^31 and 1 <= d < 2^31: m = floor(2^(31+l) / d) + 1 with l = ceil(log2 d) (m < 2^32), quotient
// (x * m) >> (31 + l) — exact because m*d - 2^(31+l) <= d and x*d < 2^(31+l).  Two multiplies and a shift instead of
// the ~35-instruction division sequence.
struct FsDiv { unsigned m; unsigned sh; };
__host__ static inline FsDiv fs_make_div(int d) {
  int l = 0;
  while ((1LL << l) < (long long)d) ++l;
  FsDiv r;
  r.m = (unsigned)(((1ULL << (31 + l)) / (unsigned long long)d) + 1ULL);
  r.sh = (unsigned)(31 + l);
  return r;
}
__device__ static inline int fs_div(int x, FsDiv d) {
  return (int)(((unsigned long long)(unsigned)x * d.m) >> d.sh);
}

// Wave-wide sums without LDS traffic.  __shfl_xor lowers to ds_bpermute_b32 (an LDS-pipeline instruction plus its
// address VALU op, six per sum); here four DPP adds fold each 16-lane row in the VALU — lane ^ 1, lane ^ 2 by
// quad_perm, then row_half_mirror and row_mirror, which pair a lane with one holding the other half's partial — and
// the four row totals are read through SGPRs.  Every lane returns the same value; call with all 64 lanes active.
template <int CTRL>
__device__ static inline float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ static inline double dpp_mov(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ static inline float read_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ static inline double read_lane(double v, int lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <typename F>
__device__ static inline F row16_sum(F v) {      // every lane: the sum over its 16-lane DPP row
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
__device__ static inline float wave_sum(float v) {
  v = row16_sum(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}
__device__ static inline double wave_sum_d(double v) {
  v = row16_sum(v);
  return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// block-wide sum (blockDim.x == 256): result valid in thread 0.  `sh` = 4 doubles of LDS.
__device__ static inline double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// 16-byte per-lane vectors: VecN<T>::N consecutive channel values (4 f32 / 8 bf16) as floats
template <typename T> struct VecN { static constexpr int N = 16 / sizeof(T); };
template <typename T> __device__ inline void loadv(const T* p, float* v);
template <> __device__ inline void loadv<float>(const float* p, float* v) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ inline void loadv<bf16>(const bf16* p, float* v) {
  uint4 t = *reinterpret_cast<const uint4*>(p);
  v[0] = bf16_bits_to_f(t.x & 0xffffu); v[1] = bf16_bits_to_f(t.x >> 16);
  v[2] = bf16_bits_to_f(t.y & 0xffffu); v[3] = bf16_bits_to_f(t.y >> 16);
  v[4] = bf16_bits_to_f(t.z & 0xffffu); v[5] = bf16_bits_to_f(t.z >> 16);
  v[6] = bf16_bits_to_f(t.w & 0xffffu); v[7] = bf16_bits_to_f(t.w >> 16);
}
template <typename T> __device__ inline void storev(T* p, const float* v);
template <> __device__ inline void storev<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ inline void storev<bf16>(bf16* p, const float* v) {
  uint4 t;
  t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]);
  t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = t;
}

// ---- 16-byte operand units as floats and back ----
template <typename T> struct Unit;
template <> struct Unit<bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4& u, float* v) {
    v[0] = bf16_bits_to_f(u.x & 0xffffu); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = bf16_bits_to_f(u.y & 0xffffu); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = bf16_bits_to_f(u.z & 0xffffu); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = bf16_bits_to_f(u.w & 0xffffu); v[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  static __device__ __forceinline__ uint4 pack(const float* v) {
    return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
};
template <> struct Unit<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4& u, float* v) {
    v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* v) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
  }
};
