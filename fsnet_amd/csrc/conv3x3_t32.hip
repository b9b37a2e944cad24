// 3x3 / stride-1 convolution (forward and data gradient) on 32x32 MFMA tiles — the round-3 rebuild of the LDS-halo
// kernel (conv3x3_halo.hip keeps the 16-channel decoder layers).  Same call sites, arguments and packed weights.
//
// What changed against the 16x16x32 kernel, and why (DESIGN section 7 has the measurements that asked for it):
//   * v_mfma_f32_32x32x16_bf16 (fp32 compute: v_mfma_f32_32x32x2_f32): a wave owns (PIX/4) pixels x CO channels as
//     TP x TC tiles of 32 x 32.  One ds_read_b128 feeds 32 rows x 16 k, so a 64 x 64 wave tile reads 1/2 LDS fragment
//     per 32 KFLOP-MFMA where the 2 x 2 tiling of 16 x 16 tiles read one per 16 KFLOP — a quarter of the LDS bytes
//     per FLOP; the old loop ran at 27 % of the matrix peak with the LDS array as co-limiter.
//   * two LDS stage buffers, ONE workgroup barrier per 64-byte channel chunk (was one buffer, two barriers).
//   * the epilogue goes through LDS: accumulators are written tile-wise as fp32 [pixel][32 channels] into the wave's
//     own region and read back as 8 consecutive channels of one pixel per lane, so addend / mask / BatchNorm-input
//     loads and the output stores are 16 bytes per lane and whole 64-byte runs per pixel (were 8-byte pieces: the
//     epilogue was a third of the kernel), and the per-channel statistics fold over 16 lanes with 8 channels each.
//   * epilogue options are template flags for the combinations the networks use (EP >= 0), run-time tests otherwise.
//   * operand prologue (FsConvArgs.pro_mode): the BatchNorm + ReLU in front of the convolution, or the second pass of
//     its backward in front of a data gradient, is applied while the halo is staged — see fsnet_hip.h.
//   * a second weight operand for the images from wgt2_from_n on (depth + pose encoder in one launch).
// Reference call sites: vision_base/networks/models/backbone/resnet.py:21-50 (BasicBlock), blocks.py:41-54,
// monodepth/networks/models/heads/depth_encoder.py:45-63, pose_decoder.py:17-37.
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
  int nitems;      // item ids to walk (pixel tiles x channel tiles, padded by the XCD mapping)
  int abl;         // development: ablation bits (1 no MFMA loop, 2 no epilogue, 4 no global operand loads, 8 no weight loads)
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
// sum over the 32 lanes of a wave half: valid in lanes 16-31 (lower half) and 48-63 (upper half)
__device__ __forceinline__ float t32_half_sum(float v) {
  v = row16_sum(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));
  return v;
}

struct T32Item { int n, y0, x0, co0, px; };

// Persistent blocks: block b works on the items b, b + gridDim.x, ... (item = pixel tile x channel tile).  While the
// last channel chunk of an item is multiplied, the first chunk of the block's NEXT item is already being fetched into
// registers, and the epilogue (global loads / stores only, no LDS) runs with those loads in flight — measured on the
// one-item-per-block version: operand staging, MFMA loop and epilogue simply added up (16 + 9.5 + 9.2 us on the
// 64 -> 64 layer at 36 images), every block of a launch being in the same phase at the same time.
// (the run-time-flag instantiations — fp32 and the rare epilogue combinations — get the whole register file: they
// would spill at the occupancy-derived register cap)
constexpr int t32_minwaves(int PIX, int CO, int EP, int PRO) {
  return EP < 0 ? 1 : (t32_occupancy(PIX, CO) - (PRO != 0 ? 1 : 0) < 1 ? 1 : t32_occupancy(PIX, CO) - (PRO != 0 ? 1 : 0));
}

template <typename T, int PIX, int CO, int EP, int PRO>
__global__ __launch_bounds__(256, t32_minwaves(PIX, CO, EP, PRO)) void conv3x3_t32_kernel(const FsConvArgs p, const T32Geom g) {
  constexpr int WPIX = PIX / 4;                 // pixels per wave
  constexpr int TP = WPIX / 32, TC = CO / 32;   // 32 x 32 MFMA tiles per wave: pixels x channels
  constexpr int HMAX = t32_hmax(PIX);           // halo pixels per stage
  constexpr int HS = 5;                         // 16-byte units per halo pixel: 4 used + 1 pad (see below)
  constexpr int LH = (HMAX * 4 + 255) / 256;
  constexpr int WU = 9 * CO * 4;                // weight units per stage
  constexpr int LW = (WU + 255) / 256;
  constexpr int OOB = 0x7fffffff;
  constexpr int UN = Unit<T>::N;                // elements per unit

  // Bank layout.  ds_read_b128 is serviced in four groups of 16 lanes — {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
  // same + 32 (MI355X_MICROARCH.md, LDS).  A 32x32 MFMA operand has lane l read row (l & 31), 16-byte k-slot
  // 2*ks + (l >> 5).  Halo pixels: consecutive pixels 5 units apart — 5 is odd, so the 16 rows of a group (all residues
  // mod 16) land on 16 distinct 16-byte bank slots at ANY tap shift, and the shift stays an immediate offset.
  // Weights: 4 units per row, slot XOR ((row >> 2) & 3): rows that share (row & 3) — the same 64-byte quarter of the
  // 256-byte bank row — differ in (row >> 2) & 3 within a group.
  __shared__ uint4 lds[t32_lds_units(PIX, CO)];
  uint4* const lds_w = lds + HMAX * HS;
  // [4 waves][CO][2] statistics partials live in the halo rows' padding units (every fifth unit, never staged)
  static_assert(4 * CO * 2 <= HMAX * 4, "statistics partials must fit the halo padding");
  auto red = [&](int f) -> float& { return reinterpret_cast<float*>(lds + (f >> 2) * HS + 4)[f & 3]; };

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, hk = lane >> 5;
  const int HW = g.TW + 2, HH = g.TH + 2;
  const int nhalo = HH * HW;
  const int ntile = g.TH * g.TW;
  const int fwd = p.sgn > 0;
  const int row_bytes = p.Cs * (int)sizeof(T);
  const int nchunk = (row_bytes + 63) / 64;
  const int q4 = t & 3;

  // ---- item decoding (XCD-aware, as conv3x3_halo.hip) ----
  const int npix = p.N * g.tiles_y * g.tiles_x, nco = p.Co_p / CO;
  const int nitems = g.nitems;
  auto decode = [&](int id, T32Item& it) -> bool {
    int px, cy;
    const int xcd = id & 7, slot = id >> 3;
    if (g.pix_major) { cy = slot % nco; px = (slot / nco) * 8 + xcd; }
    else if (nco % 8 == 0) { const int q = nco >> 3; cy = xcd + 8 * (slot % q); px = slot / q; }
    else if (8 % nco == 0) { const int q = 8 / nco; cy = xcd % nco; px = slot * q + xcd / nco; }
    else { cy = id % nco; px = id / nco; }
    if (px >= npix) return false;
    const int tq = fs_div(px, g.dTX); const int tx_i = px - tq * g.tiles_x;
    it.n = fs_div(tq, g.dTY); const int ty_i = tq - it.n * g.tiles_y;
    it.y0 = ty_i * g.TH; it.x0 = tx_i * g.TW; it.co0 = cy * CO; it.px = px;
    return true;
  };

  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_src2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(PRO == 2 ? p.pro_src2 : p.src), 0, (int)p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt), 0, (int)p.wgt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wgt2 ? p.wgt2 : p.wgt), 0, (int)p.wgt_bytes, 0x00020000);

  // ---- per-thread load units: thread t always holds 16-byte slot q = t & 3 of halo pixel (t >> 2) + 64 i and of
  // weight row (t >> 2) + 64 i (= the same channel row, 64 / CO taps further on) ----
  const int wrow_bytes = p.nchunks * p.kg * 16;
  const int wrow0 = (t >> 2) % CO, wtap0 = (t >> 2) / CO;
  const int wrel0 = q4 * 16 < row_bytes ? wrow0 * wrow_bytes + wtap0 * row_bytes + q4 * 16 : OOB;
  // fragment bases
  int hbase[TP], ety[TP], etx[TP];
#pragma unroll
  for (int b = 0; b < TP; ++b) {
    const int pi = wave * WPIX + b * 32 + l31;
    const int pv = pi < ntile ? pi : 0;              // padding lanes read a valid halo row; results are discarded
    const int ty = fs_fastdiv(pv, g.mTW), tx = pv - ty * g.TW;
    hbase[b] = (ty * HW + tx) * HS + hk;
    ety[b] = pi < ntile ? ty : -1; etx[b] = tx;
  }
  const int we = hk ^ ((l31 >> 2) & 3);
  const int wa0 = l31 * 4 + we, wa1 = l31 * 4 + (we ^ 2);

  // ---- stage state: offsets of the stage being fetched, registers in flight ----
  int hvoff[LH], wbase = 0, wsel = 0, pgo = 0;
  uint4 rh[LH], rw[LW];
  uint4 rh2[PRO == 2 ? LH : 1];
  float ka[PRO != 0 ? UN : 1], kb[PRO != 0 ? UN : 1], kc[PRO == 2 ? UN : 1];
  auto setup = [&](const T32Item& it) {
    const int oy = it.y0 + p.hb_add + (fwd ? 0 : -2), ox = it.x0 + p.hb_add + (fwd ? 0 : -2);
    const int base = (int)(((long)it.n * p.sN + (long)oy * p.sH + (long)ox * p.sW) * (long)sizeof(T)) + q4 * 16;
#pragma unroll
    for (int i = 0; i < LH; ++i) {
      const int hp = (t >> 2) + i * 64;
      const int hy = fs_fastdiv(hp, g.mHW), hx = hp - hy * HW;
      const int sy = oy + hy, sx = ox + hx;
      const bool ok = hp < nhalo && q4 * 16 < row_bytes && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      hvoff[i] = ok ? base + (hy * (int)p.sH + hx * (int)p.sW) * (int)sizeof(T) : OOB;
    }
    wbase = it.co0 * wrow_bytes;
    wsel = (p.wgt2 != nullptr && it.n >= p.wgt2_from_n) ? 1 : 0;
    if constexpr (PRO != 0) pgo = p.pro_group_imgs > 0 ? fs_div(it.n, g.dPRG) * p.Cs : 0;
  };
  auto load_regs = [&](int cc) {
    const int coff = (g.abl & 4) ? OOB : cc * 64;
#pragma unroll
    for (int i = 0; i < LH; ++i) rh[i] = t32_load16(rs_src, hvoff[i] == OOB ? OOB : hvoff[i] + coff);
    if constexpr (PRO == 2) {
#pragma unroll
      for (int i = 0; i < LH; ++i) rh2[i] = t32_load16(rs_src2, hvoff[i] == OOB ? OOB : hvoff[i] + coff);
    }
    const int wo = (wrel0 == OOB || (g.abl & 8)) ? OOB : wrel0 + wbase + coff;
    if (wsel) {
#pragma unroll
      for (int i = 0; i < LW; ++i)
        rw[i] = t32_load16(rs_wgt2, (wo == OOB || (t >> 2) + i * 64 >= 9 * CO) ? OOB : wo + i * (64 / CO) * row_bytes);
    } else {
#pragma unroll
      for (int i = 0; i < LW; ++i)
        rw[i] = t32_load16(rs_wgt, (wo == OOB || (t >> 2) + i * 64 >= 9 * CO) ? OOB : wo + i * (64 / CO) * row_bytes);
    }
    if constexpr (PRO != 0) {
      const int c0r = cc * (64 / (int)sizeof(T)) + q4 * UN;
      const bool cok = c0r < p.Cs;
      const int c0 = cok ? c0r : 0;                  // (always a valid address: the select happens on the values)
#pragma unroll
      for (int j = 0; j < UN; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(p.pro_a + pgo + c0 + j);
        const float4 b = *reinterpret_cast<const float4*>(p.pro_b + pgo + c0 + j);
        ka[j] = cok ? a.x : 0.f; ka[j + 1] = cok ? a.y : 0.f; ka[j + 2] = cok ? a.z : 0.f; ka[j + 3] = cok ? a.w : 0.f;
        kb[j] = cok ? b.x : 0.f; kb[j + 1] = cok ? b.y : 0.f; kb[j + 2] = cok ? b.z : 0.f; kb[j + 3] = cok ? b.w : 0.f;
        if constexpr (PRO == 2) {
          const float4 c = *reinterpret_cast<const float4*>(p.pro_c + pgo + c0 + j);
          kc[j] = cok ? c.x : 0.f; kc[j + 1] = cok ? c.y : 0.f; kc[j + 2] = cok ? c.z : 0.f; kc[j + 3] = cok ? c.w : 0.f;
        }
      }
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < LH; ++i) {
      const int hp = (t + i * 256) >> 2;
      uint4 u = rh[i];
      if constexpr (PRO == 1) {
        float v[UN];
        Unit<T>::unpack(u, v);
#pragma unroll
        for (int j = 0; j < UN; ++j) {
          v[j] = v[j] * ka[j] + kb[j];
          if (p.pro_relu) v[j] = fmaxf(v[j], 0.f);
        }
        u = Unit<T>::pack(v);
        if (hvoff[i] == OOB) u = make_uint4(0u, 0u, 0u, 0u);     // padding applies to the transformed tensor
      }
      if constexpr (PRO == 2) {
        float v[UN], w[UN];
        Unit<T>::unpack(u, v);
        Unit<T>::unpack(rh2[i], w);
#pragma unroll
        for (int j = 0; j < UN; ++j) v[j] = v[j] * ka[j] + (w[j] * kb[j] + kc[j]);
        u = Unit<T>::pack(v);
        if (hvoff[i] == OOB) u = make_uint4(0u, 0u, 0u, 0u);
      }
      if (hp < HMAX) lds[hp * HS + q4] = u;
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int rt = (t + i * 256) >> 2;
      if (rt < 9 * CO) lds_w[rt * 4 + (q4 ^ ((rt >> 2) & 3))] = rw[i];     // CO % 32 == 0: the swizzle follows the co row
    }
  };

  const bool has_bias = T32_FLAG(EP_BIAS, p.bias != nullptr);
  const bool has_add = T32_FLAG(EP_ADDEND, p.addend != nullptr);
  const bool has_relu = T32_FLAG(EP_RELU, p.relu != 0);
  const bool has_mask = T32_FLAG(EP_MASK, p.mask != nullptr);
  const bool has_bnb = T32_FLAG(EP_BNB, p.bnb_x != nullptr);
  const bool has_stats = T32_FLAG(EP_STATS | EP_BNB, p.stats != nullptr);
  const bool has_mbn = T32_FLAG(EP_MASKBN, p.bnb_scale != nullptr);
  const bool f32out = T32_FLAG(EP_F32, p.out_f32 != 0);

  // statistics of the item whose epilogue ran last: partial sums sit in `red`, added to the f64 slots after the
  // next workgroup barrier
  int pend_px = -1, pend_n = 0, pend_co0 = 0;
  auto flush_stats = [&]() {
    if (pend_px >= 0 && t < CO) {
      float u = 0.f, w = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { u += red((k * CO + t) * 2); w += red((k * CO + t) * 2 + 1); }
      const int co = pend_co0 + t;
      if (co < p.Co) {
        const long sg = p.stat_group_rows > 0 ? fs_div(pend_n, g.dIPG) : 0;
        double* sl = p.stats + (sg * FS_STAT_SLOTS + pend_px % FS_STAT_SLOTS) * 2 * p.Co;
        double wd = (double)w;
        if (has_bnb) wd = (wd - (double)p.bnb_mean[sg * p.Co + co] * (double)u) * (double)p.bnb_invstd[sg * p.Co + co];
        atomicAdd(sl + co, (double)u);
        atomicAdd(sl + p.Co + co, wd);
      }
    }
    pend_px = -1;
  };

  int id = blockIdx.x;
  T32Item cur, nxt;
  while (id < nitems && !decode(id, cur)) id += gridDim.x;
  if (id >= nitems) return;
  setup(cur);
  load_regs(0);

  for (;;) {
    f32x16 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
      for (int b = 0; b < TP; ++b)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
    int next = nitems;
    for (int cc = 0; cc < nchunk; ++cc) {
      t32_barrier();                 // previous stage fully multiplied (and the previous item's `red` complete)
      flush_stats();
      store_lds();
      t32_barrier();
      if (cc + 1 < nchunk) load_regs(cc + 1);
      else {
        next = id + gridDim.x;
        while (next < nitems && !decode(next, nxt)) next += gridDim.x;
        if (next < nitems) { setup(nxt); load_regs(0); }
      }
      if (g.abl & 1) continue;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int r = tap / 3, s = tap - r * 3;
        const int hoff = (fwd ? (r * HW + s) : ((2 - r) * HW + (2 - s))) * HS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          uint4 fa[TC], fb[TP];
#pragma unroll
          for (int a = 0; a < TC; ++a) fa[a] = lds_w[(ks ? wa1 : wa0) + (tap * CO + a * 32) * 4];
#pragma unroll
          for (int b = 0; b < TP; ++b) fb[b] = lds[hbase[b] + hoff + 2 * ks];
#pragma unroll
          for (int a = 0; a < TC; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) Mma32<T>::run(acc[a][b], fa[a], fb[b]);
        }
      }
    }

    // ---- epilogue of `cur`: no LDS.  A 32x32 accumulator tile has lane (pixel l31, half hk) hold channels
    // 8 g + 4 hk + (0..3), g = 0..3; two v_permlane32_swap rounds turn that into two runs of 8 consecutive channels
    // (8 hk + 0..7 and 16 + 8 hk + 0..7), so every load and store below is 16 bytes per lane ----
    if (!(g.abl & 2)) {
      const int sgoff = p.stat_group_rows > 0 ? fs_div(cur.n, g.dIPG) * p.Co : 0;
      // the swaps below read MFMA results from inline asm, where the compiler's hazard recogniser inserts nothing:
      // 20 wait states behind the last MFMA of every accumulator tile (a 16-pass MFMA needs 18 before a VALU read)
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[a][b][0]), "+v"(acc[a][b][15]));
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float lo = acc[a][b][i], hi = acc[a][b][4 + i];
            t32_swap32(lo, hi);
            acc[a][b][i] = lo; acc[a][b][4 + i] = hi;
            lo = acc[a][b][8 + i]; hi = acc[a][b][12 + i];
            t32_swap32(lo, hi);
            acc[a][b][8 + i] = lo; acc[a][b][12 + i] = hi;
          }
#pragma unroll
      for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int rn = 0; rn < 2; ++rn) {
          const int co = cur.co0 + a * 32 + rn * 16 + hk * 8;
          const bool cok = co < p.Co;                          // (Co % 8 == 0 on this path)
          const int cof = cok ? co : 0;
          float bv[8], msc[8], msh[8], s1[8], s2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { bv[j] = 0.f; msc[j] = 0.f; msh[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; }
          if (has_bias) load8<float>(p.bias + cof, bv);
          if (has_mbn) { load8<float>(p.bnb_scale + sgoff + cof, msc); load8<float>(p.bnb_shift + sgoff + cof, msh); }
#pragma unroll
          for (int b = 0; b < TP; ++b) {
            const int y = cur.y0 + ety[b], x = cur.x0 + etx[b];
            if (!(ety[b] >= 0 && y < p.Hd && x < p.Wd && cok)) continue;
            const int doff = cur.n * (int)p.dN + y * (int)p.dH + x * (int)p.dW;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = acc[a][b][rn * 8 + j] + bv[j];
            if (has_add) {
              float av[8];
              load8<T>(reinterpret_cast<const T*>(p.addend) + cur.n * (int)p.aN + y * (int)p.aH + x * (int)p.aW + co, av);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += av[j];
            }
            if (has_relu) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (has_mask) {
              float mv[8];
              load8<T>(reinterpret_cast<const T*>(p.mask) + cur.n * (int)p.mN + y * (int)p.mH + x * (int)p.mW + co, mv);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
            }
            if (has_bnb) {
              // BatchNorm-backward sums: (sum g, sum g*x) here; sum g*xhat = (sum g*x - mean * sum g) * invstd is
              // formed in f64 when the block's partials are flushed
              float cv[8];
              load8<T>(reinterpret_cast<const T*>(p.bnb_x) + doff + co, cv);       // same layout as dst
              if (has_mbn) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (cv[j] * msc[j] + msh[j]) > 0.f ? v[j] : 0.f;
              }
#pragma unroll
              for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * cv[j]; }
            } else if (has_stats) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
            }
            if (f32out) store8<float>(reinterpret_cast<float*>(p.dst) + doff + co, v);
            else store8<T>(reinterpret_cast<T*>(p.dst) + doff + co, v);
          }
          if (has_stats) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float u = t32_half_sum(s1[j]), w = t32_half_sum(s2[j]);
              if (l31 == 31) {
                const int cl = a * 32 + rn * 16 + hk * 8 + j;
                red((wave * CO + cl) * 2) = u; red((wave * CO + cl) * 2 + 1) = w;
              }
            }
          }
        }
      if (has_stats) { pend_px = cur.px; pend_n = cur.n; pend_co0 = cur.co0; }
    } else if (acc[0][0][0] == 123.456f) p.stats[0] = 1.0;
    if (next >= nitems) break;
    cur = nxt; id = next;
  }
  if (has_stats) {
    t32_barrier();
    flush_stats();
  }
}

// pixel tile (TH x TW <= PIX, halo <= hmax) that wastes the fewest MFMA lanes; wide tiles preferred (a 32-pixel
// MFMA tile that is one image row reads its halo conflict-free)
T32Geom t32_pick_geom(int Hd, int Wd, int PIX, int hmax) {
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

static const bool kNoPixMajor = [] { const char* e = getenv("FSNET_AMD_HALO_PIXMAJOR"); return e && e[0] == '0'; }();

int t32_cu_count() {
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

template <typename T, int PIX, int CO, int EP, int PRO>
int t32_launch(const FsConvArgs& a, hipStream_t st) {
  T32Geom g = t32_pick_geom(a.Hd, a.Wd, PIX, t32_hmax(PIX));
  if (g.TH == 0) return FS_EINVAL;
  g.dIPG = FsDiv{0u, 0u}; g.dPRG = FsDiv{0u, 0u};
  { const char* ae = getenv("FSNET_AMD_T32_ABL"); g.abl = ae ? atoi(ae) : 0; }
  if (a.stat_group_rows > 0) {
    const long hw = (long)a.Hd * a.Wd;
    if (a.stat_group_rows % hw != 0) return FS_EINVAL;
    g.dIPG = fs_make_div((int)(a.stat_group_rows / hw));
  }
  if (a.pro_group_imgs > 0) g.dPRG = fs_make_div(a.pro_group_imgs);
  const int npix = a.N * g.tiles_x * g.tiles_y, nco = a.Co_p / CO;
  int items = npix * nco;
  g.pix_major = (nco > 1 && a.src_bytes > 2 * a.wgt_bytes && !kNoPixMajor) ? 1 : 0;
  if (g.pix_major) items = 8 * ((npix + 7) / 8) * nco;
  else if (nco % 8 != 0 && 8 % nco == 0) { const int q = 8 / nco; items = 8 * ((npix + q - 1) / q); }
  g.nitems = items;
  // persistent grid: what the chip holds at once, in whole multiples of 8 * nco so that a block's stride keeps its
  // XCD and (with the pixel-major map) its channel tile
  const char* pe = getenv("FSNET_AMD_T32_PERSIST");
  const int persist = pe ? atoi(pe) : 1;
  int blocks = items;
  if (persist > 0) {
    const int unit = 8 * nco;
    int cap = t32_cu_count() * t32_occupancy(PIX, CO) * persist;
    cap = std::max(unit, cap / unit * unit);
    blocks = std::min(items, cap);
  }
  hipLaunchKernelGGL((conv3x3_t32_kernel<T, PIX, CO, EP, PRO>), dim3(blocks), dim3(256), 0, st, a, g);
  return fs_launch_status();
}

int t32_ep_mask(const FsConvArgs& a) {
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

// tile configuration (pixels x channels per block; blocks per CU by LDS): 0 = 256 x 64 (2), 1 = 128 x 64 (3),
// 2 = 128 x 32 (4), 3 = 256 x 32 (3)
template <typename T, int EP, int PRO>
int t32_dispatch_cfg(const FsConvArgs& a, int cfg, hipStream_t st) {
  switch (cfg) {
    case 0: return t32_launch<T, 256, 64, EP, PRO>(a, st);
    case 1: return t32_launch<T, 128, 64, EP, PRO>(a, st);
    case 2: return t32_launch<T, 128, 32, EP, PRO>(a, st);
    default: return t32_launch<T, 256, 32, EP, PRO>(a, st);
  }
}

long t32_blocks(const FsConvArgs& a, int PIX, int CO) {
  T32Geom g = t32_pick_geom(a.Hd, a.Wd, PIX, PIX == 256 ? 360 : 208);
  if (g.TH == 0) return 0;
  return (long)a.N * g.tiles_x * g.tiles_y * (a.Co_p / CO);
}
double t32_waste(const FsConvArgs& a, int PIX) {
  T32Geom g = t32_pick_geom(a.Hd, a.Wd, PIX, PIX == 256 ? 360 : 208);
  if (g.TH == 0) return 1e9;
  return (double)g.tiles_x * g.tiles_y * PIX / ((double)a.Hd * a.Wd);
}

int t32_pick_cfg(const FsConvArgs& a) {
  // (development knob, read per call: tools/probes/t32_bench.py sweeps the tile configurations in one process)
  const char* fe = getenv("FSNET_AMD_T32_CFG");
  const int kForceCfg = fe ? atoi(fe) : -1;
  if (kForceCfg >= 0) return (a.Co_p % 64 != 0 && (kForceCfg == 0 || kForceCfg == 1)) ? (kForceCfg == 0 ? 3 : 2) : kForceCfg;
  const bool c64 = a.Co_p % 64 == 0;
  // a 256-pixel tile only where it does not waste lanes (6 x 20 images fill 128-pixel tiles) and still leaves the chip
  // a wave of blocks
  const bool big = t32_waste(a, 256) <= 1.15 * t32_waste(a, 128);
  if (c64) {
    if (big && t32_blocks(a, 256, 64) >= 256) return 0;
    if (t32_blocks(a, 128, 64) >= 384) return 1;
    return 2;
  }
  if (big && t32_blocks(a, 256, 32) >= 512) return 3;
  return 2;
}

template <typename T, int PRO>
int t32_dispatch_ep(const FsConvArgs& a, int cfg, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    switch (t32_ep_mask(a)) {
      // forward: encoder (statistics), decoder (bias + statistics), pose decoder (bias + ReLU)
      case EP_STATS: return t32_dispatch_cfg<T, EP_STATS, PRO>(a, cfg, st);
      case EP_BIAS | EP_STATS: return t32_dispatch_cfg<T, EP_BIAS | EP_STATS, PRO>(a, cfg, st);
      case EP_BIAS | EP_RELU: return t32_dispatch_cfg<T, EP_BIAS | EP_RELU, PRO>(a, cfg, st);
      // data gradients
      case 0: return t32_dispatch_cfg<T, 0, PRO>(a, cfg, st);
      case EP_ADDEND: return t32_dispatch_cfg<T, EP_ADDEND, PRO>(a, cfg, st);
      case EP_MASK: return t32_dispatch_cfg<T, EP_MASK, PRO>(a, cfg, st);
      case EP_MASK | EP_BNB: return t32_dispatch_cfg<T, EP_MASK | EP_BNB, PRO>(a, cfg, st);
      case EP_ADDEND | EP_MASK | EP_BNB: return t32_dispatch_cfg<T, EP_ADDEND | EP_MASK | EP_BNB, PRO>(a, cfg, st);
      case EP_BNB | EP_MASKBN: return t32_dispatch_cfg<T, EP_BNB | EP_MASKBN, PRO>(a, cfg, st);
      default: break;
    }
  }
  return t32_dispatch_cfg<T, -1, PRO>(a, cfg, st);
}

template <typename T>
int t32_dispatch(const FsConvArgs& a, hipStream_t st) {
  const int cfg = t32_pick_cfg(a);
  switch (a.pro_mode) {
    case 0: return t32_dispatch_ep<T, 0>(a, cfg, st);
    case 1: return t32_dispatch_ep<T, 1>(a, cfg, st);
    case 2: return t32_dispatch_ep<T, 2>(a, cfg, st);
    default: return FS_EINVAL;
  }
}

}  // namespace

// internal entry: FS_EINVAL = "not mine" (fs_conv3x3_halo then runs the 16x16-tile kernel)
int fs_conv3x3_t32(const FsConvArgs& a, int dtype, hipStream_t st) {
  const int es = dtype == FS_DTYPE_BF16 ? 2 : 4;
  if ((a.Cs * es) % 64 != 0 || a.Co_p % 32 != 0 || a.Co % 8 != 0) return FS_EINVAL;
  if (a.pro_mode != 0 && (!a.pro_a || !a.pro_b || (a.pro_mode == 2 && (!a.pro_c || !a.pro_src2)))) return FS_EINVAL;
  if (a.bnb_scale && (!a.bnb_x || !a.bnb_shift || a.mask)) return FS_EINVAL;
  if (dtype == FS_DTYPE_BF16) return t32_dispatch<bf16>(a, st);
  if (dtype == FS_DTYPE_F32) return t32_dispatch<float>(a, st);
  return FS_EINVAL;
}
