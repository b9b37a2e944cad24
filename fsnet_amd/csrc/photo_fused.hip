// Fused photometric forward: warp (backproject -> project -> bilinear grid_sample, border padding) + SSIM/L1
// reprojection loss for both source frames + per-pixel minimum against the identity terms + masked sum, for ALL
// scales, in one launch — the warped images never reach HBM (fs_photo_warp + fs_photo_loss_fwd wrote and re-read
// 8 x [B,3,H,W] floats, and staged the target tile once per scale).
// Replaces (reference): MonoDepth2Decoder._generate_images_pred (monodepth2_decoder.py:61-116; FishEyeDecoder
// :355-411), compute_reprojection_loss (:118-128; SSIM monodepth_utils.py:184-215) and the per-pixel min / masking
// of compute_total_reprojection_loss (:226-272, 292).
//
// Structure: a wave owns a strip of 62 columns x 16 rows of one (batch element, scale): lane l holds column
// x0 - 1 + l, so the 3-tap horizontal window sums are two DPP adds (wave_shr / wave_shl, no LDS), and the rows are
// walked top to bottom with the last two rows' horizontal sums kept in registers (the vertical 3-tap sum is two
// adds).  Reflection padding = the halo lane / halo row evaluates the reflected pixel.  Source texels are gathered
// straight from global memory: consecutive lanes sample neighbouring texels (coalesced up to the warp's local
// stretch), the four scales of a strip run back to back on the same XCD, so the source rows stay cache resident.
// FS_HIPCC_FLAGS: -fno-slp-vectorize
// (SLP packing into v_pk_add/mul_f32 costs register shuffles here and keeps the DPP shifts from folding into the adds)
#include "photo_common.h"
#include <algorithm>

// fused multiply-adds allowed in this file (the library builds with -ffp-contract=off to mirror the unfused ATen
// arithmetic of the staged kernels; here an FMA only removes a rounding and a third of the VALU instructions)
#pragma clang fp contract(fast)

namespace {

constexpr int FW = 62;    // output columns per wave (lanes 1..62; lanes 0 and 63 are halo)
constexpr int FRH = 16;   // output rows per wave

__device__ __forceinline__ float hsum3(float v) {
  // lane i: v[i-1] + v[i] + v[i+1] (wave-wide shifts; the missing neighbour of lanes 0 / 63 reads 0: halo lanes)
  return (dpp_mov<0x138>(v) + v) + dpp_mov<0x130>(v);
}

// Two source frames per lane as one 2-vector: everything a frame owns — projection, sampler weights, the SSIM terms —
// is the same arithmetic on two independent values, and gfx950 issues v_pk_fma / v_pk_mul / v_pk_add_f32 on a register
// pair at the rate of the scalar forms (round 6; what has no packed form — v_rcp, min / max, floor, the DPP row sums —
// runs per component).  The compiler's own SLP packing of the scalar code had been measured slower (register shuffles
// to build the pairs: hence -fno-slp-vectorize above); here the pairs are the data layout.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 hsum3(f2 v) { return f2{hsum3(v.x), hsum3(v.y)}; }

// The three colour channels of one quantity as a packed pair + a scalar (backward kernel: a wave owns ONE frame there, the
// channels are what is independent): two instructions where three scalar ones stood, no register overhead.
struct C3 { f2 p; float s; };
__device__ __forceinline__ C3 c3(float a, float b, float c) { return C3{f2{a, b}, c}; }
__device__ __forceinline__ C3 c3(float a) { return C3{f2{a, a}, a}; }
__device__ __forceinline__ C3 operator+(C3 a, C3 b) { return C3{a.p + b.p, a.s + b.s}; }
__device__ __forceinline__ C3 operator-(C3 a, C3 b) { return C3{a.p - b.p, a.s - b.s}; }
__device__ __forceinline__ C3 operator*(C3 a, C3 b) { return C3{a.p * b.p, a.s * b.s}; }
__device__ __forceinline__ C3 operator+(C3 a, float b) { return C3{a.p + b, a.s + b}; }
__device__ __forceinline__ C3 operator-(C3 a, float b) { return C3{a.p - b, a.s - b}; }
__device__ __forceinline__ C3 operator*(C3 a, float b) { return C3{a.p * b, a.s * b}; }
__device__ __forceinline__ C3 operator*(float b, C3 a) { return C3{a.p * b, a.s * b}; }
__device__ __forceinline__ C3 operator-(float b, C3 a) { return C3{b - a.p, b - a.s}; }
__device__ __forceinline__ C3 operator-(C3 a) { return C3{-a.p, -a.s}; }
__device__ __forceinline__ C3 hsum3(C3 v) { return C3{f2{hsum3(v.p.x), hsum3(v.p.y)}, hsum3(v.s)}; }
__device__ __forceinline__ C3 rcp3(C3 v) {
  return C3{f2{__builtin_amdgcn_rcpf(v.p.x), __builtin_amdgcn_rcpf(v.p.y)}, __builtin_amdgcn_rcpf(v.s)};
}
__device__ __forceinline__ void put(C3& v, int c, float x) { if (c == 0) v.p.x = x; else if (c == 1) v.p.y = x; else v.s = x; }
__device__ __forceinline__ float get(const C3& v, int c) { return c == 0 ? v.p.x : (c == 1 ? v.p.y : v.s); }
__device__ __forceinline__ float dot3(C3 a, C3 b) { const f2 q = a.p * b.p; return (q.x + q.y) + a.s * b.s; }

struct Row {       // one image row of a strip: horizontal 3-tap sums + the raw values (needed when it is the centre)
  float t[3], tt[3];
  f2 x[3], xx[3], xt[3];        // component = source frame
  float rt[3];
  f2 rx[3];
  bool ov[2];
};

// load base[byte_off / 4] with a 32-bit byte offset: lets the compiler keep the (uniform) base in SGPRs and address
// with one VGPR (global_load ... s[base]) instead of a 64-bit add per load
__device__ __forceinline__ float ldg(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
// two horizontally adjacent texels in ONE 8-byte load at 4-byte alignment (global memory runs in unaligned access
// mode on gfx9): the gather is bound by the number of vector-memory instructions, not by their bytes
struct __attribute__((packed, aligned(4))) F2u { float a, b; };
__device__ __forceinline__ F2u ldg2(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const F2u*>(reinterpret_cast<const char*>(base) + byte_off);
}

// FISH = false: pinhole geometry inlined with per-lane / per-row invariants hoisted and reciprocals by v_rcp_f32
// (1 ulp; the staged kernels divide) — a sample coordinate moves by < 1e-4 px.  FISH = true: the shared ray-table
// geometry (photo_common.h).
template <bool WRITE_PRED, bool FISH>
__global__ __launch_bounds__(256) void photo_fused_fwd_kernel(const FsPhotoArgs p, int SX, int SY, int nwaves) {
  // XCD-contiguous logical block order: consecutive logical blocks (= the scales of one strip row) share an L2
  const int nblk = gridDim.x;
  const int lb = ((int)blockIdx.x & 7) * (nblk >> 3) + ((int)blockIdx.x >> 3);
  const int wv = __builtin_amdgcn_readfirstlane(lb * 4 + ((int)threadIdx.x >> 6));
  if (wv >= nwaves) return;
  const int lane = threadIdx.x & 63;
  const int sx = wv % SX;
  int rest = wv / SX;
  const int s = rest % p.S; rest /= p.S;
  const int sy = rest % SY;
  const int b = rest / SY;
  const int H = p.H, W = p.W;
  const unsigned HW = (unsigned)(H * W);
  const int h = p.dh[s], w = p.dw[s];
  const float* ge = p.geo + (long)b * GEO_STRIDE;
  const float* timg = p.img0 + (long)b * 3 * HW;
  const float* srcs[2] = {p.img_src[0] + (long)b * 3 * HW, p.img_src[1] + (long)b * 3 * HW};
  const float* dmap = p.depth[s] + (long)b * h * w;
  const float* identb = p.ident + (long)b * 2 * HW;
  uint8_t* selb = p.sel + ((long)s * p.B + b) * HW;
  const int x = sx * FW - 1 + lane;
  const int xr = min(max(refl(x, W), 0), W - 1);
  const bool col_out = lane >= 1 && lane <= FW && x < W;
  const int ys = sy * FRH;
  const int seed = p.noise_seed_ptr ? (*p.noise_seed_ptr & 0x3fffffff) : p.noise_seed;
  const float k9 = 1.f / 9.f, k3 = 1.f / 3.f;
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  const float iwm1 = 1.f / wm1, ihm1 = 1.f / hm1;
  const bool have_mask = p.warp_mask || p.patched_mask;
  const bool no_ovm = p.no_overlap_mask != 0;

  // per-lane invariants of the depth upsample (ATen upsample_bilinear2d, align_corners=True) and of the ray
  const float sh = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
  // (the tap pair starts at dx0 <= w - 2 so that it is one 8-byte load; at the right border the weight moves over)
  const float fxu = sw * (float)xr;
  const int dx0 = w > 1 ? min((int)fxu, w - 2) : 0;
  const float dlx = w > 1 ? fxu - (float)dx0 : 0.f;
  const float pxf = (float)xr;
  float ra[3] = {0.f, 0.f, 0.f};
  f2 Pm[12];                    // (frame 0, frame 1) entries of the two projection matrices
  if (!FISH) {
#pragma unroll
    for (int i = 0; i < 3; ++i) ra[i] = ge[3 * i] * pxf;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) Pm[k] = f2{ge[18 + k], ge[30 + k]};
  double acc = 0.0;

  // one row: fill `r0` (row y = ys - 1 + it) and, from the third row on, finish the window centred on row y - 1 = `r1`
  auto step = [&](const Row& r2, const Row& r1, Row& r0, const int it) {
    const int y = ys - 1 + it;
    const int yr = min(max(refl(y, H), 0), H - 1);
    const unsigned ob = (unsigned)(yr * W + xr) * 4u;
#pragma unroll
    for (int c = 0; c < 3; ++c) r0.rt[c] = ldg(timg + c * HW, ob);
    Geo g;
    if (FISH) {
      pixel_ray(p, p.depth[s], b, yr, xr, H, W, h, w, ge, g);
    } else {
      // depth upsample: the row taps are wave-uniform
      const float fyu = sh * (float)yr;
      const int dy0 = (int)fyu, dy1 = dy0 + (dy0 < h - 1 ? 1 : 0);
      const float dly = fyu - (float)dy0;
      float d00, d01, d10, d11;
      if (w > 1) {
        const F2u a = ldg2(dmap, (unsigned)(dy0 * w + dx0) * 4u), c2 = ldg2(dmap, (unsigned)(dy1 * w + dx0) * 4u);
        d00 = a.a; d01 = a.b; d10 = c2.a; d11 = c2.b;
      } else {
        d00 = d01 = ldg(dmap, (unsigned)(dy0 * w) * 4u); d10 = d11 = ldg(dmap, (unsigned)(dy1 * w) * 4u);
      }
      g.D = (1.f - dly) * ((1.f - dlx) * d00 + dlx * d01) + dly * ((1.f - dlx) * d10 + dlx * d11);
      const float pyf = (float)yr;
#pragma unroll
      for (int i = 0; i < 3; ++i) g.r[i] = (ra[i] + ge[3 * i + 1] * pyf) + ge[3 * i + 2];
    }
    unsigned og[2][2];
    f2 w01[2], w23[2];           // per frame: the sampler weights of the upper / lower tap pair
    unsigned om[2];
    bool inb[2];
    f2 ixu2, iyu2;
    if (FISH) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        project_ray(p, b, H, W, ge, f, g);
        ixu2[f] = g.ixu; iyu2[f] = g.iyu;
      }
    } else {
      const float cx_ = g.D * g.r[0], cy_ = g.D * g.r[1], cz_ = g.D * g.r[2];
      const f2 X = Pm[0] * cx_ + Pm[1] * cy_ + Pm[2] * cz_ + Pm[3];
      const f2 Y = Pm[4] * cx_ + Pm[5] * cy_ + Pm[6] * cz_ + Pm[7];
      const f2 Zp = (Pm[8] * cx_ + Pm[9] * cy_ + Pm[10] * cz_ + Pm[11]) + 1e-7f;
      const f2 iz = f2{__builtin_amdgcn_rcpf(Zp.x), __builtin_amdgcn_rcpf(Zp.y)};
      const f2 u = X * iz, v = Y * iz;
      // Project3D normalisation followed by grid_sample's align_corners=True un-normalisation
      const f2 un = (u * iwm1 - 0.5f) * 2.f, vn = (v * ihm1 - 0.5f) * 2.f;
      ixu2 = (un + 1.f) * 0.5f * wm1;
      iyu2 = (vn + 1.f) * 0.5f * hm1;
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      // bilinear taps with border clamping; the tap pair always starts at x0 <= W - 2 / y0 <= H - 2 (at the right /
      // bottom border the weight moves to the second tap: same value) so that (x0, x0 + 1) is one 8-byte load
      const float ixu = ixu2[f], iyu = iyu2[f];
      const float ix = fminf(fmaxf(ixu, 0.f), wm1), iy = fminf(fmaxf(iyu, 0.f), hm1);
      const float fx0 = fminf(floorf(ix), wm1 - 1.f), fy0 = fminf(floorf(iy), hm1 - 1.f);
      const float wx = ix - fx0, wy = iy - fy0;
      const int tx0 = (int)fx0, ty0 = (int)fy0;
      og[f][0] = (unsigned)(ty0 * W + tx0) * 4u;
      og[f][1] = og[f][0] + (unsigned)W * 4u;
      const f2 wxs = f2{1.f - wx, wx};
      w01[f] = wxs * (1.f - wy); w23[f] = wxs * wy;
      // nearest sample of patched_mask (fisheye: x ray-table mask) with zeros padding, round half to even
      const float xn = nearbyintf(ixu), yn = nearbyintf(iyu);
      inb[f] = xn >= 0.f && xn <= wm1 && yn >= 0.f && yn <= hm1;
      om[f] = inb[f] ? (unsigned)((int)yn * W + (int)xn) : 0u;
    }
    // all gathers of the row in flight together: (x0, x0 + 1) of a tap row arrive as one register pair
    f2 tap[2][3][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const F2u v2 = ldg2(srcs[f] + c * HW, og[f][k]);
          tap[f][c][k] = f2{v2.a, v2.b};
        }
    float mv[2] = {1.f, 1.f};
    if (have_mask) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
        mv[f] = p.warp_mask ? p.warp_mask[(long)b * HW + om[f]] : (float)p.patched_mask[(long)b * HW + om[f]];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      r0.t[c] = hsum3(r0.rt[c]);
      r0.tt[c] = hsum3(r0.rt[c] * r0.rt[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f2 v;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const f2 q = w01[f] * tap[f][c][0] + w23[f] * tap[f][c][1];      // (x0 column, x0 + 1 column) partial sums
        v[f] = q.x + q.y;
      }
      r0.rx[c] = v;
      r0.x[c] = hsum3(v);
      r0.xx[c] = hsum3(v * v);
      r0.xt[c] = hsum3(v * r0.rt[c]);
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
      r0.ov[f] = no_ovm || (inb[f] && mv[f] == 1.f);   // overlapped_mask=False: every sample counts (border-clamped)
    if (WRITE_PRED) {
      if (col_out && y >= ys && y < ys + FRH && y < H) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          float* pr = p.pred + (((long)s * 2 + f) * p.B + b) * 3 * HW + (unsigned)(y * W + x);
          pr[0] = r0.rx[0][f]; pr[HW] = r0.rx[1][f]; pr[2 * HW] = r0.rx[2][f];
          p.ov[(((long)s * 2 + f) * p.B + b) * HW + (unsigned)(y * W + x)] = r0.ov[f] ? 1 : 0;
        }
      }
    }
    if (it < 2) return;
    const int yc = y - 1;          // window centre row; rows y-2 (r2), y-1 (r1), y (r0)
    f2 ssim_sum = splat(0.f), l1 = splat(0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float sy_ = (r2.t[c] + r1.t[c]) + r0.t[c], syy = (r2.tt[c] + r1.tt[c]) + r0.tt[c];
      const float muy = sy_ * k9;
      const float sgy = syy * k9 - muy * muy;
      const f2 sxs = (r2.x[c] + r1.x[c]) + r0.x[c], sxx = (r2.xx[c] + r1.xx[c]) + r0.xx[c],
               sxy = (r2.xt[c] + r1.xt[c]) + r0.xt[c];
      const f2 mux = sxs * k9;
      const f2 sgx = sxx * k9 - mux * mux, sgxy = sxy * k9 - mux * muy;
      const f2 n = (2.f * mux * muy + C1) * (2.f * sgxy + C2);
      const f2 d = (mux * mux + muy * muy + C1) * (sgx + sgy + C2);
      const f2 q = FISH ? n / d : n * f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
      const f2 sv = (1.f - q) * 0.5f;
      ssim_sum += f2{fminf(fmaxf(sv.x, 0.f), 1.f), fminf(fmaxf(sv.y, 0.f), 1.f)};
      const f2 df = r1.rt[c] - r1.rx[c];
      l1 += f2{fabsf(df.x), fabsf(df.y)};
    }
    if (col_out && yc < H) {       // (yc >= ys always: it >= 2)
      const unsigned i = (unsigned)(yc * W + x);
      float best = INFINITY; int bi = 2;
      if (!p.motion_mask) {          // identity candidates (the auto-mask); a motion mask replaces them (:243-246)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          uint32_t key = (uint32_t)((((long)s * 2 + f) * p.B + b) * HW + i);
          float v = ldg(identb + f * HW, i * 4u) + tie_noise(seed, key);
          if (f == 0 || v < best) { best = v; bi = f; }
        }
      }
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float rv = 0.85f * (ssim_sum[f] * k3) + 0.15f * (l1[f] * k3);
        float v = r1.ov[f] ? rv : 100.f;
        if (v < best) { best = v; bi = 2 + f; }
      }
      // a selected reprojection term whose sample missed the source frame is the constant 100 (:231-235): value only,
      // no gradient (cannot happen next to the identity candidates, which are always below 100)
      if ((bi == 2 && !r1.ov[0]) || (bi == 3 && !r1.ov[1])) bi = 4;   // (static indices: a runtime index spills the row)
      selb[i] = (uint8_t)bi;
      double pm = p.patched_mask ? p.patched_mask[(long)b * HW + i] : 1.0;
      acc += (double)best * pm;
    }
  };

  static_assert((FRH + 2) % 3 == 0, "the row ring is rotated by unrolling three steps");
  Row A, B, C;
  for (int it = 0; it < FRH + 2; it += 3) {
    step(B, C, A, it);
    step(C, A, B, it + 1);
    step(A, B, C, it + 2);
  }
  acc = wave_sum_d(acc);
  if (lane == 0) atomicAdd(p.loss_sums + s * p.B + b, acc);
}


// =============================================================================================================
// Fused backward: d loss / d depth_s and the per-strip partials of d loss / d P_f, recomputing the warp instead
// of reading the eight warped images back (fs_photo_loss_bwd staged pred + target tiles per scale through LDS).
// A wave owns (batch element, scale, source frame, strip of 60 columns x 32 rows) and walks the rows once with two
// register rings, three stages per step at row y:
//   A  warp row y (both the values and the sampler Jacobian d pred / d (ix, iy)), horizontal window sums
//   B  window centre y-1: SSIM-derivative coefficients A + B x(q) + C t(q) for this frame where the per-pixel minimum
//      selected it (sel == 2 + f), horizontally box-summed (the transpose of the window gather)
//   C  output row y-2: vertical box sum of the coefficient rows -> d loss / d pred, + the L1 term, through the sampler
//      Jacobian, the projection and the depth upsample.
// Reflection padding: halo lanes / rows hold the reflected pixel; the virtual positions x = -1, W and y = -1, H
// (which receive window contributions of the border centres) push their gradient through the reflected pixel's
// chain — the chain is linear in d pred, so that equals folding them onto the reflected pixel.
// =============================================================================================================
constexpr int BW = 60;     // output columns per wave (lanes 2..61)
constexpr int BRH = 32;    // output rows per wave
constexpr int LDS_C = 40;  // low-res accumulation tile of a wave (scales >= 1): columns / rows
constexpr int LDS_R = 24;

struct BRow {
  C3 t, tt, x, xx, xt;                      // horizontal 3-tap sums (per colour channel)
  C3 rt, rx;                                // raw values
  C3 Jx, Jy;                                // d pred_c / d (ix, iy) with the border-clamp multipliers applied
  float X, Y, Z, D;                         // transformed point (pinhole: Z + eps) and the upsampled depth
  float pm;                                 // patched_mask at the pixel (0 for virtual positions)
  int sel;                                  // selection byte at the pixel (-1 for virtual positions)
};
struct CRow { C3 a, b, c; };                // horizontally box-summed coefficients of one centre row

template <bool FISH>
__global__ __launch_bounds__(256) void photo_fused_bwd_kernel(const FsPhotoArgs p, int SX, int SY, int nwaves) {
  __shared__ float s_dd[4][LDS_R * LDS_C];
  const int nblk = gridDim.x;
  const int lb = ((int)blockIdx.x & 7) * (nblk >> 3) + ((int)blockIdx.x >> 3);
  const int wid = (int)threadIdx.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(lb * 4 + wid);
  if (wv >= nwaves) return;
  const int lane = threadIdx.x & 63;
  const int sx = wv % SX;
  int rest = wv / SX;
  const int f = rest & 1; rest >>= 1;
  const int s = rest % p.S; rest /= p.S;
  const int sy = rest % SY;
  const int b = rest / SY;
  const int H = p.H, W = p.W;
  const unsigned HW = (unsigned)(H * W);
  const int h = p.dh[s], w = p.dw[s];
  const float* ge = p.geo + (long)b * GEO_STRIDE;
  const float* timg = p.img0 + (long)b * 3 * HW;
  const float* src = p.img_src[f] + (long)b * 3 * HW;
  const float* dmap = p.depth[s] + (long)b * h * w;
  const uint8_t* selb = p.sel + ((long)s * p.B + b) * HW;
  float* ddout = p.d_depth[s] + (long)b * h * w;
  float* lds = s_dd[wid];
  const int x0s = sx * BW, ys = sy * BRH;
  const int x = x0s - 2 + lane;
  const int xr = min(max(refl(x, W), 0), W - 1);
  const bool x_real = x >= 0 && x < W;
  const bool last_x = x0s + BW >= W, last_y = ys + BRH >= H;
  const bool col_own = (lane >= 2 && lane < 2 + BW && x < W) || (x == -1) || (x == W && last_x);   // x == -1 only when sx == 0
  const int q_lo = ys == 0 ? -1 : ys, q_hi = last_y ? H : ys + BRH - 1;
  const float k9 = 1.f / 9.f;
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
  const float iwm1 = 1.f / wm1, ihm1 = 1.f / hm1;
  double msum = 0.0;
  for (int k = 0; k < p.B; ++k) msum += p.mask_sum[k];
  const float gscale = (float)((p.gout ? *p.gout : 1.0) / ((double)p.S * (msum + 1e-6)));
  const float wss = gscale * (0.85f / 3.f), wl1 = gscale * (0.15f / 3.f);

  // depth upsample invariants (see the forward kernel)
  const float sh = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const float fxu = sw * (float)xr;
  const int dx0 = w > 1 ? min((int)fxu, w - 2) : 0;
  const float dlx = w > 1 ? fxu - (float)dx0 : 0.f;
  const bool unit_scale = (h == H && w == W);
  const int lx_min = __builtin_amdgcn_readlane(dx0, 2);                       // lane 2 holds the strip's first column
  const int ly_min = (int)(sh * (float)min(max(ys - 1, 0), H - 1));
  if (!unit_scale) {
    for (int i = lane; i < LDS_R * LDS_C; i += 64) lds[i] = 0.f;
  }
  const float pxf = (float)xr;
  float ra[3] = {0.f, 0.f, 0.f}, Pm[12];
  if (!FISH) {
#pragma unroll
    for (int i = 0; i < 3; ++i) ra[i] = ge[3 * i] * pxf;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) Pm[k] = ge[18 + f * 12 + k];
  const float* mei = FISH ? p.mei + (long)b * 8 : nullptr;
  float dPacc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) dPacc[k] = 0.f;

  auto ray_of = [&](int yr, float (&r)[3]) {
    if (FISH) {
      const float* lut = p.lut_ptrs[b];
      const unsigned o = (unsigned)(yr * W + xr) * 4u;
      r[0] = ldg(lut, o); r[1] = ldg(lut + HW, o); r[2] = ldg(lut + 2 * HW, o);
    } else {
      const float pyf = (float)yr;
#pragma unroll
      for (int i = 0; i < 3; ++i) r[i] = (ra[i] + ge[3 * i + 1] * pyf) + ge[3 * i + 2];
    }
  };

  auto step = [&](BRow& r2, const BRow& r1, BRow& r0, CRow& c2, const CRow& c1, CRow& c0, const int it) {
    // ------------------------------------------------------------------ stage A: row y
    const int y = ys - 2 + it;
    const int yr = min(max(refl(y, H), 0), H - 1);
    const bool y_real = y >= 0 && y < H;
    const unsigned ob = (unsigned)(yr * W + xr) * 4u;
    r0.rt = c3(ldg(timg, ob), ldg(timg + HW, ob), ldg(timg + 2 * HW, ob));
    r0.sel = (x_real && y_real) ? (int)selb[(unsigned)(yr * W + xr)] : -1;
    r0.pm = (x_real && y_real) ? (p.patched_mask ? (float)p.patched_mask[(long)b * HW + (unsigned)(yr * W + xr)] : 1.f) : 0.f;
    if (p.motion_mask && x_real && y_real) r0.pm *= 1.f - p.motion_mask[(long)b * HW + (unsigned)(yr * W + xr)];
    {
      const float fyu = sh * (float)yr;
      const int dy0 = h > 1 ? min((int)fyu, h - 2) : 0;
      const float dly = h > 1 ? fyu - (float)dy0 : 0.f;
      float d00, d01, d10, d11;
      if (w > 1 && h > 1) {
        const F2u a = ldg2(dmap, (unsigned)(dy0 * w + dx0) * 4u), c2_ = ldg2(dmap, (unsigned)((dy0 + 1) * w + dx0) * 4u);
        d00 = a.a; d01 = a.b; d10 = c2_.a; d11 = c2_.b;
      } else {
        d00 = d01 = d10 = d11 = ldg(dmap, (unsigned)(dy0 * w + dx0) * 4u);
      }
      r0.D = (1.f - dly) * ((1.f - dlx) * d00 + dlx * d01) + dly * ((1.f - dlx) * d10 + dlx * d11);
    }
    float ray[3];
    ray_of(yr, ray);
    float ixu, iyu;
    {
      const float cx_ = r0.D * ray[0], cy_ = r0.D * ray[1], cz_ = r0.D * ray[2];
      r0.X = Pm[0] * cx_ + Pm[1] * cy_ + Pm[2] * cz_ + Pm[3];
      r0.Y = Pm[4] * cx_ + Pm[5] * cy_ + Pm[6] * cz_ + Pm[7];
      r0.Z = Pm[8] * cx_ + Pm[9] * cy_ + Pm[10] * cz_ + Pm[11];
      if (FISH) {
        float u, v;
        mei_cam2image(mei, r0.X, r0.Y, r0.Z, u, v);
        const float un = u / (float)max(W - 1, 1) * 2.f - 1.f, vn = v / (float)max(H - 1, 1) * 2.f - 1.f;
        ixu = (un + 1.f) * 0.5f * wm1; iyu = (vn + 1.f) * 0.5f * hm1;
      } else {
        r0.Z += 1e-7f;
        const float iz = __builtin_amdgcn_rcpf(r0.Z);
        const float u = r0.X * iz, v = r0.Y * iz;
        const float un = (u * iwm1 - 0.5f) * 2.f, vn = (v * ihm1 - 0.5f) * 2.f;
        ixu = (un + 1.f) * 0.5f * wm1; iyu = (vn + 1.f) * 0.5f * hm1;
      }
    }
    {
      // border clamp of grid_sample: the coordinate gradient is zero where the sample was clipped
      const float mx = (ixu > 0.f && ixu < wm1) ? 1.f : 0.f, my = (iyu > 0.f && iyu < hm1) ? 1.f : 0.f;
      const float ix = fminf(fmaxf(ixu, 0.f), wm1), iy = fminf(fmaxf(iyu, 0.f), hm1);
      const float fx0 = fminf(floorf(ix), wm1 - 1.f), fy0 = fminf(floorf(iy), hm1 - 1.f);
      const float wx = ix - fx0, wy = iy - fy0;
      const unsigned o0 = (unsigned)((int)fy0 * W + (int)fx0) * 4u, o1 = o0 + (unsigned)W * 4u;
      float rxv[3], jxv[3], jyv[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const F2u a = ldg2(src + c * HW, o0), bb = ldg2(src + c * HW, o1);
        const float top = a.a + wx * (a.b - a.a), bot = bb.a + wx * (bb.b - bb.a);
        rxv[c] = top + wy * (bot - top);
        jxv[c] = mx * ((a.b - a.a) * (1.f - wy) + (bb.b - bb.a) * wy);
        jyv[c] = my * (bot - top);
      }
      r0.rx = c3(rxv[0], rxv[1], rxv[2]); r0.Jx = c3(jxv[0], jxv[1], jxv[2]); r0.Jy = c3(jyv[0], jyv[1], jyv[2]);
    }
    r0.t = hsum3(r0.rt);
    r0.tt = hsum3(r0.rt * r0.rt);
    r0.x = hsum3(r0.rx);
    r0.xx = hsum3(r0.rx * r0.rx);
    r0.xt = hsum3(r0.rx * r0.rt);
    // ------------------------------------------------------------------ stage B: window centre = row y - 1 (r1)
    {
      const float wgt = (it >= 2 && r1.sel == 2 + f) ? r1.pm * wss : 0.f;
      const C3 sxs = (r2.x + r1.x) + r0.x, sys_ = (r2.t + r1.t) + r0.t;
      const C3 sxx = (r2.xx + r1.xx) + r0.xx, syy = (r2.tt + r1.tt) + r0.tt;
      const C3 sxy = (r2.xt + r1.xt) + r0.xt;
      const C3 mux = sxs * k9, muy = sys_ * k9;
      const C3 sgx = sxx * k9 - mux * mux, sgy = syy * k9 - muy * muy, sgxy = sxy * k9 - mux * muy;
      const C3 n1 = 2.f * mux * muy + C1, n2 = 2.f * sgxy + C2;
      const C3 d1 = mux * mux + muy * muy + C1, d2 = sgx + sgy + C2;
      const C3 n = n1 * n2, d = d1 * d2;
      const C3 id = rcp3(d);
      const C3 sv = (1.f - n * id) * 0.5f;
      // d n / d x(q) = a1 + a2 (t(q) - muy),  d d / d x(q) = b1 + b2 (x(q) - mux)   (each tap weight 1/9)
      const C3 a1 = 2.f * muy * n2 * k9, a2 = 2.f * n1 * k9;
      const C3 b1 = 2.f * mux * d2 * k9, b2 = 2.f * d1 * k9;
      const C3 hh = -0.5f * id * id;
      C3 A = hh * ((a1 - a2 * muy) * d - n * (b1 - b2 * mux));
      C3 Bc = -(hh * n * b2);
      C3 Cc = hh * a2 * d;
      const bool on = wgt != 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float svc = get(sv, c);
        const bool ok = on && svc >= 0.f && svc <= 1.f;     // (outside the clamp of the SSIM term: no gradient)
        put(A, c, ok ? wgt * get(A, c) : 0.f);
        put(Bc, c, ok ? wgt * get(Bc, c) : 0.f);
        put(Cc, c, ok ? wgt * get(Cc, c) : 0.f);
      }
      c0.a = hsum3(A);
      c0.b = hsum3(Bc);
      c0.c = hsum3(Cc);
    }
    // ------------------------------------------------------------------ stage C: output row q = y - 2 (r2, centres c2 c1 c0)
    const int qy = y - 2;
    if (it >= 3 && qy >= q_lo && qy <= q_hi) {
      const bool l1_on = r2.sel == 2 + f;
      const C3 ga = (c2.a + c1.a) + c0.a, gb = (c2.b + c1.b) + c0.b, gc = (c2.c + c1.c) + c0.c;
      C3 dpred = ga + gb * r2.rx + gc * r2.rt;
      if (l1_on) {
        const C3 df = r2.rx - r2.rt;
        const float w1 = r2.pm * wl1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float dfc = get(df, c);
          put(dpred, c, get(dpred, c) + w1 * (dfc > 0.f ? 1.f : (dfc < 0.f ? -1.f : 0.f)));
        }
      }
      if (!col_own) dpred = c3(0.f);
      const float du = dot3(dpred, r2.Jx);
      const float dv = dot3(dpred, r2.Jy);
      if (du != 0.f || dv != 0.f) {
        float dX, dY, dZ;
        if (FISH) {
          float dq[3];
          mei_cam2image_bwd(mei, r2.X, r2.Y, r2.Z, du, dv, dq);
          dX = dq[0]; dY = dq[1]; dZ = dq[2];
        } else {
          const float iz = __builtin_amdgcn_rcpf(r2.Z);
          dX = du * iz; dY = dv * iz;
          dZ = -(du * r2.X + dv * r2.Y) * iz * iz;
        }
        const int qyr = min(max(refl(qy, H), 0), H - 1);
        float ray[3];
        ray_of(qyr, ray);
        const float pr0 = Pm[0] * ray[0] + Pm[1] * ray[1] + Pm[2] * ray[2];
        const float pr1 = Pm[4] * ray[0] + Pm[5] * ray[1] + Pm[6] * ray[2];
        const float pr2 = Pm[8] * ray[0] + Pm[9] * ray[1] + Pm[10] * ray[2];
        const float dD = dX * pr0 + dY * pr1 + dZ * pr2;
        const float cam[3] = {r2.D * ray[0], r2.D * ray[1], r2.D * ray[2]};
        const float dxyz[3] = {dX, dY, dZ};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) dPacc[i * 4 + j] += dxyz[i] * cam[j];
          dPacc[i * 4 + 3] += dxyz[i];
        }
        // transpose of the bilinear depth upsample
        if (unit_scale) {
          atomicAdd(ddout + (unsigned)(qyr * w + xr), dD);
        } else {
          const float fyu = sh * (float)qyr;
          const int dy0 = h > 1 ? min((int)fyu, h - 2) : 0;
          const float dly = h > 1 ? fyu - (float)dy0 : 0.f;
          const int li = (dy0 - ly_min) * LDS_C + (dx0 - lx_min);
          if (li >= 0 && dx0 >= lx_min && li + LDS_C + 1 < LDS_R * LDS_C && dx0 - lx_min + 1 < LDS_C) {
            atomicAdd(&lds[li], (1.f - dly) * (1.f - dlx) * dD);
            atomicAdd(&lds[li + 1], (1.f - dly) * dlx * dD);
            atomicAdd(&lds[li + LDS_C], dly * (1.f - dlx) * dD);
            atomicAdd(&lds[li + LDS_C + 1], dly * dlx * dD);
          } else {       // (strips of an unexpected aspect: straight to memory)
            const int hh1 = min(dy0 + 1, h - 1), ww1 = min(dx0 + 1, w - 1);
            atomicAdd(ddout + dy0 * w + dx0, (1.f - dly) * (1.f - dlx) * dD);
            atomicAdd(ddout + dy0 * w + ww1, (1.f - dly) * dlx * dD);
            atomicAdd(ddout + hh1 * w + dx0, dly * (1.f - dlx) * dD);
            atomicAdd(ddout + hh1 * w + ww1, dly * dlx * dD);
          }
        }
      }
    }
  };

  const int nsteps = BRH + 4 + (last_y ? 1 : 0);
  BRow A = {}, B = {}, C = {};
  CRow ca = {}, cb = {}, cc = {};
  A.sel = B.sel = C.sel = -1;
  for (int it = 0; it < nsteps; it += 3) {
    step(B, C, A, cb, cc, ca, it);
    if (it + 1 < nsteps) step(C, A, B, cc, ca, cb, it + 1);
    if (it + 2 < nsteps) step(A, B, C, ca, cb, cc, it + 2);
  }
  // ---- per-strip partial of d loss / d P_f (reduced in a fixed order by fs_photo_pose_grad) ----
  {
    const long tile = (long)sy * SX + sx, tiles = (long)SY * SX;
    float* out = p.dP + ((((long)s * p.B + b) * tiles + tile) * 2 + f) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float v = wave_sum(dPacc[k]);
      if (lane == 0) out[k] = v;
    }
  }
  if (!unit_scale) {
    for (int i = lane; i < LDS_R * LDS_C; i += 64) {
      const float v = lds[i];
      if (v != 0.f) {
        const int yy = ly_min + i / LDS_C, xx = lx_min + i % LDS_C;
        if (yy < h && xx < w) atomicAdd(ddout + yy * w + xx, v);
      }
    }
  }
}

}  // namespace

extern "C" int fs_photo_fused_fwd(const FsPhotoArgs* a, void* stream) {
  if (!a || !a->img0 || !a->img_src[0] || !a->img_src[1] || !a->geo || (!a->ident && !a->motion_mask) || !a->sel ||
      !a->loss_sums)
    return FS_EINVAL;
  if (a->S < 1 || a->S > 4 || a->B < 1 || a->H < 2 || a->W < 2) return FS_EINVAL;
  if ((a->lut_ptrs != nullptr) != (a->mei != nullptr)) return FS_EINVAL;
  if ((a->pred != nullptr) != (a->ov != nullptr)) return FS_EINVAL;
  for (int s = 0; s < a->S; ++s) if (!a->depth[s]) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int SX = (a->W + FW - 1) / FW, SY = (a->H + FRH - 1) / FRH;
  const long nwaves = (long)a->B * a->S * SY * SX;
  long nblk = (nwaves + 3) / 4;
  nblk = (nblk + 7) / 8 * 8;
  if (nblk > 0x7fffffffL) return FS_EINVAL;
  if ((long)a->H * a->W * 3 >= 0x40000000L) return FS_EINVAL;          // 32-bit plane offsets
  const dim3 grid((unsigned)nblk), blk(256);
  const bool fish = a->lut_ptrs != nullptr;
  if (a->pred) {
    if (fish) hipLaunchKernelGGL((photo_fused_fwd_kernel<true, true>), grid, blk, 0, st, *a, SX, SY, (int)nwaves);
    else hipLaunchKernelGGL((photo_fused_fwd_kernel<true, false>), grid, blk, 0, st, *a, SX, SY, (int)nwaves);
  } else {
    if (fish) hipLaunchKernelGGL((photo_fused_fwd_kernel<false, true>), grid, blk, 0, st, *a, SX, SY, (int)nwaves);
    else hipLaunchKernelGGL((photo_fused_fwd_kernel<false, false>), grid, blk, 0, st, *a, SX, SY, (int)nwaves);
  }
  return fs_launch_status();
}

extern "C" int64_t fs_photo_fused_bwd_tiles(int H, int W) {
  if (H < 2 || W < 2) return -1;
  return (int64_t)((W + BW - 1) / BW) * ((H + BRH - 1) / BRH);
}

extern "C" int fs_photo_fused_bwd(const FsPhotoArgs* a, void* stream) {
  if (!a || !a->img0 || !a->img_src[0] || !a->img_src[1] || !a->geo || !a->sel || !a->dP || !a->mask_sum)
    return FS_EINVAL;
  if (a->S < 1 || a->S > 4 || a->B < 1 || a->H < 2 || a->W < 2) return FS_EINVAL;
  if ((a->lut_ptrs != nullptr) != (a->mei != nullptr)) return FS_EINVAL;
  if ((long)a->H * a->W * 3 >= 0x40000000L) return FS_EINVAL;
  for (int s = 0; s < a->S; ++s) if (!a->depth[s] || !a->d_depth[s]) return FS_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int SX = (a->W + BW - 1) / BW, SY = (a->H + BRH - 1) / BRH;
  const long nwaves = (long)a->B * a->S * 2 * SY * SX;
  long nblk = (nwaves + 3) / 4;
  nblk = (nblk + 7) / 8 * 8;
  if (nblk > 0x7fffffffL) return FS_EINVAL;
  const dim3 grid((unsigned)nblk), blk(256);
  if (a->lut_ptrs) hipLaunchKernelGGL(photo_fused_bwd_kernel<true>, grid, blk, 0, st, *a, SX, SY, (int)nwaves);
  else hipLaunchKernelGGL(photo_fused_bwd_kernel<false>, grid, blk, 0, st, *a, SX, SY, (int)nwaves);
  return fs_launch_status();
}
