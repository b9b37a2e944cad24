// 3x3 / stride-1 convolution (forward and data gradient) on 32x32 MFMA tiles, two wave groups per workgroup that
// alternate roles — the kernel the round-3 measurements asked for:
//   * on the one-group kernels (conv3x3_t32.hip, conv3x3_halo.hip) operand staging, MFMA loop and epilogue simply ADD
//     UP (16 + 9.5 + 9.2 us on the 64 -> 64 layer at 36 images): every block of a launch is in the same phase at the
//     same time, so the matrix pipes idle while operands are staged and results stored, and vice versa;
//   * they execute 13 vector-ALU instructions per MFMA (address arithmetic of the staging, statistics reduction), and
//     a CU takes operands in at only ~15 bytes per clock.
// Here a workgroup is eight waves = two groups of four (one wave of each group per SIMD).  Each group owns a pixel
// tile x channel tile item of its own and its own LDS stage buffer; in every STEP one group multiplies the channel
// chunk it staged in the step before (72 MFMAs per wave on 64 x 64 wave tiles: half an LDS fragment read per MFMA)
// while the other group runs the epilogue of the item it just finished, writes its next chunk to LDS and issues the
// global loads of the one after — then one workgroup barrier, and the roles swap.  The matrix pipe of every SIMD
// therefore always has one wave in its MFMA loop and one wave doing memory work beside it (MI355X_MICROARCH.md, "Two
// waves per SIMD"), the staging instructions are shared by twice the MFMA work per thread (256 x 64 tiles), and a
// group's global loads have a whole compute segment to land.  Blocks are persistent: group slot s works on the items
// s, s + 2 * gridDim, ...
// Same arguments, packed weights, epilogue and prologue options as conv3x3_t32.hip.
// Reference call sites: vision_base/networks/models/backbone/resnet.py:21-50 (BasicBlock), blocks.py:41-54,
// monodepth/networks/models/heads/depth_encoder.py:45-63, pose_decoder.py:17-37.
#include "t32_common.h"

namespace {

template <typename T, int PIX, int CO, int EP, int PRO>
__global__ __launch_bounds__(512, 2) void conv3x3_d32_kernel(const FsConvArgs p, const T32Geom g) {
  constexpr int WPIX = PIX / 4;                 // pixels per wave
  constexpr int TP = WPIX / 32, TC = CO / 32;   // 32 x 32 MFMA tiles per wave: pixels x channels
  constexpr int HMAX = t32_hmax(PIX);           // halo pixels per stage
  constexpr int HS = 5;                         // 16-byte units per halo pixel: 4 used + 1 pad (see below)
  constexpr int LH = (HMAX * 4 + 255) / 256;
  constexpr int WU = 9 * CO * 4;                // weight units per stage
  constexpr int LW = (WU + 255) / 256;
  constexpr int OOB = 0x7ffff000;               // + a chunk offset (< 4096) stays out of every buffer's range
  constexpr int UN = Unit<T>::N;                // elements per unit

  // Bank layout.  ds_read_b128 is serviced in four groups of 16 lanes — {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
  // same + 32 (MI355X_MICROARCH.md, LDS).  A 32x32 MFMA operand has lane l read row (l & 31), 16-byte k-slot
  // 2*ks + (l >> 5).  Halo pixels: consecutive pixels 5 units apart — 5 is odd, so the 16 rows of a group (all residues
  // mod 16) land on 16 distinct 16-byte bank slots at ANY tap shift, and the shift stays an immediate offset.
  // Weights: 4 units per row, slot XOR ((row >> 2) & 3): rows that share (row & 3) — the same 64-byte quarter of the
  // 256-byte bank row — differ in (row >> 2) & 3 within a group.
  constexpr int BUFU = t32_lds_units(PIX, CO);
  __shared__ uint4 lds_all[2 * BUFU];
  // (readfirstlane: the group / wave index is wave-uniform, but derived from threadIdx the compiler would treat every
  // cursor, item and branch that depends on it as divergent — vector ALU arithmetic under exec masks instead of SALU)
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // wave group: 0 or 1 (waves 0-3 / 4-7)
  uint4* const lds = lds_all + grp * BUFU;         // the group's own stage buffer: halo rows, then the nine taps' weights
  uint4* const lds_w = lds + HMAX * HS;
  // [4 waves][CO][2] statistics partials live in the halo rows' padding units (every fifth unit, never staged)
  static_assert(4 * CO * 2 <= HMAX * 4, "statistics partials must fit the halo padding");
  auto red = [&](int f) -> float& { return reinterpret_cast<float*>(lds + (f >> 2) * HS + 4)[f & 3]; };

  const int t = threadIdx.x & 255, lane = t & 63;                    // thread index inside the group
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);           // wave index inside the group
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
    if (g.map_mode == 0) { const int sq = fs_div(slot, g.dMap); cy = slot - sq * g.map_div; px = sq * 8 + xcd; }
    else if (g.map_mode == 1) { const int sq = fs_div(slot, g.dMap); cy = xcd + 8 * (slot - sq * g.map_div); px = sq; }
    else if (g.map_mode == 2) { cy = xcd & (nco - 1); px = (slot << (3 - g.map_shift)) + (xcd >> g.map_shift); }
    else { px = fs_div(id, g.dMap); cy = id - px * g.map_div; }
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

  // ---- weights go straight into LDS (global_load_lds_dwordx4: no registers, no ds_write pass).  A wave instruction
  // fills 64 consecutive 16-byte units = 16 weight rows x 4 slots; the row swizzle is applied on the SOURCE side: lane
  // L of wave w, instruction i writes unit (row r = 16 w + 64 i + (L >> 2), slot L & 3) and therefore fetches slot
  // (L & 3) ^ ((r >> 2) & 3) of that row (the XOR term does not depend on i) ----
  const int wrow_bytes = p.nchunks * p.kg * 16;
  const int wr0 = wave * 16 + (lane >> 2);                       // row index within the first 64 staged rows
  const int wvoff = (wr0 % CO) * wrow_bytes + (wr0 / CO) * row_bytes + (((lane & 3) ^ ((wr0 >> 2) & 3)) << 4);
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
  uint4 rh[LH];
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
    const int coff = cc * 64;                        // scalar offset operand of the buffer loads
#pragma unroll
    for (int i = 0; i < LH; ++i)
      rh[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, hvoff[i], coff, 0));
    if constexpr (PRO == 2) {
#pragma unroll
      for (int i = 0; i < LH; ++i)
        rh2[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_src2, hvoff[i], coff, 0));
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
      const int hp = (t >> 2) + i * 64;
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
      if (64 * (i + 1) <= HMAX || hp < HMAX) lds[hp * HS + q4] = u;
    }
  };
  auto dma_weights = [&](int cc) {
    const char* wp = reinterpret_cast<const char*>(wsel ? p.wgt2 : p.wgt) + wbase + cc * 64;
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      if (i * 64 + wave * 16 < 9 * CO) {            // (wave-uniform: the last instruction of a 32-channel tile is half empty)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(wp + (long)i * (64 / CO) * row_bytes + wvoff),
            (__attribute__((address_space(3))) void*)(lds_w + wave * 64 + i * 256), 16, 0, 0);
      }
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

  // ---- the group's item sequence: slot (2 * (b >> 3) + grp) * 8 + (b & 7), stride 2 * gridDim (gridDim % 8 == 0:
  // a group keeps its XCD; the two groups of a block take neighbouring slots, i.e. with two channel tiles per pixel
  // tile the SAME pixels — the second halo fetch is a cache hit) ----
  int dbg_n = 0;
  auto stamp = [&]() {
    if (g.dbg && blockIdx.x == 0 && lane == 0 && wave == 0 && dbg_n < 60) g.dbg[grp * 64 + dbg_n] = __builtin_readcyclecounter();
    ++dbg_n;
  };
  stamp();
  const int bx = blockIdx.x, G = gridDim.x, stride = 2 * G;
  auto slot_of = [&](int gq) { return (((bx >> 3) * 2 + gq) << 3) + (bx & 7); };
  auto next_valid = [&](int id, T32Item& it) { while (id < nitems && !decode(id, it)) id += stride; return id; };
  auto count_stages = [&](int sl) {
    int c = 0; T32Item it;
    for (int id = sl; id < nitems; id += stride) c += decode(id, it) ? 1 : 0;
    return c * nchunk;
  };
  const int n_own = count_stages(slot_of(grp)), n_oth = count_stages(slot_of(1 - grp));
  // steps: group q stages its chunk m in step q + 2m, multiplies it in step q + 2m + 1 and finishes (last epilogue) in
  // step q + 2 * n_q; every wave of the workgroup runs the same number of barriers
  const int S = max(grp + 2 * n_own, (1 - grp) + 2 * n_oth) + 1;

  T32Item ld_it, ep_it;
  int ld_id = next_valid(slot_of(grp), ld_it), ld_cc = 0;      // the stage whose operands are in flight / in registers
  int ep_id = ld_id;
  ep_it = ld_it;
  int cm_cc = 0;                                               // channel chunk the next compute turn multiplies
  bool ep_ready = false;                                       // the last compute turn completed an item
  stamp();
  if (n_own > 0) { setup(ld_it); load_regs(0); }
  stamp();
  auto issue_next = [&]() {
    if (++ld_cc == nchunk) {
      ld_cc = 0;
      ld_id = next_valid(ld_id + stride, ld_it);
      if (ld_id < nitems) setup(ld_it);
    }
    if (ld_id < nitems) load_regs(ld_cc);
  };
  // epilogues that read tensors (addend, mask, BatchNorm input, bias) run BEFORE the next operand loads are issued:
  // vector-memory loads return in order, so behind them an epilogue load would wait for the whole prefetch
  constexpr bool kEpLoads = EP < 0 || (EP & (EP_BIAS | EP_ADDEND | EP_MASK | EP_BNB)) != 0;

  f32x16 acc[TC][TP];
#pragma unroll
  for (int a = 0; a < TC; ++a)
#pragma unroll
    for (int b = 0; b < TP; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  auto epilogue = [&](const T32Item& cur) {
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
            float sv[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) { sv[j] = s1[j]; sv[8 + j] = s2[j]; }
            const float tot = t32_reduce16(sv, lane);
            if ((lane & 16) == 0) {          // one row per wave half writes: lane holds (sum or sum-of-squares, channel j)
              const int j = ((lane >> 1) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1);
              const int cl = a * 32 + rn * 16 + hk * 8 + j;
              red((wave * CO + cl) * 2 + (lane & 1)) = tot;
            }
          }
        }
      if (has_stats) { pend_px = cur.px; pend_n = cur.n; pend_co0 = cur.co0; }
  };

  for (int s = 0; s < S; ++s) {
    stamp();
    const int r = s - grp;
    if (r >= 0) {
      flush_stats();                 // partials of an epilogue that ran before the last barrier
      const int m = r >> 1;
      if ((r & 1) == 0) {
        // ---- memory turn m: epilogue of the item completed by compute turn m - 1, stage chunk m, fetch chunk m + 1 ----
        // (halo registers -> LDS first: they landed a turn ago; then the weights of chunk m, straight into the group's LDS
        // buffer, which its own compute turn m - 1 released at the last barrier — issued the other way round the
        // compiler drains the LDS-DMA with vmcnt(0) before the first ds_write, 1 500 cycles of exposed latency)
        if (m < n_own) { store_lds(); dma_weights(ld_cc); }
        stamp();
        if constexpr (kEpLoads) {
          if (ep_ready && !(g.abl & 2)) { epilogue(ep_it); }
          stamp();
          if (m < n_own) issue_next();
        } else {
          if (m < n_own) issue_next();
          stamp();
          if (ep_ready && !(g.abl & 2)) { epilogue(ep_it); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the LDS-DMA of this turn is complete before the barrier
        if (ep_ready) { ep_ready = false; ep_id = next_valid(ep_id + stride, ep_it); }
      } else if (m < n_own) {
        // ---- compute turn m ----
        if (cm_cc == 0) {
#pragma unroll
          for (int a = 0; a < TC; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b)
#pragma unroll
              for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
        }
        if (!(g.abl & 1)) {
          // 18 k-steps (9 taps x 2 halves of the 64-byte chunk), software-pipelined by hand: the fragments of step
          // k + 1 are requested before the MFMAs of step k issue (left to itself the compiler reads, waits, multiplies:
          // the compute turn took 4 900 cycles for 2 304 cycles of MFMA)
          uint4 fa[2][TC], fb[2][TP];
          auto frags = [&](int k, int st) {
            const int tap = k >> 1, ks = k & 1;
            const int rr = tap / 3, ss = tap - rr * 3;
            const int hoff = (fwd ? (rr * HW + ss) : ((2 - rr) * HW + (2 - ss))) * HS;
#pragma unroll
            for (int a = 0; a < TC; ++a) fa[st][a] = lds_w[(ks ? wa1 : wa0) + (tap * CO + a * 32) * 4];
#pragma unroll
            for (int b = 0; b < TP; ++b) fb[st][b] = lds[hbase[b] + hoff + 2 * ks];
          };
          frags(0, 0);
#pragma unroll
          for (int k = 0; k < 18; ++k) {
            if (k + 1 < 18) frags(k + 1, (k + 1) & 1);
#pragma unroll
            for (int a = 0; a < TC; ++a)
#pragma unroll
              for (int b = 0; b < TP; ++b) Mma32<T>::run(acc[a][b], fa[k & 1][a], fb[k & 1][b]);
          }
          if constexpr (sizeof(T) == 2) {
            // pin the interleave (the scheduler otherwise sinks every read next to its use to save registers):
            // first fragment set, then one read of the next set per MFMA of the current one
            __builtin_amdgcn_sched_group_barrier(0x100, TC + TP, 0);
#pragma unroll
            for (int k = 0; k < 18; ++k) {
              if (k + 1 < 18) {
                t32_sched_step<TC + TP, TC * TP>();
              } else {
                __builtin_amdgcn_sched_group_barrier(0x008, TC * TP, 0);
              }
            }
          }
        }
        if (++cm_cc == nchunk) { cm_cc = 0; ep_ready = true; }
      }
    }
    stamp();
    t32_barrier();
  }
  stamp();
  flush_stats();
  if ((g.abl & 2) && acc[0][0][0] == 123.456f) p.stats[0] = 1.0;
}

template <typename T, int PIX, int CO, int EP, int PRO>
int d32_launch(const FsConvArgs& a, hipStream_t st) {
  T32Geom g = t32_pick_geom(a.Hd, a.Wd, PIX, t32_hmax(PIX));
  if (g.TH == 0) return FS_EINVAL;
  g.dIPG = FsDiv{0u, 0u}; g.dPRG = FsDiv{0u, 0u};
  { const char* ae = getenv("FSNET_AMD_T32_ABL"); g.abl = ae ? atoi(ae) : 0; }
  { const char* de = getenv("FSNET_AMD_T32_DBG"); g.dbg = de ? reinterpret_cast<unsigned long long*>(strtoull(de, nullptr, 0)) : nullptr; }
  if (a.stat_group_rows > 0) {
    const long hw = (long)a.Hd * a.Wd;
    if (a.stat_group_rows % hw != 0) return FS_EINVAL;
    g.dIPG = fs_make_div((int)(a.stat_group_rows / hw));
  }
  if (a.pro_group_imgs > 0) g.dPRG = fs_make_div(a.pro_group_imgs);
  const int npix = a.N * g.tiles_x * g.tiles_y, nco = a.Co_p / CO;
  int items = npix * nco;
  g.pix_major = (nco > 1 && a.src_bytes > 2 * a.wgt_bytes) ? 1 : 0;
  if (g.pix_major) items = 8 * ((npix + 7) / 8) * nco;
  else if (nco % 8 != 0 && 8 % nco == 0) { const int q = 8 / nco; items = 8 * ((npix + q - 1) / q); }
  g.nitems = items;
  g.map_shift = 0;
  if (g.pix_major) { g.map_mode = 0; g.map_div = nco; }
  else if (nco % 8 == 0) { g.map_mode = 1; g.map_div = nco >> 3; }
  else if (8 % nco == 0) { g.map_mode = 2; g.map_div = 1; while ((1 << g.map_shift) < nco) ++g.map_shift; }
  else { g.map_mode = 3; g.map_div = nco; }
  g.dMap = fs_make_div(g.map_div);
  // persistent grid: two group slots per block, whole multiples of 8 blocks (a group keeps its XCD)
  constexpr int occ = 163840 / (2 * t32_lds_units(PIX, CO) * 16) < 1 ? 1 : 163840 / (2 * t32_lds_units(PIX, CO) * 16);
  int blocks = ((items + 1) / 2 + 7) / 8 * 8;
  blocks = std::min(blocks, std::max(8, t32_cu_count() * occ / 8 * 8));
  hipLaunchKernelGGL((conv3x3_d32_kernel<T, PIX, CO, EP, PRO>), dim3(blocks), dim3(512), 0, st, a, g);
  return fs_launch_status();
}

// tile configuration (pixels x channels per wave group): 0 = 256 x 64, 1 = 128 x 64, 2 = 128 x 32, 3 = 256 x 32
template <typename T, int EP, int PRO>
int d32_dispatch_cfg(const FsConvArgs& a, int cfg, hipStream_t st) {
  switch (cfg) {
    case 0: return d32_launch<T, 256, 64, EP, PRO>(a, st);
    case 1: return d32_launch<T, 128, 64, EP, PRO>(a, st);
    case 2: return d32_launch<T, 128, 32, EP, PRO>(a, st);
    default: return d32_launch<T, 256, 32, EP, PRO>(a, st);
  }
}

double d32_waste(const FsConvArgs& a, int PIX) {
  T32Geom g = t32_pick_geom(a.Hd, a.Wd, PIX, t32_hmax(PIX));
  if (g.TH == 0) return 1e9;
  return (double)g.tiles_x * g.tiles_y * PIX / ((double)a.Hd * a.Wd);
}

int d32_pick_cfg(const FsConvArgs& a) {
  const char* fe = getenv("FSNET_AMD_D32_CFG");       // development knob (tools/probes/t32_ab.py)
  const bool c64 = a.Co_p % 64 == 0;
  if (fe) { const int c = atoi(fe); return (!c64 && (c == 0 || c == 1)) ? (c == 0 ? 3 : 2) : c; }
  const bool big = d32_waste(a, 256) <= 1.15 * d32_waste(a, 128);
  return c64 ? (big ? 0 : 1) : (big ? 3 : 2);
}

template <typename T, int PRO>
int d32_dispatch_ep(const FsConvArgs& a, int cfg, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    switch (t32_ep_mask(a)) {
      case EP_STATS: return d32_dispatch_cfg<T, EP_STATS, PRO>(a, cfg, st);
      case EP_BIAS | EP_STATS: return d32_dispatch_cfg<T, EP_BIAS | EP_STATS, PRO>(a, cfg, st);
      case EP_BIAS | EP_RELU: return d32_dispatch_cfg<T, EP_BIAS | EP_RELU, PRO>(a, cfg, st);
      case 0: return d32_dispatch_cfg<T, 0, PRO>(a, cfg, st);
      case EP_ADDEND: return d32_dispatch_cfg<T, EP_ADDEND, PRO>(a, cfg, st);
      case EP_MASK: return d32_dispatch_cfg<T, EP_MASK, PRO>(a, cfg, st);
      case EP_MASK | EP_BNB: return d32_dispatch_cfg<T, EP_MASK | EP_BNB, PRO>(a, cfg, st);
      case EP_ADDEND | EP_MASK | EP_BNB: return d32_dispatch_cfg<T, EP_ADDEND | EP_MASK | EP_BNB, PRO>(a, cfg, st);
      case EP_BNB | EP_MASKBN: return d32_dispatch_cfg<T, EP_BNB | EP_MASKBN, PRO>(a, cfg, st);
      default: break;
    }
  }
  return d32_dispatch_cfg<T, -1, PRO>(a, cfg, st);
}

template <typename T>
int d32_dispatch(const FsConvArgs& a, hipStream_t st) {
  const int cfg = d32_pick_cfg(a);
  switch (a.pro_mode) {
    case 0: return d32_dispatch_ep<T, 0>(a, cfg, st);
    case 1: return d32_dispatch_ep<T, 1>(a, cfg, st);
    case 2: return d32_dispatch_ep<T, 2>(a, cfg, st);
    default: return FS_EINVAL;
  }
}

}  // namespace

// internal entry: FS_EINVAL = "not mine"
int fs_conv3x3_d32(const FsConvArgs& a, int dtype, hipStream_t st) {
  const int es = dtype == FS_DTYPE_BF16 ? 2 : 4;
  if ((a.Cs * es) % 64 != 0 || a.Co_p % 32 != 0 || a.Co % 8 != 0) return FS_EINVAL;
  if (a.pro_mode != 0 && (!a.pro_a || !a.pro_b || (a.pro_mode == 2 && (!a.pro_c || !a.pro_src2)))) return FS_EINVAL;
  if (a.bnb_scale && (!a.bnb_x || !a.bnb_shift || a.mask)) return FS_EINVAL;
  if (dtype == FS_DTYPE_BF16) return d32_dispatch<bf16>(a, st);
  if (dtype == FS_DTYPE_F32) return d32_dispatch<float>(a, st);
  return FS_EINVAL;
}
