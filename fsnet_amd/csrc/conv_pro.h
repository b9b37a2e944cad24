// Operand prologue of the 3x3 / stride-1 LDS-halo convolution kernels (conv3x3_t32.hip, conv3x3_halo.hip): the
// BatchNorm (+ ReLU) in front of a convolution (pro_mode 1), applied to 16-byte units while they travel from the load
// registers to LDS.  FsConvArgs documents it (include/fsnet_hip.h).  (Mode 2 — the second pass of the backward of the
// BatchNorm behind the convolution, in the data gradient's staging — existed in rounds 3-4, measured slower and is gone.)  The per-channel coefficients live in a small LDS table that every block fills itself — from
// the f64 sums the producing kernel's epilogue left (bn_apply_kernel's preamble, bn.hip, same
// arithmetic: no fs_bn_finalize launch and no BatchNorm pass between two convolutions), or from coefficient arrays
// handed in.  Reference: nn.BatchNorm2d in train mode between conv1 and conv2 of a BasicBlock and its autograd
// backward, vision_base/networks/models/backbone/resnet.py:33-50.
#pragma once
#include "common.h"
#include "fsnet_hip_internal.h"

namespace {

template <int PRO> constexpr int pro_ncoef() { return PRO == 1 ? 2 : 0; }
// dynamic LDS a launch with this prologue needs
template <int PRO> inline unsigned pro_lds_bytes(const FsConvArgs& a) { return (unsigned)(pro_ncoef<PRO>() * a.Cs * sizeof(float)); }

__device__ inline void pro_sum_slots(const double* st, int C, int c, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int k = 0; k < FS_STAT_SLOTS; ++k) { s1 += st[(long)k * 2 * C + c]; s2 += st[(long)k * 2 * C + C + c]; }
}

// mode 1, channel c of statistics group g: bn_channel_coeffs of bn.hip
__device__ inline void pro1_channel(const FsConvArgs& p, int g, int c, float& mean, float& invstd, float& varb,
                                    float& sc, float& sh) {
  const int C = p.Cs;
  double s1, s2;
  pro_sum_slots(p.pro_stats + (long)g * FS_STAT_SLOTS * 2 * C, C, c, s1, s2);
  double m = s1 / p.pro_count, v = s2 / p.pro_count - m * m;
  if (v < 0) v = 0;
  mean = (float)m;
  varb = (float)v;
  invstd = (float)(1.0 / sqrt(v + (double)p.pro_eps));
  sc = p.pro_gamma[c] * invstd;
  sh = p.pro_beta[c] - mean * sc;
}

// fills tab[ncoef][Cs] for statistics group grp (all threads of the block; a barrier must follow before it is read)
template <int PRO>
__device__ inline void pro_build_table(const FsConvArgs& p, float* tab, int grp, int t, int nt) {
  const int C = p.Cs;
  if (p.pro_stats) {
    for (int c = t; c < C; c += nt) {
      float mean, istd, varb, sc, sh;
      pro1_channel(p, grp, c, mean, istd, varb, sc, sh);
      tab[c] = sc; tab[C + c] = sh;
    }
  } else {
    for (int c = t; c < C; c += nt) { tab[c] = p.pro_a[grp * C + c]; tab[C + c] = p.pro_b[grp * C + c]; }
  }
}

// what ONE block of the launch does besides its tile when the coefficients are derived in the kernel: mode 1 — saved
// statistics, the affine form for the backward's consumers, running statistics (one momentum update per group, in
// group order, like bn_finalize_kernel)
template <int PRO>
__device__ inline void pro_block0(const FsConvArgs& p, int t, int nt) {
  if (!p.pro_stats) return;
  const int C = p.Cs;
  const int G = p.pro_group_imgs > 0 ? p.N / p.pro_group_imgs : 1;
  if constexpr (PRO == 1) {
    const bool track = p.pro_running_mean != nullptr;
    for (int c = t; c < C; c += nt) {
      float rm = track ? p.pro_running_mean[c] : 0.f, rv = track ? p.pro_running_var[c] : 0.f;
      for (int g = 0; g < G; ++g) {
        float mean, istd, varb, sc, sh;
        pro1_channel(p, g, c, mean, istd, varb, sc, sh);
        if (p.pro_mean) { p.pro_mean[g * C + c] = mean; p.pro_invstd[g * C + c] = istd; }
        if (p.pro_save_a) { p.pro_save_a[g * C + c] = sc; p.pro_save_b[g * C + c] = sh; }
        const double unb = p.pro_count > 1.0 ? (double)varb * p.pro_count / (p.pro_count - 1.0) : (double)varb;
        rm = (1.f - p.pro_momentum) * rm + p.pro_momentum * mean;
        rv = (1.f - p.pro_momentum) * rv + p.pro_momentum * (float)unb;
      }
      if (track) { p.pro_running_mean[c] = rm; p.pro_running_var[c] = rv; }
    }
    if (t == 0 && p.pro_nbt) *p.pro_nbt += G;
  }
}

// argument checks shared by the two kernels' entry points
inline bool pro_args_ok(const FsConvArgs& a) {
  // (retired slots of the round 3-4 backward prologue: header rule — they stay NULL whatever the mode)
  if (a.pro_c || a.pro_m || a.pro_src2 || a.pro_stats_local || a.pro_dgamma || a.pro_dbeta || a.pro_dst) return false;
  if (a.pro_mode == 0) return true;
  if (a.pro_mode != 1) return false;
  // (the coefficient table is dynamic LDS on top of 38-78 KB of static LDS: 2 floats per source channel; the widest 3x3
  // layer of the networks has 512 — wider layers are declined here, the callers' can_fold_* then never ask)
  if (a.Cs % 4 != 0 || a.Cs > 512) return false;
  if (a.pro_group_imgs < 0 || (a.pro_group_imgs > 0 && a.N % a.pro_group_imgs != 0)) return false;
  if (a.pro_stats) {
    if (!a.pro_gamma || !a.pro_beta || !(a.pro_count > 0.0)) return false;
    if ((a.pro_mean == nullptr) != (a.pro_invstd == nullptr) || (a.pro_save_a == nullptr) != (a.pro_save_b == nullptr)) return false;
    if ((a.pro_running_mean == nullptr) != (a.pro_running_var == nullptr)) return false;
  } else {
    if (!a.pro_a || !a.pro_b) return false;
  }
  return true;
}

}  // namespace
