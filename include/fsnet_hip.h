/* fsnet_hip.h — C ABI of libfsnet_hip.so, the MI355X (gfx950) kernel library behind the
 * FSNet self-supervised monodepth training step.
 *
 * Everything here takes plain device pointers, sizes and a hipStream_t passed as void*.
 * No torch types cross this boundary.  Every function returns FS_OK (0) or an FS_E* code and
 * never blocks the host.  The reference (Owen-Liuyuxuan/FSNet) has no FFI of its own: its hot
 * path is stock PyTorch ATen calls issued from Python.  Each entry point therefore cites the
 * reference Python call site(s) whose ATen work it replaces; the Python host side
 * (fsnet_amd/...) binds these symbols with ctypes.
 *
 * Activation layout: NHWC, channel axis contiguous, explicit element strides for n/h/w so that
 * padded or sliced buffers can be addressed in place.  dtype codes select the element type of
 * activation/weight buffers (accumulation is always fp32).
 */
#ifndef FSNET_HIP_H
#define FSNET_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_OK 0
#define FS_EINVAL 1
#define FS_ELAUNCH 2

#define FS_DTYPE_F32 0
#define FS_DTYPE_BF16 1

/* per-channel f64 reduction buffers (BatchNorm statistics / backward sums) are spread over this many
 * address slots: layout [FS_STAT_SLOTS][2][C]; consumers add the slots up. */
#define FS_STAT_SLOTS 8

/* library/ABI version and the ISA the kernels were compiled for ("gfx950").  FS_ABI_VERSION changes whenever an
 * argument struct or a signature below does; a host binding refuses a library that reports another number. */
#define FS_ABI_VERSION 11
int fs_abi_version(void);
const char* fs_target_arch(void);
/* debugging aid: writes the device's constant-rate clock (wall_clock64, 100 MHz) into *slot (u64) on `stream`;
 * capturable.  No reference counterpart (the engine's FSNET_AMD_MARKS=1 step timeline). */
int fs_debug_timestamp(void* slot, void* stream);
/* host-side query, no launch: *token identifies the position of `stream` in the hipGraph capture it takes part in (a hash of
 * the nodes its next node would depend on; equal tokens <=> nothing captured on the stream in between), 0 outside a capture.
 * No reference counterpart (the engine orders its hand-overs to other streams by it: the graph executor assigns streams by
 * edge order). */
int fs_capture_position(void* stream, unsigned long long* token);

/* ------------------------------------------------------------------------------------------
 * Convolution forward / data-gradient (implicit GEMM on MFMA).
 * Replaces: nn.Conv2d forward and convolution_backward(input) issued by
 *   vision_base/networks/models/backbone/resnet.py:6-9,119,148-160,199-213
 *   vision_base/networks/blocks/blocks.py:41-54
 *   monodepth/networks/models/heads/depth_encoder.py:45-63,123-139
 *   monodepth/networks/models/heads/pose_decoder.py:17-21,26-37
 * ktab: one int2 per K *unit* (kg=4: one 16-byte group = 8 bf16 / 4 f32 channels; kg=8: four groups =
 * 64 bytes of one tap): { byte delta = (r'*(sH>>dshift) + s'*(sW>>dshift) + c) * sizeof(T)  (INT32_MIN = K
 * padding),  (r' & 0xffff) | s' << 16 } with (r', s') = sgn*(r, s).  Forward: src = input, rows = output
 * pixels, hb = y*hb_mul + hb_add (stride, -pad), sgn=+1.  Dgrad: src = dY, rows = input pixels, hb_mul=1,
 * hb_add=+pad, sgn=-1, dshift = log2(stride) with a parity test.  Operands are fetched with raw buffer
 * loads bounded by src_bytes / wgt_bytes (out-of-range = zero padding).  Epilogue: + bias, + addend, relu, optional
 * per-channel f64 sum / sum-of-squares (BatchNorm batch statistics), ReLU-backward mask.
 */
typedef struct FsConvArgs {
  const void* src;
  const void* wgt;      /* packed [Co_p][nchunks*kg*16 bytes], K contiguous */
  void* dst;
  const float* bias;    /* [Co] or NULL */
  const void* addend;   /* same dtype as src, or NULL */
  const void* mask;     /* same dtype as src, or NULL: out = mask > 0 ? out : 0 (ReLU backward) */
  double* stats;        /* [FS_STAT_SLOTS][2][Co] or NULL */
  const int* ktab;      /* int2 [nchunks * (kg==8 ? 2 : 4)] */
  int64_t sN, sH, sW;   /* src strides (elements) */
  int64_t dN, dH, dW;   /* dst strides */
  int64_t aN, aH, aW;   /* addend strides */
  int64_t mN, mH, mW;   /* mask strides */
  int64_t src_bytes;    /* addressable span from src (< 2 GiB) */
  int64_t wgt_bytes;    /* bytes of the packed weight operand */
  int32_t Hs, Ws;       /* src spatial size */
  int32_t Hd, Wd;       /* row-domain (dst) spatial size */
  int32_t M;            /* N*Hd*Wd */
  int32_t Co;           /* dst channels (multiple of 4) */
  int32_t Co_p;         /* packed weight rows (multiple of 16) */
  int32_t nchunks;      /* K stages; packed weight row = nchunks*kg*16 bytes */
  int32_t kg;           /* 16-byte K groups per stage: 4 or 8 (8 needs Co_p % 32 == 0) */
  int32_t hb_mul, hb_add, sgn, dshift;
  int32_t relu;
  int32_t out_f32;      /* store fp32 regardless of dtype */
  int32_t N;            /* batch size (fs_conv3x3_halo) */
  int32_t Cs;           /* source channels per tap (fs_conv3x3_halo) */
  int64_t wgt_row_bytes; /* 0: packed weight rows are nchunks*kg*16 bytes apart.  Else the row stride in bytes: wgt
                            then points at a K slice of a wider operand (one parity class of a stride-2 dgrad). */
  int32_t grp_imgs;      /* 0, or (fs_conv_igemm) images per BatchNorm statistics group when the launch carries one
                            group per blockIdx.z: M is then the row count of ONE group and N the total batch.  Used
                            instead of stat_group_rows when a group's rows are not a multiple of 256. */
  int32_t ncls;          /* <= 1: one problem.  2..4 (fs_conv_igemm): blockIdx.y selects an output-parity class of a
                            stride-2 data gradient: K stages cls_nch[c], unit table ktab + cls_ktab_off[c] (int2
                            units), operand wgt + cls_wgt_off[c] bytes, and dst / addend / mask / bnb_x advanced by
                            (c>>1)*H_stride/2 + (c&1)*W_stride/2 (their strides are those of the sub-lattice). */
  int32_t cls_nch[4];
  int32_t cls_ktab_off[4];
  int64_t cls_wgt_off[4];
  const void* bnb_x;    /* NULL: stats = (sum v, sum v^2) of the outputs (BatchNorm forward statistics).
                           Else: the raw conv output (src dtype, addressed with the dst strides dN/dH/dW: same
                           layout as dst) of the BatchNorm whose input
                           gradient this launch produces, and stats = (sum v, sum v*xhat), xhat =
                           (bnb_x - bnb_mean) * bnb_invstd: the first pass of BatchNorm backward
                           (fs_bn_bwd_reduce) fused into the data-gradient epilogue. */
  const float* bnb_mean; const float* bnb_invstd;   /* [groups][Co] saved statistics of that BatchNorm */
  int32_t stat_group_rows; /* 0: one statistics group.  >0: rows (pixels) per BatchNorm statistics group; row m
                              adds to stats + (m / stat_group_rows) * FS_STAT_SLOTS*2*Co.  Groups are whole
                              images and, for fs_conv_igemm, stat_group_rows % 256 == 0 (no tile straddles). */
  /* ---- fs_conv3x3_halo only ----
   * Operand prologue: the BatchNorm (+ ReLU) that precedes this convolution in the reference (resnet.py:33-50 conv1 ->
   * bn1 -> relu -> conv2; blocks.py:41-54) is applied to the source operand while it is staged into LDS, so the normalised
   * activation is not produced by a pass of its own.  Coefficients are fp32 [groups][Cs], group = image / pro_group_imgs
   * (0: one group).  Out-of-image taps stay zero (the padding applies to the transformed tensor).
   *   pro_mode 1: x' = a[c]*x + b[c], then max(x', 0) if pro_relu          (a = gamma*invstd, b = beta - mean*a)
   * Coefficients come from one of two places:
   *   pro_stats == NULL: pro_a / pro_b are read as given (fs_bn_finalize made them).
   *   pro_stats != NULL (ABI 6): every block derives them itself from the f64 sums [groups][FS_STAT_SLOTS][2][Cs] — (sum x,
   *     sum x^2) of src as the producing convolution's epilogue left them (after the data-parallel exchange), with
   *     pro_gamma / pro_beta / pro_count / pro_eps: bn_apply's preamble, no fs_bn_finalize launch; block 0 also writes
   *     pro_mean / pro_invstd / pro_save_a / pro_save_b ([groups][Cs]: saved for the backward) and updates
   *     pro_running_mean / pro_running_var / pro_nbt (momentum pro_momentum, once per group in group order).
   * (ABI 5-7 also had pro_mode 2 — the second pass of the backward of the BatchNorm that FOLLOWS the convolution applied
   * in a data-gradient launch's staging; measured slower than the pass it replaced and removed in ABI 8.  Its fields —
   * pro_c, pro_m, pro_src2, pro_stats_local, pro_dgamma, pro_dbeta, pro_dst — keep their slots and must be NULL.) */
  const float* pro_a; const float* pro_b; const float* pro_c; const float* pro_m;
  const void* pro_src2;
  int32_t pro_mode, pro_relu, pro_group_imgs;
  /* fs_conv3x3_halo only, for tests: 0 = the entry point chooses its kernel (always, in the product); 1 = 16x16-tile
   * kernel; 2-4 = 32x32-tile kernel, tile configuration 1-3 (FS_EINVAL where it cannot take the launch); 5 = persistent
   * one-chunk kernel whatever the tile count (shapes it does not cover: the usual choice).  (ABI 10; was reserved1.) */
  int32_t force_impl;
  /* ReLU-backward mask derived instead of read: with bnb_x set and bnb_scale != NULL the mask is
   * bnb_scale[c]*bnb_x + bnb_shift[c] > 0 (the folded forward's own expression; [groups][Co]) and `mask` must be NULL */
  const float* bnb_scale; const float* bnb_shift;
  const double* pro_stats; const double* pro_stats_local;
  const float* pro_gamma; const float* pro_beta;
  float* pro_mean; float* pro_invstd;
  float* pro_save_a; float* pro_save_b;
  float* pro_running_mean; float* pro_running_var;
  int64_t* pro_nbt;
  float* pro_dgamma; float* pro_dbeta;
  void* pro_dst;
  double pro_count;
  float pro_eps, pro_momentum;
  /* ---- fs_conv3x3_s2d only (ABI 8): the data gradient of the block's 1x1 / stride-2 downsample projection in the same
   * launch.  ds_src: its dY, same shape, strides and dtype as src; ds_wgt: its packed data-gradient operand
   * Wt[ci][co] with rows ds_wgt_row_bytes apart.  dst then receives dgrad(conv1) + dgrad(downsample) (+ addend). */
  const void* ds_src; const void* ds_wgt;
  int64_t ds_wgt_row_bytes;
} FsConvArgs;
int fs_conv_igemm(const FsConvArgs* args, int dtype, void* stream);

/* ---- Two problems per launch (ABI 7) ----
 * The reference runs the depth encoder on image 0 and the pose encoder on each (source, target) pair
 * (monodepth2_model.py:24-43: self.depth_backbone(...), self.pose_backbone(torch.cat(...))); after their stems the two
 * ResNets execute the same layer shapes (resnet.py:199-213), and nothing orders them before the loss.  Every fs_*2 entry
 * point below takes the arguments of TWO independent launches of its one-problem namesake — own tensors, weights,
 * statistics, batch sizes and statistics groups — and runs them as ONE kernel launch (the blocks of the second problem
 * follow the first's in the grid), so a pair of layers pays a launch's fixed cost once.  a1 == NULL is the one-problem
 * call.  When the two problems do not agree on what selects a kernel instantiation (shapes, epilogue options), the entry
 * point launches them one after the other: results never depend on whether a pair shared a launch. */
int fs_conv_igemm2(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, void* stream);
/* 1x1 convolutions (forward; data gradient at stride 1) as a row-streaming GEMM: same FsConvArgs and epilogue semantics
 * as fs_conv_igemm (ktab unused), bf16 only.  Returns FS_EINVAL for anything it does not take (fp32, padding, a
 * stride-2 data gradient, a K extent that is not whole 64-byte steps): the caller then uses fs_conv_igemm.
 * Replaces the Bottleneck 1x1 convolutions and downsample projections, resnet.py:52-89, 119. */
int fs_conv1x1(const FsConvArgs* args, int dtype, void* stream);

/* 3x3 / stride-1 specialisation (forward: hb_mul=1, hb_add=-pad, sgn=+1; dgrad: hb_add=+pad, sgn=-1) with an
 * LDS-resident input halo tile reused by all nine taps.  Same arguments, packed weights and epilogue as
 * fs_conv_igemm (ktab is not used); requires Cs*sizeof(T) % 64 == 0 and Co_p % 32 == 0.  The destination, addend and mask
 * views must each span fewer than 2^31 elements (32-bit element offsets in the kernels): FS_EINVAL otherwise.
 */
int fs_conv3x3_halo(const FsConvArgs* args, int dtype, void* stream);
/* the launch fs_conv3x3_halo would make for these arguments, nothing launched (pointers are only tested against NULL):
 * plan = {kernel (0: 16x16-MFMA-tile kernel, 1: 32x32-MFMA-tile kernel, 2: persistent one-chunk kernel), blocks, pixels per block tile, output channels per
 * block tile}.  bench.py splits the family's time by kernel with it; tests check that a forced configuration is the one that
 * runs. */
int fs_conv3x3_halo_plan(const FsConvArgs* args, int dtype, int32_t* plan);
/* two problems in one launch (see fs_conv_igemm2); the plan of the shared launch (FS_EINVAL if the pair would run as two) */
int fs_conv3x3_halo2(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, void* stream);
int fs_conv3x3_halo2_plan(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, int32_t* plan);

/* ---- Data gradient of a 3x3 / stride-2 / pad-1 convolution, all four output-parity classes from one staged dY halo (ABI 8)
 * Replaces convolution_backward(input) of the ResNet stage entries (resnet.py:33-50 with stride 2, instantiated at
 * resnet.py:199-213 as layer2/3/4[0].conv1).  Takes the class launch's arguments of fs_conv_igemm unchanged (ncls = 4:
 * src = dY [N,Hs,Ws,Cs], wgt = Wt[ci][class-ordered tap][co] with row stride wgt_row_bytes, dst / addend / mask / bnb_x =
 * the sub-lattice of class (0,0) with the doubled strides; Hd = Hs, Wd = Ws) and produces the same tensor; epilogue
 * options: addend, mask, bnb_x + stats (BatchNorm-backward sums per statistics group of stat_group_rows rows).
 * FS_EINVAL: not taken (ragged channel chunks, other epilogues) — the caller's chain continues with fs_conv_igemm. */
int fs_conv3x3_s2d(const FsConvArgs* args, int dtype, void* stream);
int fs_conv3x3_s2d2(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, void* stream);

/* 7x7 / stride-2 / pad-3 stem (resnet.py:118-121: conv1 of both encoders) over 8-channel bf16 pixels, forward only:
 * weights resident in LDS, im2col from an LDS input patch, persistent blocks.  Same arguments and packed forward
 * operand as fs_conv_igemm (ktab unused); requires Cs == 8, Co == Co_p == 64, bf16 output, no bias / addend / mask /
 * relu; stats and stat_group_rows as in fs_conv_igemm. */
int fs_conv_stem(const FsConvArgs* args, int dtype, void* stream);
/* both encoders' stems (3 and 6 real input channels of the same 8-channel pixels) in one launch (see fs_conv_igemm2) */
int fs_conv_stem2(const FsConvArgs* a0, const FsConvArgs* a1, int dtype, void* stream);

/* Convolution weight gradient.  Replaces convolution_backward(weight) at the same call sites.
 * dy is dense [M][Cd]; x is the forward input (strided NHWC); dw is the fp32 OIHW gradient
 * [Co][Ci][R][S] and is accumulated into (+=, deterministic).  ktab as above, one entry
 * per 16-byte group of GEMM columns (r, s, ci).
 */
typedef struct FsWgradArgs {
  const void* dy;
  const void* x;
  float* dw;
  const int* ktab;
  int64_t sN, sH, sW;   /* x strides */
  int32_t Hs, Ws;       /* x spatial size */
  int32_t Hd, Wd;       /* dy spatial size */
  int32_t M;            /* N*Hd*Wd */
  int32_t Cd;           /* dy channels (padded, multiple of 16) */
  int32_t Co, Ci, R, S; /* real weight dims */
  int32_t stride, pad;
  int32_t ncolgroups;   /* R*S*Cs / EG */
  float* workspace;     /* split-K partial slabs (or NULL: no split) */
  int64_t workspace_elems;
  int64_t x_bytes;      /* addressable span from x (< 2 GiB) — enables the 3x3/s1 LDS-halo path */
  int32_t use_halo;     /* 1: allow the 3x3/s1 LDS-halo kernel (bf16) */
  int32_t pix_per_split, nsplit, ws_rows, ws_cols;  /* filled by the library */
  /* (ABI 4) operand prologue of the 3x3/s1 LDS-halo kernel: x' = max(pro_a[c]*x + pro_b[c], 0 if pro_relu) for pixels
   * inside the image, 0 for the padding — x is then the RAW output of the convolution in front of the BatchNorm + ReLU
   * whose result this weight gradient multiplies (fs_bn_finalize).  [groups][Ci] fp32, group = image / pro_group_imgs
   * (0: one group).  fs_conv_wgrad returns FS_EINVAL if the shape is not one the halo kernel takes. */
  const float* pro_a; const float* pro_b;
  int32_t pro_relu, pro_group_imgs;
} FsWgradArgs;
int fs_conv_wgrad(const FsWgradArgs* args, int dtype, void* stream);
/* the launch fs_conv_wgrad would make for these arguments, nothing launched: plan = {kernel (0 generic tile, 1 3x3 LDS-halo,
 * 2 narrow 16/32-channel, 3 7x7 stem), blocks, threads per block, blocks of that kernel the device holds at once}.
 * A split-K grid a little above the resident count runs as two rounds; tests hold the grids to one. */
int fs_conv_wgrad_plan(const FsWgradArgs* args, int dtype, int32_t* plan);
/* the weight gradients of two convolutions of the same shape in one launch (see fs_conv_igemm2): the pixel splits of both
 * share the device's block slots and a0's workspace (slab regions back to back: a1's workspace is not used), one reduce
 * launch adds into both dW.  plan: blocks of the shared launch. */
int fs_conv_wgrad2(const FsWgradArgs* a0, const FsWgradArgs* a1, int dtype, void* stream);
int fs_conv_wgrad2_plan(const FsWgradArgs* a0, const FsWgradArgs* a1, int dtype, int32_t* plan);
/* (ABI 9 had fs_wgrad_batch_begin / _end — batched slab reductions, measured slower on the step; removed in ABI 10.) */

/* Weight packing.  OIHW fp32 master weights (the reference's state_dict layout,
 * e.g. depth_backbone.conv1.weight (64,3,7,7)) -> [rows_p][ktot_p] K-contiguous
 * MFMA operand, K = (r, s, c) with c padded to cs_p.  transpose=0: rows = co, c = ci (forward);
 * transpose=1: rows = ci, c = co (dgrad).
 */
int fs_pack_weights(const float* w_oihw, void* dst, int Co, int Ci, int R, int S, int rows_p,
                    int cs_p, int64_t ktot_p, int transpose, int dtype, void* stream);

/* The same for every convolution of a model in ONE launch.  descs lives in device memory; block_start is the
 * exclusive prefix sum of ceil((rows_f*k_f + rows_d*k_d) / 256); dst_d may be NULL with rows_d = 0.
 */
typedef struct FsPackDesc {
  const float* w;
  void* dst_f;
  void* dst_d;
  int64_t k_f, k_d;       /* padded K of the forward / dgrad operand */
  int64_t block_start;
  int32_t Co, Ci, R, S;
  int32_t rows_f, cs_f;   /* forward operand: padded rows (co), padded channels per tap (ci) */
  int32_t rows_d, cs_d;   /* dgrad operand:  padded rows (ci), padded channels per tap (co) */
  int32_t tap_order_d;    /* 0: taps of the dgrad operand in (r,s) order; 1: 3x3 stride-2 parity-class order
                             [4][3,5][1,7][0,2,6,8] (fs_pack_weights: transpose = 2) */
} FsPackDesc;
int fs_pack_weights_multi(const FsPackDesc* descs_dev, int n, int64_t total_blocks, int dtype, void* stream);
/* blocks of fs_pack_weights_multi that one layer occupies (32 co x IB ci x R*S tiles): FsPackDesc.block_start is the
 * running sum of this over the table, total_blocks the grand total.  -1 for unsupported shapes (R*S > 288). */
int64_t fs_pack_tile_blocks(int Co, int Ci, int R, int S);

/* Self-distillation.  Uncertainty head: u = sigmoid(channel 0 of the fp32 NHWC conv output with Cp
 * channels) (depth_encoder.py:186) and its backward into a Cp-channel gradient in the compute dtype (channels
 * 1..Cp-1 zero).  Distillation loss (monodepth2_decoder.py:185-203, is_unscaled_distill = False): *sum_out +=
 * sum |teacher - pred| / u + log(u + 1e-5)  (uncertain == NULL: sum |teacher - pred|); the caller divides by n.
 * Backward for loss = sum / n with upstream gradient *gout (NULL = 1): d_pred, d_uncertain are overwritten. */
int fs_sigmoid_head_fwd(const float* logits, float* u, int64_t M, int Cp, void* stream);
int fs_sigmoid_head_bwd(const float* u, const float* du, void* dl, int64_t M, int Cp, int dtype, void* stream);
int fs_distill_fwd(const float* pred, const float* teacher, const float* uncertain, int64_t n, double* sum_out,
                   void* stream);
int fs_distill_bwd(const float* pred, const float* teacher, const float* uncertain, int64_t n, const double* gout,
                   float* d_pred, float* d_uncertain, void* stream);

/* Training input pipeline: raw uint8 frames -> network / loss inputs in one launch.  Per sample b
 * and frame f: cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) with the inverted map minv[b] (f64, computed by the host
 * exactly as OpenCV inverts the forward matrix), RandomMirror, the colour chain in the drawn order, Normalize and the
 * HWC -> CHW transpose (vision_base/data/augmentations/augmentations.py:50-109, 200-226, 377-497, 527-591;
 * configs/kitti_wpose_example:129-155).  Plans (host-drawn, one row per sample):
 *   iplan[b] = { op0, op1, op2 (0 brightness, 1 contrast, 2 RGB->HSV->RGB round trip, 3 nothing; execution order),
 *                bitmask (1 brightness drawn, 2 contrast drawn, 4 saturation drawn, 8 round trip present),
 *                mirror, src_h, src_w, 0 }
 *   fplan[b] = { brightness delta, contrast alpha, saturation ratio, 0 }
 * src [B][F][Hs][Ws][3] uint8 (RGB, frames of sample b occupy the top-left src_h x src_w);
 * image / original [F][B][3][H][W] fp32 (either may be NULL); mask [B][H][W] f64 = warped + mirrored patched_mask
 * (NULL to skip).  mean / std apply to `image`; `original` is x / 255. */
#define FS_AUG_IPLAN 8
#define FS_AUG_FPLAN 4
typedef struct FsAugArgs {
  const uint8_t* src;
  const double* minv;
  const int32_t* iplan;
  const float* fplan;
  float* image;
  float* original;
  double* mask;
  float mean[3];
  float std[3];
  int32_t B, F, Hs, Ws, H, W;
} FsAugArgs;
int fs_augment_frames(const FsAugArgs* args, void* stream);

/* Validation input (configs/kitti_wpose_example:156-166): Resize (augmentations.py:112-198: cv2.resize INTER_LINEAR of
 * the float image to rh x rw, then zero pad / crop to H x W) + Normalize + CHW.  dims[b] = { src_h, src_w, rh, rw };
 * src as in FsAugArgs; image [F][B][3][H][W] fp32. */
typedef struct FsResizeArgs {
  const uint8_t* src;
  const int32_t* dims;
  float* image;
  float mean[3];
  float std[3];
  int32_t B, F, Hs, Ws, H, W;
  /* Training use of the same Resize (configs/multi_dataset_example:178-205 and the nusc / kitti360_fisheye examples:
   * Resize, Shuffle of the colour ops, RandomMirror, two Normalizes): iplan / fplan as in FsAugArgs (slots 5, 6
   * unused), original [F][B][3][H][W] = the resized frame / 255, mask [B][H][W] f64 = patched_mask (all ones)
   * resized with INTER_NEAREST and padded.  All NULL for the validation input. */
  const int32_t* iplan;
  const float* fplan;
  float* original;
  double* mask;
} FsResizeArgs;
int fs_resize_frames(const FsResizeArgs* args, void* stream);

/* Evaluation.  fs_resize_linear: single-channel fp32 [h][w] -> [H][W] with OpenCV's INTER_LINEAR
 * rule; invert != 0 resizes the inverse: dst = 1 / resize(1 / src)  (base_evaluation_hooks.py:57).
 * fs_depth_eval: per image b, pred [B][h][w] (resized on the fly to the ground truth's [H][W]) against gt [B][H][W]:
 * mask 1e-3 < gt < 80 inside the Garg crop, ratio = median(gt) / median(pred), clamp to [1e-3, 80], the seven errors of
 * compute_errors for the scaled and for the unscaled prediction (kitti_unsupervised_eval.py:47-80,
 * monodepth_utils.py:271-289).  out16[b] = { ratio, err[7], abs_err[7], n_valid }; n_valid == 0 leaves zeros.
 * scratch: B*H*W*8 bytes of device memory (the valid (gt, pred) pairs are compacted there once). */
int fs_resize_linear(const float* src, float* dst, int h, int w, int H, int W, int invert, void* stream);
int fs_depth_eval(const float* pred, const float* gt, int B, int h, int w, int H, int W, void* scratch,
                  double* out16, void* stream);

/* n <= FS_COPY_MAX device-to-device copies (16-byte aligned pointers, any byte counts) in one launch: the training
 * hook stages a batch into the static input buffers of its captured hipGraph with it
 * (base_training_hooks.py:28-31 does one .cuda() per tensor). */
#define FS_COPY_MAX 16
int fs_copy_multi(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream);
/* n (<= FS_COPY_MAX) device buffers (16-byte aligned) zeroed in one launch: the per-step scratch (BatchNorm statistics
 * pools, loss accumulators, depth-gradient maps) that torch.Tensor.zero_() calls cleared one by one along the step
 * (base_training_hooks.py:34 optimizer.zero_grad() is the reference's only such call: the rest is engine scratch). */
int fs_zero_multi(void* const* dst, const int64_t* bytes, int n, void* stream);

/* Batch images NCHW fp32 (one tensor, or two concatenated along C as the pose encoder input,
 * monodepth2_model.py:29-35) -> NHWC with Cp >= Ca+Cb zero-padded channels.
 */
int fs_nchw_to_nhwc(const float* a, const float* b, void* dst, int N, int Ca, int Cb, int H, int W,
                    int Cp, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Train-mode BatchNorm forward: y = relu?( bn(x) [+ res | + bn2(res)] ).
 * Replaces nn.BatchNorm2d(train)+relu+residual at resnet.py:33-50,70-89,201-203 and
 * blocks.py:44-52.  stats = f64 [FS_STAT_SLOTS][2][C] (sum, sumsq) from fs_conv_igemm; count = elements per
 * channel (global count under SyncBatchNorm, scripts/train.py:101).  Block 0 updates the running
 * statistics (momentum, unbiased variance) and num_batches_tracked and saves mean / invstd.
 * stats == NULL selects eval mode: normalise with running_mean / running_var (no updates).
 * pad_out: y is [N,H+2,W+2,C] and receives a replicated border (consumer conv pads 'replicate',
 * depth_encoder.py:59,62).
 */
typedef struct FsBnApplyArgs {
  const void* x;        /* dense [M][C] raw conv output */
  const void* res;      /* dense [M][C] residual (or raw downsample conv output when stats2) */
  void* y;
  const double* stats;  const double* stats2;
  const float* gamma;   const float* beta;
  const float* gamma2;  const float* beta2;
  float* running_mean;  float* running_var;
  float* running_mean2; float* running_var2;
  int64_t* num_batches_tracked; int64_t* num_batches_tracked2;
  float* save_mean;  float* save_invstd;
  float* save_mean2; float* save_invstd2;
  double count;
  float eps, momentum;
  int64_t yN, yH, yW;
  int32_t M, C, H, W;
  int32_t relu, pad_out;
  int32_t groups;       /* 0/1: one batch.  G>1: the batch is G independent BatchNorm invocations stacked along N
                           (the pose encoder's two image pairs, monodepth2_model.py:29-35): statistics, save_mean /
                           save_invstd are [G][...], count is per group, running statistics are updated G times
                           in group order. */
  /* ABI 8: the stem's MaxPool2d(3, 2, 1) in the same pass (resnet.py:201-206: relu(bn1(.)) -> features[0] -> maxpool).
   * pool_y != NULL: [N,H/2,W/2,C] receives the pooled activation and pool_idx the argmax codes (r*3+s, as fs_maxpool_fwd);
   * y may then be NULL (the activation itself is not stored: the pose encoder's features[0] has no reader).  Train mode,
   * even H x W, dense y, no residual. */
  void* pool_y; uint8_t* pool_idx;
} FsBnApplyArgs;
int fs_bn_apply(const FsBnApplyArgs* args, int dtype, void* stream);
/* two BatchNorms of the same width in one launch (see fs_conv_igemm2): bn1 of the depth encoder's block and bn1 of the
 * pose encoder's, each with its own gamma / beta / running statistics and statistics groups */
int fs_bn_apply2(const FsBnApplyArgs* a0, const FsBnApplyArgs* a1, int dtype, void* stream);
/* The statistics part of fs_bn_apply alone: save_mean / save_invstd, the affine form scale = gamma*invstd, shift =
 * beta - mean*scale ([groups][C] each) and the running-statistics update (x, res, y and the second BatchNorm's fields
 * are ignored).  The activation is then normalised by whoever reads it: FsConvArgs.pro_mode = 1 in the consuming
 * convolution (resnet.py:33-50: conv1 -> bn1 -> relu -> conv2), FsWgradArgs.pro_a in the weight gradient that takes
 * it as its input operand, bnb_scale / bnb_shift for the ReLU mask of the data gradient that flows back into it. */
int fs_bn_finalize(const FsBnApplyArgs* args, float* scale, float* shift, void* stream);

/* BatchNorm backward, two passes (the data-parallel host all-reduces `sums` in between):
 *   reduce: sums[slot][0][c] += sum g, sums[slot][1][c] += sum g*xhat, g = dout * (y > 0 if relu)
 *           (sums is f64 [FS_STAT_SLOTS][2][C], zeroed by the caller)
 *   apply : dx = gamma*invstd*(g - sums0/count - xhat*sums1/count); dgamma += sums1, dbeta += sums0
 *           (from sums_local when given); optional g_out = g (gradient of the residual branch).
 * fold: dout is a replicate-padded [N,H+2,W+2,C] gradient whose border folds onto the edge pixels.
 */
typedef struct FsBnBwdArgs {
  const void* dout; const void* y; const void* x;
  void* dx; void* g_out;
  double* sums; const double* sums_local;
  const float* gamma; const float* save_mean; const float* save_invstd;
  float* dgamma; float* dbeta;
  double count;
  int64_t gN, gH, gW;
  int64_t yN, yH, yW;
  int32_t M, C, H, W;
  int32_t relu, fold;
  int32_t groups;       /* as FsBnApplyArgs.groups: sums / save_mean / save_invstd are [G][...]; dgamma, dbeta add up */
  /* ABI 8: the gradient w.r.t. the activation gathered instead of read (the backward of the fused stem pass above):
   * pool_dy != NULL: g = maxpool backward of pool_dy [N,H/2,W/2,C] through the argmax codes pool_idx (fs_maxpool_bwd's
   * gather) + dout when dout != NULL (dense [N,H,W,C]: the gradient of the un-pooled features[0]); the ReLU mask
   * (relu = 1) from y when y != NULL, else the sign of gamma*invstd*x + (beta - mean*gamma*invstd): beta required. */
  const void* pool_dy; const uint8_t* pool_idx; const float* beta;
} FsBnBwdArgs;
int fs_bn_bwd_reduce(const FsBnBwdArgs* args, int dtype, void* stream);
int fs_bn_bwd_apply(const FsBnBwdArgs* args, int dtype, void* stream);
int fs_bn_bwd_reduce2(const FsBnBwdArgs* a0, const FsBnBwdArgs* a1, int dtype, void* stream);
int fs_bn_bwd_apply2(const FsBnBwdArgs* a0, const FsBnBwdArgs* a1, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling / upsampling / concat (NHWC, dense).
 * fs_maxpool_*: nn.MaxPool2d(3,2,1), resnet.py:122,206 (idx = argmax code r*3+s per element;
 *   backward adds `addend`, the gradient of the un-pooled skip feature).
 * fs_upcat_pad_*: F.interpolate(nearest, x2) + torch.cat(skip) (depth_encoder.py:126-133),
 *   written once into a replicate-padded [N,2h+2,2w+2,Ca+Cb] buffer for the 'replicate' conv
 *   that follows (depth_encoder.py:59); backward folds the border and splits the channels.
 * fs_channel_sum: out[c] += sum over rows (conv bias gradient).
 */
int fs_maxpool_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int dtype, void* stream);
int fs_maxpool_bwd(const void* dy, const uint8_t* idx, const void* addend, void* dx, int N, int H, int W, int C,
                   int dtype, void* stream);
/* a second tensor of N1 images with the same H, W, C in the same launch (x1 / dy1 == NULL: one tensor) */
int fs_maxpool_fwd2(const void* x, void* y, uint8_t* idx, int N, const void* x1, void* y1, uint8_t* idx1, int N1, int H,
                    int W, int C, int dtype, void* stream);
int fs_maxpool_bwd2(const void* dy, const uint8_t* idx, const void* addend, void* dx, int N, const void* dy1,
                    const uint8_t* idx1, const void* addend1, void* dx1, int N1, int H, int W, int C, int dtype,
                    void* stream);
int fs_upcat_pad_fwd(const void* a, const void* b, void* out, int N, int h, int w, int Ca, int Cb, int dtype,
                     void* stream);
int fs_upcat_pad_bwd(const void* dpad, void* da, void* db, int N, int h, int w, int Ca, int Cb, int dtype,
                     void* stream);
/* fs_upcat_pad_bwd + the first pass of the BatchNorm backward its `da` half feeds (the level's upconv(i,0) ConvBnReLU,
 * depth_encoder.py:126-133 backwards; ABI 10): da is stored masked by y > 0 (y, x: dense [N,h,w,Ca] activation and raw
 * convolution output of that BatchNorm; mean / invstd [Ca]) and sums [FS_STAT_SLOTS][2][Ca] += (sum g, sum g * xhat) of the
 * stored values — fs_bn_bwd_apply then runs as after a fused data gradient (dout already masked, sums accumulated). */
int fs_upcat_pad_bwd_bn(const void* dpad, void* da, void* db, int N, int h, int w, int Ca, int Cb, const void* y,
                        const void* x, const float* mean, const float* invstd, double* sums, int dtype, void* stream);
int fs_channel_sum(const void* x, float* out, int64_t M, int C, int Creal, int dtype, void* stream);
/* n (<= 16) column sums in one launch: the bias gradients of a hand-off batch of weight gradients (each convolution bias
 * gradient = sum over N,H,W of its dY: autograd of nn.Conv2d(bias=True), blocks.py:41-46, depth_encoder.py:45-63). */
int fs_channel_sum_multi(const void* const* x, float* const* out, const int64_t* M, const int32_t* C, const int32_t* Creal,
                         int n, int dtype, void* stream);

/* Depth-bin head: logits fp32 [M][Cl] (first K used) -> depth, disp (fp32 [M]).
 * Replaces MultiChannelDepthDecoder._gather_activation / gather_output (depth_encoder.py:76-88,
 * 114-121) and depth_to_disp (monodepth_utils.py:19-24).  Backward takes dL/d depth and dL/d disp
 * (either may be NULL) and writes dL/d logits in `dtype`.
 */
int fs_depth_head_fwd(const float* logits, const float* bins, float* depth, float* disp, int64_t M, int K, int Cl,
                      float min_depth, float max_depth, void* stream);
int fs_depth_head_bwd(const float* logits, const float* bins, const float* d_depth, const float* d_disp,
                      void* dlogits, int64_t M, int K, int Cl, float min_depth, float max_depth, int dtype,
                      void* stream);

/* The same for up to FS_HEAD_MAX decoder scales in one launch (entries 0..n-1 are used; the backward leaves depth /
 * disp unused, the forward d_depth / d_disp / dlogits; a NULL d_depth or d_disp means "no gradient from that output"). */
#define FS_HEAD_MAX 4          /* = the array extents below */
typedef struct FsHeadBatch {
  const float* logits[4];
  float* depth[4];
  float* disp[4];
  const float* d_depth[4];
  const float* d_disp[4];
  void* dlogits[4];
  int64_t M[4];
  int32_t n;
  int32_t nimg;        /* images in the batch (rows of P2); only read when P2 is given */
  const float* P2;     /* [nimg][12] intrinsics or NULL.  With P2: depth *= P2[n][0] / base_fx and the disparity is taken
                          against (min_depth, max_depth) x that scale (DepthDecoder._get_scale / gather_output,
                          depth_encoder.py:36-43,115-121; configs/multi_dataset_example:256) */
  float base_fx;
  int32_t reserved;
} FsHeadBatch;
int fs_depth_head_fwd_multi(const FsHeadBatch* batch, const float* bins, int K, int Cl, float min_depth,
                            float max_depth, void* stream);
int fs_depth_head_bwd_multi(const FsHeadBatch* batch, const float* bins, int K, int Cl, float min_depth,
                            float max_depth, int dtype, void* stream);

/* Pose tail: x = last pose conv output fp32 [B][hw][Cx]; mean over hw, x0.01, split into
 * axisangle / translation [B][nframes][3] (pose_decoder.py:39-45) and the 4x4 transform of frame 0
 * (transformation_from_parameters, monodepth_utils.py:31-63,298-337; invert for negative frame ids,
 * monodepth2_model.py:42-43).  `scale` is the 0.01 of pose_decoder.py:41 (1.0 with hw = 1 turns the
 * pair into a plain transformation_from_parameters).  Backward: dT [B][4][4] -> dx in `dtype`.
 */
int fs_pose_tail_fwd(const float* x, float* axisangle, float* translation, float* T, int B, int hw, int Cx,
                     int nframes, int invert, float scale, void* stream);
int fs_pose_tail_bwd(const float* x, const float* dT, void* dx, int B, int hw, int Cx, int nframes, int invert,
                     float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Photometric loss chain (monodepth2_decoder.py:61-128,205-292; monodepth_utils.py:101-165,184-215).
 * Images are planar NCHW fp32; all S scales are processed per launch.
 *   fs_photo_setup     K, K^-1 (f64), P_f = (K T_f)[:3] per batch element -> geo [B][48]
 *   fs_photo_identity  identity reprojection losses ident[B][2][H][W]; mask_sum[b] += sum(patched_mask[b])
 *   fs_photo_fused_fwd warp (bilinear/border + nearest/zeros overlap sample) and loss of all scales and both frames:
 *                      sel[S][B][H][W] (argmin: 0,1 identity; 2,3 reprojection), loss_sums[s][b] += masked sum;
 *                      optionally pred[S][2][B][3][H][W], ov[S][2][B][H][W]
 *   fs_photo_fused_bwd d_depth[s] += dL/d depth_s (low-res); dP[S][B][tiles][2][12] = per-strip partial sums of
 *                      dL/dP (overwritten; tiles = fs_photo_fused_bwd_tiles(H, W))
 *   fs_photo_pose_grad dT_f[B][4][4] = K^T sum_{s,tile} dP  (fixed summation order: run-to-run reproducible)
 * noise_seed < 0 disables the tie-break noise (reference: randn*1e-5, :258-259).
 */
typedef struct FsPhotoArgs {
  const float* img0;
  const float* img_src[2];
  const double* patched_mask;   /* [B][H][W] or NULL (= ones) */
  const float* depth[4];        /* per scale [B][dh][dw] */
  const float* geo;
  float* pred;
  uint8_t* ov;
  float* ident;
  uint8_t* sel;
  double* loss_sums;            /* [S][B] */
  double* mask_sum;             /* [B] */
  float* d_depth[4];
  float* dP;
  const double* gout;           /* upstream gradient of the total loss (device scalar) or NULL = 1 */
  int32_t dh[4], dw[4];
  int32_t B, H, W, S;
  int32_t noise_seed;
  const int32_t* noise_seed_ptr;  /* device-resident seed (overrides noise_seed when non-NULL; hipGraph replay) */
  /* Fisheye / Mei unified camera model (FishEyeDecoder._generate_images_pred, monodepth2_decoder.py:355-411): all three
   * NULL on the pinhole path.  `depth` is then the ray norm, the 3-D point is ray-table x norm, the transform acts on
   * it directly and mei_fisheye_utils._cam2image (:23-51) maps it to pixels (fs_photo_setup(fisheye=1) stores
   * P_f = T_f[:3] and K = identity, so fs_photo_pose_grad returns dT itself). */
  const float* const* lut_ptrs;   /* device array [B]: per-sample ray table [4][H][W] = X, Y, Z, mask (fs_mei_lut) */
  const float* mei;               /* device [B][8]: k1, k2, xi, gamma1, gamma2, u0, v0, 0 */
  const float* warp_mask;         /* [B][H][W] fp32 = patched_mask x ray-table mask (fs_mei_stage_mask): the plane the
                                     nearest-neighbour overlap sample reads (:409-411) */
  /* precomputed motion mask (monodepth2_decoder.py:243-246): [B][H][W] fp32 or NULL.  With it the
   * per-pixel minimum runs over the two reprojection terms alone (no identity auto-mask) and the gradient of a pixel
   * is scaled by (1 - motion_mask): to_optimise.detach() * m + to_optimise * (1 - m). */
  const float* motion_mask;
  /* overlapped_mask=False (monodepth2_decoder.py:110-116, 230-235; configs/multi_dataset_example, nusc_wpose_example):
   * a reprojection term is used wherever it is, with border-clamped samples — no 100.0 substitution where the sample
   * left the source frame.  0 = the masked behaviour of the KITTI configs. */
  int32_t no_overlap_mask;
  int32_t reserved0;
} FsPhotoArgs;
int fs_photo_setup(const float* P2, const float* T0, const float* T1, float* geo, int B, int* seed_counter,
                   int fisheye, void* stream);   /* seed_counter (or NULL): device int bumped by one — the noise seed of
                                                    this step */
int fs_photo_identity(const FsPhotoArgs* args, void* stream);
/* (ABI <= 9 also had the staged form fs_photo_warp / fs_photo_loss_fwd / fs_photo_loss_bwd / fs_photo_bwd_tiles.)
 * fs_photo_fused_fwd: ONE launch for all scales and both frames, the warped images stay in registers (pred / ov may be
 * NULL; when given they are written for logging and for outputs[("original_image", f, s)] of _generate_images_pred
 * :98-116).  Needs ident (or motion_mask), sel, loss_sums. */
int fs_photo_fused_fwd(const FsPhotoArgs* args, void* stream);
/* fs_photo_fused_bwd: the warp is recomputed (gathers through L2) instead of reading eight warped images back.
 * Outputs: d_depth[s] (accumulated: zero it first) and the per-strip partials
 * dP[S][B][tiles][2][12] with tiles = fs_photo_fused_bwd_tiles(H, W), reduced by fs_photo_pose_grad. */
int fs_photo_fused_bwd(const FsPhotoArgs* args, void* stream);
int64_t fs_photo_fused_bwd_tiles(int H, int W);
int fs_photo_pose_grad(const float* geo, const float* dP, float* dT0, float* dT1, int B, int S, int tiles,
                       void* stream);

/* Mei unified fisheye camera model (mei_fisheye_utils.py:14-187; configs/kitti360_fisheye_example:198-207).
 * fs_mei_lut: the per-calibration table MeiCameraProjection.image2cam caches (:150-166) — for every pixel the radial
 *   distortion is inverted by Newton (newton_methods :70-79) and the mirror equation by bisection (bisection_methods
 *   :85-101), both in f64 as numba types them; lut = [4][H][W] fp32 planes X, Y, Z, mask (invalid pixels: -1 * (xi - 1),
 *   -1 * (xi - 1), -1, 0 exactly as the reference leaves them).  gamma1/gamma2/u0/v0 are the fp32 entries of P2.
 * fs_mei_stage_mask: warp_mask[b] = float(patched_mask[b] * table_mask[b]) (monodepth2_decoder.py:409).
 * fs_mei_points: points[B][H][W][3] = table x norm (image2cam :183-187; get_prediction reads z, decoder :413-420).
 */
int fs_mei_lut(float* lut, int H, int W, float gamma1, float gamma2, float u0, float v0, double k1, double k2,
               double xi, void* stream);
int fs_mei_stage_mask(const float* const* lut_ptrs, const double* patched_mask, float* warp_mask, int B, int H, int W,
                      void* stream);
int fs_mei_points(const float* const* lut_ptrs, const float* norm, float* points, int B, int H, int W, void* stream);

/* Edge-aware smoothness (monodepth_utils.py:168-181; monodepth2_decoder.py:214-219,294-299) and
 * loss assembly (:292-304).  fs_color_pyramid = adaptive_avg_pool2d with an integer ratio.
 * fs_loss_finalize: out[0..S) loss/s, out[S..2S) smooth_loss/s, out[2S] total_loss (f64).
 */
typedef struct FsSmoothArgs {
  const float* disp[4];
  const float* color[4];
  float* d_disp[4];
  double* disp_sum;    /* [S][B] */
  double* sm_sums;     /* [S][B][2] */
  double* dot;         /* [S][B] */
  const double* gout;
  int32_t h[4], w[4], scale_id[4];
  int32_t B, S;
} FsSmoothArgs;
int fs_color_pyramid(const float* img, float* out, int B, int H, int W, int h, int w, void* stream);
int fs_smooth_mean(const FsSmoothArgs* args, void* stream);
int fs_smooth_fwd(const FsSmoothArgs* args, void* stream);
int fs_smooth_bwd(const FsSmoothArgs* args, void* stream);
int fs_loss_finalize(const double* loss_sums, const double* mask_sum, const double* sm_sums,
                     const FsSmoothArgs* args, double* out, double* total_out, void* stream);

/* Optimizer over a flat fp32 arena: global grad sum-of-squares, then clip + Adam in one pass
 * (clip_grad_norm_ + torch.optim.Adam.step, base_training_hooks.py:46-49; optimizers.py:7-8).
 * grad_scale multiplies every gradient first (1/world_size after a SUM all-reduce).
 * max_norm <= 0 or sumsq == NULL disables clipping.
 * step_ptr / lr_ptr: optional device-resident step count / learning rate (override the scalars) so that the
 * launch can be replayed from a hipGraph; fs_counter_incr bumps such a counter on the stream.
 */
int fs_sumsq(const float* g, int64_t n, double* out, int* step_counter, void* stream);   /* step_counter (or NULL): +1 */
int fs_counter_incr(int32_t* counter, void* stream);
int fs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float max_norm, const double* sumsq, float grad_scale,
                 const int32_t* step_ptr, const float* lr_ptr, void* stream);

#ifdef __cplusplus
}
#endif
#endif
