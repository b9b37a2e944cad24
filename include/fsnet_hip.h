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

/* library/ABI version and the ISA the kernels were compiled for ("gfx950") */
int fs_abi_version(void);
const char* fs_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * Convolution forward / data-gradient (implicit GEMM on MFMA).
 * Replaces: nn.Conv2d forward and convolution_backward(input) issued by
 *   vision_base/networks/models/backbone/resnet.py:6-9,119,148-160,199-213
 *   vision_base/networks/blocks/blocks.py:41-54
 *   monodepth/networks/models/heads/depth_encoder.py:45-63,123-139
 *   monodepth/networks/models/heads/pose_decoder.py:17-21,26-37
 * ktab: one int per 16-byte K group (8 bf16 / 4 f32 channels): c | r<<16 | s<<24, or -1 for
 * zero padding of K.  Forward: src = input, rows = output pixels, hb = y*hb_mul + hb_add
 * (stride, -pad), sgn=+1.  Dgrad: src = dY, rows = input pixels, hb_mul=1, hb_add=+pad, sgn=-1,
 * dshift = log2(stride) with a parity test.  Epilogue: + bias, + addend, relu, optional
 * per-channel f64 sum / sum-of-squares (BatchNorm batch statistics).
 */
typedef struct FsConvArgs {
  const void* src;
  const void* wgt;      /* packed [Co_p][nchunks*64 bytes], K contiguous */
  void* dst;
  const float* bias;    /* [Co] or NULL */
  const void* addend;   /* same dtype as src, or NULL */
  double* stats;        /* [2][Co] or NULL */
  const int* ktab;      /* [nchunks*4] */
  int64_t sN, sH, sW;   /* src strides (elements) */
  int64_t dN, dH, dW;   /* dst strides */
  int64_t aN, aH, aW;   /* addend strides */
  int32_t Hs, Ws;       /* src spatial size */
  int32_t Hd, Wd;       /* row-domain (dst) spatial size */
  int32_t M;            /* N*Hd*Wd */
  int32_t Co;           /* dst channels (multiple of 4) */
  int32_t Co_p;         /* packed weight rows (multiple of 16) */
  int32_t nchunks;
  int32_t hb_mul, hb_add, sgn, dshift;
  int32_t relu;
  int32_t out_f32;      /* store fp32 regardless of dtype */
} FsConvArgs;
int fs_conv_igemm(const FsConvArgs* args, int dtype, void* stream);

/* Convolution weight gradient.  Replaces convolution_backward(weight) at the same call sites.
 * dy is dense [M][Cd]; x is the forward input (strided NHWC); dw is the fp32 OIHW gradient
 * [Co][Ci][R][S] and is accumulated into (atomics) — zero it first.  ktab as above, one entry
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
  int32_t pix_per_split;/* filled by the library */
} FsWgradArgs;
int fs_conv_wgrad(const FsWgradArgs* args, int dtype, void* stream);

/* Weight packing.  OIHW fp32 master weights (the reference's state_dict layout,
 * e.g. depth_backbone.conv1.weight (64,3,7,7)) -> [rows_p][ktot_p] K-contiguous
 * MFMA operand, K = (r, s, c) with c padded to cs_p.  transpose=0: rows = co, c = ci (forward);
 * transpose=1: rows = ci, c = co (dgrad).
 */
int fs_pack_weights(const float* w_oihw, void* dst, int Co, int Ci, int R, int S, int rows_p,
                    int cs_p, int64_t ktot_p, int transpose, int dtype, void* stream);

/* Batch images NCHW fp32 (one tensor, or two concatenated along C as the pose encoder input,
 * monodepth2_model.py:29-35) -> NHWC with Cp >= Ca+Cb zero-padded channels.
 */
int fs_nchw_to_nhwc(const float* a, const float* b, void* dst, int N, int Ca, int Cb, int H, int W,
                    int Cp, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
