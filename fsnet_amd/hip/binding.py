"""Loader for libfsnet_hip.so.  There is no fallback: if the library is missing the product
path raises (build it with `python -m fsnet_amd.csrc.build` / __graft_entry__.build())."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libfsnet_hip.so")

FS_DTYPE_F32 = 0
FS_DTYPE_BF16 = 1


class FsError(RuntimeError):
    pass


class FsConvArgs(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("wgt", C.c_void_p), ("dst", C.c_void_p), ("bias", C.c_void_p),
        ("addend", C.c_void_p), ("mask", C.c_void_p), ("stats", C.c_void_p), ("ktab", C.c_void_p),
        ("sN", C.c_int64), ("sH", C.c_int64), ("sW", C.c_int64),
        ("dN", C.c_int64), ("dH", C.c_int64), ("dW", C.c_int64),
        ("aN", C.c_int64), ("aH", C.c_int64), ("aW", C.c_int64),
        ("mN", C.c_int64), ("mH", C.c_int64), ("mW", C.c_int64),
        ("src_bytes", C.c_int64), ("wgt_bytes", C.c_int64),
        ("Hs", C.c_int32), ("Ws", C.c_int32), ("Hd", C.c_int32), ("Wd", C.c_int32),
        ("M", C.c_int32), ("Co", C.c_int32), ("Co_p", C.c_int32), ("nchunks", C.c_int32), ("kg", C.c_int32),
        ("hb_mul", C.c_int32), ("hb_add", C.c_int32), ("sgn", C.c_int32), ("dshift", C.c_int32),
        ("relu", C.c_int32), ("out_f32", C.c_int32), ("N", C.c_int32), ("Cs", C.c_int32),
        ("wgt_row_bytes", C.c_int64),
        ("grp_imgs", C.c_int32), ("ncls", C.c_int32), ("cls_nch", C.c_int32 * 4), ("cls_ktab_off", C.c_int32 * 4), ("cls_wgt_off", C.c_int64 * 4),
        ("bnb_x", C.c_void_p), ("bnb_mean", C.c_void_p), ("bnb_invstd", C.c_void_p),
        ("stat_group_rows", C.c_int32),
        ("pro_a", C.c_void_p), ("pro_b", C.c_void_p), ("pro_c", C.c_void_p), ("pro_m", C.c_void_p), ("pro_src2", C.c_void_p),
        ("pro_mode", C.c_int32), ("pro_relu", C.c_int32), ("pro_group_imgs", C.c_int32), ("force_impl", C.c_int32),
        ("bnb_scale", C.c_void_p), ("bnb_shift", C.c_void_p),
        ("pro_stats", C.c_void_p), ("pro_stats_local", C.c_void_p),
        ("pro_gamma", C.c_void_p), ("pro_beta", C.c_void_p),
        ("pro_mean", C.c_void_p), ("pro_invstd", C.c_void_p),
        ("pro_save_a", C.c_void_p), ("pro_save_b", C.c_void_p),
        ("pro_running_mean", C.c_void_p), ("pro_running_var", C.c_void_p),
        ("pro_nbt", C.c_void_p),
        ("pro_dgamma", C.c_void_p), ("pro_dbeta", C.c_void_p),
        ("pro_dst", C.c_void_p),
        ("pro_count", C.c_double), ("pro_eps", C.c_float), ("pro_momentum", C.c_float),
        ("ds_src", C.c_void_p), ("ds_wgt", C.c_void_p), ("ds_wgt_row_bytes", C.c_int64),
    ]


class FsWgradArgs(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("ktab", C.c_void_p),
        ("sN", C.c_int64), ("sH", C.c_int64), ("sW", C.c_int64),
        ("Hs", C.c_int32), ("Ws", C.c_int32), ("Hd", C.c_int32), ("Wd", C.c_int32),
        ("M", C.c_int32), ("Cd", C.c_int32),
        ("Co", C.c_int32), ("Ci", C.c_int32), ("R", C.c_int32), ("S", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("ncolgroups", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_elems", C.c_int64), ("x_bytes", C.c_int64), ("use_halo", C.c_int32),
        ("pix_per_split", C.c_int32), ("nsplit", C.c_int32), ("ws_rows", C.c_int32), ("ws_cols", C.c_int32),
        ("pro_a", C.c_void_p), ("pro_b", C.c_void_p), ("pro_relu", C.c_int32), ("pro_group_imgs", C.c_int32),
    ]


class FsPackDesc(C.Structure):
    _fields_ = [
        ("w", C.c_void_p), ("dst_f", C.c_void_p), ("dst_d", C.c_void_p),
        ("k_f", C.c_int64), ("k_d", C.c_int64), ("block_start", C.c_int64),
        ("Co", C.c_int32), ("Ci", C.c_int32), ("R", C.c_int32), ("S", C.c_int32),
        ("rows_f", C.c_int32), ("cs_f", C.c_int32), ("rows_d", C.c_int32), ("cs_d", C.c_int32),
        ("tap_order_d", C.c_int32),
    ]


class FsAugArgs(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("minv", C.c_void_p), ("iplan", C.c_void_p), ("fplan", C.c_void_p),
        ("image", C.c_void_p), ("original", C.c_void_p), ("mask", C.c_void_p),
        ("mean", C.c_float * 3), ("std", C.c_float * 3),
        ("B", C.c_int32), ("F", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
    ]


class FsHeadBatch(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p * 4), ("depth", C.c_void_p * 4), ("disp", C.c_void_p * 4),
        ("d_depth", C.c_void_p * 4), ("d_disp", C.c_void_p * 4), ("dlogits", C.c_void_p * 4),
        ("M", C.c_int64 * 4), ("n", C.c_int32), ("nimg", C.c_int32), ("P2", C.c_void_p), ("base_fx", C.c_float),
        ("reserved", C.c_int32),
    ]


class FsResizeArgs(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dims", C.c_void_p), ("image", C.c_void_p),
        ("mean", C.c_float * 3), ("std", C.c_float * 3),
        ("B", C.c_int32), ("F", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("iplan", C.c_void_p), ("fplan", C.c_void_p), ("original", C.c_void_p), ("mask", C.c_void_p),
    ]


class FsBnApplyArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
        ("stats", C.c_void_p), ("stats2", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("gamma2", C.c_void_p), ("beta2", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
        ("running_mean2", C.c_void_p), ("running_var2", C.c_void_p),
        ("num_batches_tracked", C.c_void_p), ("num_batches_tracked2", C.c_void_p),
        ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p),
        ("save_mean2", C.c_void_p), ("save_invstd2", C.c_void_p),
        ("count", C.c_double), ("eps", C.c_float), ("momentum", C.c_float),
        ("yN", C.c_int64), ("yH", C.c_int64), ("yW", C.c_int64),
        ("M", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("relu", C.c_int32), ("pad_out", C.c_int32), ("groups", C.c_int32),
        ("pool_y", C.c_void_p), ("pool_idx", C.c_void_p),
    ]


class FsBnBwdArgs(C.Structure):
    _fields_ = [
        ("dout", C.c_void_p), ("y", C.c_void_p), ("x", C.c_void_p), ("dx", C.c_void_p), ("g_out", C.c_void_p),
        ("sums", C.c_void_p), ("sums_local", C.c_void_p),
        ("gamma", C.c_void_p), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p),
        ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
        ("count", C.c_double),
        ("gN", C.c_int64), ("gH", C.c_int64), ("gW", C.c_int64),
        ("yN", C.c_int64), ("yH", C.c_int64), ("yW", C.c_int64),
        ("M", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("relu", C.c_int32), ("fold", C.c_int32), ("groups", C.c_int32),
        ("pool_dy", C.c_void_p), ("pool_idx", C.c_void_p), ("beta", C.c_void_p),
    ]


class FsPhotoArgs(C.Structure):
    _fields_ = [
        ("img0", C.c_void_p), ("img_src", C.c_void_p * 2), ("patched_mask", C.c_void_p),
        ("depth", C.c_void_p * 4), ("geo", C.c_void_p),
        ("pred", C.c_void_p), ("ov", C.c_void_p), ("ident", C.c_void_p), ("sel", C.c_void_p),
        ("loss_sums", C.c_void_p), ("mask_sum", C.c_void_p),
        ("d_depth", C.c_void_p * 4), ("dP", C.c_void_p), ("gout", C.c_void_p),
        ("dh", C.c_int32 * 4), ("dw", C.c_int32 * 4),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("S", C.c_int32),
        ("noise_seed", C.c_int32), ("noise_seed_ptr", C.c_void_p),
        ("lut_ptrs", C.c_void_p), ("mei", C.c_void_p), ("warp_mask", C.c_void_p), ("motion_mask", C.c_void_p),
        ("no_overlap_mask", C.c_int32), ("reserved0", C.c_int32),
    ]


class FsSmoothArgs(C.Structure):
    _fields_ = [
        ("disp", C.c_void_p * 4), ("color", C.c_void_p * 4), ("d_disp", C.c_void_p * 4),
        ("disp_sum", C.c_void_p), ("sm_sums", C.c_void_p), ("dot", C.c_void_p), ("gout", C.c_void_p),
        ("h", C.c_int32 * 4), ("w", C.c_int32 * 4), ("scale_id", C.c_int32 * 4),
        ("B", C.c_int32), ("S", C.c_int32),
    ]


ABI_VERSION = 11     # FS_ABI_VERSION of include/fsnet_hip.h (tests/test_abi.py holds the two together)
_lib = None


def load_library(path=None):
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("FSNET_HIP_LIB", LIB_PATH)
    # torch bundles its own libamdhip64; it must be the HIP runtime of this process BEFORE our library is
    # dlopen'ed (same SONAME), otherwise kernels register with a second runtime and every launch fails.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise FsError(
            "libfsnet_hip.so not found at %s — the HIP kernel library is required (no CPU fallback). "
            "Build it with `python fsnet_amd/csrc/build.py` or __graft_entry__.build()." % path)
    handle = C.CDLL(path)
    from . import signatures
    signatures.declare(handle)
    got = int(handle.fs_abi_version())
    if got != ABI_VERSION:
        raise FsError("libfsnet_hip.so at %s reports ABI %d, this binding is written against ABI %d (FS_ABI_VERSION "
                      "in include/fsnet_hip.h): rebuild it with `python fsnet_amd/csrc/build.py`" % (path, got, ABI_VERSION))
    _lib = handle
    return _lib


class _LazyLib:
    def __getattr__(self, name):
        return getattr(load_library(), name)


lib = _LazyLib()


def check(status, what=""):
    if status != 0:
        raise FsError("libfsnet_hip call %s failed with status %d" % (what, status))


def raw_stream(device_index=None):
    """integer hipStream_t of torch's current stream on the (current) device.  torch.cuda.current_stream() builds a
    Stream object per call (~9 us); with ~500 launches per step that alone was ~2 ms of host time per eager step."""
    import torch
    if device_index is None:
        device_index = torch._C._cuda_getDevice()
    return torch._C._cuda_getCurrentRawStream(device_index)


def stream_ptr():
    """Raw hipStream_t of torch's current stream (kernels are launched on it so that torch's
    stream ordering, events and graph capture all apply)."""
    return C.c_void_p(raw_stream())
