"""argtypes/restype table for every symbol declared in include/fsnet_hip.h.  The CPU test
suite checks this table against the header and against the built library's exports."""
import ctypes as C

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
D = C.c_double

SIGNATURES = {
    "fs_abi_version": (C.c_int, []),
    "fs_target_arch": (C.c_char_p, []),
    "fs_debug_timestamp": (C.c_int, [P, P]),
    "fs_capture_position": (C.c_int, [P, P]),
    "fs_conv_igemm": (C.c_int, [P, I, P]),
    # two problems per launch (ABI 7): (args0, args1 | NULL, dtype, stream)
    "fs_conv_igemm2": (C.c_int, [P, P, I, P]),
    "fs_conv3x3_halo2": (C.c_int, [P, P, I, P]),
    "fs_conv3x3_halo2_plan": (C.c_int, [P, P, I, P]),
    "fs_conv3x3_s2d": (C.c_int, [P, I, P]),
    "fs_conv3x3_s2d2": (C.c_int, [P, P, I, P]),
    "fs_conv_stem2": (C.c_int, [P, P, I, P]),
    "fs_conv_wgrad2": (C.c_int, [P, P, I, P]),
    "fs_conv_wgrad2_plan": (C.c_int, [P, P, I, P]),
    "fs_bn_apply2": (C.c_int, [P, P, I, P]),
    "fs_bn_bwd_reduce2": (C.c_int, [P, P, I, P]),
    "fs_bn_bwd_apply2": (C.c_int, [P, P, I, P]),
    "fs_maxpool_fwd2": (C.c_int, [P, P, P, I, P, P, P, I, I, I, I, I, P]),
    "fs_maxpool_bwd2": (C.c_int, [P, P, P, P, I, P, P, P, P, I, I, I, I, I, P]),
    "fs_conv3x3_halo": (C.c_int, [P, I, P]),
    "fs_conv3x3_halo_plan": (C.c_int, [P, I, P]),
    "fs_conv1x1": (C.c_int, [P, I, P]),
    "fs_conv_stem": (C.c_int, [P, I, P]),
    "fs_conv_wgrad": (C.c_int, [P, I, P]),
    "fs_conv_wgrad_plan": (C.c_int, [P, I, P]),
    "fs_pack_weights": (C.c_int, [P, P, I, I, I, I, I, I, L, I, I, P]),
    "fs_sigmoid_head_fwd": (C.c_int, [P, P, L, I, P]),
    "fs_sigmoid_head_bwd": (C.c_int, [P, P, P, L, I, I, P]),
    "fs_distill_fwd": (C.c_int, [P, P, P, L, P, P]),
    "fs_distill_bwd": (C.c_int, [P, P, P, L, P, P, P, P]),
    "fs_resize_linear": (C.c_int, [P, P, I, I, I, I, I, P]),
    "fs_depth_eval": (C.c_int, [P, P, I, I, I, I, I, P, P, P]),
    "fs_copy_multi": (C.c_int, [P, P, P, I, P]),
    "fs_zero_multi": (C.c_int, [P, P, I, P]),
    "fs_pack_tile_blocks": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "fs_pack_weights_multi": (C.c_int, [P, I, L, I, P]),
    "fs_nchw_to_nhwc": (C.c_int, [P, P, P, I, I, I, I, I, I, I, P]),
    "fs_bn_apply": (C.c_int, [P, I, P]),
    "fs_bn_finalize": (C.c_int, [P, P, P, P]),
    "fs_bn_bwd_reduce": (C.c_int, [P, I, P]),
    "fs_bn_bwd_apply": (C.c_int, [P, I, P]),
    "fs_maxpool_fwd": (C.c_int, [P, P, P, I, I, I, I, I, P]),
    "fs_maxpool_bwd": (C.c_int, [P, P, P, P, I, I, I, I, I, P]),
    "fs_upcat_pad_fwd": (C.c_int, [P, P, P, I, I, I, I, I, I, P]),
    "fs_upcat_pad_bwd": (C.c_int, [P, P, P, I, I, I, I, I, I, P]),
    "fs_upcat_pad_bwd_bn": (C.c_int, [P, P, P, I, I, I, I, I, P, P, P, P, P, I, P]),
    "fs_channel_sum": (C.c_int, [P, P, L, I, I, I, P]),
    "fs_channel_sum_multi": (C.c_int, [P, P, P, P, P, I, I, P]),
    "fs_depth_head_fwd": (C.c_int, [P, P, P, P, L, I, I, F, F, P]),
    "fs_depth_head_bwd": (C.c_int, [P, P, P, P, P, L, I, I, F, F, I, P]),
    "fs_depth_head_fwd_multi": (C.c_int, [P, P, I, I, F, F, P]),
    "fs_depth_head_bwd_multi": (C.c_int, [P, P, I, I, F, F, I, P]),
    "fs_pose_tail_fwd": (C.c_int, [P, P, P, P, I, I, I, I, I, F, P]),
    "fs_pose_tail_bwd": (C.c_int, [P, P, P, I, I, I, I, I, F, I, P]),
    "fs_photo_setup": (C.c_int, [P, P, P, P, I, P, I, P]),
    "fs_mei_lut": (C.c_int, [P, I, I, F, F, F, F, D, D, D, P]),
    "fs_mei_stage_mask": (C.c_int, [P, P, P, I, I, I, P]),
    "fs_mei_points": (C.c_int, [P, P, P, I, I, I, P]),
    "fs_photo_identity": (C.c_int, [P, P]),
    "fs_photo_fused_fwd": (C.c_int, [P, P]),
    "fs_photo_fused_bwd": (C.c_int, [P, P]),
    "fs_photo_fused_bwd_tiles": (C.c_int64, [I, I]),
    "fs_photo_pose_grad": (C.c_int, [P, P, P, P, I, I, I, P]),
    "fs_augment_frames": (C.c_int, [P, P]),
    "fs_resize_frames": (C.c_int, [P, P]),
    "fs_color_pyramid": (C.c_int, [P, P, I, I, I, I, I, P]),
    "fs_smooth_mean": (C.c_int, [P, P]),
    "fs_smooth_fwd": (C.c_int, [P, P]),
    "fs_smooth_bwd": (C.c_int, [P, P]),
    "fs_loss_finalize": (C.c_int, [P, P, P, P, P, P, P]),
    "fs_sumsq": (C.c_int, [P, L, P, P, P]),
    "fs_counter_incr": (C.c_int, [P, P]),
    "fs_adam_step": (C.c_int, [P, P, P, P, L, F, F, F, F, F, I, F, P, F, P, P, P]),
}


def declare(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
