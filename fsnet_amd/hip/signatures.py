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
    "fs_conv_igemm": (C.c_int, [P, I, P]),
    "fs_conv_wgrad": (C.c_int, [P, I, P]),
    "fs_pack_weights": (C.c_int, [P, P, I, I, I, I, I, I, L, I, I, P]),
    "fs_nchw_to_nhwc": (C.c_int, [P, P, P, I, I, I, I, I, I, I, P]),
}


def declare(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
