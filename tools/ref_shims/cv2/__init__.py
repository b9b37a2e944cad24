"""Import shim for tools/gen_golden.py ONLY (never on the product path, never on the GPU box).

OpenCV is not installed in this image.  The reference's augmentation module imports cv2 at module level and calls
two of its functions on the training path (warpAffine, cvtColor) and resize on the validation path.  This stand-in lets the module import and
routes those two calls to oracle/augment_oracle.py's restatement of OpenCV's algorithms — so golden vectors that
pass through them pin the reference's OWN arithmetic around the calls (draw order, P2 bookkeeping, mirror, colour
ops, Normalize) but NOT OpenCV itself: the oracle header and DESIGN.md mark warpAffine / cvtColor "parity unpinned".
"""
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT = 0
COLOR_RGB2HSV, COLOR_HSV2RGB = 41, 55


def warpAffine(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT):
    from oracle import augment_oracle as A
    assert borderMode == BORDER_CONSTANT
    w, h = dsize
    if flags == INTER_NEAREST:
        return A.warp_affine_nearest(src, M, w, h)
    assert flags == INTER_LINEAR
    return A.warp_affine_linear(src, M, w, h)


def cvtColor(src, code):
    from oracle import augment_oracle as A
    if code == COLOR_RGB2HSV:
        return A.rgb2hsv(src)
    assert code == COLOR_HSV2RGB
    return A.hsv2rgb(src)


def resize(src, dsize, interpolation=INTER_LINEAR):
    from oracle import augment_oracle as A
    if interpolation == INTER_NEAREST:
        return A.resize_nearest(src, dsize[0], dsize[1])
    assert interpolation == INTER_LINEAR and src.dtype.name == "float32"
    return A.resize_linear(src, dsize[0], dsize[1])
