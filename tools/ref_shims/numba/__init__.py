"""Dev-only stand-in for `numba.jit` so the reference's fisheye module imports (tools/gen_golden.py only).

numba types the root finders of mei_fisheye_utils.py:66-120 with float64 arithmetic: the float32 radius read from the
array meets float64 calibration scalars and every local is unified to float64.  Run as plain Python the numpy float32
scalar would instead keep the arithmetic in float32 (NEP 50: Python floats are weak), so the wrapper promotes
numpy floating scalars to Python floats at every call, which reproduces numba's types."""
import functools

import numpy as np


def _wrap(f):
    @functools.wraps(f)
    def g(*a):
        return f(*[float(x) if isinstance(x, np.floating) else x for x in a])
    return g


def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return _wrap(a[0])
    return _wrap
