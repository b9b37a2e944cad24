"""Dev-only no-op `numba.jit` so the reference's fisheye module imports (tools/gen_golden.py only)."""


def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
