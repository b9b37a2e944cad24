"""Knock-out study: step time of bench.py with the launches of some C-ABI entry points dropped (the results of the step are
garbage, the replayed graph simply lacks those kernels) — what a kernel family costs the step, overlap included.
usage: python tools/probes/knockout.py fs_bn_bwd_apply[,fs_bn_apply] [bench.py args...]   (`none` = nothing dropped)
The hook lives here, not in the package: the product binding has no such switch."""
import os, runpy, sys
sys.path.insert(0, os.getcwd())
from fsnet_amd.hip import binding

skip = set(sys.argv[1].split(","))
_orig = binding._LazyLib.__getattr__


def _getattr(self, name):
    if name in skip:
        return lambda *a: 0
    return _orig(self, name)


binding._LazyLib.__getattr__ = _getattr
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
