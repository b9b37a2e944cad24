"""The hot kernels must not spill registers to scratch memory: a runtime-indexed register array or an unlucky
occupancy heuristic turns a 160 us kernel into a 360 us one without failing any numerical test (it happened:
photo_fused_fwd, `r1.ov[bi - 2]`).  Compiles the device code of the hot files to assembly and reads the
per-kernel resource summary."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from fsnet_amd.csrc import build as B

HOT = ["photo_fused.hip", "conv3x3_halo.hip", "conv3x3_t32.hip", "conv3x3_p1.hip", "conv_igemm.hip", "conv1x1.hip", "conv_wgrad.hip", "bn.hip"]


@pytest.mark.skipif(shutil.which(B.HIPCC) is None and not os.path.exists(B.HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("name", HOT)
def test_no_scratch(name):
    src = os.path.join(os.path.dirname(B.__file__), name)
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        flags = [f for f in B.FLAGS if f != "-fPIC"] + B.extra_flags(src)
        subprocess.run([B.HIPCC] + flags + ["-S", "--cuda-device-only", src, "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        txt = open(out).read()
    kernels = re.findall(r"^\s*\.amdhsa_kernel (\S+)", txt, re.M)
    vspill = [int(x) for x in re.findall(r"^\s*\.vgpr_spill_count:\s*(\d+)", txt, re.M)]
    scratch = [int(x) for x in re.findall(r"^; ScratchSize: (\d+)", txt, re.M)]
    assert kernels and len(vspill) >= len(kernels) and len(scratch) >= len(kernels)
    assert max(vspill) == 0, "vector register spills in %s: %s" % (name, vspill)
    # no instruction touches scratch memory.  (A kernel at the scalar-register limit may still declare a few bytes of
    # private segment: hipcc reserves an emergency slot when it spills SGPRs into VGPR lanes — v_writelane, no memory
    # traffic.  Which instantiation gets it changes with unrelated edits; what matters is that nothing is stored there.)
    assert not re.search(r"^\s*(scratch_(load|store)|buffer_(load|store)\S* .*\boffen\b.*s\[0:3\])", txt, re.M), name
    # (the two-problem kernels address their argument set through a computed kernarg offset: two more live scalars, and
    # bn_bwd_apply's emergency slot grew from 36 to 68 bytes with 16-18 SGPRs parked in VGPR lanes — still no scratch
    # instruction, which the search above establishes)
    assert max(scratch) <= 128, "stack objects in %s: %s" % (name, scratch)
