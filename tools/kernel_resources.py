"""Per-kernel register / LDS / scratch summary of one HIP source of the library (device assembly of hipcc for gfx950):
python tools/kernel_resources.py conv3x3_t32 [filter].  Occupancy = waves per SIMD the registers allow (512 / allocated,
8-register granules)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fsnet_amd.csrc import build as B


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n") if r.returncode == 0 else names


def main():
    name = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    src = os.path.join(ROOT, "fsnet_amd", "csrc", name + ".hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        flags = [f for f in B.FLAGS if f != "-fPIC"] + B.extra_flags(src)
        subprocess.run([B.HIPCC] + flags + ["-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    rows = []
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.M | re.S):
        body = m.group(2)
        get = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        rows.append((m.group(1), get("next_free_vgpr"), get("next_free_sgpr"), get("group_segment_fixed_size"),
                     get("private_segment_fixed_size")))
    names = demangle([r[0] for r in rows])
    for (raw, v, s, l, sc), dn in zip(rows, names):
        dn = re.sub(r"\(anonymous namespace\)::", "", dn)
        dn = re.sub(r"\(.*$", "", dn)
        if flt and flt not in dn:
            continue
        alloc = (v + 7) // 8 * 8
        print("%-70s vgpr %3d (waves/SIMD %d) sgpr %3d lds %6d scratch %d" % (dn[:70], v, min(8, 512 // alloc), s, l, sc))


if __name__ == "__main__":
    main()
