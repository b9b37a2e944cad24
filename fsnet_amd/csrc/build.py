"""Build libfsnet_hip.so (gfx950 only) with hipcc: one object per .hip file, compiled in
parallel, linked in-tree at fsnet_amd/lib/libfsnet_hip.so.  No torch, no cmake."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libfsnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-I", os.path.join(ROOT, "include")]


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    h.update(" ".join(f for f in FLAGS if not f.startswith("/")).encode())
    return h.hexdigest()


def extra_flags(src):
    """per-file compiler flags: a `// FS_HIPCC_FLAGS: ...` line in the first lines of the source"""
    with open(src) as f:
        for _ in range(40):
            line = f.readline()
            if "FS_HIPCC_FLAGS:" in line:
                return line.split("FS_HIPCC_FLAGS:", 1)[1].split()
    return []


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "fsnet_hip.h"))
    return hs


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_digest = _digest(headers())
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        stamp = obj + ".sha"
        want = _digest([src]) + hdr_digest
        objs.append(obj)
        have = open(stamp).read() if os.path.exists(stamp) and os.path.exists(obj) else ""
        if force or have != want:
            jobs.append((src, obj, stamp, want))

    def compile_one(job):
        src, obj, stamp, want = job
        cmd = [HIPCC] + FLAGS + extra_flags(src) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
        with open(stamp, "w") as f:
            f.write(want)
        return src

    if jobs:
        if verbose:
            print("[fsnet_amd] compiling %d HIP source(s) for gfx950" % len(jobs), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
