"""CPU checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol that
include/fsnet_hip.h declares, and the ctypes signature table + struct mirrors agree with the header."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fsnet_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", src, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(fs_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        fns[m.group(2)] = (m.group(1), args)
    return fns


def declared_structs():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \w+;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ty = decl.rsplit(" ", 1)[0] if "," not in decl else None
            if "," in decl:
                first = decl.split(",")[0]
                ty = first.rsplit(" ", 1)[0].strip()
                names = [first.rsplit(" ", 1)[1]] + [n.strip() for n in decl.split(",")[1:]]
            else:
                names = [decl.rsplit(" ", 1)[1]]
            for n in names:
                star = n.count("*")
                n = n.replace("*", "").strip()
                arr = re.search(r"\[(\d+)\]", n)
                fields.append((re.sub(r"\[\d+\]", "", n), ty.strip() + "*" * star, int(arr.group(1)) if arr else 0))
        out[m.group(1)] = fields
    return out


def ctype_of(ty):
    ty = ty.replace("const ", "").strip()
    if ty.endswith("*"):
        return C.c_void_p
    return {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "float": C.c_float,
            "double": C.c_double}[ty]


def test_library_builds_and_exports_every_declared_symbol():
    from fsnet_amd.csrc import build as B
    path = B.build(verbose=False)
    lib = C.CDLL(path)
    fns = declared_functions()
    assert len(fns) >= 30
    for name in fns:
        assert hasattr(lib, name), "libfsnet_hip.so does not export %s" % name
    lib.fs_abi_version.restype = C.c_int
    lib.fs_target_arch.restype = C.c_char_p
    from fsnet_amd.hip import binding
    header = open(HEADER).read()
    declared = int(re.search(r"#define FS_ABI_VERSION (\d+)", header).group(1))
    assert lib.fs_abi_version() == declared == binding.ABI_VERSION
    assert lib.fs_target_arch() == b"gfx950"


def test_signature_table_matches_header():
    from fsnet_amd.hip.signatures import SIGNATURES
    fns = declared_functions()
    assert set(SIGNATURES) == set(fns)
    for name, (ret, args) in fns.items():
        res, argtypes = SIGNATURES[name]
        assert len(argtypes) == len(args), name
        for decl, ct in zip(args, argtypes):
            ty = decl.rsplit(" ", 1)[0] if not decl.endswith("*") else decl
            ty = ty + ("*" if decl.rsplit(" ", 1)[-1].startswith("*") else "")
            assert ctype_of(ty) == ct, (name, decl, ct)


def test_struct_mirrors_match_header():
    import fsnet_amd.hip.binding as L
    for sname, fields in declared_structs().items():
        mirror = getattr(L, sname)
        got = [(n, t) for n, t in mirror._fields_]
        assert [f[0] for f in fields] == [g[0] for g in got], sname
        for (n, ty, arr), (_, ct) in zip(fields, got):
            want = ctype_of(ty)
            if arr:
                want = want * arr
                assert C.sizeof(ct) == C.sizeof(want), (sname, n)
            else:
                assert ct == want, (sname, n, ty, ct)


def test_invalid_arguments_are_rejected_without_a_gpu():
    from fsnet_amd.hip import lib
    assert lib.fs_conv_igemm(None, 0, None) == 1
    assert lib.fs_conv_wgrad(None, 1, None) == 1
    assert lib.fs_adam_step(None, None, None, None, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 1, 0.0, None, 1.0, None, None, None) == 1
    assert lib.fs_conv_stem(None, 1, None) == 1 and lib.fs_conv3x3_halo(None, 1, None) == 1
    assert lib.fs_augment_frames(None, None) == 1 and lib.fs_resize_frames(None, None) == 1
    assert lib.fs_sumsq(None, 0, None, None, None) == 1
    assert lib.fs_photo_fused_bwd_tiles(192, 640) == 11 * 6 and lib.fs_photo_fused_bwd_tiles(1, 1) == -1
    assert lib.fs_pack_tile_blocks(64, 64, 3, 3) == 4 and lib.fs_pack_tile_blocks(1, 1, 20, 20) == -1


def test_missing_library_fails_loudly(monkeypatch):
    import fsnet_amd.hip.binding as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setenv("FSNET_HIP_LIB", "/nonexistent/libfsnet_hip.so")
    with pytest.raises(L.FsError):
        L.load_library()
    monkeypatch.delenv("FSNET_HIP_LIB")
    monkeypatch.setattr(L, "_lib", None)
    L.load_library()


def test_graft_entry_build_passes():
    """the driver's "does it build" check, run the way the driver runs it (incremental: seconds)"""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    entry = importlib.import_module("__graft_entry__")
    entry.build()
