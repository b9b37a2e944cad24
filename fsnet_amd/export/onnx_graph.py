"""ONNX interchange of the depth network (reference scripts/onnx_export.py:49-52, which
traces `meta_arch.dummy_forward` = depth encoder -> depth decoder -> get_prediction with opset 11).

The HIP engine is opaque to the ONNX tracer (its kernels are reached through ctypes), so the exported file is traced
from a DESCRIPTION of the same inference graph in ONNX-mappable torch operators, written over the very parameter
modules (nn.Conv2d / nn.BatchNorm2d) the HIP runners read.  `describe_depth_network` exists for the tracer only:
`dummy_forward` switches to it while `torch.onnx.is_in_onnx_export()` holds and nowhere else — no training,
inference or evaluation path of this package executes it (tests/test_onnx_export_cpu.py interprets the exported file
against the real reference's output; tests/test_onnx_gpu.py holds the HIP forward to the same file).

torch's TorchScript exporter serialises the ModelProto in C++; the only thing it wants the `onnx` Python package for
is splicing onnxscript custom functions into the file, which this graph has none of, so `export` skips that step
when the package is absent (it is absent in the ROCm image).  `read_model` / `check_model` / `printable_graph` stand
in for `onnx.load` / `onnx.checker.check_model` / `onnx.helper.printable_graph` of the reference script with a
small protobuf wire-format reader (ONNX's field numbers are part of its published IR)."""
import contextlib
import io
import struct

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- graph description
def _block(blk, x):
    """BasicBlock / Bottleneck (reference resnet.py:34-50, 69-89)"""
    out = F.relu(blk.bn1(blk.conv1(x)))
    if hasattr(blk, "conv3"):
        out = F.relu(blk.bn2(blk.conv2(out)))
        out = blk.bn3(blk.conv3(out))
    else:
        out = blk.bn2(blk.conv2(out))
    residual = x if blk.downsample is None else blk.downsample(x)
    return F.relu(out + residual)


def describe_resnet(net, image):
    """feature pyramid of the reference's ResNet.forward (resnet.py:199-213)"""
    x = F.relu(net.bn1(net.conv1(image)))
    outs = [x] if -1 in net.out_indices else []
    x = net.maxpool(x)
    for i in range(net.num_stages):
        for blk in getattr(net, "layer%d" % (i + 1)):
            x = _block(blk, x)
        if i in net.out_indices:
            outs.append(x)
    return outs


def _conv_bn_relu(unit, x):
    return F.relu(unit.sequence(x))          # blocks.py:49-54


def describe_depth_decoder(dec, feats):
    """('depth', 0, 0) of DepthDecoder / MultiChannelDepthDecoder(.Uncertain) with P2=None, i.e. depth_scale = 1
    (depth_encoder.py:90-111, 124-139, 73-86)"""
    x = feats[-1]
    for i in range(4, -1, -1):
        x = _conv_bn_relu(dec.convs[("upconv", i, 0)], x)
        x = [F.interpolate(x, scale_factor=2, mode="nearest")]
        if dec.use_skips and i > 0:
            x += [feats[i - 1]]
        x = _conv_bn_relu(dec.convs[("upconv", i, 1)], torch.cat(x, 1))
    logits = dec.convs[("dispconv", 0)](x)
    if dec.sigmoid_head:
        disp = torch.sigmoid(logits)
        min_disp, max_disp = 1 / dec.max_depth, 1 / dec.min_depth
        return 1 / (min_disp + (max_disp - min_disp) * disp)
    act = torch.softmax(torch.clamp(logits, -10.0, 10.0), dim=1)
    return torch.sum(act * dec.depth_bins.reshape(1, -1, 1, 1), dim=1, keepdim=True)


def describe_depth_network(meta_arch, image):
    """what `dummy_forward` computes (monodepth2_model.py:48-52), as the tracer sees it"""
    head = meta_arch.head
    if 0 not in head.depth_decoder.scales:
        raise ValueError("get_prediction reads ('depth', 0, 0): the decoder must emit scale 0")
    if type(head).get_prediction is not _pinhole_get_prediction():
        raise NotImplementedError("%s.get_prediction needs the batch dictionary; the reference's dummy_forward "
                                  "passes None and cannot export it either" % type(head).__name__)
    feats = describe_resnet(meta_arch.depth_backbone, image)
    return dict(depth=describe_depth_decoder(head.depth_decoder, feats))


def _pinhole_get_prediction():
    from fsnet_amd.monodepth.networks.models.heads.monodepth2_decoder import MonoDepth2Decoder
    return MonoDepth2Decoder.get_prediction


# ----------------------------------------------------------------------------------------------------------- export
@contextlib.contextmanager
def _without_onnx_package():
    """torch.onnx.export(dynamo=False) imports `onnx` only to splice onnxscript functions into the serialised model
    (torchscript_exporter/onnx_proto_utils.py `_add_onnxscript_fn`); with no such functions the bytes pass through."""
    try:
        import onnx  # noqa: F401
        yield
        return
    except ImportError:
        pass
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    saved = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    try:
        yield
    finally:
        onnx_proto_utils._add_onnxscript_fn = saved


def export(meta_arch, dummy_input, onnx_file, input_names=("input",), output_names=("output",), opset_version=11):
    """the reference's export call (onnx_export.py:49-52): eval-mode meta-arch, forward := dummy_forward"""
    if meta_arch.training:
        raise RuntimeError("export an eval()-mode meta-arch (BatchNorm on running statistics), as the reference does")
    saved = meta_arch.forward
    meta_arch.forward = meta_arch.dummy_forward
    try:
        with _without_onnx_package(), torch.no_grad():
            torch.onnx.export(meta_arch, dummy_input, onnx_file, input_names=list(input_names),
                              output_names=list(output_names), opset_version=opset_version, dynamo=False)
    finally:
        meta_arch.forward = saved


# -------------------------------------------------------------------------------------------- reading the file back
def _fields(buf):
    """protobuf wire format: yields (field number, wire type, value) — varint -> int, 64-bit / 32-bit -> raw bytes,
    length-delimited -> memoryview"""
    buf = memoryview(buf)
    i, n = 0, len(buf)
    while i < n:
        key = shift = 0
        while True:
            b = buf[i]; i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if b < 0x80:
                break
        no, wt = key >> 3, key & 7
        if wt == 0:
            v = shift = 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if b < 0x80:
                    break
            yield no, wt, v
        elif wt == 1:
            yield no, wt, bytes(buf[i:i + 8]); i += 8
        elif wt == 5:
            yield no, wt, bytes(buf[i:i + 4]); i += 4
        elif wt == 2:
            ln = shift = 0
            while True:
                b = buf[i]; i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if b < 0x80:
                    break
            yield no, wt, buf[i:i + ln]; i += ln
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)


def _varints(v, wt):
    """a repeated integer field: one element, or a packed run"""
    if wt == 0:
        return [_signed(v)]
    out, cur, shift = [], 0, 0
    for c in bytes(v):
        cur |= (c & 0x7F) << shift
        shift += 7
        if c < 0x80:
            out.append(_signed(cur))
            cur = shift = 0
    return out


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16,
           11: np.float64, 12: np.uint32, 13: np.uint64}


def _tensor(buf):
    dims, dt, name, raw = [], 1, "", None
    f32, i32, i64, f64 = [], [], [], []
    for no, wt, v in _fields(buf):
        if no == 1:
            dims += _varints(v, wt)
        elif no == 2:
            dt = v
        elif no == 8:
            name = bytes(v).decode()
        elif no == 9:
            raw = bytes(v)
        elif no == 4:
            f32 += list(struct.unpack("<%df" % (len(v) // 4), bytes(v))) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif no == 5:
            i32 += _varints(v, wt)
        elif no == 7:
            i64 += _varints(v, wt)
        elif no == 10:
            f64 += list(struct.unpack("<%dd" % (len(v) // 8), bytes(v))) if wt == 2 else [struct.unpack("<d", v)[0]]
    np_dt = _DTYPES[dt]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_dt).copy()
    else:
        arr = np.array(f32 or i32 or i64 or f64, dtype=np_dt)
    return name, arr.reshape(dims)


def _attribute(buf):
    name, out, typ = "", None, 0
    floats, ints, strings = [], [], []
    for no, wt, v in _fields(buf):
        if no == 1:
            name = bytes(v).decode()
        elif no == 2:
            out = struct.unpack("<f", v)[0]
        elif no == 3:
            out = _signed(v)
        elif no == 4:
            out = bytes(v)
        elif no == 5:
            out = _tensor(v)[1]
        elif no == 7:
            floats += list(struct.unpack("<%df" % (len(v) // 4), bytes(v))) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif no == 8:
            ints += _varints(v, wt)
        elif no == 9:
            strings.append(bytes(v))
        elif no == 20:
            typ = v
    if typ == 6 or (out is None and floats):
        out = floats
    elif typ == 7 or (out is None and ints):
        out = ints
    elif typ == 8 or (out is None and strings):
        out = strings
    return name, out


def _value_info(buf):
    name, elem, shape = "", 0, []
    for no, wt, v in _fields(buf):
        if no == 1:
            name = bytes(v).decode()
        elif no == 2:
            for no2, _, v2 in _fields(v):
                if no2 != 1:
                    continue                      # tensor_type
                for no3, _, v3 in _fields(v2):
                    if no3 == 1:
                        elem = v3
                    elif no3 == 2:
                        for _, _, dim in _fields(v3):
                            d = None
                            for no5, _, v5 in _fields(dim):
                                d = _signed(v5) if no5 == 1 else bytes(v5).decode()
                            shape.append(d)
    return dict(name=name, elem_type=elem, shape=shape)


def read_model(source):
    """ModelProto -> dict(ir_version, producer, opsets, graph=dict(nodes, initializers, inputs, outputs))"""
    if isinstance(source, (bytes, bytearray, memoryview)):
        buf = bytes(source)
    elif isinstance(source, io.BytesIO):
        buf = source.getvalue()
    else:
        with open(source, "rb") as fh:
            buf = fh.read()
    model = dict(ir_version=None, producer="", opsets={}, graph=None)
    for no, wt, v in _fields(buf):
        if no == 1:
            model["ir_version"] = v
        elif no == 2:
            model["producer"] = bytes(v).decode()
        elif no == 8:
            dom, ver = "", 0
            for no2, _, v2 in _fields(v):
                if no2 == 1:
                    dom = bytes(v2).decode()
                elif no2 == 2:
                    ver = v2
            model["opsets"][dom] = ver
        elif no == 7:
            g = dict(name="", nodes=[], initializers={}, inputs=[], outputs=[])
            for no2, _, v2 in _fields(v):
                if no2 == 1:
                    node = dict(op_type="", name="", domain="", inputs=[], outputs=[], attrs={})
                    for no3, _, v3 in _fields(v2):
                        if no3 == 1:
                            node["inputs"].append(bytes(v3).decode())
                        elif no3 == 2:
                            node["outputs"].append(bytes(v3).decode())
                        elif no3 == 3:
                            node["name"] = bytes(v3).decode()
                        elif no3 == 4:
                            node["op_type"] = bytes(v3).decode()
                        elif no3 == 5:
                            k, a = _attribute(v3)
                            node["attrs"][k] = a
                        elif no3 == 7:
                            node["domain"] = bytes(v3).decode()
                    g["nodes"].append(node)
                elif no2 == 2:
                    g["name"] = bytes(v2).decode()
                elif no2 == 5:
                    k, arr = _tensor(v2)
                    g["initializers"][k] = arr
                elif no2 == 11:
                    g["inputs"].append(_value_info(v2))
                elif no2 == 12:
                    g["outputs"].append(_value_info(v2))
            model["graph"] = g
    return model


def check_model(model):
    """the structural part of onnx.checker.check_model: versions present, standard-domain operators only, SSA value
    names, nodes in topological order, graph outputs produced"""
    if not model["ir_version"] or "" not in model["opsets"] or model["graph"] is None:
        raise ValueError("not an ONNX model: ir_version / default-domain opset / graph missing")
    g = model["graph"]
    known = set(g["initializers"]) | {v["name"] for v in g["inputs"]}
    if len(known) != len(g["initializers"]) + len([v for v in g["inputs"] if v["name"] not in g["initializers"]]):
        raise ValueError("duplicate value names among graph inputs / initializers")
    for node in g["nodes"]:
        if node["domain"] not in ("", "ai.onnx"):
            raise ValueError("node %s uses non-standard domain %r" % (node["name"], node["domain"]))
        for name in node["inputs"]:
            if name and name not in known:
                raise ValueError("node %s (%s) reads %r before it is defined" % (node["name"], node["op_type"], name))
        for name in node["outputs"]:
            if name in known:
                raise ValueError("value %r assigned twice" % name)
            known.add(name)
    for v in g["outputs"]:
        if v["name"] not in known:
            raise ValueError("graph output %r is never produced" % v["name"])
    return True


def printable_graph(model):
    g = model["graph"]
    lines = ["graph %s (" % g["name"]]
    lines += ["  %%%s[%s]" % (v["name"], ", ".join(str(d) for d in v["shape"])) for v in g["inputs"]
              if v["name"] not in g["initializers"]]
    lines.append(") initializers (")
    lines += ["  %%%s[%s]" % (k, ", ".join(str(d) for d in a.shape)) for k, a in g["initializers"].items()]
    lines.append(") {")
    for n in g["nodes"]:
        attrs = ", ".join("%s=%s" % (k, v if not isinstance(v, np.ndarray) else "<tensor %s>" % (v.shape,))
                          for k, v in n["attrs"].items())
        lines.append("  %s = %s%s(%s)" % (", ".join("%" + o for o in n["outputs"]), n["op_type"],
                                          "[%s]" % attrs if attrs else "", ", ".join("%" + i for i in n["inputs"])))
    lines.append("  return %s\n}" % ", ".join("%" + v["name"] for v in g["outputs"]))
    return "\n".join(lines)
