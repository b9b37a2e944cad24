"""Seeded inputs of the ONNX export case (the construction tools/gen_golden.py's gen_onnx ran the reference's
MonoDepthWPose on to make tests/golden/onnx_dummy_forward.npz and onnx_graph.json), and a small interpreter of the
ONNX operators the depth network exports to — an independent evaluation of the exported FILE (onnxruntime, which the
reference's script uses for that, is not in the image)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import fsnet_oracle as O


def case(depth=18):
    sd0 = O.init_state(seed=11, depth=depth, with_pose=False)
    g = torch.Generator().manual_seed(12)
    for k in sd0:
        if k.endswith("running_mean"):
            sd0[k] = 0.1 * torch.randn(sd0[k].shape, generator=g)
        elif k.endswith("running_var"):
            sd0[k] = 0.5 + torch.rand(sd0[k].shape, generator=g)
    image = (O.synthetic_batch(1, 64, 128, seed=400)[("image", 0)]).float()
    return sd0, image


_CAST = {1: torch.float32, 6: torch.int32, 7: torch.int64, 9: torch.bool, 11: torch.float64}


def _ints(t):
    return [int(v) for v in torch.as_tensor(t).reshape(-1).tolist()]


def run(model, feeds):
    """evaluate a `read_model` dictionary (opset 11 semantics of the operators below) on torch CPU tensors"""
    g = model["graph"]
    env = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g["initializers"].items()}
    env.update({k: torch.as_tensor(v) for k, v in feeds.items()})
    for n in g["nodes"]:
        a, op = n["attrs"], n["op_type"]
        x = [env[i] if i else None for i in n["inputs"]]
        if op == "Constant":
            y = torch.from_numpy(np.ascontiguousarray(a["value"]))
        elif op == "Identity":
            y = x[0]
        elif op == "Conv":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3]
            y = F.conv2d(x[0], x[1], x[2] if len(x) > 2 else None, stride=a.get("strides", [1, 1]), padding=p[:2],
                         dilation=a.get("dilations", [1, 1]), groups=a.get("group", 1))
        elif op == "Relu":
            y = torch.relu(x[0])
        elif op == "Sigmoid":
            y = torch.sigmoid(x[0])
        elif op == "Reciprocal":
            y = 1 / x[0]
        elif op in ("Add", "Mul", "Sub", "Div"):
            y = {"Add": torch.add, "Mul": torch.mul, "Sub": torch.sub, "Div": torch.div}[op](x[0], x[1])
        elif op == "MaxPool":
            p = a.get("pads", [0, 0, 0, 0])
            assert p[0] == p[2] and p[1] == p[3]
            y = F.max_pool2d(x[0], a["kernel_shape"], a.get("strides", [1, 1]), p[:2], ceil_mode=bool(a.get("ceil_mode", 0)))
        elif op == "Concat":
            y = torch.cat(x, a["axis"])
        elif op == "ConstantOfShape":
            v = torch.from_numpy(np.ascontiguousarray(a["value"])) if "value" in a else torch.zeros(1)
            y = v.reshape(()).expand(_ints(x[0])).clone()
        elif op == "Reshape":
            shape = _ints(x[1])
            shape = [x[0].shape[k] if d == 0 else d for k, d in enumerate(shape)]
            y = x[0].reshape(shape)
        elif op == "Slice":
            starts, ends = _ints(x[1]), _ints(x[2])
            axes = _ints(x[3]) if len(x) > 3 and x[3] is not None else list(range(len(starts)))
            steps = _ints(x[4]) if len(x) > 4 and x[4] is not None else [1] * len(starts)
            y = x[0]
            for s, e, ax, st in zip(starts, ends, axes, steps):
                size = y.shape[ax]
                if st > 0:
                    s = min(max(s + size if s < 0 else s, 0), size)
                    e = min(max(e + size if e < 0 else e, 0), size)
                    idx = torch.arange(s, e, st)
                else:
                    s = min(max(s + size if s < 0 else s, 0), size - 1)
                    e = min(max(e + size if e < 0 else e, -1), size - 1)
                    idx = torch.arange(s, e, st)
                y = y.index_select(ax, idx)
        elif op == "Transpose":
            y = x[0].permute(a["perm"])
        elif op == "Cast":
            y = x[0].to(_CAST[a["to"]])
        elif op == "Pad":
            pads = _ints(x[1])
            r = x[0].dim()
            assert all(pads[k] == 0 and pads[r + k] == 0 for k in range(r - 2)), "spatial padding only"
            mode = a.get("mode", b"constant").decode()
            tp = (pads[r - 1], pads[2 * r - 1], pads[r - 2], pads[2 * r - 2])
            if mode == "constant":
                y = F.pad(x[0], tp, value=float(x[2]) if len(x) > 2 and x[2] is not None else 0.0)
            else:
                y = F.pad(x[0], tp, mode={"edge": "replicate", "reflect": "reflect"}[mode])
        elif op == "Resize":
            assert a.get("mode", b"nearest") == b"nearest"
            assert a.get("coordinate_transformation_mode", b"half_pixel") == b"asymmetric"
            assert a.get("nearest_mode", b"round_prefer_floor") == b"floor"
            scales = [float(v) for v in x[2].reshape(-1).tolist()]
            y = x[0]
            for ax, sc in enumerate(scales):
                if sc != 1.0:
                    n_out = int(np.floor(y.shape[ax] * sc))
                    src = torch.floor(torch.arange(n_out, dtype=torch.float64) / sc).long().clamp_(max=y.shape[ax] - 1)
                    y = y.index_select(ax, src)
        elif op == "Clip":
            y = x[0]
            if len(x) > 1 and x[1] is not None:
                y = torch.maximum(y, x[1].to(y.dtype))
            if len(x) > 2 and x[2] is not None:
                y = torch.minimum(y, x[2].to(y.dtype))
        elif op == "Softmax":
            ax = a.get("axis", 1)
            ax = ax + x[0].dim() if ax < 0 else ax
            flat = x[0].reshape(int(np.prod(x[0].shape[:ax])), -1)        # opset < 13: coerced to 2-D at `axis`
            y = torch.softmax(flat, 1).reshape(x[0].shape)
        elif op == "ReduceSum":
            y = x[0].sum(dim=a["axes"], keepdim=bool(a.get("keepdims", 1)))
        else:
            raise NotImplementedError("ONNX operator %s" % op)
        env[n["outputs"][0]] = y
        assert len(n["outputs"]) == 1 or op == "MaxPool"
    return [env[v["name"]] for v in g["outputs"]]
