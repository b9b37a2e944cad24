"""ONNX interchange (reference scripts/onnx_export.py:39-52): the file `fsnet_amd` writes for
`dummy_forward` against (a) the file the REAL reference writes for the same weights — same operator histogram,
same graph signature, same initializer shapes (tests/golden/onnx_graph.json) — and (b) the reference's
`dummy_forward` output, by evaluating the exported file with an independent operator interpreter
(tests/golden/onnx_dummy_forward.npz)."""
import collections
import io
import json
import os
import warnings

import numpy as np
import pytest
import torch

from fsnet_amd.configs import meta_arch_cfg
from fsnet_amd.export import onnx_graph as G
from fsnet_amd.vision_base.utils.builder import build
from tests import helpers_onnx

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def exported():
    sd0, image = helpers_onnx.case()
    m = build(**meta_arch_cfg(64, 128, with_pose=False))
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m.eval()
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G.export(m, torch.zeros(1, 3, 64, 128), f)
    return m, image, f.getvalue()


def test_exported_graph_matches_the_reference_export(exported):
    _, _, blob = exported
    ref = json.load(open(os.path.join(GOLD, "onnx_graph.json")))
    model = G.read_model(blob)
    assert G.check_model(model)
    g = model["graph"]
    assert model["ir_version"] == ref["ir_version"] and model["opsets"] == ref["opsets"]
    assert dict(collections.Counter(n["op_type"] for n in g["nodes"])) == ref["ops"]
    assert [v for v in g["inputs"] if v["name"] not in g["initializers"]] == ref["inputs"]
    assert [dict(name=v["name"], elem_type=v["elem_type"], rank=len(v["shape"])) for v in g["outputs"]] == ref["outputs"]
    assert sorted([list(a.shape) for a in g["initializers"].values()]) == ref["initializer_shapes"]
    assert abs(len(blob) - ref["n_bytes"]) < 1 << 16          # value names differ, tensors do not
    text = G.printable_graph(model)
    assert "Softmax" in text and text.rstrip().endswith("}")


def test_exported_file_reproduces_reference_dummy_forward(exported):
    _, image, blob = exported
    want = np.load(os.path.join(GOLD, "onnx_dummy_forward.npz"))["depth"]
    (got,) = helpers_onnx.run(G.read_model(blob), {"input": image})
    assert tuple(got.shape) == want.shape == (1, 1, 64, 128)
    err = np.abs(got.numpy() - want) / np.abs(want)
    assert err.max() < 2e-4, err.max()


def test_export_refuses_training_mode_and_leaves_forward_alone(exported):
    m, _, _ = exported
    fwd = m.forward
    m.train()
    with pytest.raises(RuntimeError):
        G.export(m, torch.zeros(1, 3, 64, 128), io.BytesIO())
    m.eval()
    assert m.forward == fwd


def test_checker_rejects_broken_graphs(exported):
    _, _, blob = exported
    model = G.read_model(blob)
    model["graph"]["nodes"].reverse()
    with pytest.raises(ValueError):
        G.check_model(model)
    with pytest.raises(ValueError):
        G.check_model(G.read_model(b""))


def test_resnet50_sigmoid_head_export_evaluates_to_oracle():
    """Bottleneck encoder + the sigmoid-disparity DepthDecoder (configs/multi_dataset_example's encoder with the base
    class head): the exported file against the oracle's eval-mode forward (the oracle restates resnet.py:69-89 and
    depth_encoder.py:90-111 and is pinned to the reference by model_r50fx.npz / sigmoid_decoder.npz)."""
    from oracle import fsnet_oracle as O
    sd0 = O.init_state(seed=21, depth=50, with_pose=False, num_out=1, gain=0.8)    # keeps the sigmoid off its rails
    g = torch.Generator().manual_seed(22)
    for k in sd0:
        if k.endswith("running_mean"):
            sd0[k] = 0.1 * torch.randn(sd0[k].shape, generator=g)
        elif k.endswith("running_var"):
            sd0[k] = 0.5 + torch.rand(sd0[k].shape, generator=g)
    cfg = meta_arch_cfg(64, 128, with_pose=False, depth=50, num_output_channels=1)
    dd = cfg.head_cfg.depth_decoder_cfg
    dd.name = dd.name.replace("MultiChannelDepthDecoder", "DepthDecoder")
    m = build(**cfg)
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    m.eval()
    f = io.BytesIO()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G.export(m, torch.zeros(1, 3, 64, 128), f)
    model = G.read_model(f)
    assert G.check_model(model)
    ops = collections.Counter(n["op_type"] for n in model["graph"]["nodes"])
    assert ops["Conv"] == 53 + 10 + 1 and ops["Sigmoid"] == 1 and "Softmax" not in ops
    image = O.synthetic_batch(1, 64, 128, seed=401)[("image", 0)].float()
    (got,) = helpers_onnx.run(model, {"input": image})
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd0.items()}
    feats = O.resnet_forward(sd, "depth_backbone.", image.double(), depth=50, train=False)
    want = O.depth_decoder_forward(sd, "head.depth_decoder.", feats, 0.5, 100.0, train=False, sigmoid=True)[("depth", 0, 0)]
    err = ((got.double() - want).abs() / want.abs()).max()
    assert float(err) < 1e-4, float(err)
    assert float(want.max() / want.min()) > 5
