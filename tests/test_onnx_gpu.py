"""The HIP engine's `dummy_forward` and the ONNX file exported from it (reference
scripts/onnx_export.py): the engine reproduces the real reference's `dummy_forward` (golden), and the export driver
writes a file that — evaluated by the independent operator interpreter — returns what the engine returns."""
import os
import tempfile
import warnings

import numpy as np
import pytest
import torch

from fsnet_amd.configs import meta_arch_cfg
from fsnet_amd.engine.runtime import RT
from fsnet_amd.export import onnx_graph as G
from fsnet_amd.scripts import onnx_export
from fsnet_amd.vision_base.utils.builder import build
from tests import helpers_onnx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

CFG = """
from easydict import EasyDict
from fsnet_amd.configs import meta_arch_cfg
cfg = EasyDict()
cfg.data = EasyDict(rgb_shape=(64, 128, 3))
cfg.trainer = EasyDict(gpu=0)
cfg.meta_arch = meta_arch_cfg(64, 128, with_pose=False)
"""


@pytest.fixture()
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    RT.set_compute_dtype("fp32")
    yield torch.device("cuda", 0)
    RT.set_compute_dtype("fp32")


def test_engine_dummy_forward_matches_reference(dev):
    sd0, image = helpers_onnx.case()
    m = build(**meta_arch_cfg(64, 128, with_pose=False)).to(dev)
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    RT.bump_weights()
    m.eval()
    with torch.no_grad():
        got = m.dummy_forward(image.to(dev))
    assert list(got.keys()) == ["depth"]
    want = np.load(os.path.join(GOLD, "onnx_dummy_forward.npz"))["depth"]
    err = np.abs(got["depth"].cpu().numpy() - want) / np.abs(want)
    assert err.max() < 1e-3, err.max()


def test_export_driver_file_evaluates_to_engine_output(dev):
    sd0, image = helpers_onnx.case()
    with tempfile.TemporaryDirectory() as d:
        cfgp, ckpt, onx = os.path.join(d, "cfg.py"), os.path.join(d, "ck.pth"), os.path.join(d, "metaarch.onnx")
        open(cfgp, "w").write(CFG)
        torch.save({"model_state_dict": sd0, "optimizer_state_dict": {}}, ckpt)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = onnx_export.main(["--config", cfgp, "--checkpoint_path", ckpt, "--onnx_file", onx])
        assert os.path.getsize(onx) > 50 << 20
        (from_file,) = helpers_onnx.run(G.read_model(onx), {"input": image})
    assert model["graph"]["inputs"][0]["shape"] == [1, 3, 64, 128]
    m = build(**meta_arch_cfg(64, 128, with_pose=False)).to(dev)
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    RT.bump_weights()
    m.eval()
    with torch.no_grad():
        got = m.dummy_forward(image.to(dev))["depth"].cpu()
    err = ((got - from_file).abs() / from_file.abs()).max()
    assert float(err) < 1e-3, float(err)
