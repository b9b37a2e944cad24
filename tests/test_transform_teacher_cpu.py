"""Stage-1 checkpoint -> teacher weights (mirror of monodepth/transform_teacher.py) against the reference's own
output on the reference meta-arch's keys (tests/golden/teacher_keys.json, made by tools/gen_golden.py)."""
import json
import os

import torch

from fsnet_amd.monodepth.transform_teacher import teacher_state_dict, transform_teacher_model
from oracle import fsnet_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "teacher_keys.json")


def test_teacher_keys_match_reference(tmp_path):
    rec = json.load(open(GOLD))
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.vision_base.utils.builder import build
    m = build(**meta_arch_cfg(64, 128, with_pose=True))
    m.load_state_dict(O.init_state(seed=1, with_pose=True), strict=True)
    sd = m.state_dict()
    assert list(sd.keys()) == rec["src_keys"]                 # the same checkpoint the reference was given
    src, dst = str(tmp_path / "stage1.pth"), str(tmp_path / "teacher.pth")
    torch.save({"model_state_dict": sd, "optimizer_state_dict": {}}, src)
    transform_teacher_model(src, dst)
    out = torch.load(dst, map_location="cpu")
    assert rec["is_bare_state_dict"] and "model_state_dict" not in out
    assert list(out.keys()) == rec["dst_keys"]
    assert [float(v.double().sum()) for v in out.values()] == rec["dst_sums"]
    assert not any(k.startswith(("head.", "pose")) for k in out)


def test_teacher_weights_load_into_the_teacher_model():
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.vision_base.utils.builder import build
    cfg = meta_arch_cfg(64, 128, with_pose=True)
    teacher = build(name="fsnet_amd.monodepth.networks.models.meta_archs.teacher_model.MonoDepthInference",
                    backbone_cfg=cfg.depth_backbone_cfg, depth_head_cfg=cfg.head_cfg.depth_decoder_cfg)
    sd = teacher_state_dict(O.init_state(seed=1, with_pose=True))
    missing, unexpected = teacher.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
