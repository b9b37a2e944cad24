"""Host-side plugin surface on CPU: registry, cfg helpers, schedulers, collate, checkpoint names, and the
'no CPU fallback' guarantee of the product path."""
import os
import tempfile

import numpy as np
import pytest
import torch

from fsnet_amd.vision_base.utils import builder, utils
from fsnet_amd.vision_base.utils.utils import EasyDict


def test_build_and_find_object():
    assert builder.build("numpy.exp", 0.0) == 1.0
    assert utils.find_object("torch.sigmoid") is torch.sigmoid
    with pytest.raises(ModuleNotFoundError) as e:
        utils.find_object("fsnet_amd.does_not.exist")
    assert "error traces" in str(e.value)


def test_combinators():
    seq = builder.Sequential([dict(name="operator.methodcaller", ), ], ) if False else None
    cfg = [dict(name="functools.partial", func=np.add), ]
    # children built through the registry with shared keywords merged in
    s = builder.Sequential([dict(name="tests.helpers_cpu.AddOne"), dict(name="tests.helpers_cpu.Double")])
    assert s(3) == 8
    p = builder.Parallel([dict(name="tests.helpers_cpu.AddOne"), dict(name="tests.helpers_cpu.Double")])
    assert p(3) == [4, 6]
    sh = builder.Shuffle([dict(name="tests.helpers_cpu.AddOne"), dict(name="tests.helpers_cpu.Double")])
    np.random.seed(0)
    assert sh(3) in (7, 8)


def test_cfg_from_file_and_update_cfg():
    src = "from easydict import EasyDict\ncfg = EasyDict()\ncfg.a = 1\ncfg.b = EasyDict(c=0, f=2)\ncfg.c = 3\n"
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "example.py")
        open(path, "w").write(src)
        cfg = utils.cfg_from_file(path)
    assert isinstance(cfg, EasyDict) and cfg.b.f == 2
    cfg = utils.update_cfg(cfg, **{"a": 2, "b.c": 3, "d.e.f": 4, "c.g": 1})
    assert cfg["b"]["f"] == 2 and cfg["a"] == 2 and cfg["b"]["c"] == 3
    assert isinstance(cfg["d"]["e"], dict) and cfg["d"]["e"]["f"] == 4 and cfg["c"]["g"] == 1


def test_schedulers_and_optimizer_names():
    from fsnet_amd.vision_base.networks.optimizers import optimizers, schedulers
    m = torch.nn.Linear(2, 2)
    opt = optimizers.build_optimizer(m, name="adam", lr=1e-4, weight_decay=0)
    assert isinstance(opt, optimizers.FusedAdam) and opt.param_groups[0]["lr"] == 1e-4
    sch = schedulers.build_scheduler(opt, name="StepLR", step_size=2)
    for _ in range(2):
        sch.step()
    assert abs(opt.param_groups[0]["lr"] - 1e-5) < 1e-12
    assert isinstance(schedulers.build_scheduler(opt), torch.optim.lr_scheduler.ExponentialLR)
    with pytest.raises(NotImplementedError):
        schedulers.build_scheduler(opt, name="nope")
    with pytest.raises(NotImplementedError):
        optimizers.build_optimizer(m, name="lion")
    poly = schedulers.build_scheduler(optimizers.build_optimizer(m, name="sgd", lr=1.0), name="PolyLR", gamma=0.9, n_iteration=10)
    poly.step()
    assert 0 < poly.get_last_lr()[0] < 1.0


def test_collate_fn_keeps_shared_keys_only():
    from fsnet_amd.vision_base.data.datasets.dataset_utils import collate_fn
    a = {("image", 0): torch.zeros(3, 4, 5), "P2": np.eye(3, 4, dtype=np.float32), "calib_meta": {"k": 1}, "only_a": 1}
    b = {("image", 0): torch.ones(3, 4, 5), "P2": np.eye(3, 4, dtype=np.float32), "calib_meta": {"k": 2}}
    out = collate_fn([a, b])
    assert set(out) == {("image", 0), "P2", "calib_meta"}
    assert out[("image", 0)].shape == (2, 3, 4, 5) and out["P2"].dtype == torch.float32 and out["calib_meta"][1]["k"] == 2


def test_product_path_has_no_cpu_fallback():
    from fsnet_amd.configs import meta_arch_cfg
    m = builder.build(**meta_arch_cfg(64, 128, with_pose=True)).train()
    from oracle import fsnet_oracle as O
    data = O.synthetic_batch(1, 64, 128, seed=0)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(data, dict(is_training=True))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.depth_backbone(data[("image", 0)])


def test_product_path_never_imports_the_oracle():
    import subprocess, sys
    code = ("import sys; import fsnet_amd.configs, fsnet_amd.vision_base.utils.builder as b; "
            "from fsnet_amd.configs import meta_arch_cfg; b.build(**meta_arch_cfg(64,128,True)); "
            "import fsnet_amd.vision_base.pipeline_hooks.train_val_hooks.base_training_hooks; "
            "assert not any(k == 'oracle' or k.startswith('oracle.') for k in sys.modules), 'oracle imported'")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_checkpoint_roundtrip_keeps_reference_format():
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.networks.utils.utils import load_models, save_models
    m = builder.build(**meta_arch_cfg(64, 128, with_pose=False))
    opt = build_optimizer(m, name="adam", lr=1e-4)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ck.pth")
        save_models(path, m, opt)
        ck = torch.load(path, map_location="cpu")
        assert set(ck) == {"model_state_dict", "optimizer_state_dict"}
        assert "depth_backbone.layer2.0.downsample.0.weight" in ck["model_state_dict"]
        assert "head.depth_decoder.decoder.10.weight" in ck["model_state_dict"]
        m2 = builder.build(**meta_arch_cfg(64, 128, with_pose=False))
        load_models(path, m2, None, map_location="cpu", strict=True)
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
