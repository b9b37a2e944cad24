"""The training driver (mirror of scripts/train.py) end to end on a tiny synthetic config: cfg_from_file ->
build(**cfg...) -> dataloader -> hook -> scheduler -> checkpoint, loss decreasing."""
import os
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = '''
import numpy as np
from easydict import EasyDict as edict
from fsnet_amd.configs import meta_arch_cfg, training_cfg
cfg = edict()
cfg.path = edict(checkpoint_path=%r)
tc = training_cfg()
cfg.trainer = edict(gpu=0, max_epochs=2, disp_iter=1000, save_iter=1, test_iter=0, max_iters=0,
                    training_hook=tc.training_hook)
cfg.optimizer = edict(name='adam', lr=1e-4, weight_decay=0)
cfg.scheduler = edict(name='StepLR', step_size=15)
cfg.data = edict(batch_size=2, num_workers=0)
cfg.train_dataset = edict(name='fsnet_amd.vision_base.data.datasets.synthetic.SyntheticTripletDataset', size=8,
                          height=64, width=128)
cfg.meta_arch = meta_arch_cfg(64, 128, with_pose=False)
'''


def test_driver_runs_and_checkpoints(dev):
    from fsnet_amd.scripts import train
    with tempfile.TemporaryDirectory() as d:
        cfgp = os.path.join(d, "cfg.py")
        open(cfgp, "w").write(CFG % os.path.join(d, "ck"))
        model = train.main(["--config", cfgp, "--optimizer.lr=0.0002"])
        files = os.listdir(os.path.join(d, "ck"))
        assert any(f.endswith("_latest.pth") for f in files) and len(files) == 3
        ck = torch.load(os.path.join(d, "ck", [f for f in files if f.endswith("_latest.pth")][0]), map_location="cpu")
        assert "depth_backbone.conv1.weight" in ck["model_state_dict"]
        st = ck["optimizer_state_dict"]["state"]
        assert len(st) == len(list(model.parameters())) and float(st[0]["step"]) == 8.0
        assert ck["optimizer_state_dict"]["param_groups"][0]["lr"] == 0.0002
