"""Checkpoint format of the reference (vision_base/networks/utils/utils.py:3-19):
torch.save({'model_state_dict', 'optimizer_state_dict'}); keys identical to the reference's modules, so
checkpoints interchange in both directions."""
import torch

from fsnet_amd.engine.runtime import RT


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def save_models(output_path, model, optimizer):
    torch.save({'model_state_dict': _unwrap(model).state_dict(),
                'optimizer_state_dict': optimizer.state_dict()}, output_path)


def load_models(path, model, optimizer=None, map_location="cuda:0", strict=False):
    checkpoint = torch.load(path, map_location=map_location)
    _unwrap(model).load_state_dict(checkpoint['model_state_dict'], strict=strict)
    RT.bump_weights()
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
