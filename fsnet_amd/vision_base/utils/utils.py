"""cfg / registry helpers with the reference's names and behaviour (vision_base/utils/utils.py:12-169)."""
import importlib
import os
import random
import shutil
import sys
import tempfile

import numpy as np
import torch

from ._easydict import get_easydict

EasyDict = get_easydict()


def get_num_parameters(model):
    m = model.module if hasattr(model, "module") else model
    return sum(p.numel() for p in m.parameters() if p.requires_grad)


def set_random_seed(seed, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def cfg_from_file(cfg_filename):
    """A config is a python file defining `cfg = EasyDict()`; it is copied to a temp dir and imported."""
    assert cfg_filename.endswith(".py")
    with tempfile.TemporaryDirectory() as tmp:
        name = "fsnet_cfg_%s" % next(tempfile._get_candidate_names())
        shutil.copyfile(cfg_filename, os.path.join(tmp, name + ".py"))
        sys.path.insert(0, tmp)
        try:
            cfg = getattr(importlib.import_module(name), "cfg")
        finally:
            sys.path.pop(0)
            sys.modules.pop(name, None)
    assert isinstance(cfg, EasyDict)
    return cfg


def update_dict(obj, key, rest_items, value):
    if len(rest_items) == 0:
        obj[key] = value
        return obj
    if not (key in obj and isinstance(obj[key], dict)):
        obj[key] = EasyDict()
    obj[key] = update_dict(obj[key], rest_items[0], rest_items[1:], value)
    return obj


def update_cfg(cfg, **kwargs):
    """dotted-key command line overrides: update_cfg(cfg, **{'a.b.c': 1})."""
    for key, value in kwargs.items():
        items = key.split(".")
        cfg = update_dict(cfg, items[0], items[1:], value)
    return cfg


def merge_name(names):
    return ".".join(names)


def find_object(object_string):
    """Import the longest importable dotted prefix, then getattr down the rest."""
    parts = object_string.split(".")
    traces = []
    for i in range(len(parts), 0, -1):
        prefix = merge_name(parts[:i])
        try:
            obj = importlib.import_module(prefix)
            for name in parts[i:]:
                obj = getattr(obj, name)
            return obj
        except Exception as e:  # noqa: BLE001 - mirrored behaviour: collect and report every attempt
            traces.append((prefix, e))
    log = "".join("%s : %s \n" % (n, e) for n, e in traces)
    raise ModuleNotFoundError("%s not imported, error traces: \n%s" % (object_string, log))
