"""Meta-arch contract (reference vision_base/networks/models/meta_archs/base_meta.py:3-23):
forward(data, meta) dispatches on meta['is_training']."""
import torch.nn as nn


class BaseMetaArch(nn.Module):
    def forward_train(self, data, meta):
        raise NotImplementedError

    def forward_test(self, data, meta):
        raise NotImplementedError

    def dummy_forward(self, data):
        return dict()

    def forward(self, data, meta):
        return self.forward_train(data, meta) if meta["is_training"] else self.forward_test(data, meta)
