"""ConvBnReLU parameter container (reference vision_base/networks/blocks/blocks.py:33-54): keys
`sequence.0.{weight,bias}` (conv) and `sequence.1.*` (BatchNorm2d).  The arithmetic runs in the
HIP engine (fsnet_amd/engine/nets.py); standalone forward is a single fused conv+BN+ReLU unit."""
import torch.nn as nn


class ConvBnReLU(nn.Module):
    def __init__(self, input_features=1, output_features=1, kernel_size=(1, 1), stride=[1, 1], padding='SAME',
                 dilation=1, groups=1, relu=True, **kwargs):
        super().__init__()
        if isinstance(kernel_size, int):
            kernel_size = (kernel_size, kernel_size)
        if dilation != 1 or groups != 1:
            raise NotImplementedError("HIP ConvBnReLU supports dilation=1, groups=1")
        pad = int((kernel_size[0] - 1) / 2) if padding.lower() == 'same' else 0
        self.sequence = nn.Sequential(
            nn.Conv2d(input_features, output_features, kernel_size=kernel_size, stride=stride, padding=pad, **kwargs),
            nn.BatchNorm2d(output_features),
        )
        self.relu = True

    def forward(self, x):
        raise NotImplementedError(
            "ConvBnReLU is executed by its owning network's HIP runner (DepthDecoder); standalone use is not on "
            "the monodepth hot path")
