"""CNN feature encoder: the parameter container and the PyTorch-ROCm / MIOpen forward (`model.encoder_backend = "torch"`).

The DEFAULT back end since round 3 is `encoder_hip.py` (implicit-GEMM convolutions on SH NHWC activations, csrc/conv_pp.hip +
encoder.hip: `model.encoder_backend = "hip"`); this module defines the layers -- so that reference checkpoints load into the
same `fnet.*` keys -- and remains the selectable torch path and the reference the HIP encoder is tested against.

Architecture of the reference ``BasicEncoder`` (cotracker/models/core/cotracker/blocks.py:141-219,
residual unit :79-138) re-expressed with the same parameter names so that reference
checkpoints load unchanged (``fnet.*`` keys): 7x7/2 stem -> four 2-unit residual
stages (64, 96, 128, 128 channels; strides 1,2,2,2) with parameter-free instance norm ->
the four stage outputs resized to 1/stride resolution, concatenated (416 ch) -> 3x3 conv to
256 -> instance norm, ReLU -> 1x1 conv to 128.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


class _ResUnit(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        # reference: nn.Sequential(conv1x1, norm3) where norm3 has no parameters -> key "downsample.0.*"
        self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride)) if stride != 1 else None

    def forward(self, x):
        y = F.relu(_inorm(self.conv1(x)))
        y = F.relu(_inorm(self.conv2(y)))
        if self.downsample is not None:
            x = _inorm(self.downsample(x))
        return F.relu(x + y)


class BasicEncoder(nn.Module):
    def __init__(self, input_dim=3, output_dim=128, stride=4):
        super().__init__()
        self.stride = stride
        c = output_dim
        self.conv1 = nn.Conv2d(input_dim, c // 2, 7, stride=2, padding=3)
        self.layer1 = nn.Sequential(_ResUnit(c // 2, c // 2, 1), _ResUnit(c // 2, c // 2, 1))
        self.layer2 = nn.Sequential(_ResUnit(c // 2, c // 4 * 3, 2), _ResUnit(c // 4 * 3, c // 4 * 3, 1))
        self.layer3 = nn.Sequential(_ResUnit(c // 4 * 3, c, 2), _ResUnit(c, c, 1))
        self.layer4 = nn.Sequential(_ResUnit(c, c, 2), _ResUnit(c, c, 1))
        self.conv2 = nn.Conv2d(c * 3 + c // 4, c * 2, 3, padding=1)
        self.conv3 = nn.Conv2d(c * 2, c, 1)

    def forward(self, x):
        H, W = x.shape[-2:]
        size = (H // self.stride, W // self.stride)
        x = F.relu(_inorm(self.conv1(x)))
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
            feats.append(F.interpolate(x, size, mode="bilinear", align_corners=True))
        x = self.conv2(torch.cat(feats, dim=1))
        x = F.relu(_inorm(x))
        return self.conv3(x)
