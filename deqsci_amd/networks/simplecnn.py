"""DE-GAP-CNN denoiser ("SimpleCNN") with the reference's module API
(networks/provable/model/SimpleCNN_models.py:6-61): `DnCNN(channels, num_of_layers, lip, no_bn,
adaptive, tag)` with the layers in `self.dncnn`, so `cnn.ckpt` keys `dncnn.{0,2,4,6}.weight` load
unchanged.  `lip > 0` selects real-spectral-norm convolutions (conv_sn_chen.py:16-93); in eval mode
those use their stored, already normalised `weight` buffer, which is all inference needs
(`rsn_cnn.ckpt` keys `weight_orig / weight / weight_u`).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class RealSNConv2d(nn.Module):
    """Inference-only stand-in for conv_spectral_norm(nn.Conv2d(..., bias=False)): same state-dict
    entries; eval-mode forward = conv2d with the stored normalised weight (conv_sn_chen.py:65-67)."""

    def __init__(self, cin, cout, sigma=1.0):
        super().__init__()
        self.sigma = sigma
        w = torch.empty(cout, cin, 3, 3)
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight_orig = nn.Parameter(w)
        self.register_buffer("weight", w.detach().clone())
        self.register_buffer("weight_u", torch.zeros(1, 1 if cout == 1 else 64, 40, 40))

    def forward(self, x):
        if self.training:
            raise NotImplementedError("RealSN power iteration (training) is outside the inference hot path")
        return F.conv2d(x, self.weight, padding=1)


class DnCNN(nn.Module):
    def __init__(self, channels, num_of_layers=17, lip=1.0, no_bn=False, adaptive=False, tag='denoiser'):
        super().__init__()
        self.tag = tag
        features = 64
        sigmas = [pow(lip, 1.0 / num_of_layers) if lip > 0.0 else 0.0 for _ in range(num_of_layers)]
        if adaptive:
            sigmas = [5.0, 2.0, 1.0, 0.681, 0.464, 0.316]
            if len(sigmas) != num_of_layers:
                raise AssertionError(f"adaptive spectral-norm schedule has {len(sigmas)} entries, the network {num_of_layers} layers")

        def conv_layer(cin, cout, sigma):
            if sigma > 0.0:
                return RealSNConv2d(cin, cout, sigma)
            return nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False)

        layers = [conv_layer(channels, features, sigmas[0]), nn.ReLU(inplace=True)]
        for i in range(1, num_of_layers - 1):
            layers.append(conv_layer(features, features, sigmas[i]))
            if not no_bn:
                layers.append(nn.BatchNorm2d(features))
            layers.append(nn.ReLU(inplace=True))
        layers.append(conv_layer(features, channels, sigmas[-1]))
        self.dncnn = nn.Sequential(*layers)

    def forward(self, x):
        return self.dncnn(x)
