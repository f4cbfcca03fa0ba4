"""FFDNet denoiser with the reference's module API (networks/ffdnet/models.py:27-108).

Same constructor (`FFDNet(num_input_channels, tag)`), same parameter names
(`intermediate_dncnn.itermediate_dncnn.{0,2,3,5,6,...}` - the reference's spelling), same
forward contract: `forward(x, noise_sigma)` with x (N,C,H,W), noise_sigma (N,) returns the
predicted NOISE.  The reference's explicit strided-slice loops for the 2x2 de-interleave
(functions.py:16-53, channel = 4c + 2i + j, sigma map first) and its inverse (:62-81) are
exactly pixel_unshuffle / pixel_shuffle, which is what runs here; the convolutions are
torch.nn.Conv2d on PyTorch-ROCm (MIOpen).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class IntermediateDnCNN(nn.Module):
    def __init__(self, input_features, middle_features, num_conv_layers):
        super().__init__()
        if input_features == 5:
            output_features = 4          # grayscale
        elif input_features == 15:
            output_features = 12         # RGB
        else:
            raise Exception('Invalid number of input features')
        self.input_features = input_features
        self.middle_features = middle_features
        self.num_conv_layers = num_conv_layers
        self.output_features = output_features
        conv = lambda cin, cout: nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False)
        layers = [conv(input_features, middle_features), nn.ReLU(inplace=True)]
        for _ in range(num_conv_layers - 2):
            layers += [conv(middle_features, middle_features), nn.BatchNorm2d(middle_features), nn.ReLU(inplace=True)]
        layers.append(conv(middle_features, output_features))
        self.itermediate_dncnn = nn.Sequential(*layers)

    def forward(self, x):
        return self.itermediate_dncnn(x)


class FFDNet(nn.Module):
    def __init__(self, num_input_channels, tag):
        super().__init__()
        self.num_input_channels = num_input_channels
        self.tag = tag
        if num_input_channels == 1:
            self.num_feature_maps, self.num_conv_layers = 64, 15
            self.downsampled_channels, self.output_features = 5, 4
        elif num_input_channels == 3:
            self.num_feature_maps, self.num_conv_layers = 96, 12
            self.downsampled_channels, self.output_features = 15, 12
        else:
            raise Exception('Invalid number of input features')
        self.intermediate_dncnn = IntermediateDnCNN(input_features=self.downsampled_channels,
                                                    middle_features=self.num_feature_maps,
                                                    num_conv_layers=self.num_conv_layers)

    @staticmethod
    def concatenate_input_noise_map(x, noise_sigma):
        N, C, H, W = x.shape
        noise_map = noise_sigma.reshape(N, 1, 1, 1).expand(N, C, H // 2, W // 2)
        return torch.cat((noise_map, F.pixel_unshuffle(x, 2)), 1)

    def forward(self, x, noise_sigma):
        h = self.intermediate_dncnn(self.concatenate_input_noise_map(x.detach(), noise_sigma))
        return F.pixel_shuffle(h, 2)
