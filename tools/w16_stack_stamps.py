#!/usr/bin/env python3
"""Phase timing of the STACK launch of csrc/conv_w16.hip (build with -DW16_STAMP: tools/w16_variants.sh "stamp:-DW16_STAMP") at ONE tile per CU -
8 images of 128 x 128 = one measurement per call, the reference's usage: shader cycles each wave spends per LAYER in its MFMA streams, waiting for
DMA, at barriers, in the epilogue, in the prologue / tile setup, and in the slow path (= waiting for the neighbours' progress words)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEQSCI_HIP_LIB", os.path.join(ROOT, "build/w16v/lib_stamp.so"))
from deqsci_amd import _hip  # noqa: E402

n, L = int(os.environ.get("W16_IMAGES", "8")), int(os.environ.get("W16_LAYERS", "13"))
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.relu(torch.randn(n, 64, 128, 128, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
ws = [_hip.Wino16Weights(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.04) for _ in range(L)]
NW = 8
stamps = torch.zeros(256 * NW * 8, dtype=torch.int32, device="cuda")           # layer 0's "bias": the kernel writes its stamps over it
bs = [stamps.view(torch.float32)] + [torch.zeros(64, device="cuda") for _ in range(L - 1)]
stack = _hip.Wino16Stack(list(zip(ws, bs, [True] * L)), "cuda")
xp = _hip.P32.from_nchw(x)
for _ in range(3):
    stamps.zero_()
    _hip.conv3x3_c64_wino16_stack(xp, stack, check=False)
torch.cuda.synchronize()
s = stamps.view(256, NW, 8).double().cpu()
tiles = n * 32 / 256
names = ["MFMA stream h = 0", "MFMA stream h = 1", "DMA wait h = 0", "DMA wait h = 1", "barrier", "epilogue", "rest (prologue, tile setup, stores out)", "slow path (neighbours' words)"]
tot = s.sum(-1)
print("stack of %d layers, %d images (%.0f tile(s) per workgroup and layer): cycles per wave over the launch: mean %.0f (min %.0f max %.0f) = %.0f per layer"
      % (L, n, tiles, tot.mean(), tot.min(), tot.max(), tot.mean() / L))
for i, nme in enumerate(names):
    print("  %-42s %5.1f %%  per layer %8.1f   (waves: %s)" % (nme, 100 * s[..., i].sum() / tot.sum(), s[..., i].mean() / L,
                                                                 " ".join("%8.1f" % (s[:, k, i].mean() / L) for k in range(NW))))
