#!/usr/bin/env python3
"""Phase timing of the split-fp16 Winograd kernel (build with -DW16_STAMP: tools/w16_variants.sh "stamp:-DW16_STAMP"): shader cycles each wave
spends in the MFMA stream of a half-stage (72 MFMAs + transform + DMA issue + operand reads), waiting for the DMA, at the barrier, in the
epilogue, in the rest.  W16_IMAGES (64)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEQSCI_HIP_LIB", os.path.join(ROOT, "build/w16v/lib_stamp.so"))
from deqsci_amd import _hip  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(5)
w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
n = int(os.environ.get("W16_IMAGES", "64"))
fmt = "p32"
x = torch.relu(torch.randn(n, 64, 128, 128, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
xs = _hip.P32.from_nchw(x)
out = _hip.P32.empty(n, 128, 128, "cuda")
Ww = _hip.Wino16Weights(w)
NW = 8
stamps = torch.zeros(256 * NW * 8, dtype=torch.int32, device="cuda")
for _ in range(3):
    _hip.conv3x3_c64_wino16(xs, Ww, stamps.view(torch.float32), True, out=out)
torch.cuda.synchronize()
s = stamps.view(256, NW, 8).double().cpu()
tiles = n * 32 / 256
names = ["MFMA stream h = 0 (36 MFMAs)", "MFMA stream h = 1 (36 MFMAs)", "DMA wait h = 0", "DMA wait h = 1", "barrier", "epilogue", "rest (prologue, tile setup)", "slow path"]
per = [4 * tiles, 4 * tiles, 4 * tiles, 4 * tiles, 8 * tiles, tiles, 1, tiles]
unit = ["half-stage"] * 5 + ["tile", "launch", "tile"]
tot = s.sum(-1)
print("%s, %d images: cycles per wave over the launch: mean %.0f (min %.0f max %.0f); %.0f tiles per workgroup; ideal MFMA issue per half-stage and SIMD 2 x 36 x 32 = 2304"
      % (fmt, n, tot.mean(), tot.min(), tot.max(), tiles))
for i, nme in enumerate(names):
    print("  %-34s %5.1f %%  per %-10s %8.1f   (waves: %s)" % (nme, 100 * s[..., i].sum() / tot.sum(), unit[i], s[..., i].mean() / per[i],
                                                                     " ".join("%8.1f" % (s[:, k, i].mean() / per[i]) for k in range(NW))))
