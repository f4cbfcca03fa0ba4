#!/usr/bin/env python3
"""Phase timing of the F(4x4,3x3) kernel (build with -DW44_STAMP: tools/w44_variants.sh "stamp:-DW44_STAMP"): cycles each wave spends in
the first MFMA part, at barrier X1 (with its vmcnt wait), in the second MFMA part + input transform, at X2, and in the rest."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402

_hip._LIB_PATH = os.path.join(ROOT, os.environ.get("W44_LIB", "build/w44v/lib_stamp.so"))
g = torch.Generator(device="cuda").manual_seed(5)
w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
x = torch.randn(64, 64, 128, 128, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
U = _hip.pack_winograd44_weights(w)
out = torch.empty_like(x)
stamps = torch.zeros(256 * 8 * 5, dtype=torch.int32, device="cuda")
for _ in range(3):
    _hip.conv3x3_c64_winograd44(x, U, stamps.view(torch.float32), True, out=out)
torch.cuda.synchronize()
s = stamps.view(256, 8, 5).double().cpu()
names = ["mfma part 1", "wait + X1", "mfma part 2 + transform", "wait + X2", "DMA issue + (1 stage in 8) output transform, last input transform"]
tot = s.sum(-1)
print("cycles per wave over the launch: mean %.0f (min %.0f max %.0f); 64 stages per workgroup" % (tot.mean(), tot.min(), tot.max()))
for i, nme in enumerate(names):
    print("  %-28s mean %8.0f per stage %7.1f   (waves 0-3: %7.1f, waves 4-7: %7.1f)" % (nme, s[..., i].mean(), s[..., i].mean() / 64, s[:, :4, i].mean() / 64, s[:, 4:, i].mean() / 64))
