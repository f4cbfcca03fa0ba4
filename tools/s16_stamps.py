#!/usr/bin/env python3
"""Phase timing of the split-fp16 conv kernel (build with -DS16_STAMP: tools/s16_variants.sh "stamp:-DS16_STAMP"): shader cycles each wave spends
in the 18 MFMA groups of a stage, waiting for the next chunk's DMA (s_waitcnt vmcnt(0)), at the stage barrier, in the epilogue, and in the rest."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEQSCI_HIP_LIB", os.path.join(ROOT, "build/s16v/lib_stamp.so"))
from deqsci_amd import _hip  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(5)
w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
n = int(os.environ.get("S16_IMAGES", "64"))
x = torch.randn(n, 64, 128, 128, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
xs = _hip.to_split16(x)
out = _hip.Sp16.empty(n, 128, 128, "cuda")
Wsp = _hip.Split16Weights(w)
NW = int(os.environ.get("S16_NW", "8"))                 # waves per workgroup of the build under test (-DS16_GEOM_NW)
stamps = torch.zeros(256 * NW * 5, dtype=torch.int32, device="cuda")
for _ in range(3):
    _hip.conv3x3_c64_split16(xs, Wsp, stamps.view(torch.float32), True, out=out)
torch.cuda.synchronize()
s = stamps.view(256, NW, 5).double().cpu()
tiles = n * 32 / 256
stages = 4 * tiles
names = ["18 MFMA groups (+ DMA issue, operand reads)", "s_waitcnt vmcnt(0) (next chunk's DMA)", "stage barrier", "epilogue", "rest (tile setup, prologue)"]
tot = s.sum(-1)
print("cycles per wave over the launch: mean %.0f (min %.0f max %.0f); %.0f tiles = %.0f stages per workgroup; ideal MFMA issue per stage and SIMD 216 x 32 = 6912"
      % (tot.mean(), tot.min(), tot.max(), tiles, stages))
for i, nme in enumerate(names):
    per = stages if i < 3 else tiles
    print("  %-46s %5.1f %%  per %s %8.1f   (first half of the waves: %8.1f, second half: %8.1f)" % (nme, 100 * s[..., i].sum() / tot.sum(), "stage" if i < 3 else "tile ",
                                                                                       s[..., i].mean() / per, s[:, :NW // 2, i].mean() / per, s[:, NW // 2:, i].mean() / per))
blk = tot.mean(1)
print("per-workgroup total: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (blk.min(), blk.quantile(0.1), blk.median(), blk.quantile(0.9), blk.max()))
