#!/usr/bin/env python3
"""Phase timing of the split-fp16 STACK launch (a run of 64->64 layers in one launch, csrc/conv_s16.hip; build with -DS16_STAMP:
tools/s16_variants.sh "stamp:-DS16_STAMP"): shader cycles per layer each wave spends in the MFMA groups, at the stage barriers, in the
epilogue, and BETWEEN two layers - waiting for its stores, at the barrier behind them, publishing its progress word and waiting for the
eight neighbours', fetching the first chunk of the next layer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEQSCI_HIP_LIB", os.path.join(ROOT, "build/s16v/lib_stamp.so"))
from deqsci_amd import _hip  # noqa: E402

n, H, W, L = int(os.environ.get("S16_IMAGES", "8")), 128, 128, int(os.environ.get("S16_LAYERS", "13"))
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.relu(torch.randn(n, 64, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
xs = _hip.to_split16(x)
Ws = [_hip.Split16Weights(torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.04) for _ in range(L)]
tiles = n * (H // 16) * (W // 32)
stamps = torch.zeros(tiles * 8 * 9, dtype=torch.int32, device="cuda")
stack = _hip.Split16Stack([(w, stamps.view(torch.float32) if i == 0 else None, True) for i, w in enumerate(Ws)], "cuda")
for _ in range(3):
    _hip.conv3x3_c64_split16_stack(xs, stack)
torch.cuda.synchronize()
assert not stack.timed_out()
s = stamps.view(tiles, 8, 9).double().cpu()
tot = s.sum(-1)
names = ["MFMA groups (+ DMA issue, operand reads)", "s_waitcnt vmcnt(0) (next chunk's DMA)", "stage barrier", "epilogue", "rest (prologue of the launch)",
         "between layers: wait for own stores", "between layers: barrier behind the stores", "between layers: publish + wait for the 8 neighbours",
         "between layers: fetch chunk 0 of the next layer"]
print("== stack launch, %d images of %d x %d, %d layers, %d tiles" % (n, H, W, L, tiles))
print("cycles per wave over the launch: mean %.0f = %.0f per layer (ideal MFMA issue per layer and SIMD 4 x 6912 = 27648)" % (tot.mean(), tot.mean() / L))
for i, nme in enumerate(names):
    per = L if i != 4 else 1
    print("  %-55s %5.1f %%  per layer %8.1f   (waves 0-3: %8.1f, waves 4-7: %8.1f)" % (nme, 100 * s[..., i].sum() / tot.sum(), s[..., i].mean() / per,
                                                                                    s[:, :4, i].mean() / per, s[:, 4:, i].mean() / per))
