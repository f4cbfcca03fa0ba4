#!/usr/bin/env python3
"""Random shapes through the split-fp16 kernels (64->64 conv both output forms, matrix-core head and tails) against float64 torch: tile edges,
tiny images, many images per workgroup run.  Prints the worst error per kernel."""
import os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip
random.seed(7)
g = torch.Generator(device="cuda").manual_seed(7)
worst = {"conv sp16": 0.0, "conv f32": 0.0, "head": 0.0, "tail4": 0.0, "tail1": 0.0}
rel = lambda a, b: float((a.double() - b).norm() / b.norm())
w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
b = torch.randn(64, device="cuda", generator=g) * 0.2
W16 = _hip.Split16Weights(w)
wh = torch.randn(64, 5, 3, 3, device="cuda", generator=g) * 0.1
wt4 = torch.randn(4, 64, 3, 3, device="cuda", generator=g) * 0.05
wt1 = torch.randn(1, 64, 3, 3, device="cuda", generator=g) * 0.05
H16, T4, T1 = _hip.HeadSplit16Weights(wh), _hip.TailSplit16Weights(wt4), _hip.TailSplit16Weights(wt1)
shapes = [(random.randint(1, 40), random.randint(1, 150), random.randint(1, 150)) for _ in range(30)] + [(1, 15, 31), (1, 16, 33), (1, 17, 32), (2, 1, 64), (2, 64, 1), (520, 16, 32)]
for (n, H, Wd) in shapes:
    x = torch.randn(n, 64, H, Wd, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    xs = _hip.to_split16(x)
    o = _hip.Sp16.empty(n, H, Wd, "cuda"); o.t.fill_(float("nan"))
    _hip.conv3x3_c64_split16(xs, W16, b, True, out=o)
    assert bool(torch.isfinite(o.t).all()), (n, H, Wd)
    worst["conv sp16"] = max(worst["conv sp16"], rel(o.to_nchw(), want))
    worst["conv f32"] = max(worst["conv f32"], rel(_hip.conv3x3_c64_split16(xs, W16, b, True, out_f32=True), want))
    worst["tail4"] = max(worst["tail4"], rel(_hip.tail_split16(xs, T4), F.pixel_shuffle(F.conv2d(x.double(), wt4.double(), padding=1), 2)))
    worst["tail1"] = max(worst["tail1"], rel(_hip.tail_split16(xs, T1), F.conv2d(x.double(), wt1.double(), padding=1)))
    img = torch.randn(n, 1, 2 * H, 2 * Wd, device="cuda", generator=g)
    sig = torch.rand(n, device="cuda", generator=g)
    inp = torch.cat((sig.view(n, 1, 1, 1).expand(n, 1, H, Wd), F.pixel_unshuffle(img, 2)), 1)
    hw = torch.relu(F.conv2d(inp.double(), wh.double(), padding=1))
    hs = _hip.Sp16.empty(n, H, Wd, "cuda"); hs.t.fill_(float("nan"))
    _hip.ffdnet_head_split16(img, H16, sig, out=hs)
    assert bool(torch.isfinite(hs.t).all()), (n, H, Wd)
    worst["head"] = max(worst["head"], rel(hs.to_nchw(), hw))
print({k: "%.2e" % v for k, v in worst.items()}, "over", len(shapes), "shapes")
assert all(v < 4e-7 for v in worst.values()), worst
print("fuzz: all ok")
