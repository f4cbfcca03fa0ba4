#!/usr/bin/env python3
"""Where is a Winograd variant wrong?  Error map by image / row / column / channel for small shapes."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip
from tools.wg_variant_bench import load, run
fn = load(os.path.join(ROOT, "build", "wgv", f"lib_{sys.argv[1]}.so"))
g = torch.Generator(device="cuda").manual_seed(3)
for mode in ("delta_center", "random"):
    w = torch.zeros(64, 64, 3, 3, device="cuda")
    if mode == "delta_center":
        for c in range(64):
            w[c, c, 1, 1] = 1.0                      # identity conv: out == in
    else:
        w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    U = _hip.pack_winograd_weights(w)
    for shape, use_b, relu in (((1, 16, 16), False, 0), ((1, 32, 32), True, 1), ((3, 40, 56), True, 0), ((2, 17, 23), True, 1), ((70, 64, 80), True, 1), ((64, 128, 128), True, 1), ((300, 16, 16), True, 0)):
        x = torch.randn(shape[0], 64, shape[1], shape[2], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        b = torch.randn(64, device="cuda", generator=g) if use_b else None
        want = F.conv2d(x.double(), w.double(), b.double() if use_b else None, padding=1)
        if relu:
            want = torch.relu(want)
        out = torch.empty_like(x)
        run(fn, x, U, b, out, relu=relu)
        torch.cuda.synchronize()
        err = (out.double() - want).abs()
        print(mode, shape, "bias", use_b, "relu", relu, "rel err", float((out.double() - want).norm() / want.norm()))
        if float(err.max()) > 1e-3:
            print("  by image:", [round(float(v), 2) for v in err.amax((1, 2, 3))][:80])
            e = err[int(err.amax((1, 2, 3)).argmax())]   # (64, H, W) of the worst image
            print("  by channel (first 16):", [round(float(v), 2) for v in e.amax((1, 2))[:16]])
            print("  by row:", [round(float(v), 2) for v in e.amax((0, 2))])
            print("  by col:", [round(float(v), 2) for v in e.amax((0, 1))])
            if mode == "delta_center":
                # which input pixel does each output pixel equal?  check channel 0 against shifted inputs
                o = out[0, 0]; xi = x[0, 0]
                for (dy, dx) in ((0, 0), (0, 1), (1, 0), (0, -1), (-1, 0), (1, 1)):
                    sh = torch.roll(xi, shifts=(-dy, -dx), dims=(0, 1))
                    print(f"   ch0 matches input shifted by ({dy},{dx}) on", int(((o - sh).abs() < 1e-5).sum()), "of", o.numel(), "pixels")
                for cs in (1, 2, 4, 8, 32):
                    print(f"   out ch0 vs in ch{cs}:", int(((out[0, 0] - x[0, cs]).abs() < 1e-5).sum()))
