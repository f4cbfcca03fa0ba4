#!/usr/bin/env python3
"""Rounding error of ONE f-call's denoiser along the real DEQ loop: the actual denoiser inputs z1 = GAP(X_k) of an FFDNet + Anderson
run (traffic measurement 0, 180 iterations) are captured at a few call indices, and FFDNet's noise prediction on each is computed
with every 64->64 implementation and compared with the same network in float64.  This is the quantity the conv64 policy of
DEQSCIEngine is chosen by: how noisy each kernel makes f on the data it really sees, early (blocky Phi^T y-like iterates) and late."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip, checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip  # noqa: E402

DATA = os.path.join(ROOT, "data", "test_gray")
CALLS = (0, 1, 2, 3, 5, 8, 12, 20, 30, 40, 60, 100, 150, 180)


def ffdnet64(den, x, sigma):
    """The folded network of the engine in float64 on torch (reference for the rounding of one call)."""
    n, _, H2, W2 = x.shape
    h = torch.cat((sigma.double().view(1, 1, 1, 1).expand(n, 1, H2 // 2, W2 // 2), F.pixel_unshuffle(x.double(), 2)), 1)
    for w, b, relu in den.fast:
        h = F.conv2d(h, w.double(), None if b is None else b.double(), padding=1)
        if relu:
            h = torch.relu(h)
    return F.pixel_shuffle(h, 2)


def main():
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    clip = [as_clip(c) for c in SCITestDataset(DATA)][-1]
    Phi = clip["mask"].to("cuda")[None].contiguous()
    y = clip["meas"][..., 0].to("cuda")[None].contiguous()
    eng = DEQSCIEngine(net, max_iter=180, use_graph=False, conv64="f22")
    den = eng.den
    captured = {}
    orig = den.run

    def spy(z1, call):
        if call in CALLS:
            captured[call] = z1.clone()
        return orig(z1, call)
    den.run = spy
    eng.reconstruct(y, Phi)
    den.run = orig
    rows = []
    for call, z1 in sorted(captured.items()):
        x = z1.view(8, 1, 256, 256)
        sig = den.sigma_table[call:call + 1]
        ref = ffdnet64(den, x, sig)
        row = {"call": call, "sigma": float(sig), "z1_rms": float(z1.pow(2).mean().sqrt()),
               "z1_roughness": float((z1[..., 1:] - z1[..., :-1]).pow(2).mean().sqrt() / z1.pow(2).mean().sqrt())}
        for pol in ("f22", "f44", "s16"):
            den.conv64 = den._policy = pol
            den.f22_calls = None
            out, _ = orig(z1, call)
            row[pol] = float((out.double().view_as(ref) - ref).norm() / ref.norm())
        e2 = DEQSCIEngine(net, max_iter=180, use_graph=False, winograd=False)       # MIOpen direct convolutions, edges still HIP
        e2.den.prepare(200, "cuda")
        out, _ = e2.den.run(z1, call)
        row["miopen"] = float((out.double().view_as(ref) - ref).norm() / ref.norm())
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "fcall_error_along_loop.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
