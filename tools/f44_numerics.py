#!/usr/bin/env python3
"""Would Winograd F(4x4,3x3) pass the parity gates?  Numerics only: the 64->64 layers of the engine are replaced by an
fp32 EMULATION of F(4x4,3x3) in torch (transforms and the 36 channel contractions in fp32, weights transformed in fp64
and rounded once, as a host-packed kernel would have them), and the end-to-end FFDNet gates of tests/test_gpu_parity.py
are evaluated against the reference's own outputs (tests/golden/e2e_ffdnet_*): Anderson @10 / @30 and Picard @180 on
traffic measurement 0.  Prints one JSON line per (variant, gate); no timing - the emulation is slow."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deqsci_amd  # noqa: E402,F401
from deqsci_amd import _hip, checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip, psnr  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data", "test_gray")

BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def pack44(w):
    g = torch.tensor(G, dtype=torch.float64, device=w.device)
    return (g @ w.detach().double() @ g.t()).float().permute(2, 3, 0, 1).reshape(36, 64, 64).contiguous()   # [xi][cout][cin]


def conv44(x, U, bias=None, relu=True, out=None):
    n, C, H, W = x.shape
    assert H % 4 == 0 and W % 4 == 0
    bt = torch.tensor(BT, dtype=torch.float32, device=x.device)
    at = torch.tensor(AT, dtype=torch.float32, device=x.device)
    xp = torch.nn.functional.pad(x.contiguous(), (1, 1, 1, 1))
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                        # (n,C,th,tw,6,6)
    th, tw = d.shape[2], d.shape[3]
    V = torch.einsum('ij,nctujk,lk->ilnctu', bt, d, bt).reshape(36, n, C, th * tw)
    M = torch.einsum('xoc,xnct->xnot', U, V)                      # 36 contractions over cin, fp32
    M = M.reshape(6, 6, n, 64, th, tw)
    Y = torch.einsum('ij,jknotu,lk->notiul', at, M, at).reshape(n, 64, th * 4, tw * 4)
    if bias is not None:
        Y = Y + bias.view(1, -1, 1, 1)
    if relu:
        Y = torch.relu(Y)
    return Y.contiguous(memory_format=torch.channels_last)


def main():
    dev = "cuda"
    clip = [as_clip(c) for c in SCITestDataset(DATA) if c["file"].startswith("traffic")][0]
    Phi = clip["mask"].to(dev)[None].contiguous()
    y = clip["meas"][..., 0].to(dev)[None].contiguous()
    gt = clip["gt"][..., :8].numpy()
    gates = [("anderson", 10), ("anderson", 30), ("picard", 180)]
    plain = (_hip.pack_winograd_weights, _hip.conv3x3_c64_winograd)
    for variant in ("F(2x2,3x3) HIP kernel", "F(4x4,3x3) fp32 emulation"):
        if variant.startswith("F(4"):
            _hip.pack_winograd_weights, _hip.conv3x3_c64_winograd = pack44, conv44
        else:
            _hip.pack_winograd_weights, _hip.conv3x3_c64_winograd = plain
        for iterator, iters in gates:
            solver, _ = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), iters)
            eng = DEQSCIEngine(solver.nonlinear_op, iterator=iterator, m=5, beta=1.0, lam=1e-2, max_iter=iters, tol=1e-5,
                               use_graph=False)
            rec = eng.reconstruct(y, Phi).cpu().numpy()
            ref = np.load(os.path.join(GOLDEN, f"e2e_ffdnet_{iterator}_{iters}" + ("_first" if iterator == "picard" else "") + "_rec.npz"))["traffic_m0"]
            print(json.dumps({"conv64": variant, "gate": f"ffdnet {iterator}@{iters} traffic m0", "rel_l2_vs_reference": rel_l2(rec, ref),
                              "psnr": psnr(np.clip(rec[0], 0, 1), gt), "psnr_reference": psnr(np.clip(ref[0], 0, 1), gt),
                              "limit": 1e-4}), flush=True)
    # one layer against float64, FFDNet's own weights and a realistic activation
    solver, _ = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 1)
    convs = [m for m in solver.nonlinear_op.modules() if isinstance(m, torch.nn.Conv2d) and tuple(m.weight.shape) == (64, 64, 3, 3)]
    g = torch.Generator().manual_seed(0)
    x = torch.relu(torch.randn(8, 64, 128, 128, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    w = convs[3].weight.detach().to(dev)
    exact = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    for name, r in (("F(2x2,3x3) HIP kernel", plain[1](x, plain[0](w), None, False)), ("F(4x4,3x3) fp32 emulation", conv44(x, pack44(w), None, False)),
                    ("direct fp32 (MIOpen)", torch.nn.functional.conv2d(x, w, padding=1))):
        print(json.dumps({"one_layer": name, "rel_l2_vs_fp64": float((r.double() - exact).norm() / exact.norm())}), flush=True)


if __name__ == "__main__":
    main()
