#!/usr/bin/env python3
"""Rounding error of every implementation of FFDNet's 64->64 layers ON THE NETWORK'S OWN DATA (folded weights of net_gray, the
activations a real iterate produces), layer by layer, against a float64 convolution of the same fp32 operands.  Random data say
little here: the F(4x4,3x3) form is 6-8x noisier than a direct convolution on randn inputs and on par with it on these.
Prints one JSON line per (input, layer) and a by-position breakdown (row / column of the output inside its 4x4 tile, image border)."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deqsci_amd  # noqa: E402
from deqsci_amd import _hip, checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine, SIGMA0  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip  # noqa: E402

DATA = os.path.join(ROOT, "data", "test_gray")


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def main():
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=30, use_graph=False)
    den = eng.den
    clip = [as_clip(c) for c in SCITestDataset(DATA)][-1]                # traffic
    Phi = clip["mask"].to("cuda")[None].contiguous()
    y = clip["meas"][..., 0].to("cuda")[None].contiguous()
    x0 = deqsci_amd.initial_point(y, Phi, None, None)
    rec = eng.reconstruct(y, Phi)                                        # a 30-iteration iterate
    inputs = {"x0": x0, "iterate30": rec}
    for k in (3, 5, 10, 20, 60, 120):                                    # how the rounding of each form moves as the iterate settles
        e = DEQSCIEngine(net, max_iter=k, use_graph=False, conv64="f22")
        inputs[f"iterate{k}"] = e.reconstruct(y, Phi)
    out = []
    for name, z in inputs.items():
        x = z.permute(0, 3, 1, 2).reshape(8, 1, 256, 256).contiguous()
        k_it = 0 if name == "x0" else int(name[7:])
        sig = torch.full((1,), SIGMA0 * 0.971 ** k_it, device="cuda")   # (the sigma the loop would hand FFDNet at that call)
        h = _hip.ffdnet_head(x, den.head_w, sig)                         # (8,64,128,128) channels_last
        for li in range(1, len(den.fast) - 1):
            w, b, relu = den.fast[li]
            ref = F.conv2d(h.double(), w.double(), b.double(), padding=1)
            if relu:
                ref = torch.relu(ref)
            row = {"input": name, "layer": li, "act_max": float(h.abs().max()), "act_rms": float(h.pow(2).mean().sqrt())}
            got = {}
            got["f22"] = _hip.conv3x3_c64_winograd(h, den.wino[li].f22, b, relu)
            got["f44"] = _hip.conv3x3_c64_winograd44(h, den.wino[li].f44, b, relu)
            g = F.conv2d(h, w, b, padding=1)
            got["miopen"] = torch.relu(g) if relu else g
            got["s16"] = _hip.conv3x3_c64_split16(_hip.to_split16(h), _hip.Split16Weights(w), b, relu, out_f32=True)
            for k, v in got.items():
                row[k] = rel(v, ref)
            # where the F(4x4) error sits
            e = (got["f44"].double() - ref)
            e2 = e.pow(2).mean(dim=(0, 1))                               # (H, W)
            tot = float(e2.mean())
            row["f44_by_row_mod4"] = [round(float(e2[r::4].mean()) / tot, 3) for r in range(4)]
            row["f44_by_col_mod4"] = [round(float(e2[:, c::4].mean()) / tot, 3) for c in range(4)]
            row["f44_border_share"] = round(float((e2[0].sum() + e2[-1].sum() + e2[1:-1, 0].sum() + e2[1:-1, -1].sum()) / e2.sum()), 4)
            row["f44_mean_err_over_rms"] = float(e.mean() / e.pow(2).mean().sqrt())
            row["f22_mean_err_over_rms"] = float((got["f22"].double() - ref).mean() / (got["f22"].double() - ref).pow(2).mean().sqrt())
            out.append(row)
            print(json.dumps(row))
            h = got["f22"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "conv_error_real.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
