#!/usr/bin/env python3
"""Winograd F(4x4,3x3) kernel (csrc/winograd44.hip): error maps against a float64 convolution on several shapes, and its time
next to the F(2x2,3x3) kernel at the bench shape (interleaved, medians).  `python tools/w44_check.py [check|time|both]`."""
import json
import os
import statistics
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402

if os.environ.get("W44_LIB"):                      # a variant build (tools: build/w44v/lib_<name>.so) instead of the product library
    _hip._LIB_PATH = os.path.join(ROOT, os.environ["W44_LIB"])


def check():
    g = torch.Generator(device="cuda").manual_seed(3)
    bad = 0
    for mode in ("delta_center", "random"):
        w = torch.zeros(64, 64, 3, 3, device="cuda")
        if mode == "delta_center":
            for c in range(64):
                w[c, c, 1, 1] = 1.0
        else:
            w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
        U = _hip.pack_winograd44_weights(w)
        shapes = (((1, 16, 32), False, 0), ((1, 32, 32), True, 1), ((3, 40, 56), True, 0), ((2, 17, 23), True, 1),
                                   ((70, 64, 80), True, 1), ((64, 128, 128), True, 1), ((300, 16, 16), True, 0), ((2, 250, 130), True, 1))
        if os.environ.get("W44_SHAPES"):
            shapes = shapes[:int(os.environ["W44_SHAPES"])]
        for shape, use_b, relu in shapes:
            x = torch.randn(shape[0], 64, shape[1], shape[2], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
            b = torch.randn(64, device="cuda", generator=g) if use_b else None
            want = F.conv2d(x.double(), w.double(), b.double() if use_b else None, padding=1)
            if relu:
                want = torch.relu(want)
            out = torch.full_like(x, float("nan"))
            _hip.conv3x3_c64_winograd44(x, U, b, bool(relu), out=out)
            torch.cuda.synchronize()
            err = (out.double() - want).abs()
            rel = float((out.double() - want).norm() / want.norm())
            print(mode, shape, "bias", use_b, "relu", relu, "rel err", rel, flush=True)
            if not rel < 1e-5:
                bad += 1
                err = torch.nan_to_num(err, nan=99.0)
                print("  by image:", [round(float(v), 2) for v in err.amax((1, 2, 3))][:40])
                e = err[int(err.amax((1, 2, 3)).argmax())]
                print("  by channel:", [round(float(v), 2) for v in e.amax((1, 2))])
                print("  by row:", [round(float(v), 2) for v in e.amax((0, 2))][:64])
                print("  by col:", [round(float(v), 2) for v in e.amax((0, 1))][:64])
                if mode == "delta_center":
                    o, xi = out[0, 0], x[0, 0]
                    for (dy, dx) in ((0, 0), (0, 1), (1, 0), (0, -1), (-1, 0)):
                        sh = torch.roll(xi, shifts=(-dy, -dx), dims=(0, 1))
                        print(f"   ch0 == input shifted by ({dy},{dx}) on", int(((o - sh).abs() < 1e-5).sum()), "of", o.numel())
    print("check:", "FAILED" if bad else "ok", bad)
    return bad


def timeit():
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g)
    shapes = ((64, 128, 128), (8, 128, 128), (32, 128, 128), (8, 256, 256), (1, 128, 128), (2, 128, 128), (4, 128, 128), (6, 128, 128), (16, 128, 128), (1, 256, 256), (2, 256, 256))
    for shape in shapes[:int(os.environ.get("W44_SHAPES", "11"))]:
        x = torch.randn(shape[0], 64, shape[1], shape[2], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        if os.environ.get("W44_ZERO_X"):                 # power experiment: all-zero activations
            x.zero_()
        out = torch.empty_like(x)
        fns = {"f22": (lambda: _hip.conv3x3_c64_winograd(x, U2, b, True, out=out)), "f44": (lambda: _hip.conv3x3_c64_winograd44(x, U4, b, True, out=out))}
        if os.environ.get("W44_LIB"):
            fns.pop("f22")
        U2, U4 = _hip.pack_winograd_weights(w), _hip.pack_winograd44_weights(w)
        res = {k: [] for k in fns}
        for rnd in range(9):
            for k, fn in fns.items():
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
        print(json.dumps({"lib": os.environ.get("W44_LIB", "product"), "shape": shape, **{k + "_us": round(statistics.median(v), 1) for k, v in res.items()}}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    bad = check() if what in ("check", "both") else 0
    if what in ("time", "both") and not bad:
        timeit()
