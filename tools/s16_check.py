#!/usr/bin/env python3
"""Split-fp16 direct convolution (csrc/conv_s16.hip): error against a float64 convolution on random data and several shapes (both
output forms, a two-layer chain), and its time next to the two Winograd kernels at the bench shape.  `python tools/s16_check.py [check|time|both]`"""
import json
import os
import statistics
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def check():
    g = torch.Generator(device="cuda").manual_seed(3)
    bad = 0
    for shape in ((1, 16, 32), (1, 32, 32), (2, 17, 23), (3, 40, 56), (8, 128, 128), (70, 64, 80), (2, 250, 130), (300, 16, 16), (1, 1, 1)):
        n, H, W = shape
        x = torch.randn(n, 64, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        w1 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
        w2 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
        b1 = torch.randn(64, device="cuda", generator=g)
        b2 = torch.randn(64, device="cuda", generator=g)
        xs = _hip.to_split16(x)
        back = rel(xs.to_nchw(), x.double())
        W1, W2 = _hip.Split16Weights(w1), _hip.Split16Weights(w2)
        ref1 = torch.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
        o_f32 = _hip.conv3x3_c64_split16(xs, W1, b1, True, out_f32=True)
        o_sp = _hip.conv3x3_c64_split16(xs, W1, b1, True)
        ref2 = F.conv2d(ref1, w2.double(), b2.double(), padding=1)
        o2 = _hip.conv3x3_c64_split16(o_sp, W2, b2, False, out_f32=True)
        d32 = torch.relu(F.conv2d(x, w1, b1, padding=1))
        e = {"split roundtrip": back, "f32 out": rel(o_f32, ref1), "sp16 out": rel(o_sp.to_nchw(), ref1), "2-layer chain": rel(o2, ref2), "miopen fp32": rel(d32, ref1)}
        ok = e["f32 out"] < 1e-6 and e["sp16 out"] < 1e-6 and e["2-layer chain"] < 2e-6
        bad += not ok
        print(shape, {k: "%.2e" % v for k, v in e.items()}, "ok" if ok else "FAILED", flush=True)
    print("check:", "FAILED" if bad else "ok")
    return bad


def timeit():
    g = torch.Generator(device="cuda").manual_seed(5)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g)
    for shape in ((64, 128, 128), (8, 128, 128), (32, 128, 128)):
        x = torch.randn(shape[0], 64, shape[1], shape[2], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        xs = _hip.to_split16(x)
        os_ = _hip.Sp16.empty(*shape, "cuda")
        of = torch.empty_like(x)
        xb = _hip.Blk32.from_nchw(x)
        ob = _hip.Blk32.empty(*shape, "cuda")
        Wsp, U4, U2 = _hip.Split16Weights(w), _hip.pack_winograd44_weights(w), _hip.pack_winograd_weights(w)
        fns = {"s16 sp16->sp16": lambda: _hip.conv3x3_c64_split16(xs, Wsp, b, True, out=os_),
               "s16 sp16->f32": lambda: _hip.conv3x3_c64_split16(xs, Wsp, b, True, out=of, out_f32=True),
               "f44 blk->blk": lambda: _hip.conv3x3_c64_winograd44(xb, U4, b, True, out=ob, out_blk=True),
               "f22": lambda: _hip.conv3x3_c64_winograd(x, U2, b, True, out=of),
               "f32->sp16 convert": lambda: _hip.to_split16(x, out=os_)}
        res = {k: [] for k in fns}
        for rnd in range(7):
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
        med = {k: round(statistics.median(v), 1) for k, v in res.items()}
        fl = 3 * 2.0 * 64 * 64 * 9 * shape[0] * shape[1] * shape[2]
        med["s16_f16_mfma_util_of_2.5PF"] = round(fl / (med["s16 sp16->sp16"] * 1e-6) / 2.5e15, 3)
        print(json.dumps({"shape": shape, **med}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    bad = check() if what in ("check", "both") else 0
    if what in ("time", "both"):
        timeit()
