#!/usr/bin/env python3
"""Split-fp16 Winograd F(2,3) x direct convolution (csrc/conv_w16.hip): error against a float64 convolution on random data and several
shapes (a two-layer chain, the stack launch against per-layer launches), and its time next to the split-fp16
direct kernel at the bench shapes.  `python tools/w16_check.py [check|time|both]`"""
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
    for shape in ((1, 8, 64), (1, 16, 64), (2, 17, 23), (3, 40, 56), (8, 128, 128), (5, 64, 80), (2, 250, 130), (40, 16, 16), (1, 1, 1), (1, 9, 65)):
        n, H, W = shape
        x = torch.relu(torch.randn(n, 64, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
        w1 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
        w2 = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
        b1 = torch.randn(64, device="cuda", generator=g)
        b2 = torch.randn(64, device="cuda", generator=g)
        W1, W2 = _hip.Wino16Weights(w1), _hip.Wino16Weights(w2)
        ref1 = torch.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
        ref2 = F.conv2d(ref1, w2.double(), b2.double(), padding=1)
        e = {}
        for name, xin in (("p32", _hip.P32.from_nchw(x)),):
            o1 = _hip.conv3x3_c64_wino16(xin, W1, b1, True)
            o2 = _hip.conv3x3_c64_wino16(o1, W2, b2, False)
            e[name] = rel(o1.to_nchw(), ref1)
            e[name + " chain"] = rel(o2.to_nchw(), ref2)
        d16 = _hip.conv3x3_c64_split16(_hip.to_split16(x), _hip.Split16Weights(w1), b1, True, out_f32=True)
        e["s16 direct"] = rel(d16, ref1)
        e["miopen fp32"] = rel(torch.relu(F.conv2d(x, w1, b1, padding=1)), ref1)
        ok = all(v < 1e-6 for k, v in e.items() if "chain" not in k) and all(v < 2e-6 for k, v in e.items() if "chain" in k)
        bad += not ok
        print(shape, {k: "%.2e" % v for k, v in e.items()}, "ok" if ok else "FAILED", flush=True)
    # the stack launch against a launch per layer: bit-identical
    for shape, nl in (((8, 128, 128), 13), ((3, 40, 56), 4), ((32, 128, 128), 13), ((2, 64, 192), 5), ((1, 24, 200), 3)):
        n, H, W = shape
        x = torch.relu(torch.randn(n, 64, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
        ws = [torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.04 for _ in range(nl)]
        bs = [torch.randn(64, device="cuda", generator=g) * 0.1 for _ in range(nl)]
        packs = [_hip.Wino16Weights(w) for w in ws]
        for act, xin in ((_hip.P32, _hip.P32.from_nchw(x)),):
            h = xin
            for i in range(nl):
                h = _hip.conv3x3_c64_wino16(h, packs[i], bs[i], i % 3 != 2)
            st = _hip.Wino16Stack([(packs[i], bs[i], i % 3 != 2) for i in range(nl)], "cuda")
            for rep in range(2):
                for b in st.state(n, H, W):
                    b.t.fill_(float("nan"))
                o = _hip.conv3x3_c64_wino16_stack(xin, st, check=False)
                same = bool(torch.equal(o.to_nchw(), h.to_nchw()))     # (the padding columns of a block are nobody's)
                to = st.timed_out()
                bad += (not same) or to
                print("stack", shape, nl, act.__name__, "rep", rep, "bit-identical" if same else "DIFFERS", "TIMED OUT" if to else "", flush=True)
    print("check:", "FAILED" if bad else "ok")
    return bad


def timeit():
    g = torch.Generator(device="cuda").manual_seed(5)
    ws = [torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.04 for _ in range(13)]
    bs = [torch.randn(64, device="cuda", generator=g) * 0.1 for _ in range(13)]
    w, b = ws[0], bs[0]
    for shape in ((64, 128, 128), (32, 128, 128), (8, 128, 128)):
        x = torch.relu(torch.randn(shape[0], 64, shape[1], shape[2], device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
        xs, xp = _hip.to_split16(x), _hip.P32.from_nchw(x)
        os_, op = _hip.Sp16.empty(*shape, "cuda"), _hip.P32.empty(*shape, "cuda")
        Wsp, Ww = _hip.Split16Weights(w), _hip.Wino16Weights(w)
        st16 = _hip.Split16Stack([(_hip.Split16Weights(ws[i]), bs[i], True) for i in range(13)], "cuda")
        stw = _hip.Wino16Stack([(_hip.Wino16Weights(ws[i]), bs[i], True) for i in range(13)], "cuda")
        fns = {"s16 direct": lambda: _hip.conv3x3_c64_split16(xs, Wsp, b, True, out=os_),
               "w16 p32": lambda: _hip.conv3x3_c64_wino16(xp, Ww, b, True, out=op)}
        if shape[0] <= 32:
            fns["s16 stack13 /13"] = lambda: _hip.conv3x3_c64_split16_stack(xs, st16, check=False)
            fns["w16 p32 stack13 /13"] = lambda: _hip.conv3x3_c64_wino16_stack(xp, stw, check=False)
        res = {k: [] for k in fns}
        for rnd in range(5):
            for k, fn in fns.items():
                for _ in range(2):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / 10 * 1e3 / (13 if "stack" in k else 1))
        med = {k: round(statistics.median(v), 1) for k, v in res.items()}
        print(json.dumps({"shape": shape, "us_per_layer": med, "timed_out": [st16.timed_out(), stw.timed_out()]}), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "both"
    rc = 0
    with torch.no_grad():
        if mode in ("check", "both"):
            rc = check()
        if mode in ("time", "both"):
            timeit()
    sys.exit(1 if rc else 0)
