#!/usr/bin/env python3
"""(debug) conv_w16 with delta weights: which channels / pixels of the output differ from the input."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip
torch.manual_seed(0)
n, H, W = 1, 8, 64
x = torch.rand(n, 64, H, W, device="cuda").contiguous(memory_format=torch.channels_last) + 0.5
w = torch.zeros(64, 64, 3, 3, device="cuda")
for c in range(64):
    w[c, c, 1, 1] = 1.0
Ww = _hip.Wino16Weights(w)
for name, xin in (("p32", _hip.P32.from_nchw(x)),):
    o = _hip.conv3x3_c64_wino16(xin, Ww, None, False).to_nchw()
    err = (o - x).abs()
    print(name, "max err", float(err.max()), "per channel max:", [round(float(v), 3) for v in err.amax(dim=(0, 2, 3))[:16]])
    print("   per column max:", [round(float(v), 3) for v in err.amax(dim=(0, 1, 2))[:12]], " per row:", [round(float(v), 3) for v in err.amax(dim=(0, 1, 3))])
    if name == "p32":
        # which input channel does output channel c hold?
        xi = x[0, :, 3, 10]
        oi = o[0, :, 3, 10]
        print("   out channel c at (3,10) matches input channel:", [int((xi - oi[c]).abs().argmin()) if float((xi - oi[c]).abs().min()) < 1e-5 else -1 for c in range(64)])
        xr = _hip.P32.from_nchw(x).to_nchw()
        print("   roundtrip err", float((xr - x).abs().max()))

# ---- stack vs per-layer, layer by layer
g = torch.Generator(device="cuda").manual_seed(3)
for shape, nl in (((8, 128, 128), 1), ((8, 128, 128), 2), ((8, 128, 128), 3), ((32, 128, 128), 2)):
    n, H, W = shape
    x = torch.relu(torch.randn(n, 64, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
    ws = [torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.04 for _ in range(nl)]
    bs = [torch.randn(64, device="cuda", generator=g) * 0.1 for _ in range(nl)]
    packs = [_hip.Wino16Weights(w) for w in ws]
    xin = _hip.P32.from_nchw(x)
    hs, h = [], xin
    for i in range(nl):
        h = _hip.conv3x3_c64_wino16(h, packs[i], bs[i], True)
        hs.append(h)
    st = _hip.Wino16Stack([(packs[i], bs[i], True) for i in range(nl)], "cuda")
    bufs = st.state(n, H, W)
    for b in bufs:
        b.t.fill_(float("nan"))
    o = _hip.conv3x3_c64_wino16_stack(xin, st)
    torch.cuda.synchronize()
    for i in range(max(0, nl - 2), nl):
        got, ref = bufs[i % 2].t, hs[i].t
        bad = (got != ref) | torch.isnan(got)
        print("stack", shape, "layers", nl, "layer", i, "mismatching elements", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(got).sum()),
              "timed out", st.timed_out())
        if int(bad.sum()):
            idx = bad.nonzero()[:5].tolist()
            print("   first bad [n, b8, j, y, x, k]:", idx, "rows with bad:", sorted(set(bad.nonzero()[:, 3].tolist()))[:20], "cols:", sorted(set(bad.nonzero()[:, 4].tolist()))[:20])
