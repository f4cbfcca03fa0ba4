#!/usr/bin/env python3
"""The denoisers' edge layers alone, 64 images of 128 x 128 (FFDNet) / 64 of 256 x 256 (SimpleCNN): heads with fp32 / sp16 output, tails
with fp32 / sp16 input (vector-ALU form) and the matrix-core tail on sp16; HIP-event medians, interleaved."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip
g = torch.Generator(device="cuda").manual_seed(5)
n, H, W = 64, 128, 128
x = torch.randn(n, 1, 2 * H, 2 * W, device="cuda", generator=g)
wh = torch.randn(64, 5, 3, 3, device="cuda", generator=g) * 0.1
wt = torch.randn(4, 64, 3, 3, device="cuda", generator=g) * 0.05
sig = torch.rand(1, device="cuda", generator=g)
h = torch.relu(torch.randn(n, 64, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
hs = _hip.to_split16(h)
hw, tw, tw16 = _hip.pack_head_weights(wh), _hip.pack_tail_weights(wt), _hip.TailSplit16Weights(wt)
hw16 = _hip.HeadSplit16Weights(wh)
oh = torch.empty_like(h); ohs = _hip.Sp16.empty(n, H, W, "cuda"); ot = torch.empty(n, 1, 2 * H, 2 * W, device="cuda")
fns = {"head fp32 out": lambda: _hip.ffdnet_head(x, hw, sig, out=oh),
       "head sp16 out (f16 MFMA)": lambda: _hip.ffdnet_head_split16(x, hw16, sig, out=ohs),
       "tail fp32 in (VALU)": lambda: _hip.ffdnet_tail(h, tw, out=ot),
       "tail sp16 in (MFMA)": lambda: _hip.tail_split16(hs, tw16, out=ot)}
res = {k: [] for k in fns}
for rnd in range(7):
    for k, fn in fns.items():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
print(json.dumps({"shape": [n, H, W], **{k + " us": round(statistics.median(v), 1) for k, v in res.items()}}))
