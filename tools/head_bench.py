#!/usr/bin/env python3
"""FFDNet head kernel (sigma map + pixel_unshuffle + conv 5->64 + ReLU) at the bench shape: matrix-core vs vector-ALU variant
(DEQSCI_HEAD_VALU=1 forces the latter in the diagnostic build of the library: `make diag`, then run the script twice with
DEQSCI_HIP_LIB=build/diag/libdeqsci_hip_diag.so)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(64, 1, 256, 256, device="cuda", generator=g)
w = torch.randn(64, 5, 3, 3, device="cuda", generator=g) * 0.1
sig = torch.rand(64, device="cuda", generator=g)
wp = _hip.pack_head_weights(w)
out = torch.empty((64, 64, 128, 128), device="cuda").contiguous(memory_format=torch.channels_last)
ts = []
for r in range(7):
    for _ in range(3):
        _hip.ffdnet_head(x, wp, sig, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _hip.ffdnet_head(x, wp, sig, out=out)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(json.dumps({"head_variant": "vector ALU" if os.environ.get("DEQSCI_HEAD_VALU") == "1" else "MFMA", "us": round(statistics.median(ts), 1)}))
