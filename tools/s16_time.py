#!/usr/bin/env python3
"""Time of the split-fp16 conv kernel alone (sp16 -> sp16, 64 x 128 x 128) for the library in DEQSCI_HIP_LIB (variants of tools/s16_variants.sh);
S16_ZERO=1 runs it on all-zero operands (less switching power: what the clock gives back)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip
g = torch.Generator(device="cuda").manual_seed(5)
w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
b = torch.randn(64, device="cuda", generator=g)
shape = (64, 128, 128)
x = torch.randn(shape[0], 64, shape[1], shape[2], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
if os.environ.get("S16_ZERO"):
    x.zero_(); w.zero_()
xs = _hip.to_split16(x); os_ = _hip.Sp16.empty(*shape, "cuda"); Wsp = _hip.Split16Weights(w)
ts = []
for r in range(7):
    for _ in range(5): _hip.conv3x3_c64_split16(xs, Wsp, b, True, out=os_)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): _hip.conv3x3_c64_split16(xs, Wsp, b, True, out=os_)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 30 * 1e3)
print(json.dumps({"lib": os.environ.get("DEQSCI_HIP_LIB", "product"), "zero": bool(os.environ.get("S16_ZERO")), "us": round(statistics.median(ts), 1)}))
