#!/usr/bin/env python3
"""Time of the conv_w16 stack launch (13 layers, 64 images in slices of 32, relu(randn), fixed ranges) for an ablation / what-if build selected by
DEQSCI_HIP_LIB (tools/w16_variants.sh "share:-DW16_ABL=16" "mix4:-DW16_ABL=32" ...): no correctness check - the what-if builds are wrong by construction."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip  # noqa: E402

n, H, W, L = 64, 128, 128, 13
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.relu(torch.randn(n, 64, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
ws_raw = [torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.04 for _ in range(L)]
bs = [torch.randn(64, device="cuda", generator=g) * 0.05 for _ in range(L)]
rng = torch.ones(L + 1, n, device="cuda") * 4.0
stack = _hip.Wino16Stack(list(zip([_hip.Wino16Weights(w) for w in ws_raw], bs, [True] * L)), "cuda")
p0 = _hip.P32.from_nchw(x, rng=rng[0])
timer = _hip.KernelTimer(capacity=2000)
_hip.CONV64_EVENT_HOOK = lambda kind, m, hh, ww, layers=1: timer.pair()
for _ in range(int(os.environ.get("WHATIF_PASSES", "12"))):
    _hip.conv3x3_c64_wino16_stack(p0, stack, rng, check=False)
torch.cuda.synchronize()
ms = timer.durations_ms()[2:]
print(json.dumps({"lib": os.path.basename(os.environ.get("DEQSCI_HIP_LIB", "product")), "avg_launch_us": round(1e3 * sum(ms) / len(ms), 1), "timed_out": bool(stack.timed_out())}))
