#!/usr/bin/env python3
"""The stack launch alone, as bench.py's step uses it: FFDNet's 13 layers over a batch of 64 images of 128 x 128 in slices of 32 (two
launches per pass, each reading an input the other's 13 layers have since pushed out of the caches), HIP-event time per launch.
STACK_KERNEL = w16 (default: deqsci::w16::conv_w16_kernel<1>, split-fp16 under Winograd F(2,3) x direct) | s16 (deqsci::s16::conv_s16_kernel<0, 0, 1>,
split-fp16 direct).  Under `rocprofv3 --pmc ...` (tools/pmc_winograd.sh, tools/gpu_round5.sh) this is what the counters of the launch are
measured on."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402

n, H, W, L = int(os.environ.get("STACK_IMAGES", "64")), 128, 128, int(os.environ.get("STACK_LAYERS", "13"))
passes = int(os.environ.get("STACK_PASSES", "12"))
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.relu(torch.randn(n, 64, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
ws_raw = [torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.04 for _ in range(L)]
Ws = [_hip.Split16Weights(w) for w in ws_raw]
bs = [torch.randn(64, device="cuda", generator=g) * 0.05 for _ in range(L)]
rng = torch.zeros(L + 1, n, device="cuda")
_hip.absmax(x, rng[0])
h = h0 = _hip.to_split16(x, rng=rng[0])
for i in range(L):                                              # the ranges, as the engine's first f-call measures them
    _hip.conv3x3_c64_split16(h, Ws[i], bs[i], True, track=rng[i + 1])
    h = _hip.conv3x3_c64_split16(h, Ws[i], bs[i], True, out_rng=rng[i + 1])
kernel = os.environ.get("STACK_KERNEL", "w16")
per = _hip.split16_stack_per_launch(n, H, W)
timer = _hip.KernelTimer(capacity=4 * passes * (-(-n // per)))
_hip.CONV64_EVENT_HOOK = lambda kind, m, hh, ww, layers=1: timer.pair()
if kernel == "s16":
    stack = _hip.Split16Stack(list(zip(Ws, bs, [True] * L)), "cuda")
    for _ in range(passes):
        out = _hip.conv3x3_c64_split16_stack(h0, stack, rng, check=False)
    same = torch.equal(out.t, h.t)
else:
    Ww = [_hip.Wino16Weights(w) for w in ws_raw]
    stack = _hip.Wino16Stack(list(zip(Ww, bs, [True] * L)), "cuda")
    p0 = _hip.P32.from_nchw(x, rng=rng[0])
    for _ in range(passes):
        out = _hip.conv3x3_c64_wino16_stack(p0, stack, rng, check=False)
    d = (out.to_nchw().double() - h.to_nchw().double()).norm() / h.to_nchw().double().norm()
    same = float(d) < 3e-6                                      # (the direct kernel's chain of 13 layers: another arithmetic of the same accuracy)
torch.cuda.synchronize()
_hip.CONV64_EVENT_HOOK = None
ms = timer.durations_ms()[2:]
assert not stack.timed_out() and same
print(json.dumps({"kernel": kernel, "images": n, "images_per_launch": per, "layers": L, "launches_timed": len(ms), "avg_launch_us": round(1e3 * sum(ms) / len(ms), 1),
                  "us_per_layer_and_launch": round(1e3 * sum(ms) / len(ms) / L, 2)}))
timer.close()
