#!/usr/bin/env python3
"""FFDNet's first and last layer in their p32 forms (what the engine runs around a conv_w16 stack launch) alone: slices of 32 and 64 images of
128 x 128 (half resolution), THREE rotating activation buffers so that the 256 MiB Infinity Cache cannot serve a launch, HIP-event time per
launch, bytes moved / time against the 8 TB/s HBM peak.  DEQSCI_HIP_LIB selects a variant build (tools/lib_variants.sh)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
hw = _hip.HeadSplit16Weights(torch.randn(64, 5, 3, 3, device="cuda", generator=g) * 0.1)
tw = _hip.TailSplit16Weights(torch.randn(4, 64, 3, 3, device="cuda", generator=g) * 0.05)
sig = torch.rand(1, device="cuda", generator=g)
for n in (32, 64):
    x = torch.rand(n, 1, 256, 256, device="cuda", generator=g)
    rng = torch.ones(2, n, device="cuda") * 2.0
    acts = [_hip.P32.empty(n, 128, 128, "cuda") for _ in range(3)]
    for a in acts:
        a.t.normal_(generator=g)
        a.rng = rng[1]
    out = torch.empty(n, 1, 256, 256, device="cuda")
    res = {"images": n}
    for name, fn, nbytes in (("head_p32", lambda i: _hip.ffdnet_head_p32(x, hw, sig, out=acts[i % 3], in_rng=rng[0], out_rng=rng[1]), n * 128 * 128 * 256 + x.numel() * 4),
                             ("tail_p32", lambda i: _hip.ffdnet_tail_p32(acts[i % 3], tw, out=out), n * 128 * 128 * 256 + out.numel() * 4)):
        for i in range(6):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 60
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        res[name] = {"us": round(us, 1), "GB/s": round(nbytes / us / 1e3, 0), "of 8 TB/s": round(nbytes / us / 1e3 / 8000, 3)}
    print(json.dumps(res), flush=True)
