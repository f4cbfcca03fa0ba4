#!/usr/bin/env python3
"""Does the denoiser run faster on sub-batches whose activations stay inside the 256 MiB Infinity Cache?  FFDNet's 64-channel activation of a
bsz-8 call (64 images of 128 x 128) is 268 MB per layer (537 MB ping-pong); on 4 / 2 / 1 measurements at a time it is 134 / 67 / 34 MB.
Times one whole denoiser call (head + 13 layers + tail) on bsz 8 as 1, 2, 4 and 8 sub-batches, same data, HIP events."""
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402


def main():
    bsz = int(os.environ.get("SUBBATCH_BSZ", "8"))
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    eng = DEQSCIEngine(net, max_iter=180, use_graph=False)
    den = eng.den
    den.prepare(200, "cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    z1 = torch.rand(bsz, 8, 256, 256, device="cuda", generator=g)
    out = {"bsz": bsz}
    for parts in (1, 2, 4, 8):
        if bsz % parts:
            continue
        step = bsz // parts
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                for p in range(parts):
                    den.run(z1[p * step:(p + 1) * step], 20)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        out[f"{parts} x {step} measurements: us per denoiser call over the batch"] = round(statistics.median(ts[1:]), 1)
    # the same halves / quarters on CONCURRENT streams: does one stream's kernel fill the other's launch ramp, drain and cache write-back?
    for parts in (2, 4):
        if bsz % parts:
            continue
        step = bsz // parts
        streams = [torch.cuda.Stream() for _ in range(parts)]
        ts = []
        for rep in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for st in streams:
                st.wait_event(e0)
            for _ in range(5):
                for p, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        den.run(z1[p * step:(p + 1) * step], 20)
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        out[f"{parts} concurrent streams x {step} measurements: us per denoiser call over the batch"] = round(statistics.median(ts[1:]), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
