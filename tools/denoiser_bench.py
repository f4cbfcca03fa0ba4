#!/usr/bin/env python3
"""Times the denoiser step (FFDNet / SimpleCNN on PyTorch-ROCm/MIOpen) in the variants the engine could
use: default, MIOpen find mode (cudnn.benchmark), channels_last, fused conv+bias+relu op."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import checkpoint  # noqa: E402
from deqsci_amd.cli import build_denoiser  # noqa: E402
from deqsci_amd.engine import _Denoiser  # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    dev = "cuda"
    net = build_denoiser("ffdnet").eval()
    net.load_state_dict(checkpoint.read_state_dict(checkpoint.shipped("ffdnet_gray"))[0])
    net = net.to(dev)
    den = _Denoiser(net, fused_epilogue=False)
    den.prepare(8, dev)
    variants = {"nchw+fused_epilogue": _Denoiser(net), "channels_last+fused_epilogue": _Denoiser(net, channels_last=True),
                "channels_last unfused": _Denoiser(net, channels_last=True, fused_epilogue=False)}
    for v in variants.values():
        v.prepare(8, dev)
    for nimg in (8, 64):
        xx = torch.rand(nimg // 8, 8, 256, 256, device=dev)
        with torch.no_grad():
            ref = den.run(xx, 3)[0]
            for name, v in variants.items():
                got = v.run(xx, 3)[0]
                err = float((got - ref).norm() / ref.norm())
                tt = timeit(lambda: v.run(xx, 3))
                print(json.dumps({"variant": "ffdnet folded " + name, "images": nimg, "ms": round(tt * 1e3, 3), "rel_err_vs_unfused": err}), flush=True)
    for nimg in (8, 64):
        x = torch.rand(nimg // 8, 8, 256, 256, device=dev)
        gflop = 127.0 * nimg / 8
        with torch.no_grad():
            for bench in (False, True):
                torch.backends.cudnn.benchmark = bench
                t = timeit(lambda: den.run(x, 0))
                print(json.dumps({"variant": f"ffdnet folded nchw benchmark={bench}", "images": nimg, "ms": round(t * 1e3, 3),
                                  "TFLOPs_direct": round(gflop / t / 1e3, 1)}), flush=True)
            torch.backends.cudnn.benchmark = False
            t = timeit(lambda: net(x.view(nimg, 1, 256, 256), den.sigma_table[0:1].expand(nimg)))
            print(json.dumps({"variant": "ffdnet module (unfolded BN)", "images": nimg, "ms": round(t * 1e3, 3)}), flush=True)
            # channels_last
            layers = [(w.contiguous(memory_format=torch.channels_last), b, r) for w, b, r in den.fast]

            def run_cl():
                sig = den.sigma_table[0:1].expand(nimg)
                h = torch.cat((sig.view(-1, 1, 1, 1).expand(nimg, 1, 128, 128), F.pixel_unshuffle(x.view(nimg, 1, 256, 256), 2)), 1)
                h = h.contiguous(memory_format=torch.channels_last)
                for w, b, r in layers:
                    h = F.conv2d(h, w, b, padding=1)
                    if r:
                        h = F.relu_(h)
                return F.pixel_shuffle(h, 2)
            try:
                t = timeit(run_cl)
                print(json.dumps({"variant": "ffdnet folded channels_last", "images": nimg, "ms": round(t * 1e3, 3)}), flush=True)
            except Exception as e:
                print("channels_last failed", e)
            # fused conv+bias+relu
            if hasattr(torch.ops.aten, "miopen_convolution_relu"):
                def run_fused():
                    sig = den.sigma_table[0:1].expand(nimg)
                    h = torch.cat((sig.view(-1, 1, 1, 1).expand(nimg, 1, 128, 128), F.pixel_unshuffle(x.view(nimg, 1, 256, 256), 2)), 1)
                    for w, b, r in den.fast:
                        if r:
                            bb = b if b is not None else torch.zeros(w.shape[0], device=dev)
                            h = torch.ops.aten.miopen_convolution_relu(h, w, bb, [1, 1], [1, 1], [1, 1], 1)
                        else:
                            h = F.conv2d(h, w, b, padding=1)
                    return F.pixel_shuffle(h, 2)
                try:
                    ref = den.run(x, 0)[0]
                    got = run_fused().reshape(ref.shape)
                    err = float((got - ref).norm() / ref.norm())
                    t = timeit(run_fused)
                    print(json.dumps({"variant": "ffdnet folded miopen_convolution_relu", "images": nimg, "ms": round(t * 1e3, 3), "rel_err": err}), flush=True)
                except Exception as e:
                    print("fused failed", repr(e)[:300])
    # single 64->64 layer alone
    with torch.no_grad():
        h = torch.rand(64, 64, 128, 128, device=dev)
        w = torch.rand(64, 64, 3, 3, device=dev)
        b = torch.rand(64, device=dev)
        fl = 2 * 64 * 64 * 9 * 128 * 128 * 64 / 1e12
        t = timeit(lambda: F.conv2d(h, w, None, padding=1), n=20)
        print(json.dumps({"variant": "conv 64->64 3x3 only, 64x128x128", "ms": round(t * 1e3, 3), "TFLOPs_direct": round(fl / t, 1)}))
        t = timeit(lambda: F.conv2d(h, w, b, padding=1), n=20)
        print(json.dumps({"variant": "conv+bias", "ms": round(t * 1e3, 3)}))
        t = timeit(lambda: F.relu_(F.conv2d(h, w, b, padding=1)), n=20)
        print(json.dumps({"variant": "conv+bias+relu_", "ms": round(t * 1e3, 3)}))


if __name__ == "__main__":
    main()
