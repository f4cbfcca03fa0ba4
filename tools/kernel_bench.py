#!/usr/bin/env python3
"""Per-kernel HBM roofline microbenchmark (SURVEY 8(d) "making the GB/s honest"): every HIP kernel of
the path on a working set far beyond the 256 MiB Infinity Cache, HIP-event timed over many launches,
algorithmic bytes / time against the 8 TB/s HBM3E peak.  Prints one JSON object per kernel.

    python tools/kernel_bench.py [--bsz 64] [--size 256x256x8] [--launches 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip  # noqa: E402

HWB, BHW = 0, 1
PEAK = 8000.0


def timeit(fns, launches, warm=6):
    """fns: one closure per rotating buffer set (consecutive launches touch different memory, so the
    256 MiB Infinity Cache cannot serve a launch from the previous one's lines)."""
    for i in range(warm):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(launches):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bsz", type=int, default=64)
    ap.add_argument("--size", default="256x256x8")
    ap.add_argument("--launches", type=int, default=50)
    ap.add_argument("--m", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--sets", type=int, default=3, help="rotating buffer sets (defeats the 256 MiB Infinity Cache)")
    args = ap.parse_args()
    H, W, B = (int(v) for v in args.size.split("x"))
    bsz, m = args.bsz, args.m
    P, N = H * W, H * W * B
    dev = "cuda"
    px = bsz * P

    def make_set(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        z = torch.randn(bsz, H, W, B, device=dev, generator=g)
        Phi = (torch.rand(bsz, H, W, B, device=dev, generator=g) < 0.5).float()
        y = torch.rand(bsz, H, W, device=dev, generator=g) * 4
        Ps = _hip.phi_sum(Phi, HWB)
        zp, Phip = _hip.transpose(z, BHW), _hip.transpose(Phi, BHW)
        out = torch.empty_like(z)
        outp = torch.empty_like(zp)
        ws = _hip.AndersonWorkspace(bsz, N, m, dev)
        ws.F.normal_(generator=g)
        ws.G.normal_(generator=g)
        ws.alpha[:, :m] = 1.0 / m
        noise = torch.randn_like(zp)
        return {
            "sci_forward_hwb": lambda: _hip.sci_forward(z, Phi, HWB, out=y),
            "sci_forward_bhw": lambda: _hip.sci_forward(zp, Phip, BHW, out=y),
            "sci_adjoint_hwb": lambda: _hip.sci_adjoint(y, Phi, HWB, out=out),
            "sci_adjoint_bhw": lambda: _hip.sci_adjoint(y, Phip, BHW, out=outp),
            "gap_update_hwb": lambda: _hip.gap_update(z, Phi, y, Ps, HWB, HWB, out=out),
            "gap_update_bhw": lambda: _hip.gap_update(zp, Phip, y, Ps, BHW, BHW, out=outp),
            "gap_update_hwb2bhw": lambda: _hip.gap_update(z, Phi, y, Ps, HWB, BHW, out=outp),
            "transpose_hwb2bhw": lambda: _hip.transpose(z, BHW, out=outp),
            "transpose_bhw2hwb": lambda: _hip.transpose(zp, HWB, out=out),
            "residual_out_hwb": lambda: _hip.residual_out(zp, noise, HWB, out=out),
            "mix_gap_bhw_n5": lambda: _hip.anderson_mix_gap(ws, 1.0, m, Phip, y, Ps, outp, zp, BHW),
            "mix_gap_hwb_n5": lambda: _hip.anderson_mix_gap(ws, 1.0, m, Phi, y, Ps, out, z, HWB),
            "mix_n5": lambda: _hip.anderson_mix(ws, outp.view(bsz, N), 1.0, m),
            "residual_store_nf5": lambda: _hip.residual_store(ws, zp.view(bsz, N), noise.view(bsz, N), outp.view(bsz, N), 2, m, None),
            "anderson_solve": lambda: _hip.anderson_solve(ws, 2, m, m, 1e-2, 1e-5),
            "torch_copy_reference": lambda: outp.copy_(zp),
        }
    sets = [make_set(s) for s in range(args.sets)]
    nbytes_of = {
        "sci_forward_hwb": px * (8 * B + 4), "sci_forward_bhw": px * (8 * B + 4),
        "sci_adjoint_hwb": px * (8 * B + 4), "sci_adjoint_bhw": px * (8 * B + 4),
        "gap_update_hwb": px * (12 * B + 8), "gap_update_bhw": px * (12 * B + 8), "gap_update_hwb2bhw": px * (12 * B + 8),
        "transpose_hwb2bhw": px * 8 * B, "transpose_bhw2hwb": px * 8 * B, "residual_out_hwb": px * 12 * B,
        "mix_gap_bhw_n5": px * (4 * B * (m + 3) + 8), "mix_gap_hwb_n5": px * (4 * B * (m + 3) + 8),
        "mix_n5": px * 4 * B * (m + 1), "residual_store_nf5": px * 4 * B * (3 + (m - 1) + 2),
        "anderson_solve": 0, "torch_copy_reference": px * 8 * B,
    }
    cases = {name: ([st[name] for st in sets], nbytes_of[name]) for name in nbytes_of}
    for name, (fn, nbytes) in cases.items():
        if args.only and args.only not in name:
            continue
        t = timeit(fn, args.launches)
        rec = {"kernel": name, "bsz": bsz, "size": args.size, "sets": args.sets, "avg_us": round(t * 1e6, 2)}
        if nbytes:
            rec.update({"algorithmic_MB": round(nbytes / 1e6, 2), "GBps": round(nbytes / t / 1e9, 1),
                        "frac_of_8TBps": round(nbytes / t / 1e9 / PEAK, 4)})
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
