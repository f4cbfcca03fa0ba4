#!/usr/bin/env python3
"""Correctness + timing of every Winograd-kernel variant library under build/wgv/ (tools/wg_variants.sh)."""
import ctypes
import glob
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402  (weight packing only; the kernels come from the variant libraries)

c_i64, c_int, c_ptr = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p


def load(path):
    lib = ctypes.CDLL(path)
    fn = lib.deqsci_conv3x3_c64_winograd_f32
    fn.argtypes = [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr]
    fn.restype = c_int
    return fn


def run(fn, x, U, b, out, relu=1):
    n, _, H, W = x.shape
    rc = fn(x.data_ptr(), U.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(), n, H, W, relu,
            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def main():
    only = sys.argv[1:] or None
    g = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g)
    U = _hip.pack_winograd_weights(w)
    checks = []
    for shape in ((3, 40, 56), (2, 17, 23), (70, 64, 80), (1, 128, 128)):
        x = torch.randn(shape[0], 64, shape[1], shape[2], device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
        checks.append((x, want))
    shapes = ((64, 128, 128), (8, 128, 128), (64, 256, 256))
    xs = {s: torch.randn(s[0], 64, s[1], s[2], device="cuda", generator=g).contiguous(memory_format=torch.channels_last) for s in shapes}
    if os.environ.get("WG_ZERO_DATA"):        # DVFS experiment: all-zero activations and weights draw less power -> higher clock
        for t in xs.values():
            t.zero_()
        U = torch.zeros_like(U)
    outs = {s: torch.empty_like(xs[s]) for s in shapes}
    libs = []
    for path in sorted(glob.glob(os.path.join(ROOT, "build", "wgv", "lib_*.so"))):
        name = os.path.basename(path)[4:-3]
        if (only and name not in only) or (not only and name.startswith("stamp")):
            continue                                   # stamp builds scribble over the bias pointer: never next to a check
        libs.append((name, load(path)))
    recs = {}
    for name, fn in libs:
        err = 0.0
        for x, want in checks:
            out = torch.empty_like(x)
            run(fn, x, U, b, out)
            torch.cuda.synchronize()
            e = float((out.double() - want).norm() / want.norm())
            err = max(err, e if e == e else 9.0)
        recs[name] = {"variant": name, "max_rel_err_vs_fp64": err}
    # timing: the variants take turns (round-robin over ROUNDS rounds) so that clock / thermal drift of the box hits all of
    # them alike; per shape the median over rounds of the mean of 10 back-to-back launches
    ROUNDS = 9
    times = {name: {s: [] for s in shapes} for name, _ in libs}
    for name, fn in libs:
        for s in shapes:
            for _ in range(3):
                run(fn, xs[s], U, b, outs[s])
    torch.cuda.synchronize()
    for rnd in range(ROUNDS):
        for s in shapes:
            for name, fn in (libs if rnd % 2 == 0 else libs[::-1]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(fn, xs[s], U, b, outs[s])
                e1.record()
                torch.cuda.synchronize()
                times[name][s].append(e0.elapsed_time(e1) / 10 * 1e3)
    for name, _ in libs:
        for s in shapes:
            t = sorted(times[name][s])
            med = t[len(t) // 2]
            fl = 2 * 64 * 64 * 9 * s[0] * s[1] * s[2] / 2.25
            recs[name]["x".join(map(str, s))] = {"us": round(med, 1), "min_us": round(t[0], 1), "mfma_frac": round(fl / med / 1e6 / 157.3, 4)}
        print(json.dumps(recs[name]), flush=True)


if __name__ == "__main__":
    main()
