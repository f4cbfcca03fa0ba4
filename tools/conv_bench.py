#!/usr/bin/env python3
"""64->64 3x3 conv layer (+bias+ReLU): MIOpen (channels_last igemm + HIP epilogue) vs the two Winograd MFMA kernels."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip  # noqa: E402


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


# `--shape N H W` runs one shape only (what the PMC script profiles)
SHAPES = ((64, 128, 128), (8, 128, 128), (64, 256, 256))
if "--shape" in sys.argv:
    i = sys.argv.index("--shape")
    SHAPES = (tuple(int(v) for v in sys.argv[i + 1:i + 4]),)
for (N, H, W) in SHAPES:
    x = torch.randn(N, 64, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    wcl = w.contiguous(memory_format=torch.channels_last)
    b = torch.randn(64, device="cuda")
    U = _hip.pack_winograd_weights(w)
    U44 = _hip.pack_winograd44_weights(w)
    out = torch.empty_like(x, memory_format=torch.channels_last)
    fl = 2 * 64 * 64 * 9 * H * W * N
    t_mi = timeit(lambda: _hip.bias_relu_(F.conv2d(x, wcl, None, padding=1), b, True))
    t_wg = timeit(lambda: _hip.conv3x3_c64_winograd(x, U, b, True, out=out))
    t_44 = timeit(lambda: _hip.conv3x3_c64_winograd44(x, U44, b, True, out=out))
    xb, ob = _hip.Blk32.from_nchw(x), _hip.Blk32.empty(N, H, W, "cuda")
    t_44b = timeit(lambda: _hip.conv3x3_c64_winograd44(xb, U44, b, True, out=ob, out_blk=True))
    xs, os_, Ws = _hip.to_split16(x), _hip.Sp16.empty(N, H, W, "cuda"), _hip.Split16Weights(w)
    t_s16 = timeit(lambda: _hip.conv3x3_c64_split16(xs, Ws, b, True, out=os_))
    print(json.dumps({"images": N, "HxW": f"{H}x{W}", "miopen_igemm_plus_epilogue_us": round(t_mi, 1), "winograd_f22_us": round(t_wg, 1),
                      "winograd_f44_us": round(t_44, 1), "winograd_f44_blk32_us": round(t_44b, 1), "split16_us": round(t_s16, 1),
                      "front_end_picks": _hip.conv64_kernel_for(N, H, W), "front_end_picks_fast32": _hip.conv64_kernel_for(N, H, W, policy="fast32"),
                      "speedup_vs_miopen": round(t_mi / min(t_wg, t_44b, t_s16), 2),
                      "split16_f16_mfma_util": round(3 * fl / t_s16 / 1e6 / 2500.0, 3), "split16_direct_equiv_TFLOPs": round(fl / t_s16 / 1e6, 1),
                      "f22_direct_equiv_TFLOPs": round(fl / t_wg / 1e6, 1), "f22_mfma_util": round((fl / 2.25) / t_wg / 1e6 / 157.3, 3),
                      "f44_direct_equiv_TFLOPs": round(fl / t_44 / 1e6, 1), "f44_mfma_util": round((fl / 4.0) / t_44 / 1e6 / 157.3, 3)}), flush=True)
