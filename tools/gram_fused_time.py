#!/usr/bin/env python3
"""K4 alone, K4 + the reference Gram's first pass as two launches, and the fused launch (deqsci_residual_store_ref_f32), timed on a history like the
loop's (correlated heavy-tailed rows), bsz 8 and 4 at N = 2^19, m = 5; then the apply kernel + solve.  DEQSCI_HIP_LIB selects an ablation build."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip  # noqa: E402


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for bsz, N in ((8, 1 << 19), (4, 1 << 19), (1, 1 << 19)) + (((2, 1 << 22), (8, 1 << 22)) if os.environ.get("GRAM_BIG") else ()):
    m = 5
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randn(bsz, 1, N, device="cuda", generator=g) ** 3
    rows = ((base * (1 + 0.05 * torch.arange(m, device="cuda").view(1, m, 1)) + 0.3 * torch.randn(bsz, m, N, device="cuda", generator=g) ** 3) * 1e-3).contiguous()
    zero = torch.zeros(bsz, N, device="cuda")
    ws = _hip.AndersonWorkspace(bsz, N, m, "cuda")
    for k in range(m):
        _hip.residual_store(ws, rows[:, k].contiguous(), None, zero, k, k + 1, None, ref=True)
        _hip.anderson_solve(ws, k, k + 1, k + 1 if k else 0, 1e-2, 1e-5, ref=True)
    z = rows[:, 2].contiguous()
    noise = torch.zeros_like(z)
    out = {"bsz": bsz, "N": N}
    out["K4"] = timed(lambda: _hip.residual_store(ws, z, noise, zero, 2, m, None))
    out["K4 + first pass fused"] = timed(lambda: _hip.residual_store(ws, z, noise, zero, 2, m, None, ref=True))

    def two():
        _hip.residual_store(ws, z, noise, zero, 2, m, None)
        _hip.anderson_solve(ws, 2, m, m, 1e-2, 1e-5, ref=True)

    def fused():
        _hip.residual_store(ws, z, noise, zero, 2, m, None, ref=True)
        _hip.anderson_solve(ws, 2, m, m, 1e-2, 1e-5, ref=True)
    out["K4, round, apply, solve"] = timed(two)
    out["fused K4, apply, solve"] = timed(fused)
    out["K4, solve (float64)"] = timed(lambda: (_hip.residual_store(ws, z, noise, zero, 2, m, None), _hip.anderson_solve(ws, 2, m, m, 1e-2, 1e-5)))
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)
