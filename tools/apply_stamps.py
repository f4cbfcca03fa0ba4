#!/usr/bin/env python3
"""(diag build: `make diag`, DEQSCI_HIP_LIB=build/diag/libdeqsci_hip_diag.so) where gram_chain_apply_kernel's time goes, chain (entry 0, chain 0) of
sample 0, along a real reconstruction (FFDNet + Anderson, traffic m2, one measurement): cycles in the record window load, the prefetch of the
predicted walks, the 64-block acceptance rounds, the walks; counts of rounds, walks, and where the walked terms came from."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip, checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip  # noqa: E402

clip = [as_clip(c) for c in SCITestDataset(os.path.join(ROOT, "data", "test_gray")) if "traffic" in as_clip(c)["file"]][0]
Phi, y = clip["mask"][None].to("cuda"), clip["meas"][None, ..., 2].contiguous().to("cuda")
net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 60)[0].nonlinear_op
eng = DEQSCIEngine(net, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=60, tol=1e-5, anderson_arith="reference", use_graph=False)
orig = _hip.anderson_solve
rows = []


def spy(ws, slot, n_filled, n, lam, eps, res_row=0, gram32=None, ref=False):
    r = orig(ws, slot, n_filled, n, lam, eps, res_row, gram32=gram32, ref=ref)
    if ref:
        d = ws.chains_walked()[0, 6, :13].cpu().tolist()
        w = ws.chains_walked()[0, :n_filled].float()
        wl = ws.chains_walked()[0, 7, :16].cpu().tolist() + ws.chains_walked()[0, 8 - 1, :0].cpu().tolist()
        codes = ws.ref_state().view(ws.bsz, -1)[0].view(torch.int32)
        rows.append(dict(zip(("window", "prefetch", "rounds", "walks", "n_rounds", "n_walked", "from_prefetch", "prefetched_without_slot", "missed", "total", "prefetch:marking", "prefetch:slot_loads", "n_prefetched"), d),
                         blocks=[(v & 0xfff, 'P' if v & 0x1000 else '-', 'S' if v & 0x2000 else '-') for v in wl[:d[5]]], n_filled=n_filled, walked_mean=round(float(w.mean()), 1), walked_max=int(w.max())))
    return r


_hip.anderson_solve = spy
with torch.no_grad():
    eng.reconstruct(y, Phi)
for i, r in enumerate(rows):
    if i < 12 or i % 8 == 0:
        print(i, json.dumps(r))
