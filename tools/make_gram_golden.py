#!/usr/bin/env python3
"""tests/golden/gram_bmm_cpu.npz: the fp32 torch.bmm(G, G^T) of solvers/new_equilibrium_utils_yaping.py:178 on this container's CPU (torch 2.10 / MKL,
AVX-512) for five heavy-tailed, correlated rows of N = 2^19 elements - what Anderson's residual history looks like in the FFDNet loop - generated
from a seed (numpy RandomState: frozen streams), together with the float64 Gram.  Pins oracle.gram_chain16 (the summation order restated) to the
reference's own line as this machine executes it.  Round 6: the same for other shapes - N = 2^17 ... 2^22 (512 x 512 x 16 is 2^22), n = 2 ... 8
(m up to DEQSCI_MAX_M) - in tests/golden/gram_bmm_cpu_shapes.npz: MKL's kernel choice could depend on the shape; on this CPU it does not (the
16-chain order holds within two ulps on every entry of every shape, whatever the thread count).  `python tools/make_gram_golden.py`"""
import os

import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import deqsci_oracle as orc  # noqa: E402  (the generator of the rows lives with the checker)


if __name__ == "__main__":
    G = orc.heavy_tailed_rows()
    t = torch.from_numpy(G)[None]
    bmm = torch.bmm(t, t.transpose(1, 2))[0].numpy()
    exact = G.astype(np.float64) @ G.astype(np.float64).T
    out = os.path.join(ROOT, "tests", "golden", "gram_bmm_cpu.npz")
    np.savez(out, seed=7, n=5, N=2 ** 19, bmm=bmm, exact=exact, torch=torch.__version__, cpu_capability=torch.backends.cpu.get_cpu_capability())
    print("diag error x 1e6:", np.round(np.diag((bmm - exact) / exact) * 1e6, 2), "->", out)
    shapes = [(5, 17), (8, 19), (2, 19), (3, 21), (5, 22)]
    rec = {"shapes": np.array(shapes), "torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability()}
    for n, lg in shapes:
        G = orc.heavy_tailed_rows(seed=11 + n + lg, n=n, N=2 ** lg)
        t = torch.from_numpy(G)[None]
        rec[f"bmm_{n}_{lg}"] = torch.bmm(t, t.transpose(1, 2))[0].numpy()
        rec[f"exact_{n}_{lg}"] = G.astype(np.float64) @ G.astype(np.float64).T
    out = os.path.join(ROOT, "tests", "golden", "gram_bmm_cpu_shapes.npz")
    np.savez(out, **rec)
    print("->", out)
