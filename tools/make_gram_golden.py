#!/usr/bin/env python3
"""tests/golden/gram_bmm_cpu.npz: the fp32 torch.bmm(G, G^T) of solvers/new_equilibrium_utils_yaping.py:178 on this container's CPU (torch 2.10 / MKL,
AVX-512) for five heavy-tailed, correlated rows of N = 2^19 elements - what Anderson's residual history looks like in the FFDNet loop - generated
from a seed (numpy RandomState: frozen streams), together with the float64 Gram.  Pins oracle.gram_chain16 (the summation order restated) to the
reference's own line as this machine executes it.  `python tools/make_gram_golden.py`"""
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
