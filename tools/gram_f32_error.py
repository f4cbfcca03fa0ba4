#!/usr/bin/env python3
"""What "the reference's fp32 Gram" is, numerically (CPU; no GPU needed): the relative error against float64 of the entries of G G^T for five
correlated rows of N = 2^19 elements (what Anderson's residual history looks like) (a) as ONE fp32 torch.bmm forms them - the reference's
line, solvers/new_equilibrium_utils_yaping.py:177-178 - and (b) as a flat chain of fp32 additions of block partials along K, for several
block sizes - what deqsci_anderson_solve_ref_f32 does with K4's 64-element partials.  The engine's float64 Gram is exact on this scale
(1e-8); its block-wise fp32 tree of round 2 was 1e-7."""
import numpy as np
import torch

torch.manual_seed(0)
N = 2 ** 19
base = torch.randn(N)
G = torch.stack([base * (1 + 0.05 * k) + 0.3 * torch.randn(N) for k in range(5)])[None].float()
ref = (G.double() @ G.double().transpose(1, 2))[0]
got = torch.bmm(G, G.transpose(1, 2))[0].double()
rel = ((got - ref) / ref).abs()
print("one fp32 torch.bmm (torch %s, CPU): mean %.2e max %.2e of an entry" % (torch.__version__, rel.mean(), rel.max()))
g = G[0].numpy()
for bs in (64, 128, 256, 2048):
    out = np.zeros((5, 5))
    for i in range(5):
        for j in range(5):
            p = (g[i].astype(np.float64) * g[j].astype(np.float64)).reshape(-1, bs).sum(1).astype(np.float32)
            acc = np.float32(0)
            for v in p:
                acc = np.float32(acc + v)
            out[i, j] = acc
    r = np.abs((out - ref.numpy()) / ref.numpy())
    print("flat fp32 chain of %5d-element partials (%5d additions): mean %.2e max %.2e" % (bs, N // bs, r.mean(), r.max()))
