"""Round 4's form of the reference's Anderson arithmetic, kept for A/B runs only (it was `anderson_arith="reference-bmm"` of the engine until
round 6): the Gram block G G^T of solvers/new_equilibrium_utils_yaping.py:178 as ONE fp32 torch.bmm (rocBLAS, ~200 us per iteration) handed
to K6's fp32 LU.  Not a product path: the engine's own "reference" arithmetic reproduces the summation order of the reference's CPU GEMM with
hand-written kernels (csrc/anderson.hip); this one has rocBLAS's order (a larger diagonal bias: DESIGN section 5, "Config 2", item 6)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deqsci_amd import _hip                      # noqa: E402
from deqsci_amd.engine import DEQSCIEngine       # noqa: E402


class ReferenceBmmEngine(DEQSCIEngine):
    """DEQSCIEngine whose alpha comes from a rocBLAS torch.bmm Gram block + fp32 LU (everything else as shipped)."""

    def __init__(self, denoiser, **kw):
        kw["anderson_arith"] = "float64"          # (the exact sums still feed the residual and the persistent float64 Gram)
        kw.setdefault("groups", 1)
        super().__init__(denoiser, **kw)

    def _store_solve_steps(self, ws, x_in, call, slot, n_filled, n_solve, x_next, eps, res_row):
        out, is_noise = yield from self.den.run_steps(ws.z1, call, calibrate=(call == 0))
        out = _hip.f32c(out)
        if is_noise:
            _hip.residual_store(ws, ws.z1, out, x_in, slot, n_filled, x_next)
        else:
            _hip.residual_store(ws, out, None, x_in, slot, n_filled, x_next)
        gram32 = None
        if n_solve > 0:
            G = ws.G[:, :n_solve]                                  # rows in the reference's slot order k % m
            gram32 = torch.bmm(G, G.transpose(1, 2))               # :178, ONE fp32 GEMM over N elements
        _hip.anderson_solve(ws, slot, n_filled, n_solve, self.lam, eps, res_row, gram32=gram32)
