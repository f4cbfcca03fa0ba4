"""CPU: the plain-C oracle (oracle/deqsci_oracle.c) against the reference goldens and the torch oracle."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, rel_l2
from oracle import deqsci_oracle as orc

LIB = os.path.join(ROOT, "oracle", "libdeqsci_oracle.so")


@pytest.fixture(scope="module")
def clib():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.run(["make", "-C", ROOT, "oracle/libdeqsci_oracle.so"], check=True)
    return ctypes.CDLL(LIB)


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_c_ops_vs_reference_golden(clib):
    g = np.load(os.path.join(GOLDEN, "ops.npz"))
    for c in ("c0_", "c1_", "c2_"):
        Phi, z, y, x = (np.ascontiguousarray(g[c + k]) for k in ("Phi", "z", "y", "x"))
        B = Phi.shape[-1]
        npix = y.size
        i64 = ctypes.c_int64
        out_y = np.empty_like(y)
        clib.orc_sci_forward(P(z), P(Phi), P(out_y), i64(npix), i64(B))
        np.testing.assert_allclose(out_y, g[c + "Az"], rtol=1e-6, atol=2e-6)
        out_x = np.empty_like(z)
        clib.orc_sci_adjoint(P(y), P(Phi), P(out_x), i64(npix), i64(B))
        assert np.array_equal(out_x, g[c + "Aty"])
        ps = np.empty_like(y)
        clib.orc_phi_sum(P(Phi), P(ps), i64(npix), i64(B))
        assert np.array_equal(ps, g[c + "Phi_sum"])
        z1 = np.empty_like(z)
        clib.orc_gap_update(P(z), P(Phi), P(y), P(ps), P(z1), i64(npix), i64(B))
        np.testing.assert_allclose(z1, g[c + "z1"], rtol=1e-5, atol=1e-5)


def test_c_anderson_step_vs_torch_oracle(clib):
    """One Anderson alpha-solve + mix in C equals the same step done with bmm/linalg.solve."""
    g = torch.Generator().manual_seed(9)
    n, N, lam, beta = 4, 333, 1e-2, 0.8
    F_, X_ = torch.randn(n, N, generator=g), torch.randn(n, N, generator=g)
    G_ = F_ - X_
    H = torch.zeros(n + 1, n + 1)
    H[0, 1:] = 1
    H[1:, 0] = 1
    H[1:, 1:] = G_ @ G_.T + lam * torch.eye(n)
    rhs = torch.zeros(n + 1, 1)
    rhs[0] = 1
    want_alpha = torch.linalg.solve(H, rhs)[1:, 0]
    want_x = beta * (want_alpha[None] @ F_)[0] + (1 - beta) * (want_alpha[None] @ X_)[0]
    Fn, Xn = F_.numpy().copy(), X_.numpy().copy()
    alpha = np.empty(n, dtype=np.float32)
    assert clib.orc_anderson_alpha(P(Fn), P(Xn), n, ctypes.c_int64(N), ctypes.c_double(lam), P(alpha)) == 0
    np.testing.assert_allclose(alpha, want_alpha.numpy(), rtol=2e-4, atol=1e-6)
    assert abs(alpha.sum() - 1) < 1e-5
    out = np.empty(N, dtype=np.float32)
    clib.orc_anderson_mix(P(Fn), P(Xn), P(alpha), n, ctypes.c_int64(N), ctypes.c_float(beta), P(out))
    assert rel_l2(out, want_x.numpy()) < 1e-4
