#!/usr/bin/env python3
"""WHICH fp32 rounding of the Gram matrix moves Anderson's alpha on the loop's own data (config 2: FFDNet + Anderson, traffic m2)?  The engine runs
with round 4's form of the reference's arithmetic (tools/reference_bmm.py, until round 6 the engine's anderson_arith="reference-bmm": G G^T as one fp32 torch.bmm, solvers/new_equilibrium_utils_yaping.py:178);
every few iterations the residual history G (n x N, N = 2^19) is taken aside and its Gram matrix formed in float64 (exact on this scale), by that
torch.bmm, and as flat fp32 chains along K over partials of 1 .. 4096 elements (numpy cumsum: sequential) - the candidates for a hand-written kernel -
and the bordered system of :179-180 solved in fp32 for each.  Printed per sampled iteration: the relative error of the Gram entries and the
deviation of alpha from the exact-Gram alpha."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip, checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip  # noqa: E402

LAM = 1e-2
SIZES = (1, 16, 64, 256, 4096)


def alpha_of(gram):
    """fp32 LU of the bordered system, as torch.solve on the reference's H (:175-180)."""
    n = gram.shape[0]
    H = np.zeros((n + 1, n + 1), np.float32)
    H[0, 1:] = H[1:, 0] = 1
    H[1:, 1:] = gram.astype(np.float32) + np.float32(LAM) * np.eye(n, dtype=np.float32)
    y = np.zeros(n + 1, np.float32)
    y[0] = 1
    return torch.linalg.solve(torch.from_numpy(H), torch.from_numpy(y)).numpy()[1:].astype(np.float64)


def alpha64(gram):
    n = gram.shape[0]
    H = np.zeros((n + 1, n + 1))
    H[0, 1:] = H[1:, 0] = 1
    H[1:, 1:] = gram + LAM * np.eye(n)
    y = np.zeros(n + 1)
    y[0] = 1
    return np.linalg.solve(H, y)[1:]


def chain(G, size):
    n, N = G.shape
    out = np.zeros((n, n))
    for i in range(n):
        for j in range(i, n):
            p = (G[i].astype(np.float64) * G[j].astype(np.float64)) if size > 1 else None
            if size == 1:
                terms = (G[i] * G[j]).astype(np.float32)                   # (a chain of single products: an FMA chain rounds once less per step)
            else:
                terms = p.reshape(-1, size).sum(1).astype(np.float32)
            out[i, j] = out[j, i] = np.cumsum(terms, dtype=np.float32)[-1]
    return out


def chain16(G):
    """MKL's order for this shape (and csrc/anderson.hip gram_row_chain16_kernel's): 16 interleaved FMA chains per entry, halves onto halves at the end."""
    n, N = G.shape
    A = G.reshape(n, -1, 16).astype(np.longdouble)
    S = np.zeros((n, n, 16), np.float32)
    for k in range(A.shape[1]):
        a = A[:, k, :]
        S = (S.astype(np.longdouble) + a[:, None, :] * a[None, :, :]).astype(np.float32)
    while S.shape[-1] > 1:
        h = S.shape[-1] // 2
        S = (S[..., :h] + S[..., h:]).astype(np.float32)
    return S[..., 0].astype(np.float64)


def main():
    fi = int(os.environ.get("MEAS", "2"))
    every = int(os.environ.get("EVERY", "12"))
    clip = [as_clip(c) for c in SCITestDataset(os.path.join(ROOT, "data", "test_gray")) if "traffic" in as_clip(c)["file"]][0]
    Phi, y = clip["mask"][None].to("cuda"), clip["meas"][None, ..., fi].contiguous().to("cuda")
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    from reference_bmm import ReferenceBmmEngine
    eng = ReferenceBmmEngine(net, iterator="anderson", m=5, beta=1.0, lam=LAM, max_iter=180, tol=1e-5, use_graph=False)
    taken = []
    orig = _hip.anderson_solve
    count = [0]

    def spy(ws, slot, n_filled, n, lam, eps, res_row=0, gram32=None, ref=False):
        if gram32 is not None and n == 5 and count[0] % every == 0:
            taken.append((count[0], ws.G[0, :n].cpu().numpy().copy(), gram32[0].cpu().numpy().copy()))
        count[0] += 1
        return orig(ws, slot, n_filled, n, lam, eps, res_row, gram32=gram32, ref=ref)
    _hip.anderson_solve = spy
    import deqsci_amd.engine as E
    E._hip.anderson_solve = spy
    eng.reconstruct(y, Phi)
    _hip.anderson_solve = orig
    dump = os.environ.get("DUMP")                                  # e.g. DUMP=36,96,156: those histories (10 MB each) to gpurun_out/, for a look on the CPU
    if dump:
        keep = [int(v) for v in dump.split(",")]
        np.savez(os.path.join(ROOT, "gpurun_out", "real_history_m%d.npz" % fi), **{"G%d" % c: G for c, G, _ in taken if c in keep},
                 **{"bmm%d" % c: b for c, G, b in taken if c in keep})
        return
    rows = []
    for call, G, bmm in taken:
        exact = G.astype(np.float64) @ G.astype(np.float64).T
        a_exact = alpha64(exact)
        a_exact32 = alpha_of(exact)
        row = {"f_call": call, "|G_k|^2": float(exact[0, 0]), "alpha exact": [round(float(v), 4) for v in a_exact],
               "exact Gram, fp32 LU: alpha dev": float(np.abs(a_exact32 - a_exact).max())}
        t = torch.from_numpy(G)[None]
        cands = {"torch.bmm (rocBLAS)": bmm.astype(np.float64), "torch.bmm on the CPU (MKL)": torch.bmm(t, t.transpose(1, 2))[0].double().numpy(),
                 "16 interleaved FMA chains per entry": chain16(G)}
        for s in SIZES:
            cands[f"flat chain of {s}-element partials"] = chain(G, s)
        for name, g in cands.items():
            rel = np.abs((g - exact) / exact)
            # what alpha sees is the error of the DIFFERENCES between entries: in units of the spread of the entries
            spread = exact.max() - exact.min()
            row[name] = {"entry err mean": float(rel.mean()), "max": float(rel.max()), "err / spread of entries": float(np.abs(g - exact).max() / spread),
                         "diagonal, signed x 1e6": [round(float(v) * 1e6, 2) for v in np.diag((g - exact) / exact)],
                         "off-diagonal mean |err|": float(rel[~np.eye(len(g), dtype=bool)].mean()),
                         "alpha dev": float(np.abs(alpha_of(g) - a_exact).max())}
        rows.append(row)
        print(json.dumps(row), flush=True)
    names = [k for k in rows[0] if isinstance(rows[0][k], dict)]
    print("SUMMARY", json.dumps({k: {"median alpha dev": float(np.median([r[k]["alpha dev"] for r in rows])), "max alpha dev": max(r[k]["alpha dev"] for r in rows),
                                     "median entry err": float(np.median([r[k]["entry err mean"] for r in rows]))} for k in names}))


if __name__ == "__main__":
    with torch.no_grad():
        main()
