#!/usr/bin/env python3
"""anderson_arith = "reference" with the build's own kernels (deqsci_anderson_solve_ref_f32: gram_row_chain16_kernel, csrc/anderson.hip)
on the GPU:
  gram   - the fp32 Gram those kernels form at N = 2^19 (five correlated rows, as Anderson's residual history) against float64, next to ONE fp32
           torch.bmm on the same rows (rocBLAS - what round 4's "reference" used; the reference itself: solvers/new_equilibrium_utils_yaping.py:178)
           and the resulting alpha of all three (kernel fp32, bmm fp32, exact);
  time   - K4 and K5 in the three arithmetics, one and eight measurements of 256 x 256 x 8;
  engine - SimpleCNN @ 180 on traffic m0 against the reference's own reconstruction (tests/golden) in the three arithmetics.
`python tools/anderson_ref_check.py [gram] [time] [engine]` (default: all)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import _hip, checkpoint  # noqa: E402

MAXM = 8
DEV = "cuda"


def fill(ws, rows_f, rows_x, ref):
    """Five store + solve steps (slots 0..4), as the engine's f-calls issue them; rows_f: f(x_k), rows_x: x_k, both (bsz, m, N)."""
    m = rows_f.shape[1]
    for k in range(m):
        nf = k + 1
        _hip.residual_store(ws, rows_f[:, k].contiguous(), None, rows_x[:, k].contiguous(), k, nf, None)
        _hip.anderson_solve(ws, k, nf, nf if nf >= 2 else 0, 1e-2, 1e-5, ref=ref)


def gram():
    torch.manual_seed(0)
    for bsz in (1, 8):
        N, m = 2 ** 19, 5
        base = torch.randn(bsz, 1, N, device=DEV)
        x = torch.rand(bsz, m, N, device=DEV)
        G = (base * (1 + 0.05 * torch.arange(m, device=DEV).view(1, m, 1)) + 0.3 * torch.randn(bsz, m, N, device=DEV)) * 1e-2       # residual rows f - x
        f = x + G
        Gs = f - x                                                                       # what K4 stores (fp32 subtraction)
        exact = Gs.double() @ Gs.double().transpose(1, 2)
        bmm = torch.bmm(Gs, Gs.transpose(1, 2))
        ws = _hip.AndersonWorkspace(bsz, N, m, DEV)
        fill(ws, f, x, True)
        g32 = ws.gram32_state()[:, :m, :m].double()
        lanes = (Gs.view(bsz, m, -1, 16).permute(0, 1, 3, 2)).cpu()                      # the same order on the CPU: 16 interleaved chains, then halves onto halves
        emu = torch.zeros(bsz, m, m)
        import numpy as np
        for b in range(bsz if bsz == 1 else 1):
            L = lanes[b].numpy()
            for i in range(m):
                for j in range(m):
                    acc = np.zeros(16, np.float32)
                    a64, b64 = L[i].astype(np.longdouble), L[j].astype(np.longdouble)
                    for k in range(L.shape[2]):
                        acc = (acc.astype(np.longdouble) + a64[:, k] * b64[:, k]).astype(np.float32)
                    while acc.size > 1:
                        acc = (acc[:acc.size // 2] + acc[acc.size // 2:]).astype(np.float32)
                    emu[b, i, j] = float(acc[0])
        bit_equal = int((emu[0].double() == g32[0].cpu()).sum())
        two = _hip.gram_row_chain16(ws, m - 1, m, serial=False)[:, :m].clone()           # the 16 chain sums per entry: two-pass form against the chains as written
        walked = ws.chains_walked()[:, :m].clone().float()
        ser = _hip.gram_row_chain16(ws, m - 1, m, serial=True)[:, :m].clone()
        chains_equal = "%d / %d" % (int((two == ser).sum()), two.numel())
        a_ref = ws.alpha.clone().view(bsz, -1)[:, :m]
        ws2 = _hip.AndersonWorkspace(bsz, N, m, DEV)
        fill(ws2, f, x, False)
        a_exact = ws2.alpha.clone().view(bsz, -1)[:, :m]
        _hip.anderson_solve(ws2, m - 1, m, m, 1e-2, 1e-5, gram32=bmm.contiguous())
        a_bmm = ws2.alpha.clone().view(bsz, -1)[:, :m]

        def err(a):
            r = ((a - exact) / exact).abs()
            return {"mean": float(r.mean()), "max": float(r.max())}
        print(json.dumps({"what": "fp32 Gram against float64, relative error of an entry", "bsz": bsz, "N": N,
                          "kernel (16 interleaved FMA chains per entry)": err(g32), "entries bit-equal to a numpy emulation of that order (sample 0)": "%d / %d" % (bit_equal, m * m),
                          "chain sums, two-pass form bit-equal to the serial chains": chains_equal, "blocks walked per chain (of %d): mean, max" % (N // 2048): [float(walked.mean()), float(walked.max())], "signed error of the diagonal x 1e6 (sample 0)": [round(float(v) * 1e6, 2) for v in ((g32 - exact) / exact)[0].diagonal()], "one fp32 torch.bmm (rocBLAS)": err(bmm.double()),
                          "alpha kernels vs exact": float((a_ref - a_exact).abs().max()), "alpha bmm vs exact": float((a_bmm - a_exact).abs().max()),
                          "alpha": a_ref[0].tolist(), "alpha exact": a_exact[0].tolist()}), flush=True)


def time_():
    for bsz in (1, 8):
        N, m = 2 ** 19, 5
        g = torch.Generator(device=DEV).manual_seed(1)
        f, x = torch.rand(bsz, N, device=DEV, generator=g), torch.rand(bsz, N, device=DEV, generator=g)
        Gh = None
        out = {"bsz": bsz}
        for name, fine, kw in (("float64", False, {}), ("reference", True, {"ref": True}), ("reference-bmm", False, {"bmm": True})):     # (fine: ref=True)
            ws = _hip.AndersonWorkspace(bsz, N, m, DEV)
            for k in range(m):
                _hip.residual_store(ws, f, None, x, k, k + 1, None)
                _hip.anderson_solve(ws, k, k + 1, 0, 1e-2, 1e-5, ref=fine)

            def k4():
                _hip.residual_store(ws, f, None, x, 2, m, None)

            def k5():
                if kw.get("bmm"):
                    Gh = ws.G[:, :m]
                    _hip.anderson_solve(ws, 2, m, m, 1e-2, 1e-5, gram32=torch.bmm(Gh, Gh.transpose(1, 2)))
                else:
                    _hip.anderson_solve(ws, 2, m, m, 1e-2, 1e-5, ref=fine)
            extra = ()
            if fine:
                extra = (("the 16 x n chains, serial", lambda: _hip.gram_row_chain16(ws, 2, m, serial=True)), ("the 16 x n chains, two passes", lambda: _hip.gram_row_chain16(ws, 2, m, serial=False)))
            for label, fn in (("K4 residual_store", k4), ("K5 solve", k5)) + extra:
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out[f"{name}: {label} us"] = round(e0.elapsed_time(e1) * 10, 2)
        print(json.dumps(out), flush=True)


def engine():
    from deqsci_amd.cli import build_pipeline
    from deqsci_amd.engine import DEQSCIEngine
    from deqsci_amd.harness import SCITestDataset, as_clip
    gold = np.load(os.path.join(ROOT, "tests", "golden", "e2e_SimpleCNN_anderson_180_rec.npz"))
    clip = [as_clip(c) for c in SCITestDataset(os.path.join(ROOT, "data", "test_gray")) if "traffic" in as_clip(c)["file"]][0]
    Phi, y = clip["mask"][None].to(DEV), clip["meas"][None, ..., 0].contiguous().to(DEV)
    net = build_pipeline("SimpleCNN", checkpoint.shipped("cnn"), 180)[0].nonlinear_op
    want = gold["traffic_m0"]
    recs = {}
    for aa in ("float64", "reference", "reference-bmm"):
        if aa == "reference-bmm":
            from reference_bmm import ReferenceBmmEngine       # (tools/reference_bmm.py)
            eng = ReferenceBmmEngine(net, max_iter=180)
        else:
            eng = DEQSCIEngine(net, max_iter=180, anderson_arith=aa)
        recs[aa] = eng.reconstruct(y, Phi).cpu().numpy()

    def rel(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    print(json.dumps({"what": "SimpleCNN @ 180 Anderson iterations, traffic m0, against the reference's run",
                      **{aa + " vs reference": rel(r, want) for aa, r in recs.items()},
                      "reference vs float64": rel(recs["reference"], recs["float64"]), "reference vs reference-bmm": rel(recs["reference"], recs["reference-bmm"])}), flush=True)


if __name__ == "__main__":
    modes = sys.argv[1:] or ["gram", "time", "engine"]
    with torch.no_grad():
        if "gram" in modes:
            gram()
        if "time" in modes:
            time_()
        if "engine" in modes:
            engine()
