#!/usr/bin/env python3
"""(round-4 diagnostic) FFDNet @30 / SimpleCNN @180 on scaled measurements: every conv64 policy against the reference's golden and against one
another - is a 1e-4 miss arithmetic or conditioning?"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import checkpoint
from deqsci_amd.cli import build_pipeline
from deqsci_amd.engine import DEQSCIEngine
from deqsci_amd.harness import SCITestDataset, as_clip
gold = np.load(os.path.join(ROOT, "tests/golden/e2e_scaled_measurements.npz"))
clip = [as_clip(c) for c in SCITestDataset(os.path.join(ROOT, "data/test_gray")) if "traffic" in as_clip(c)["file"]][0]
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for kind, w, iters, crop in (("ffdnet", "ffdnet_gray", 30, 256), ("SimpleCNN", "cnn", 180, 128)):
    net = build_pipeline(kind, checkpoint.shipped(w), iters)[0].nonlinear_op
    for scale in (1e-2, 1e-4, 1e-6, 1e2):
        Phi = clip["mask"][None, :crop, :crop].contiguous().cuda()
        y = (clip["meas"][None, :crop, :crop, 0] * np.float32(scale)).contiguous().cuda()
        want = gold[f"{kind}_{iters}_s{scale:g}_rec"]
        recs = {}
        for name, kw in (("data", {}), ("fixed", dict(act_range="fixed", conv64="fast")), ("fast32", dict(conv64="fast32")), ("f22", dict(conv64="f22")),
                         ("miopen", dict(winograd=False))):
            for it in ((iters, 10) if kind == "ffdnet" else (iters,)):
                eng = DEQSCIEngine(net, max_iter=it, use_graph=False, **kw)
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    recs[(name, it)] = eng.reconstruct(y, Phi).cpu().numpy()
        out = {f"{n}@{it} vs gold": "%.2e" % rel(r, want) for (n, it), r in recs.items() if it == iters}
        out.update({f"{n}@{it} vs f22@{it}": "%.2e" % rel(r, recs[("f22", it)]) for (n, it), r in recs.items() if n != "f22"})
        print(kind, scale, json.dumps(out), flush=True)
