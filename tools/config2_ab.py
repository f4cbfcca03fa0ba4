#!/usr/bin/env python3
"""BASELINE config 2 (FFDNet, Anderson @180, all 8 shipped measurements) under A/B variants of the HIP path:
default engine, BN not folded, MIOpen convs instead of Winograd, clip-batched, the generic (non-engine) DEQ path and
1e-7 perturbations of x0.  Writes gpurun_out/config2_ab.json: per-measurement PSNR / residual per variant, next to
the reference's ensemble band (tests/golden/e2e_ffdnet_anderson_180_spread.json) when that file exists."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deqsci_amd  # noqa: E402
from deqsci_amd import checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip, psnr, scored_measurements  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data", "test_gray")


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 180
    dev = "cuda"
    solver, _ = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), iters)
    net = solver.nonlinear_op
    variants = {
        "default": dict(),
        "no_fold_bn": dict(fold_bn=False),
        "no_winograd": dict(winograd=False),
        "no_winograd_no_edges": dict(winograd=False, fused_edges=False),
    }
    out = {"iters": iters, "variants": {}}
    clips = [as_clip(c) for c in SCITestDataset(DATA)]
    for name, kw in variants.items():
        eng = DEQSCIEngine(net, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=iters, tol=1e-5, **kw)
        rows = {}
        for clip in clips:
            Phi = clip["mask"].to(dev)[None].contiguous()
            for fi in scored_measurements(clip["file"], clip["meas"].shape[-1]):
                y = clip["meas"][..., fi].to(dev)[None].contiguous()
                rec = eng.reconstruct(y, Phi)
                gt = clip["gt"][..., 8 * fi:8 * fi + 8].numpy()
                rows[f"{clip['file']}:{fi}"] = {"psnr": psnr(rec.clamp(0, 1).cpu().numpy()[0], gt), "res": eng.last_info["res"]}
        out["variants"][name] = rows
        print(name, {k: round(v["psnr"], 4) for k, v in rows.items()}, flush=True)
    # x0 perturbations through the default engine (the reference ensemble's recipe)
    eng = DEQSCIEngine(net, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=iters, tol=1e-5)
    for seed in range(1, 9):
        rows = {}
        for clip in clips:
            Phi = clip["mask"].to(dev)[None].contiguous()
            for fi in scored_measurements(clip["file"], clip["meas"].shape[-1]):
                y = clip["meas"][..., fi].to(dev)[None].contiguous()
                x0 = deqsci_amd.initial_point(y, Phi, None, None)
                g = torch.Generator().manual_seed(seed)
                x0 = x0 * (1 + 1e-7 * torch.randn(x0.shape, generator=g)).to(dev)
                rec = eng.reconstruct(y, Phi, initial_point=x0)
                gt = clip["gt"][..., 8 * fi:8 * fi + 8].numpy()
                rows[f"{clip['file']}:{fi}"] = {"psnr": psnr(rec.clamp(0, 1).cpu().numpy()[0], gt), "res": eng.last_info["res"]}
        out["variants"][f"x0_seed{seed}"] = rows
        print(f"x0_seed{seed}", {k: round(v["psnr"], 4) for k, v in rows.items()}, flush=True)
    ref = json.load(open(os.path.join(GOLDEN, f"e2e_ffdnet_anderson_{iters}.json")))
    out["reference_base"] = {m["id"]: {"psnr": m["psnr"], "res": m["res"]} for m in ref["measurements"]}
    sp = os.path.join(GOLDEN, f"e2e_ffdnet_anderson_{iters}_spread.json")
    if os.path.exists(sp):
        s = json.load(open(sp))["measurements"]
        out["reference_band"] = {k: [v["psnr_min"], v["psnr_max"]] for k, v in s.items()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"config2_ab_{iters}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    ids = list(out["reference_base"])
    print("%-24s" % "measurement", " ".join("%12s" % v[:12] for v in ["ref"] + list(out["variants"])))
    for mid in ids:
        print("%-24s" % mid, " ".join("%12.4f" % v for v in [out["reference_base"][mid]["psnr"]] + [out["variants"][k][mid]["psnr"] for k in out["variants"]]))


if __name__ == "__main__":
    main()
