#!/usr/bin/env python3
"""BASELINE config 2 (FFDNet + Anderson @180) as ENSEMBLES on the HIP path: unperturbed x0 and x0 (1 + 1e-7 randn), seeds 1..8 (the
recipe of the reference ensembles in tests/golden), for several engine configurations; prints the ensemble mean per measurement next
to the two reference ensemble means.  `python tools/config2_ensemble.py [name=key:value,...] ...`"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deqsci_amd  # noqa: E402
from deqsci_amd import _hip, checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip, psnr, scored_measurements  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data", "test_gray")


N_SEEDS = int(os.environ.get("DEQSCI_ENSEMBLE_SEEDS", "9"))          # 9 = the recipe of the reference ensembles (seed 0 = unperturbed)
TRAFFIC_ONLY = os.environ.get("DEQSCI_ENSEMBLE_TRAFFIC_ONLY", "0") == "1"
ONLY = os.environ.get("DEQSCI_ENSEMBLE_CONFIGS")                      # comma-separated substrings of the configuration names


def ensemble(eng):
    out = {}
    for clip in (as_clip(c) for c in SCITestDataset(DATA)):
        if TRAFFIC_ONLY and "traffic" not in clip["file"]:
            continue
        Phi = clip["mask"].to("cuda")[None].contiguous()
        for fi in scored_measurements(clip["file"], clip["meas"].shape[-1]):
            y = clip["meas"][..., fi].to("cuda")[None].contiguous()
            gt = clip["gt"][..., 8 * fi:8 * fi + 8].numpy()
            x0 = deqsci_amd.initial_point(y, Phi, None, None)
            ps = []
            for seed in range(N_SEEDS):
                xs = x0 if seed == 0 else x0 * (1 + 1e-7 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(seed))).to("cuda")
                ps.append(psnr(eng.reconstruct(y, Phi, initial_point=xs).clamp(0, 1).cpu().numpy()[0], gt))
            out[f"{clip['file']}:{fi}"] = ps
    return out


def main():
    a = json.load(open(os.path.join(GOLDEN, "e2e_ffdnet_anderson_180_spread.json")))["measurements"]
    b = json.load(open(os.path.join(GOLDEN, "e2e_ffdnet_anderson_180_spread_gram64.json")))["measurements"]
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    configs = {"engine default: conv64='auto' = 'fast' (split-fp16 direct conv on the f16 matrix cores at this size)": {},
               "conv64='fast32' (fp32 MFMA only: F(4x4,3x3) at this size)": {"kw": dict(conv64="fast32")},
               "conv64='f22' (F(2x2,3x3) throughout)": {"kw": dict(conv64="f22")},
               "MIOpen direct fp32 conv, BN not folded": {"kw": dict(winograd=False, fold_bn=False)}}
    for k in [int(v) for v in os.environ.get("DEQSCI_ENSEMBLE_HYBRID", "").split(",") if v]:
        configs[f"conv64='fast32', first {k} f-calls on F(2x2,3x3)"] = {"kw": dict(conv64="fast32", conv64_f22_calls=k)}
    for k in [int(v) for v in os.environ.get("DEQSCI_ENSEMBLE_HYBRID_S16", "").split(",") if v]:
        configs[f"conv64='s16', first {k} f-calls on F(2x2,3x3)"] = {"kw": dict(conv64="s16", conv64_f22_calls=k)}
    res = {}
    if ONLY:
        configs = {k: v for k, v in configs.items() if any(o in k for o in ONLY.split(","))}
    for name, c in configs.items():
        _hip.FORCE_CONV64 = c.get("force")
        eng = DEQSCIEngine(net, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=180, tol=1e-5, **c.get("kw", {}))
        res[name] = ensemble(eng)
        _hip.FORCE_CONV64 = None
    rows = {}
    for mid in a:
        if not all(mid in v for v in res.values()):
            continue
        ra = [v["psnr"] for k, v in a[mid]["variants"].items() if k != "gram_fp64"]        # (that row is an exact-Gram run)
        rb = [v["psnr"] for v in b[mid]["variants"].values()]
        rows[mid] = {"reference fp32 Gram mean": round(float(np.mean(ra)), 4), "reference exact Gram mean": round(float(np.mean(rb)), 4),
                     **{k: {"mean": round(float(np.mean(v[mid])), 4), "min": round(min(v[mid]), 4), "max": round(max(v[mid]), 4)} for k, v in res.items()}}
        print(mid, json.dumps(rows[mid]))
    # the six chaotic measurements together: mean of the per-measurement ensemble means, standard error from the per-measurement variances
    tr = [m for m in rows if m.startswith("traffic")]
    summary = {}
    for k, v in res.items():
        means = [np.mean(v[m]) for m in tr]
        se = np.sqrt(sum(np.var(v[m], ddof=1) / len(v[m]) for m in tr)) / len(tr)
        summary[k] = {"traffic_mean": round(float(np.mean(means)), 4), "se": round(float(se), 4), "runs_per_measurement": N_SEEDS,
                      "per_measurement": {m: [round(float(np.mean(v[m])), 4), round(float(np.std(v[m], ddof=1) / np.sqrt(len(v[m]))), 4)] for m in tr}}
    for tag, ref in (("reference fp32 Gram", a), ("reference exact Gram", b)):
        vals = {m: [x["psnr"] for kk, x in ref[m]["variants"].items() if kk != "gram_fp64"] for m in tr}   # (an exact-Gram row inside the as-is file)
        se = np.sqrt(sum(np.var(vals[m], ddof=1) / len(vals[m]) for m in tr)) / len(tr)
        summary[tag] = {"traffic_mean": round(float(np.mean([np.mean(vals[m]) for m in tr])), 4), "se": round(float(se), 4)}
    print("SUMMARY", json.dumps(summary))
    rows["summary"] = summary
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "config2_ensemble.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
