#!/usr/bin/env python3
"""the ZERO-ROUNDING LIMIT of the denoiser on BASELINE config 2 (FFDNet + Anderson, and_maxiters=180, the six chaotic
`traffic` measurements).  The engine runs as it always does (K3 / K4 / K5+K6 / K7 HIP kernels, fp32 state) but the denoiser is evaluated
in FLOAT64 - every convolution as nine (pixels x cin) @ (cin x cout) float64 matrix products, BatchNorm unfolded, the result rounded to fp32
once - and the x0-perturbation ensemble of tools/config2_ensemble.py is run on it.  If the pooled mean PSNR of this ensemble lands on the
reference's (21.434 exact Gram / 21.439 as it is) the build's -0.010 ... -0.015 dB is the denoiser's rounding noise; if it stays at the
build's 21.42 the bias is elsewhere (GAP / mix fusion, fp64 Gram finish, BN folding) and has to be bisected.

    python tools/config2_fp64_denoiser.py [seeds=50] [seed0=0] [variants=fp64,fp64_gramf32,...]

Variants: fp64 (denoiser in float64, everything else as shipped); default (the shipped engine: for the same seeds, same box).
Output: gpurun_out/config2_fp64_denoiser.json (per-run PSNRs, pooled mean +- SE, the reference ensembles beside them)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deqsci_amd  # noqa: E402
from deqsci_amd import checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.engine import DEQSCIEngine  # noqa: E402
from deqsci_amd.harness import SCITestDataset, as_clip, psnr  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data", "test_gray")


def conv3x3_f64(x, w):
    """x (N,H,W,Cin) float64, w (Cout,Cin,3,3) float64 -> (N,H,W,Cout): nine shifted matrix products (rocBLAS dgemm), zero padding."""
    N, H, W, C = x.shape
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    out = None
    for dy in range(3):
        for dx in range(3):
            t = xp[:, dy:dy + H, dx:dx + W, :].reshape(-1, C) @ w[:, :, dy, dx].t()
            out = t if out is None else out + t
    return out.view(N, H, W, -1)


class Float64FFDNet(torch.nn.Module):
    """A plugin with tag 'ffdnet' (solvers/equilibrium_solvers_yaping.py:408-417) that evaluates networks/ffdnet/models.py:70-108 in
    float64 from the fp32 parameters: the engine sees an opaque nn.Module and calls net(x, sigma)."""
    tag = "ffdnet"

    def __init__(self, net):
        super().__init__()
        self.layers = []
        mods = list(net.intermediate_dncnn.itermediate_dncnn)
        i = 0
        while i < len(mods):
            w = mods[i].weight.detach().double()
            i += 1
            bn = None
            if i < len(mods) and isinstance(mods[i], torch.nn.BatchNorm2d):
                m = mods[i]
                bn = (m.running_mean.double(), m.running_var.double(), m.weight.detach().double(), m.bias.detach().double(), float(m.eps))
                i += 1
            relu = i < len(mods) and isinstance(mods[i], torch.nn.ReLU)
            i += int(relu)
            self.layers.append((w, bn, relu))

    @torch.no_grad()
    def forward(self, x, sigma):
        N, C, H, W = x.shape
        h = torch.cat((sigma.double().view(N, 1, 1, 1).expand(N, 1, H // 2, W // 2), F.pixel_unshuffle(x.double(), 2)), 1)
        h = h.permute(0, 2, 3, 1).contiguous()
        for w, bn, relu in self.layers:
            h = conv3x3_f64(h, w)
            if bn is not None:
                mean, var, g, b, eps = bn
                h = (h - mean) / torch.sqrt(var + eps) * g + b
            if relu:
                h = torch.relu_(h)
        return F.pixel_shuffle(h.permute(0, 3, 1, 2), 2).float()


def ensemble(eng, n_seeds, log, seed0=0):
    out = {}
    clip = [as_clip(c) for c in SCITestDataset(DATA) if "traffic" in as_clip(c)["file"]][0]
    Phi = clip["mask"].to("cuda")[None].contiguous()
    for fi in range(clip["meas"].shape[-1]):
        y = clip["meas"][..., fi].to("cuda")[None].contiguous()
        gt = clip["gt"][..., 8 * fi:8 * fi + 8].numpy()
        x0 = deqsci_amd.initial_point(y, Phi, None, None)
        ps = []
        t0 = time.time()
        for seed in range(seed0, seed0 + n_seeds):
            xs = x0 if seed == 0 else x0 * (1 + 1e-7 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(seed))).to("cuda")
            ps.append(float(psnr(eng.reconstruct(y, Phi, initial_point=xs).clamp(0, 1).cpu().numpy()[0], gt)))
        out[f"traffic_cacti.mat:{fi}"] = ps
        print(f"{log} m{fi}: mean {np.mean(ps):.4f} +- {np.std(ps, ddof=1) / np.sqrt(len(ps)):.4f}  ({time.time() - t0:.0f} s)", flush=True)
    return out


def pooled(per):
    means = np.array([np.mean(v) for v in per.values()])
    ses = np.array([np.std(v, ddof=1) / np.sqrt(len(v)) for v in per.values()])
    return float(means.mean()), float(np.sqrt((ses ** 2).sum()) / len(ses))


def reference(path):
    book = json.load(open(path))["measurements"]
    return {m: [d["psnr"] for k, d in book[m]["variants"].items() if k != "gram_fp64"] for m in book if m.startswith("traffic")}


def main():
    args = dict(a.split("=", 1) for a in sys.argv[1:])
    n_seeds = int(args.get("seeds", 50))
    seed0 = int(args.get("seed0", 0))                          # seeds seed0 .. seed0 + seeds - 1 (seed 0 = the unperturbed x0)
    variants = args.get("variants", "fp64,default").split(",")
    net = build_pipeline("ffdnet", checkpoint.shipped("ffdnet_gray"), 180)[0].nonlinear_op
    kw = dict(iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=180, tol=1e-5)
    runs = {}
    for v in variants:
        if v == "fp64":
            eng = DEQSCIEngine(Float64FFDNet(net), use_graph=False, **kw)
        elif v == "default":
            eng = DEQSCIEngine(net, **kw)
        elif v == "default_s16":                               # round 4's denoiser kernels (the direct split-fp16 stack), exact Gram
            eng = DEQSCIEngine(net, stack_kernel="s16", **kw)
        elif v == "refarith_s16":                              # ... with the reference's Anderson arithmetic
            eng = DEQSCIEngine(net, stack_kernel="s16", anderson_arith="reference", **kw)
        elif v == "fixed":                                     # the round-3 arithmetic: activation scales pinned at 2^8
            eng = DEQSCIEngine(net, act_range="fixed", **kw)
        elif v == "refarith":                                  # the reference's Anderson arithmetic (fp32 bmm Gram, fp32 LU) around the shipped denoiser
            eng = DEQSCIEngine(net, anderson_arith="reference", **kw)
        elif v == "refbmm":                                    # round 4's form of it: the Gram as one fp32 torch.bmm (rocBLAS)
            from reference_bmm import ReferenceBmmEngine       # (tools/reference_bmm.py)
            eng = ReferenceBmmEngine(net, **kw)
        elif v == "refarith_fp64":                             # ... around the float64 denoiser
            eng = DEQSCIEngine(Float64FFDNet(net), anderson_arith="reference", **kw)
        elif "+" in v:                                         # "fast+3": the first 3 f-calls on F(2x2,3x3), then the policy - a family of
            pol, k = v.split("+")                              # equivalent arithmetics (the scatter of the pooled mean between implementations)
            eng = DEQSCIEngine(net, conv64=pol, conv64_f22_calls=int(k), **kw)
        else:
            eng = DEQSCIEngine(net, conv64=v, **kw)            # fast32 / f22 / f44 / s16
        runs[v] = ensemble(eng, n_seeds, v, seed0)
    refs = {"reference as it is (fp32 bmm Gram)": reference(os.path.join(GOLDEN, "e2e_ffdnet_anderson_180_spread.json")),
            "reference, exact Gram": reference(os.path.join(GOLDEN, "e2e_ffdnet_anderson_180_spread_gram64.json"))}
    summary = {}
    for name, per in list(runs.items()) + list(refs.items()):
        mu, se = pooled(per)
        summary[name] = {"pooled_mean_psnr": round(mu, 4), "se": round(se, 4), "runs_per_measurement": len(next(iter(per.values()))),
                         "per_measurement": {m: [round(float(np.mean(p)), 4), round(float(np.std(p, ddof=1) / np.sqrt(len(p))), 4)] for m, p in per.items()}}
    diffs = {}
    for v in runs:
        for r in refs:
            d = summary[v]["pooled_mean_psnr"] - summary[r]["pooled_mean_psnr"]
            se = float(np.hypot(summary[v]["se"], summary[r]["se"]))
            diffs[f"{v} - {r}"] = {"dB": round(d, 4), "se_of_difference": round(se, 4), "in_se": round(d / se, 2)}
        for u in runs:
            if u < v:
                d = summary[v]["pooled_mean_psnr"] - summary[u]["pooled_mean_psnr"]
                se = float(np.hypot(summary[v]["se"], summary[u]["se"]))
                diffs[f"{v} - {u}"] = {"dB": round(d, 4), "se_of_difference": round(se, 4), "in_se": round(d / se, 2)}
    fam = [summary[v]["pooled_mean_psnr"] for v in runs if "+" in v]
    if len(fam) >= 3:                                          # the between-implementation scatter of the pooled statistic, measured
        summary["family of equivalent arithmetics (policy+K)"] = {"members": len(fam), "mean_of_pooled_means": round(float(np.mean(fam)), 4),
                                                                    "sd_of_pooled_means": round(float(np.std(fam, ddof=1)), 4),
                                                                    "min": round(min(fam), 4), "max": round(max(fam), 4)}
        print("FAMILY", json.dumps(summary["family of equivalent arithmetics (policy+K)"]))
    print("SUMMARY", json.dumps({k: (v["pooled_mean_psnr"], v["se"]) for k, v in summary.items() if "pooled_mean_psnr" in v}))
    print("DIFFS", json.dumps(diffs))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"what": __doc__.split("\n\n")[0], "seeds": n_seeds, "summary": summary, "differences": diffs, "runs": runs},
              open(os.path.join(ROOT, "gpurun_out", args.get("out", "config2_fp64_denoiser.json")), "w"), indent=1)


if __name__ == "__main__":
    main()
