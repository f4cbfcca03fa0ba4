#!/usr/bin/env python3
"""Runs the build's own evaluation harness on the GPU for every configuration the reference goldens cover and writes
gpurun_out/parity_report.md: per-measurement PSNR / residual next to the reference's (and, for BASELINE config 2, next to
the reference's own perturbation band), relative L2 where a full reference reconstruction is committed, and the wall-clock
frames/s of the harness measurement by measurement (the reference's schedule; hipGraph replay) and clip-batched."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deqsci_amd import checkpoint  # noqa: E402
from deqsci_amd.cli import build_pipeline  # noqa: E402
from deqsci_amd.harness import SCITestDataset, evaluate  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data", "test_gray")
WEIGHTS = {"ffdnet": "ffdnet_gray", "SimpleCNN": "cnn", "RealSN_SimpleCNN": "rsn_cnn"}


def rel_l2(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def timed(deq, clips, batch, records=None):
    """-> (average PSNR, wall seconds of the whole evaluation incl. upload + PSNR, seconds inside the reconstructions only)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    avg, results = evaluate(deq, clips, batch=batch)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if records is not None:
        for r in results:
            for i, m in enumerate(r.info["measurements"]):
                records.append({"id": f"{r.name}:{m}", "psnr": r.psnr[i], "res": r.res[i], "rec": r.rec[i:i + 1].cpu()})
    return avg, wall, sum(r.seconds for r in results)


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "2"
    out = [f"# Parity report (round {rnd}) - build harness on MI355X vs the reference's own `test_solver_sci` run on CPU", "",
           "Weights: `cnn.ckpt` (SimpleCNN), `rsn_cnn.ckpt` (RealSN_SimpleCNN), `net_gray.pth` (FFDNet substitute for the missing "
           "`ffdnet.ckpt`). 3 shipped clips = 8 measurements = 64 frames. Reference numbers: `tests/golden/e2e_*.json`.", ""]
    ds = list(SCITestDataset(DATA))                          # the three clips, read from disk once (file I/O is not what is timed)
    only = sys.argv[2] if len(sys.argv) > 2 else None         # e.g. "ffdnet:180"
    for kind, iters in (("SimpleCNN", 10), ("SimpleCNN", 100), ("SimpleCNN", 180), ("RealSN_SimpleCNN", 10), ("RealSN_SimpleCNN", 100),
                        ("ffdnet", 10), ("ffdnet", 30), ("ffdnet", 180)):
        if only and only != f"{kind}:{iters}":
            continue
        meta_path = os.path.join(GOLDEN, f"e2e_{kind}_anderson_{iters}.json")
        if not os.path.exists(meta_path):
            continue
        meta = json.load(open(meta_path))
        rec_path = os.path.join(GOLDEN, f"e2e_{kind}_anderson_{iters}_rec.npz")
        recs = np.load(rec_path) if os.path.exists(rec_path) else {}
        spread_path = os.path.join(GOLDEN, f"e2e_{kind}_anderson_{iters}_spread.json")
        spread = json.load(open(spread_path))["measurements"] if os.path.exists(spread_path) else {}
        exact_path = os.path.join(GOLDEN, f"e2e_{kind}_anderson_{iters}_spread_gram64.json")
        exact = json.load(open(exact_path))["measurements"] if os.path.exists(exact_path) else {}
        _, deq = build_pipeline(kind, checkpoint.shipped(WEIGHTS[kind]), iters)
        for batch in (False, False, True, True):          # eager warm-up, then the hipGraph capture, for both schedules
            timed(deq, ds, batch)
        records = []
        avg, dt, dtr = timed(deq, ds, False, records)
        avg_b, dtb, dtbr = timed(deq, ds, True)
        out += [f"## {kind}, Anderson m=5, and_maxiters={iters}", "",
                f"avg PSNR: reference **{meta['avg_psnr']:.4f} dB**, build **{avg:.4f} dB** (clip-batched harness {avg_b:.4f} dB); "
                f"build wall {dt:.2f} s for 64 frames = **{64 / dt:.1f} frames/s** measurement-by-measurement (the reference's schedule; "
                f"{64 / dtr:.1f} inside the reconstruction calls, i.e. without upload and PSNR), **{64 / dtb:.1f} frames/s** clip-batched "
                f"(the build's default; {64 / dtbr:.1f} inside the calls); reference CPU wall {meta['wall_s']:.0f} s = "
                f"{64 / meta['wall_s']:.3f} frames/s ({meta['threads']} threads, build container).", "",
                "| measurement | ref PSNR | build PSNR | d dB | reference band (x0 +-1e-7) | reference band with an exact Gram | ref res | build res | rel-L2 vs ref rec |",
                "|---|---|---|---|---|---|---|---|---|"]
        for r, m in zip(records, meta["measurements"]):
            key = m["id"].replace(".mat:", "_m").replace("_cacti", "")
            rl = f"{rel_l2(r['rec'].numpy(), recs[key]):.2e}" if key in recs else "-"
            band = f"[{spread[m['id']]['psnr_min']:.4f}, {spread[m['id']]['psnr_max']:.4f}]" if m["id"] in spread else "-"
            band2 = f"[{exact[m['id']]['psnr_min']:.4f}, {exact[m['id']]['psnr_max']:.4f}]" if m["id"] in exact else "-"
            out.append(f"| {m['id']} | {m['psnr']:.4f} | {r['psnr']:.4f} | {r['psnr'] - m['psnr']:+.4f} | {band} | {band2} | {m['res']:.3e} | {r['res']:.3e} | {rl} |")
        out.append("")
        print(f"{kind}@{iters}: ref {meta['avg_psnr']:.4f} build {avg:.4f}  {64 / dt:.1f} ({64 / dtr:.1f}) fps seq, {64 / dtb:.1f} ({64 / dtbr:.1f}) fps batched", flush=True)
    out += ["FFDNet + Anderson beyond ~30 iterations is chaotic in the reference itself on the `traffic` clip (a 1e-7 relative input",
            "perturbation moves its own 180-iteration output by 4e-2 rel-L2 / 0.1 dB, SURVEY F9): the 180-iteration FFDNet rows are gated",
            "against the reference's own perturbation bands - as it is (fp32 `torch.bmm` Gram) and with an exactly accumulated Gram matrix,",
            "which is what this build computes (`tests/golden/e2e_ffdnet_anderson_180_spread.json`, `..._spread_gram64.json`, DESIGN.md section 5,",
            "`tests/test_gpu_parity.py::test_config2_ffdnet_anderson_180_all_measurements`)."]
    with open(os.path.join(ROOT, "gpurun_out", "parity_report.md"), "w") as fh:
        fh.write("\n".join(out) + "\n")


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
