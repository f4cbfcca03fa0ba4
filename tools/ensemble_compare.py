#!/usr/bin/env python3
"""Pooled comparison of the build's config-2 ensembles (tools/config2_ensemble.py output) with the reference's (spread files written by
tests/golden/make_golden.py g10, possibly extended with more seeds): per measurement and pooled over the six chaotic `traffic` measurements,
mean, standard error, difference in units of the standard error of the difference.
    python tools/ensemble_compare.py <build json> <reference spread json> [<reference spread json> ...]"""
import json
import sys

import numpy as np


def ref_runs(path):
    book = json.load(open(path))
    out = {}
    for mid, m in book["measurements"].items():
        v = m["variants"]
        out[mid] = np.array([d["psnr"] for k, d in v.items() if k != "gram_fp64"])      # (the fp64-Gram row of the as-is file is another variant)
    return out


def build_runs(path):
    book = json.load(open(path))
    out = {}
    for mid, m in book.items():
        if not mid.startswith("traffic"):
            continue
        for name, d in m.items():
            if isinstance(d, dict) and "runs" in d:
                out.setdefault(name, {})[mid] = np.array(d["runs"])
            elif isinstance(d, dict) and "mean" in d:
                out.setdefault(name, {})[mid] = d
    return out


def pooled(per):
    means = np.array([np.mean(v) for v in per.values()])
    ses = np.array([np.std(v, ddof=1) / np.sqrt(len(v)) for v in per.values()])
    return float(means.mean()), float(np.sqrt((ses ** 2).sum()) / len(ses))


def main():
    build = json.load(open(sys.argv[1]))
    mids = [f"traffic_cacti.mat:{i}" for i in range(6)]
    refs = {p: ref_runs(p) for p in sys.argv[2:]}
    rows = {}
    for p, r in refs.items():
        per = {m: r[m] for m in mids}
        mu, se = pooled(per)
        rows["reference " + p.split("/")[-1]] = (mu, se, {m: (float(np.mean(per[m])), float(np.std(per[m], ddof=1) / np.sqrt(len(per[m]))), len(per[m])) for m in mids})
    summ = build.get("summary", {})
    for name, d in summ.items():
        if "per_measurement" in d:
            rows["build " + name[:60]] = (d["traffic_mean"], d["se"], {m: (d["per_measurement"][m][0], d["per_measurement"][m][1], d["runs_per_measurement"]) for m in mids})
    for k, (mu, se, per) in rows.items():
        print(f"{k:90s} pooled {mu:.4f} +- {se:.4f}   " + " ".join(f"{per[m][0]:.3f}({per[m][2]})" for m in mids))
    names = list(rows)
    print()
    for a in names:
        if not a.startswith("build"):
            continue
        for b in names:
            if not b.startswith("reference"):
                continue
            d = rows[a][0] - rows[b][0]
            sd = float(np.hypot(rows[a][1], rows[b][1]))
            rms = float(np.sqrt(np.mean([(rows[a][2][m][0] - rows[b][2][m][0]) ** 2 for m in mids])))
            print(f"{a[:50]:50s} - {b:55s}: {d:+.4f} dB = {d / sd:+.2f} SE of the difference ({sd:.4f}); RMS per-measurement offset {rms:.4f}")


if __name__ == "__main__":
    main()
