#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/kernel_bench.py into per-launch HBM
bytes per kernel, calibrated as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the
counters are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads, so both counters are
calibrated on a kernel with a known byte count in the same access pattern measured in the same run - this
repo's LDS transpose (16-B-per-lane reads of exactly N bytes, writes of exactly N bytes).  The calibration
reproduces the guide's factor (FETCH_SIZE x 2.000, WRITE_SIZE x 1.000)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"deqsci::(\w+)(<[^>]*>)?", name)
    if m:                                     # template args: <shape, cache policy> -> keep the shape argument only
        arg = m.group(2) or ""
        first = arg.strip("<>").split(",")[0].strip() if arg else ""
        return m.group(1) + (f"<{first}>" if first and m.group(1) not in ("forward_bhw_kernel", "adjoint_bhw_kernel", "phisum_bhw_kernel", "sub_flat_kernel", "mix_kernel") else "")
    return None


def load(path):
    acc = defaultdict(list)
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            if k:
                acc[k].append(float(row["Counter_Value"]))
    return acc


def main(root, bsz, H=256, W=256, B=8, m=5):
    fetch = load(f"{root}/FETCH_SIZE_b{bsz}/k_counter_collection.csv")
    write = load(f"{root}/WRITE_SIZE_b{bsz}/k_counter_collection.csv")
    px = bsz * H * W
    copy_bytes = px * B * 4
    med = lambda v: sorted(v)[len(v) // 2]
    cal = "transpose_hwb2bhw_kernel<%d>" % (B // 4)
    f_cal = copy_bytes / (med(fetch[cal]) * 1024)
    w_cal = copy_bytes / (med(write[cal]) * 1024)
    alg = {"gap_bhw_kernel<8>": (px * (8 * B + 8), px * 4 * B), "gap_hwb_kernel<2>": (px * (8 * B + 8), px * 4 * B),
           "gap_hwb2bhw_kernel<2>": (px * (8 * B + 8), px * 4 * B),
           "mix_gap_bhw_kernel<8>": (px * (4 * B * (m + 1) + 8), px * 8 * B), "mix_gap_hwb_kernel<2>": (px * (4 * B * (m + 1) + 8), px * 8 * B),
           "residual_store_kernel<5>": (px * 4 * B * (3 + m - 1), px * 8 * B), "forward_hwb_kernel<2>": (px * 8 * B, px * 4),
           "adjoint_hwb_kernel<2>": (px * (4 * B + 4), px * 4 * B), "mix_kernel": (px * 4 * B * m, px * 4 * B)}
    out = {"bsz": bsz, "size": f"{H}x{W}x{B}", "calibration": {"kernel": "%s moving %d bytes each way" % (cal, copy_bytes),
           "fetch_factor": round(f_cal, 4), "write_factor": round(w_cal, 4)}, "kernels": {}}
    for k in sorted(set(fetch) & set(write)):
        fb = med(fetch[k]) * 1024 * f_cal
        wb = med(write[k]) * 1024 * w_cal
        rec = {"launches": len(fetch[k]), "FETCH_SIZE_KiB_raw": med(fetch[k]), "WRITE_SIZE_KiB_raw": med(write[k]),
               "hbm_read_bytes": int(fb), "hbm_write_bytes": int(wb), "hbm_bytes_per_launch": int(fb + wb)}
        if k in alg:
            rec["algorithmic_bytes"] = alg[k][0] + alg[k][1]
            rec["traffic_over_algorithmic"] = round((fb + wb) / (alg[k][0] + alg[k][1]), 3)
        out["kernels"][k] = rec
    return out


if __name__ == "__main__":
    root = sys.argv[1]
    res = [main(root, int(b)) for b in sys.argv[2:]]
    print(json.dumps(res, indent=1))
