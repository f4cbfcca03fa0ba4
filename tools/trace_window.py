#!/usr/bin/env python3
"""What runs between two stack launches?  Reads a `rocprofv3 --kernel-trace --output-format csv` trace of bench.py (…_kernel_trace.csv) and
prints, for a few steady-state f-calls of the middle of the run, every kernel between the end of one conv_w16/conv_s16 stack launch and the
start of the next-but-one: start / end relative to the first stack's end (us), queue, short name - and the totals per kernel name over the
whole trace (count, mean, share of the traced span).  `python tools/trace_window.py <kernel_trace.csv> [n windows]`"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("deqsci::", "")
    return name[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ks = []
    for r in rows:
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))))
    ks.sort()
    t0, t1 = ks[0][0], max(k[1] for k in ks)
    tot = defaultdict(lambda: [0, 0])
    for a, b, n, q in ks:
        tot[n][0] += 1
        tot[n][1] += b - a
    print("traced span %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(ks)))
    for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
        print("  %-60s %6d x %9.1f us = %6.2f %% of the span" % (n, c, d / c / 1e3, 100.0 * d / (t1 - t0)))
    stacks = [i for i, k in enumerate(ks) if re.search(r"conv_[ws]16_kernel<1>|conv_s16_kernel<0, 0, 1>", k[2])]
    if len(stacks) < 8:
        return
    mid = len(stacks) // 2
    for w in range(nwin):
        i0, i1 = stacks[mid + 2 * w], stacks[mid + 2 * w + 2]
        base = ks[i0][1]
        print("window %d: from the end of a stack launch to the start of the next-but-one (%.1f us; the stack between them %.1f us)"
              % (w, (ks[i1][0] - base) / 1e3, (ks[stacks[mid + 2 * w + 1]][1] - ks[stacks[mid + 2 * w + 1]][0]) / 1e3))
        for a, b, n, q in ks[i0 + 1:i1 + 1]:
            print("   %9.1f .. %9.1f  (%7.1f us)  q%-3s %s" % ((a - base) / 1e3, (b - base) / 1e3, (b - a) / 1e3, q, n))


if __name__ == "__main__":
    main()
