#!/usr/bin/env python3
"""Where the time between kernels goes: gaps of a rocprofv3 --kernel-trace CSV.

    python3 tools/trace_gaps.py <dir-or-csv> [--tail 0.6] > summary.json

Takes the last `tail` fraction of the dispatches (the timed steps of a bench run: set-up and warm-up come first), orders
them by start time and reports: the span, the time inside kernels, the time between them by gap size, which kernel
precedes the long gaps (a host read-back sits behind it), and count / average duration per kernel name."""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("mpp::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--tail", type=float, default=0.6)
    a = ap.parse_args()
    files = [a.path] if a.path.endswith(".csv") else sorted(glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % a.path)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    rows = rows[int(len(rows) * (1.0 - a.tail)):]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    buckets = [(2, "<2us"), (5, "2-5us"), (10, "5-10us"), (20, "10-20us"), (50, "20-50us"), (200, "50-200us"), (1e18, ">200us")]
    gaps = {lbl: [0, 0.0] for _, lbl in buckets}
    before = {}
    per = {}
    end_prev = None
    for s, e, nm in rows:
        busy += e - s
        p = per.setdefault(nm, [0, 0])
        p[0] += 1
        p[1] += e - s
    prev = None
    covered_end = rows[0][1]
    for s, e, nm in rows:
        if prev is not None:
            g = (s - covered_end) / 1e3
            if g > 0:
                for lim, lbl in buckets:
                    if g < lim:
                        gaps[lbl][0] += 1
                        gaps[lbl][1] += g
                        break
                if g >= 10:
                    b = before.setdefault(prev, [0, 0.0])
                    b[0] += 1
                    b[1] += g
        covered_end = max(covered_end, e)
        prev = nm
    out = {
        "files": [os.path.basename(f) for f in files],
        "dispatches": len(rows), "span_ms": span / 1e6, "in_kernels_ms": busy / 1e6,
        "between_kernels_ms": (span - busy) / 1e6, "between_frac": (span - busy) / span,
        "gaps": {k: {"count": v[0], "total_ms": v[1] / 1e3} for k, v in gaps.items()},
        "long_gaps_follow": sorted(({"kernel": k, "count": v[0], "total_ms": v[1] / 1e3} for k, v in before.items()),
                                   key=lambda d: -d["total_ms"])[:12],
        "kernels": sorted(({"kernel": k, "count": v[0], "avg_us": v[1] / v[0] / 1e3, "total_ms": v[1] / 1e6} for k, v in per.items()),
                          key=lambda d: -d["total_ms"])[:30],
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
