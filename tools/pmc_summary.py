#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel HBM bytes per launch.
gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md), so reads = 2 x FETCH_SIZE KB; the
factor is calibrated on k_copy of the same run (read bytes must equal written bytes).  Usage:
    pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [kernel substring ...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    per = defaultdict(lambda: defaultdict(float))   # kernel -> dispatch -> value (summed over counter instances)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            per[row["Kernel_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):   # rocprofv3's default rocpd (sqlite) output
        import sqlite3
        cur = sqlite3.connect(f).cursor()
        for name, disp, val in cur.execute("select kernel_name, dispatch_id, value from counters_collection where counter_name = ?", (counter,)):
            per[name][disp] += float(val)
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    want = sys.argv[3:]
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    copyk = [k for k in fetch if "k_copy" in k]
    factor = 2.0
    if copyk and copyk[0] in write and fetch[copyk[0]] > 0:
        factor = write[copyk[0]] / fetch[copyk[0]]
    out = {"read_factor": factor, "kernels": {}}
    for k in sorted(fetch):
        if want and not any(w in k for w in want):
            continue
        rd = fetch[k] * factor * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"fetch_size_kb": fetch[k], "write_size_kb": write.get(k, 0.0), "hbm_bytes_per_launch": rd + wr}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
