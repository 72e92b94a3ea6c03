#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ/GRBM counter passes (CSV output) per kernel: averages per dispatch over the dispatches of a
kernel, plus derived figures.  Usage: pmc_sq_summary.py <dir> [<dir> ...] [--match substr ...]
Derived (MI355X_MICROARCH.md): effective clock = GRBM_GUI_ACTIVE / 8 / kernel time (the counter is summed over 8 XCDs);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, summed over waves: shares of wave lifetime."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict


def main():
    args = sys.argv[1:]
    match = []
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1:]
        args = args[:i]
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> dispatch -> counter -> value
    meta = {}
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if match and not any(m in k for m in match):
                    continue
                key = (d, row["Dispatch_Id"])
                per[k][key][row["Counter_Name"]] += float(row["Counter_Value"])
                per[k][key]["_ns"] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                meta[k] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Workgroup_Size"], row["Grid_Size"])
    for k in sorted(per):
        disp = list(per[k].values())
        # skip the first dispatch of a kernel in each pass (warm-up)
        names = sorted({c for d in disp for c in d if c != "_ns"})
        avg = {}
        for c in names + ["_ns"]:
            vals = [d[c] for d in disp if c in d]
            if len(vals) > 1:
                vals = vals[1:]
            avg[c] = sum(vals) / len(vals)
        short = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", k.replace("void ", "").replace("mpp::(anonymous namespace)::", ""))
        print("## %s" % short)
        print("VGPR %s (+%s AGPR), SGPR %s, LDS %s B/block, block %s, grid %s threads; %d dispatches; %.4f ms per dispatch (under the profiler)"
              % (meta[k] + (len(disp), avg["_ns"] * 1e-6)))
        for c in names:
            print("  %-22s %.4g" % (c, avg[c]))
        if "GRBM_GUI_ACTIVE" in avg:
            print("  effective clock        %.2f GHz" % (avg["GRBM_GUI_ACTIVE"] / 8.0 / avg["_ns"]))
        if "SQ_WAVES" in avg and "SQ_INSTS_VALU" in avg:
            print("  VALU instructions/wave %.0f" % (avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]))
        if "SQ_WAVE_CYCLES" in avg:
            wc = avg["SQ_WAVE_CYCLES"]
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in avg:
                    print("  %-22s %.1f %% of wave lifetime" % (c + " share", 100.0 * avg[c] / wc))
        print()


if __name__ == "__main__":
    main()
