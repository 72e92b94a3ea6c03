#!/usr/bin/env python3
"""Per-launch view of the ILU(0) box solves from a rocprofv3 --kernel-trace CSV: for the last complete application, the duration
of every k_box_lower / k_box_upper launch against the number of boxes in it (grid size), and the gap to the previous launch.

    python3 tools/box_trace.py <dir-or-csv> > summary.json"""
import csv
import glob
import json
import os
import sys


def main():
    path = sys.argv[1]
    files = [path] if path.endswith(".csv") else sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                             int(r.get("Workgroup_Size_X", 64) or 64)))
    rows.sort()
    out = {}
    for tag in ("k_box_lower", "k_box_upper"):
        sel = [(s, e, g // max(w, 1)) for s, e, nm, g, w in rows if tag + "<" in nm and "queue" not in nm]
        if not sel:
            continue
        # applications are separated by other kernels: cut where the gap to the previous box launch exceeds 100 us
        apps, cur = [], [sel[0]]
        for a, b in zip(sel, sel[1:]):
            if b[0] - a[1] > 100000:
                apps.append(cur); cur = []
            cur.append(b)
        apps.append(cur)
        full = [a for a in apps if len(a) == max(len(x) for x in apps)]
        app = full[-1]
        dur = [(e - s) / 1e3 for s, e, _ in app]
        gap = [0.0] + [(b[0] - a[1]) / 1e3 for a, b in zip(app, app[1:])]
        nbx = [n for _, _, n in app]
        out[tag] = {"launches": len(app), "applications_seen": len(full), "span_us": (app[-1][1] - app[0][0]) / 1e3, "kernel_us": sum(dur), "gap_us": sum(gap),
                    "avg_gap_us": sum(gap) / max(len(gap) - 1, 1), "boxes": nbx, "duration_us": [round(d, 2) for d in dur], "gap_before_us": [round(g, 2) for g in gap]}
    for tag in ("k_box_lower_queue", "k_box_upper_queue", "k_box_gather_in", "k_gather<", "k_gather("):
        sel = [(e - s) / 1e3 for s, e, nm, g, w in rows if tag in nm]
        if sel:
            out[tag] = {"launches": len(sel), "avg_us": sum(sel) / len(sel), "min_us": min(sel)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
