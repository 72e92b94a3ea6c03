#!/usr/bin/env python3
"""Register / LDS / occupancy table of the HIP kernels (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).

    python tools/resource_usage.py [pattern ...] > profiles/rNN_kernel_resource_usage.md

Cross-compiles mpp_amd/csrc/*.hip (no GPU needed) and prints one markdown row per kernel instantiation whose demangled
name contains one of the patterns (default: the kernels of the 1024x1024x100 solve).  Waves/SIMD allowed by the
register allocation follows MI355X_MICROARCH.md (granule 8, 512 registers per lane and SIMD).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpp_amd", "csrc")
DEFAULT = ["k_assemble_cols", "k_tri_", "k_spmv_c", "k_spmv_pipe_mb", "k_bcgs_xr", "k_waxpy", "k_dots_pat", "k_bc<", "k_th_cols", "k_box_", "k_finalize",
           "k_spmv_tri"]


def remarks(src: str) -> str:
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                            os.path.join(d, "x.o"), "-Rpass-analysis=kernel-resource-usage"],
                           cwd=CSRC, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise SystemExit(r.returncode)
        return r.stderr


def main():
    pats = sys.argv[1:] or DEFAULT
    rows = []
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        blocks = re.split(r"remark: Function Name: ", remarks(src))[1:]
        for b in blocks:
            mangled = b.split(" ", 1)[0]

            def g(key):
                m = re.search(re.escape(key) + r": (\d+)", b)
                return int(m.group(1)) if m else -1
            rows.append((src, mangled, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"),
                         g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
    names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.split("\n")
    print("| kernel | file | VGPR | AGPR | SGPR | scratch B/lane | waves/SIMD (compiler) | LDS B/block |")
    print("|---|---|---|---|---|---|---|---|")
    for r, name in zip(rows, names):
        name = name.replace("mpp::(anonymous namespace)::", "").replace("void ", "")
        if not any(p in name for p in pats):
            continue
        short = re.sub(r"\(.*$", "", name)
        print("| `%s` | %s | %d | %d | %d | %d | %d | %d |" % ((short, r[0]) + r[2:]))


if __name__ == "__main__":
    main()
