#!/usr/bin/env python3
"""Fill the measured numbers of DESIGN.md from the evidence files of a round (the @@NAME@@ placeholders of DESIGN.md.in).

    python3 tools/fill_design.py profiles/r05_evidence > DESIGN.md

Reads bench.json, bench2.json, bench_256x256x50*.json, bench_1024x128x100.json, kernel_traffic.json,
rocprofv3_kernel_stats_bench.csv, th_512x512x80.json, thermal_512.json, pmc_th_512x512x80.json, pmc_thermal_512.json and the
rocprofv3_kernel_stats_{th_512x512x80,thermal_512}.csv of that directory, and profiles/r03_kbench_laws*.json (the column march
per constitutive law: those kernels did not change in round 4)."""
from __future__ import annotations

import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 104857600
NNZ = 3 * N - 2 * 1024 * 1024


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("mpp::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    ev = sys.argv[1]
    J = lambda f: json.load(open(os.path.join(ev, f)))
    bench, bench2 = J("bench.json"), J("bench2.json")
    c2, c2s, s8 = J("bench_256x256x50.json"), J("bench_256x256x50_spmv.json"), J("bench_1024x128x100.json")
    traffic = J("kernel_traffic.json")
    th = J("th_512x512x80.json")
    stats = {}
    for r in csv.DictReader(open(os.path.join(ev, "rocprofv3_kernel_stats_bench.csv"))):
        stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) * 1e-6)
    moved = {}
    for cls in traffic["classes"].values():
        moved.update(cls["kernels"])

    def row(kern, what, alg_per_cell, note=""):
        calls, ms = stats.get(kern, (0, None))
        mv = moved.get(kern)
        if ms is None:
            return None
        alg = alg_per_cell * N
        fa = alg / (ms * 1e-3) / 8e12
        fm = (mv / (ms * 1e-3) / 8e12) if mv else None
        return "| `%s` | %s | %.1f·N | %s | %.3f ms (%d launches) | %.2f / %s |%s" % (
            kern, what, alg_per_cell, ("%.1f·N" % (mv / N)) if mv else "—", ms, calls, fa, ("%.2f" % fm) if fm else "—", note)

    bspmv = (12.0 * NNZ + 20.0 * N + 4.0) / N
    rows = ["| kernel | what it restates | algorithmic bytes / launch | moved (PMC) | rocprofv3 average in the bench | frac of 8 TB/s (algorithmic / moved) |",
            "|---|---|---|---|---|---|"]
    for r in (
        row("k_spmv_cols_dots", "PETSc `MatMult` + the line search's `VecDot` / `VecNorm`s (A x not written), from the CSR values at their computed places — the line search's own product pass: since `ls_fold` (round 5) only where the Krylov solve took more than one iteration or ‖X‖ was not at hand (the roofline kernel of round 4)", bspmv),
        row("k_spmv_c<3, false, true>", "… on the compact 16-bit index (grids that are not soil columns)", bspmv),
        row("k_spmv_c<3, false, false>", "PETSc `MatMult_SeqAIJ` (y = A x): multi-iteration Krylov solves, 3-D problems; timed stand-alone behind the bench", bspmv),
        row("k_tri_solve_lds<2, 4, 3, 4, 2, true>", "start of `KSPSolve_BCGS`: `PCApply` (back substitution of L⁻¹F), `MatMult`, `PCApply`, 3 dots + the line search's F·(J p) (`ls_fold`) — **the `roofline` kernels** (this one and the next, averaged)", 32 + bspmv + 48),
        row("k_tri_solve_lds<2, 5, 2, 4, 3, true>", "`VecWAXPY` (s = r − αv), `MatMult`, `PCApply`, 2 dots + F·(J s)", 24 + bspmv + 48),
        row("k_tri_solve_lds<2, 4, 3, 4, 2, false>", "… without the line search's sum (a solve whose sums are not folded)", 32 + bspmv + 48),
        row("k_tri_solve_lds<2, 5, 2, 4, 3, false>", "… likewise", 24 + bspmv + 48),
        row("k_tri_solve_lds<2, 4, 3, 4, 1, false>", "`MatMult` + `PCApply` + 2 dots (later iterations, GMRES)", bspmv + 48),
        row("k_assemble_cols<2, 3, 0, true, false>", "GE:388-453, 1603-2200 (residual + Jacobian), `PCSetUp`, forward elimination, `VecWAXPY`, 2 `VecNorm`", 148.0),
        row("k_assemble_cols<0, 3, 0, true, false>", "GE:388-421, 1603-1938 (residual), `VecWAXPY`, 2 `VecNorm`", 104.0),
        row("k_bcgs_xr<true, true>", "`VecAXPBYPCZ`, `VecWAXPY`, `VecNorm` (r not written when the solve is expected to end) + the line search's Y·Y", 40.0),
        row("k_bcgs_xr<true, false>", "… without Y·Y", 40.0),
        row("k_bc<2>", "boundary fluxes + derivatives (GE:1815-1877, 2076-2139) into side arrays", 0.0),
    ):
        if r:
            rows.append(r)
    rows.append("")
    rows.append("The rows above 1.0 on algorithmic bytes move fewer bytes than SURVEY's count (A v never written, soil classes, per-level "
                "geometry, computed row slots): the honest figure for them is the moved one.  `k_finalize` (5–8 µs, "
                "one per reduction) and the thermal / TH / ILU(0) kernels are in `profiles/`.")
    kernel_table = "\n".join(rows)

    laws = json.load(open(os.path.join(ROOT, "profiles", "r03_kbench_laws.json")))["laws"]
    names = [("vg_tgdpb01", "van Genuchten m = ½, TGDPB01, soil classes (the benchmark's soil)"),
             ("vg_tgdpb01_percell", "… per-cell soils"),
             ("vgm_tgdpb01", "van Genuchten m = 0.4 (general powers)"),
             ("bc_tgdpb01", "Brooks–Corey"),
             ("sbc2_tgdpb01", "smoothed Brooks–Corey bz2"),
             ("sbc3_tgdpb01", "smoothed Brooks–Corey bz3"),
             ("sbc3_tgdpb01_percell", "smoothed Brooks–Corey bz3, per-cell soils (**ELM's default**)"),
             ("sbc3_ifc67", "smoothed Brooks–Corey bz3, IFC-67 density"),
             ("vg_ifc67", "van Genuchten m = ½, IFC-67 density"),
             ("mixed_tgdpb01", "three functions mixed per column (general-soil march)")]
    pmc = {}
    pf = os.path.join(ROOT, "profiles", "r03_kbench_laws_pmc.json")
    if os.path.exists(pf):
        pmc = json.load(open(pf))["laws"]
    lt = ["| constitutive law | residual + Jacobian march, round-2 kernels (ms) | round 3 (ms) | HBM bytes moved per cell (PMC) | frac of 8 TB/s on moved bytes | residual only (ms) |",
          "|---|---|---|---|---|---|"]
    for k, label in names:
        if k in laws and "final" in laws[k]:
            e = pmc.get(k, {})
            both = e.get("assemble_both", {})
            lt.append("| %s | %.2f | **%.2f** | %s | %s | %.2f |" % (
                label, laws[k]["r02_code"]["assemble_both"], both.get("ms", laws[k]["final"]["assemble_both"]),
                ("%.1f" % both["moved_bytes_per_cell"]) if both else "—", ("%.2f" % both["frac_of_peak_moved"]) if both else "—",
                e.get("residual", {}).get("ms", laws[k]["final"]["residual"])))
    laws_table = "\n".join(lt)

    rb = lambda j: j.get("host_readbacks_per_newton_iteration")
    results = ["| configuration (one MI355X) | Newton it/s | ms per Newton iteration | host waits per Newton iteration | file |", "|---|---|---|---|---|"]
    for label, j, f in (("1024×1024×100 (BASELINE config 3's grid; `bench.py` default, with the CPU legs and the answer check)", bench, "bench.json"),
                        ("… second run of the same box, GPU only", bench2, "bench2.json"),
                        ("256×256×50 (BASELINE config 2), every kernel class timed", c2, "bench_256x256x50.json"),
                        ("256×256×50, SpMV class timed only", c2s, "bench_256x256x50_spmv.json"),
                        ("1024×128×100 (one rank's share of an 8-GPU run), SpMV class timed only", s8, "bench_1024x128x100.json")):
        results.append("| %s | **%.1f** | %.3f | %s | `%s/%s` |" % (label, j["value"], 1e3 / j["value"], rb(j), ev, f))
    dfile = os.path.join(ev, "bench_driver_form_steps20_warmup5.json")
    if os.path.exists(dfile):
        dj = json.load(open(dfile))
        results.append("| 1024×1024×100, **the driver's command** (`bench.py --gpus 1 --steps 20 --warmup 5`: later steps backtrack less than the first four) | **%.1f** | %.3f | %s | `%s/bench_driver_form_steps20_warmup5.json` |"
                       % (dj["value"], 1e3 / dj["value"], rb(dj), ev))
    roof = bench["roofline"]
    cb = bench.get("cpu_baseline", {})
    results.append("")
    results.append("Driver runs: 105.7 (`BENCH_r02.json`), 125.5 (`BENCH_r03.json`), 143.6 (`BENCH_r04.json`) Newton it/s; box-to-box spread of this pool ≈ ±4 %% (the column march "
                   "follows the box's clocks: %.2f ms here, 1.83–2.30 ms over the boxes of rounds 4–5)." % bench["kernels"]["jacobian"]["avg_ms"])
    results.append("`roofline` of that line: %s → **%.0f GB/s on the kernels' own byte model = %.3f of peak** (%.5f ms per launch, HIP events; "
                   "`frac` is a fraction of peak of bytes the timed kernels can be said to move — round 5; SURVEY's CSR count over the same time "
                   "is `achieved_algorithmic` %.0f GB/s); %.0f GB/s = %.3f of peak on the bytes they move (PMC %.3f GB per launch); the plain "
                   "`y = A x` kernel timed behind the region: %.5f ms = %.3f of peak; the CSR kernel inside a timed region of its own (7-point "
                   "grid, `extra_configs`): %s." % (
                       ("the fused product + column-solve launches of the timed region (`k_tri_solve_lds<…, SPMV, FDOT>`: t = M⁻¹(A v) from the CSR values at their computed places with the Krylov dots and the line search's F·(J p), F·(J s) — with `ls_fold` no product launch of the search's own is left in the region)" if roof.get("roofline_kernel_class") == "spmv_pc" else "the line search's product launches of the timed region (`k_spmv_cols_dots`)"),
                       roof["achieved"], roof["frac"], roof["avg_launch_ms"], roof.get("achieved_algorithmic", 0),
                       roof.get("achieved_physical", 0), roof.get("frac_physical", 0), (roof.get("traffic") or 0) / 1e9,
                       roof.get("plain_spmv_kernel", {}).get("avg_launch_ms", 0), roof.get("plain_spmv_kernel", {}).get("frac", 0),
                       "; ".join("%.0f GB/s = %.3f of peak over %d launches of %.3f ms" % (e["roofline"]["achieved"], e["roofline"]["frac"], e["roofline"].get("launches", 0), e["roofline"]["avg_launch_ms"])
                                 for e in (bench.get("extra_configs") or []) if "roofline" in e) or "not in this line"))
    if cb.get("value"):
        results.append("`cpu_baseline` (the oracle, \"port\"): %.3f Newton it/s on %d threads of an %s (GMRES(30)+ILU(0), the reference's "
                       "defaults; %.3f with the GPU's own BiCGStab + block-Jacobi; %.4f on one core — the reference itself is serial) → "
                       "%.0f× / %.0f×." % (cb["value"], cb["cores"], cb.get("cpu_model", "?"), cb["same_solver_as_gpu"]["value"],
                                           cb["value_one_core"], bench["value"] / cb["value"], bench["value"] / cb["same_solver_as_gpu"]["value"]))
        xp = cb.get("extrapolated_all_physical_cores")
        if xp:
            results[-1] += ("  The box has %d physical cores of which the process may use %d: `extrapolated_all_physical_cores` scales "
                            "linearly (an upper bound for a bandwidth-bound solve) to %.2f–%.2f Newton it/s → **%.0f×** against the "
                            "whole host (north-star: ≥ 10×)." % (xp["physical_cores"], cb["cores"], xp["from_threads_measured"],
                                                                 xp["from_one_core"], xp["gpu_over_cpu"]))
    results_txt = "\n".join(results)

    ex = bench.get("extra_configs") or []
    extra_rows = []
    for e in ex:
        extra_rows.append("%s: **%.1f** Newton it/s (%.3f ms per Newton iteration, set-up %.1f s)" % (
            e.get("config", "?"), e.get("value", 0.0), e.get("ms_per_newton_iteration", 0.0), e.get("setup_s", 0.0)))
    extra_txt = "; ".join(extra_rows) if extra_rows else "(none in this line)"
    per_newton = 0.0
    out = open(os.path.join(ROOT, "DESIGN.md.in")).read()
    th_line = "%.1f Newton it/s (BiCGStab + per-column 2×2 block-Jacobi; round 2: 95.8, round 3: 103.6, round 4: 122.9, round 5 before the factorisation left the iteration: 128.5; this evidence run's box — boxes differ by ± 3 %%, 144–151 were seen on the final sources, the A/Bs below are each on one box), column march %.2f ms" % (
        th["newton_its_per_s"], th["kernels"]["jacobian"]["avg_ms"])

    def kstats(fname):
        d = {}
        f = os.path.join(ev, fname)
        if os.path.exists(f):
            for r in csv.DictReader(open(f)):
                d[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) * 1e-6)
        return d

    def pmc(fname):
        f = os.path.join(ev, fname)
        if not os.path.exists(f):
            return {}
        return {short(k): v["hbm_bytes_per_launch"] for k, v in json.load(open(f))["kernels"].items()}

    def frac_rows(stats_file, pmc_file, cells, picks):
        st, mv = kstats(stats_file), pmc(pmc_file)
        out = []
        for kern, label in picks:
            if kern in st and kern in mv:
                calls, ms = st[kern]
                out.append("%s %.2f ms, %.0f B per cell, **%.2f** of peak" % (label, ms, mv[kern] / cells, mv[kern] / (ms * 1e-3) / 8e12))
        return "; ".join(out)

    th_pmc = frac_rows("rocprofv3_kernel_stats_th_512x512x80.csv", "pmc_th_512x512x80.json", 512 * 512 * 80, (
        ("k_th_spmv_tri<2, true, true>", "`k_th_spmv_tri<2, true, true>` (SpMV + block-Thomas solve with the pivots redone + dots + the line search's sum)"),
        ("k_th_spmv_tri<2, true, false>", "`k_th_spmv_tri<2, true>` (SpMV + block-Thomas solve + dots + the line search's sum)"),
        ("k_th_tri_factor<true, 1>", "block factorisation + first solve + norm in one sweep"), ("k_th_spmv_cols<true>", "index-free SpMV with the line search's sums"),
        ("k_th_spmv_cols<false>", "index-free SpMV"), ("k_th_tri_solve_dots<2>", "block-Thomas solve + dots"),
        ("k_th_tri_solve_dots<1>", "… with one dot"), ("k_th_tri_factor<false, 0>", "block factorisation alone"), ("k_bcgs_xr<true>", "`k_bcgs_xr`"),
        ("k_th_cols<2, 1>", "column march `k_th_cols<2>` (residual + Jacobian)"), ("k_th_cols<0, 1>", "`k_th_cols<0>` (residual)")))
    thermal_pmc = frac_rows("rocprofv3_kernel_stats_thermal_512.csv", "pmc_thermal_512.json", 512 ** 3, (
        ("k_box_lower_stream<8, false>", "lower stream"), ("k_box_upper_stream<8, false>", "upper stream"),
        ("k_box_gather_in", "into the box layout"), ("k_box_gather_out", "out of it"), ("k_spmv_pipe_mb<7>", "SpMV")))
    gs = frac_rows("rocprofv3_kernel_stats_thermal_512_short.csv", "pmc_thermal_512.json", 512 ** 3, (
        ("k_mdot", "`k_mdot`"), ("k_maxpy_neg", "`k_maxpy_neg`")))
    if gs:   # (their work per launch grows with the basis: averages of the short command the PMC passes ran, 20 iterations)
        thermal_pmc += "; Gram–Schmidt kernels, averaged over the 20 iterations of the PMC passes' own command: " + gs
    thermal = "see `%s/thermal_512.json`" % ev
    tf = os.path.join(ev, "thermal_512.json")
    if not os.path.exists(tf):
        tf = os.path.join(ROOT, "profiles", "r03_thermal_512.json")
    if os.path.exists(tf):
        t = json.load(open(tf))
        sw = {("%g" % x["ksp_rtol"]): x for x in t.get("rtol_sweep", [])}
        its = t["linear_iterations"]
        thermal = ("library set-up %.1f s in all (r02: 88, r03: 34.3; `mppgpu_setup` itself 25.6 → 3.0 s with the index analysis on "
                   "the device, `profiles/r04/setup_phase_timing_device.log`), %d GMRES(30) iterations in %.1f s at PETSc's default "
                   "`ksp_rtol` 1e-5 = %.1f ms per iteration (r03: 16.3, r04: 14.0; error vs the manufactured solution %.2f K), %d iterations / "
                   "%.1f s at 1e-8 (%.4f K); ILU(0) application %.2f ms in %d launches (r03: 6.59 ms in 270, r04: 4.60 in 4)" % (
                       t["setup_s"], its, t["step_s"], 1e3 * t["step_s"] / max(its, 1),
                       t["max_error_vs_manufactured_solution"], sw.get("1e-08", {}).get("linear_iterations", 0),
                       sw.get("1e-08", {}).get("step_s", 0), sw.get("1e-08", {}).get("max_error_vs_manufactured_solution", 0),
                       t["roofline"]["avg_apply_ms"], t["roofline"].get("launches_per_application", 0)))
    share = "%.0f Newton it/s on 1024×128 columns = %.1f %% of 8× the full-grid rate of the same box" % (s8["value"], 100.0 * s8["value"] / (8 * bench2["value"]))
    # bytes per cell and Newton iteration from the moved bytes and the launch counts of the profiled run
    tot = 0.0
    for k, (calls, ms) in stats.items():
        if k in moved:
            tot += moved[k] * calls
    newton_prof = None
    for k, (calls, ms) in stats.items():
        if k.startswith("k_tri_solve_lds<2, 5, 2, 4, 3"):
            newton_prof = calls
    bytes_txt = "%.0f" % (tot / N / newton_prof) if newton_prof else "?"
    v3d_pmc = frac_rows("rocprofv3_kernel_stats_vsfm3d_512x512x100.csv", "pmc_vsfm3d_512x512x100.json", 512 * 512 * 100, (
        ("k_assemble_cols<2, 3, 0, false, true>", "march (residual + Jacobian)"), ("k_flux_rows_ell<2, 4, true>", "lateral pass (residual + Jacobian)"),
        ("k_assemble_cols<0, 3, 0, false, true>", "march (residual)"), ("k_flux_rows_ell<0, 4, true>", "lateral pass (residual)"),
        ("k_spmv_pipe_mb<7>", "7-point CSR SpMV `k_spmv_pipe_mb<7>`"), ("k_tri_solve_lds<2, 4, 6, 4, 0, false>", "LDS column solve + 2 dots"),
        ("k_tri_factor", "column factorisation")))
    df = os.path.join(ev, "bench_driver_form_steps20_warmup5.json")   # `python bench.py --gpus 1 --steps 20 --warmup 5`, as the driver runs it
    driver_form = ("%.1f" % json.load(open(df))["value"]) if os.path.exists(df) else "(not in the evidence directory)"
    subs = {"V3D_PMC": v3d_pmc or "(PMC passes not in the evidence directory)", "KERNEL_TABLE": kernel_table, "LAWS_TABLE": laws_table, "RESULTS": results_txt,
            "READBACKS": "%.2f–%.2f" % (min(rb(s8), rb(c2s)), max(rb(bench), rb(bench2))),
            "ANSWER": "%.1e" % bench["answer_check"]["max_rel_err"], "SHARE": share, "THERMAL": thermal, "TH": th_line, "BYTES": bytes_txt,
            "TH_PMC": th_pmc or "(PMC passes not in the evidence directory)", "THERMAL_PMC": thermal_pmc or "(PMC passes not in the evidence directory)",
            "EXTRA": extra_txt, "DRIVER_FORM": driver_form}
    for k, v in subs.items():
        out = out.replace("@@%s@@" % k, v)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
