#!/bin/bash
# round 2, GPU run 2: why does the heterogeneous 1024x1024x100 answer differ from per-column oracle solves? + SQ counters + graph trace
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run2; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== diag default"; timeout -k 10 400 python3 tools/gpu/diag_answer.py > $O/diag_default.log 2>&1; echo "rc=$?"; cat $O/diag_default.log
echo "== diag tri_lds=0"; timeout -k 10 400 python3 tools/gpu/diag_answer.py tri_lds=0 > $O/diag_trilds0.log 2>&1; echo "rc=$?"; cat $O/diag_trilds0.log
echo "== diag tight"; timeout -k 10 400 python3 tools/gpu/diag_answer.py snes_rtol=1e-13 snes_stol=1e-14 > $O/diag_tight.log 2>&1; echo "rc=$?"; cat $O/diag_tight.log
echo "== diag unfused"; timeout -k 10 400 python3 tools/gpu/diag_answer.py tri_lds=0 fused_rows=0 fused_ls_dots=0 ls_ahead=0 fused_assembly=0 wide_xr=0 spmv_variant=17 > $O/diag_unfused.log 2>&1; echo "rc=$?"; cat $O/diag_unfused.log
echo "== pmc sq"; timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq1 -o sq1 --output-format csv -- python3 tools/kbench.py --reps 3 --only assemble_both residual pc_apply_dots:0 pc_apply_dots:1 spmv:20 > $O/pmc_sq1.log 2>&1; echo "rc=$?"; tail -3 $O/pmc_sq1.log
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/pmc_sq2 -o sq2 --output-format csv -- python3 tools/kbench.py --reps 3 --only assemble_both residual pc_apply_dots:0 pc_apply_dots:1 spmv:20 > $O/pmc_sq2.log 2>&1; echo "rc=$?"; tail -3 $O/pmc_sq2.log
echo "== rocprofv3 kernel-trace over the hipGraph replay (thermal 24^3, ilu_graph=1, ilu_chain=0)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_graph -o g --output-format csv -- python3 tools/thermal_bench.py --n 24 --cpu-n 0 --opt ilu_chain=0 --opt ilu_graph=1 > $O/prof_graph.log 2>&1; echo "rc=$?"; tail -30 $O/prof_graph.log
ls $O
