#!/bin/bash
# round 2, GPU run 1: parity after the refactoring + LDS column solve, micro-benchmarks, SQ counters of the fused march,
# a bench line, and ONE rocprofv3 kernel-trace of the hipGraph replay of the ILU(0) level solves (log kept).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run1; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest r02"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py -x -q -m gpu > $O/pytest_r02.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_r02.log
echo "== pytest all"; timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_r02.py > $O/pytest_all.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_all.log
echo "== kbench"; timeout -k 10 600 python3 tools/kbench.py --reps 20 --out $O/kbench.json > $O/kbench.log 2>&1; echo "rc=$?"; cat $O/kbench.log
echo "== bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cat $O/bench.json; tail -5 $O/bench.err
echo "== pmc sq"; timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq1 -o sq1 --output-format csv -- python3 tools/kbench.py --reps 3 --only assemble_both residual pc_apply_dots:0 pc_apply_dots:1 spmv:20 > $O/pmc_sq1.log 2>&1; echo "rc=$?"; tail -3 $O/pmc_sq1.log
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/pmc_sq2 -o sq2 --output-format csv -- python3 tools/kbench.py --reps 3 --only assemble_both residual pc_apply_dots:0 pc_apply_dots:1 spmv:20 > $O/pmc_sq2.log 2>&1; echo "rc=$?"; tail -3 $O/pmc_sq2.log
echo "== rocprofv3 kernel-trace over the hipGraph replay (thermal 24^3, ilu_graph=1, ilu_chain=0)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_graph -o g --output-format csv -- python3 tools/thermal_bench.py --n 24 --cpu-n 0 --opt ilu_chain=0 --opt ilu_graph=1 > $O/prof_graph.log 2>&1; echo "rc=$?"; tail -30 $O/prof_graph.log
ls $O
