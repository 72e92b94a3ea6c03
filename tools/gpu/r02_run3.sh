#!/bin/bash
# round 2, GPU run 3: parity of the pipelined march (+ soil classes, level geometry, analytic row slots), new soil models, line
# searches, standalone driver; micro-benchmarks; a bench line; rocprofv3 kernel-trace over the hipGraph replay at 128^3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run3; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest r02"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py -q -m gpu > $O/pytest_r02.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_r02.log
echo "== pytest all"; timeout -k 10 900 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_r02.py > $O/pytest_all.log 2>&1; echo "rc=$?"; tail -8 $O/pytest_all.log
echo "== kbench"; timeout -k 10 600 python3 tools/kbench.py --reps 20 --out $O/kbench.json --only spmv:20 pc_apply_dots:0 pc_apply_dots:1 residual jacobian assemble_both > $O/kbench.log 2>&1; echo "rc=$?"; cat $O/kbench.log
echo "== kbench, byte savers off"; timeout -k 10 600 python3 tools/kbench.py --reps 20 --opt soil_classes=0 --opt level_geometry=0 --out $O/kbench_nosave.json --only residual jacobian assemble_both > $O/kbench_nosave.log 2>&1; echo "rc=$?"; cat $O/kbench_nosave.log
echo "== bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cat $O/bench.json; tail -5 $O/bench.err
echo "== rocprofv3 kernel-trace over the hipGraph replay (thermal 128^3, ilu_graph=1, ilu_chain=0)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_graph128 -o g --output-format csv -- python3 tools/thermal_bench.py --n 128 --cpu-n 0 --opt ilu_chain=0 --opt ilu_graph=1 > $O/prof_graph128.log 2>&1; echo "rc=$?"; tail -12 $O/prof_graph128.log | cut -c1-300
ls $O
