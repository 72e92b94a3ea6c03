#!/bin/bash
# round 3, run 12: where the set-up time of the thermal 256^3 problem goes (library phases with MPPGPU_TIMING, Python side with cProfile)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run12; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
MPPGPU_TIMING=1 timeout -k 10 600 python3 -c "
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
sys.argv = ['thermal_bench.py', '--n', '256', '--cpu-n', '0', '--no-events']
import runpy
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path('tools/thermal_bench.py', run_name='__main__')
finally:
    pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
" > $O/setup_256.log 2>&1; echo "rc=$?"; grep "mppgpu\]" $O/setup_256.log | head -40; grep -A50 "cumulative" $O/setup_256.log | cut -c1-150 | head -60
