#!/bin/bash
# round 3, run 7: the whole GPU suite; then the thermal 128^3 step under rocprofv3 --kernel-trace with the level solves replayed
# from hipGraphs of <= 128 kernel nodes (the default path; no profiler detection any more): boxes, and level-scheduled (ilu_box=0)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run7; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log
echo "== thermal 128 plain"; timeout -k 10 300 python3 tools/thermal_bench.py --n 128 --cpu-n 0 --out $O/thermal_128.json > $O/thermal_128.log 2>&1; echo "rc=$?"; tail -2 $O/thermal_128.log | cut -c1-300
echo "== thermal 128 levels, under rocprofv3"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_levels -o t --output-format csv -- python3 tools/thermal_bench.py --n 128 --cpu-n 0 --opt ilu_box=0 --out $O/thermal_128_levels_prof.json > $O/rocprofv3_graph_replay_128_levels.log 2>&1; echo "rc=$?"; tail -3 $O/rocprofv3_graph_replay_128_levels.log | cut -c1-300
echo "== thermal 128 boxes, under rocprofv3"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_boxes -o t --output-format csv -- python3 tools/thermal_bench.py --n 128 --cpu-n 0 --out $O/thermal_128_boxes_prof.json > $O/rocprofv3_graph_replay_128_boxes.log 2>&1; echo "rc=$?"; tail -3 $O/rocprofv3_graph_replay_128_boxes.log | cut -c1-300
rm -f $O/prof_*/*kernel_trace.csv
ls $O $O/prof_levels | head -30
