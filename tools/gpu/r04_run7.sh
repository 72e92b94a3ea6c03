#!/bin/bash
# round 4, run 7: box ILU(0) solves as streams of pencils -- bit-identity tests, then A/B at 256^3 and 512^3
source "$(dirname "$0")/_common.sh"
build_all
echo "== tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_more.py -q -m gpu -x -k "box_ilu or four_in_process or same_pair" > $O/pytest_box.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest_box.log
if [ $rc -ne 0 ]; then keep_failure $O/pytest_box.log; exit 1; fi
echo "== thermal 256^3"; timeout -k 10 300 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --then ilu_box_stream=1 --then ilu_box_stream=1,ilu_box_workers=512 --then ilu_box_stream=1,ilu_box_workers=2048 --then ilu_box_stream=0 --out $O/thermal_256.json > $O/thermal_256.log 2>&1 || { echo "FAILED"; keep_failure $O/thermal_256.log; exit 1; }
grep "^then" $O/thermal_256.log; python3 -c "import json; d=json.load(open('$O/thermal_256.json')); print('base', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'])"
echo "== thermal 512^3"; timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --then ilu_box_stream=1 --then ilu_box_stream=1,ilu_box_workers=512 --then ilu_box_stream=1,ilu_box_workers=2048 --then ilu_box_stream=0 --out $O/thermal_512.json > $O/thermal_512.log 2>&1 || { echo "FAILED"; keep_failure $O/thermal_512.log; exit 1; }
grep "^then" $O/thermal_512.log; python3 -c "import json; d=json.load(open('$O/thermal_512.json')); print('base', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'])"
