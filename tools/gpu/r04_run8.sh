#!/bin/bash
# round 4, run 8: coupled TH with the shared IFC-67 evaluation and the two-wave march: tests, then the config-5 size; box-stream tuning at 512^3
source "$(dirname "$0")/_common.sh"
build_all
echo "== TH tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_th.py -q -m gpu -x > $O/pytest_th.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest_th.log
if [ $rc -ne 0 ]; then keep_failure $O/pytest_th.log; fi
echo "== th 512x512x80"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"; tail -1 $O/th.log | cut -c1-900
echo "== thermal 512^3 stream tuning"
timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --opt ilu_box_stream=1 --then ilu_box_workers=1024 --then ilu_box_workers=4096 --out $O/thermal_512_len64.json > $O/thermal_512_len64.log 2>&1 || { echo "FAILED"; keep_failure $O/thermal_512_len64.log; }
grep "^then" $O/thermal_512_len64.log; python3 -c "import json; d=json.load(open('$O/thermal_512_len64.json')); print('base', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'])"
timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --opt ilu_box_stream=1 --opt ilu_box_len=128 --then ilu_box_stream=0 --out $O/thermal_512_len128.json > $O/thermal_512_len128.log 2>&1 || { echo "FAILED"; keep_failure $O/thermal_512_len128.log; }
grep "^then" $O/thermal_512_len128.log; python3 -c "import json; d=json.load(open('$O/thermal_512_len128.json')); print('base', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'])"
