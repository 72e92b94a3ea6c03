#!/bin/bash
# round 4, run 12: padded gather passes around the streamed box solves; the 512^3 test; ring depth of the stream kernels 8 / 4 / 12
source "$(dirname "$0")/_common.sh"
build_all
echo "== tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py tests/test_gpu_more.py tests/test_gpu_th.py -q -m gpu -x -k "box_ilu or config_4 or ranks or restart or th_" > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && keep_failure $O/pytest.log
echo "== th 512x512x80"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"; python3 -c "import json; d=json.load(open('$O/th_512x512x80.json')); print(d['newton_its_per_s'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
for dp in 8 4 12; do
  if [ $dp -ne 8 ]; then touch mpp_amd/csrc/kernels.hip; make -C mpp_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DMPP_BOX_STREAM_DP=$dp" > $O/build_dp$dp.log 2>&1 || { echo "build failed"; keep_failure $O/build_dp$dp.log; break; }; fi
  W=2048; [ $dp -eq 4 ] && W=3072; [ $dp -eq 12 ] && W=1024
  echo "== thermal 512^3, ring depth $dp, $W workers"
  timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --opt ilu_box_workers=$W --out $O/thermal_512_dp$dp.json > $O/thermal_512_dp$dp.log 2>&1 || { echo "FAILED"; keep_failure $O/thermal_512_dp$dp.log; }
  python3 -c "import json; d=json.load(open('$O/thermal_512_dp$dp.json')); print('dp $dp', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'], d['setup_s'], d['build_s'])"
done
