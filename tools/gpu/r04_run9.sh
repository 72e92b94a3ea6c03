#!/bin/bash
# round 4, run 9: streamed box solves reading r / writing z themselves (tests, 512^3); TH after un-parking the residual march; the
# residual-only VSFM march 2 / 3 / 4 levels deep; a line-search trace of the headline
source "$(dirname "$0")/_common.sh"
build_all
echo "== tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_th.py -q -m gpu -x -k "box_ilu or th_" > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then keep_failure $O/pytest.log; exit 1; fi
echo "== thermal 512^3, streams with fused io"
timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --opt ilu_box_stream=1 --then ilu_box_fused_io=0 --then ilu_box_fused_io=1 --out $O/thermal_512.json > $O/thermal_512.log 2>&1 || { echo "FAILED"; keep_failure $O/thermal_512.log; }
grep "^then" $O/thermal_512.log; python3 -c "import json; d=json.load(open('$O/thermal_512.json')); print('base', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'], d['kernels'])"
echo "== th 512x512x80"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"; python3 -c "import json; d=json.load(open('$O/th_512x512x80.json')); print(d['newton_its_per_s'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
echo "== headline, line-search trace"; timeout -k 10 600 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-extra-configs --opt ls_trace=1 > $O/bench_trace.json 2> $O/bench_trace.err; echo "rc=$?"; cut -c1-200 $O/bench_trace.json; grep -c "mppgpu ls" $O/bench_trace.err
echo "== residual march depth"
for d in 3 2 4; do
  if [ $d -ne 3 ]; then touch mpp_amd/csrc/kernels.hip; make -C mpp_amd/csrc -s CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DMPP_MARCH_DEPTH_RESIDUAL=$d" > $O/build_depth$d.log 2>&1 || { echo "build failed"; keep_failure $O/build_depth$d.log; break; }; fi
  timeout -k 10 300 python3 tools/kbench.py --only residual jacobian assemble_both --reps 30 --out $O/kbench_depth$d.json > $O/kbench_depth$d.log 2>&1; echo "depth $d rc=$?"; cat $O/kbench_depth$d.log
done
