#!/bin/bash
# round 4, run 17: set-up after the no-initialisation host arrays, parallel prefix sums and the device-side permutations
# (connection end cells, thermal / TH connection geometry, thermal cell properties): the whole GPU suite, then the set-up phases
# of thermal 512^3 and of the headline grid
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log; [ $rc -ne 0 ] && { keep_failure $O/pytest_gpu.log; exit 1; }
MPPGPU_TIMING=1 timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --out $O/thermal_512.json > $O/thermal_512.log 2>&1; echo "rc=$?"
grep "mppgpu\]" $O/thermal_512.log | head -40
python3 -c "import json; d=json.load(open('$O/thermal_512.json')); print('setup_s', d['setup_s'], 'build_s', d['build_s'], 'step_s', d['step_s'], 'its', d['linear_iterations'], 'err', d['max_error_vs_manufactured_solution'])"
MPPGPU_TIMING=1 timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs > $O/bench.json 2> $O/bench.err; echo "rc=$?"
grep "mppgpu\]" $O/bench.err | head -30
python3 -c "import json; d=json.load(open('$O/bench.json')); print('setup_s', d['setup_s'], 'value', d['value'])"
