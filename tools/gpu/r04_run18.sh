#!/bin/bash
# round 4, run 18: incidence lists built on the device (setup_kernels.hip): the new tests, the suites of the general-assembly paths,
# then the set-up phases of thermal 512^3
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log; [ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 1; }
MPPGPU_TIMING=1 timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol 1e-2 --out $O/thermal_512.json > $O/thermal_512.log 2>&1; echo "rc=$?"
grep "mppgpu\]" $O/thermal_512.log | head -40
python3 -c "import json; d=json.load(open('$O/thermal_512.json')); print('setup_s', d['setup_s'], 'build_s', d['build_s'], 'step_s', d['step_s'], 'its', d['linear_iterations'], 'err', d['max_error_vs_manufactured_solution'])"
MPPGPU_TIMING=1 timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs > $O/bench.json 2> $O/bench.err; echo "rc=$?"
grep "mppgpu\]" $O/bench.err | head -14
python3 -c "import json; d=json.load(open('$O/bench.json')); print('setup_s', d['setup_s'], 'value', d['value'])"
