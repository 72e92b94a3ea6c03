#!/bin/bash
# round 3, run 15: TH with the linear porosity model and mixed saturation functions (tests), set-up phases of thermal 256^3 and of
# the VSFM full grid with finer timers
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run15; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_th.py tests/test_gpu_parity.py tests/test_gpu_thermal.py tests/test_gpu_more.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
echo "== thermal 256"; MPPGPU_TIMING=1 timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --out $O/thermal_256.json > $O/thermal_256.log 2>&1; echo "rc=$?"; grep "mppgpu\]" $O/thermal_256.log; python3 -c "import json;j=json.load(open('$O/thermal_256.json'));print('build',j['build_s'],'setup',j['setup_s'],'step',j['step_s'],'its',j['linear_iterations'])"
echo "== vsfm full set-up"; MPPGPU_TIMING=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/full.json 2> $O/full.err; echo "rc=$?"; grep "mppgpu\]" $O/full.err | head -12; python3 -c "import json;j=json.load(open('$O/full.json'));print(j['value'],j['setup_s'])"
