#!/bin/bash
# round 3, run 14: set-up times after the second pass over the set-up code (VSFM full grid, thermal 256^3), then config 4 at full
# size on one GPU: thermal 512^3, GMRES(30) + ILU(0) by boxes, with the set-up seconds, apply ms and the rtol sweep
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run14; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== quick tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_thermal.py tests/test_gpu_th.py -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
echo "== vsfm full set-up"; MPPGPU_TIMING=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/full.json 2> $O/full.err; echo "rc=$?"; grep "mppgpu\]" $O/full.err | head -6; python3 -c "import json;j=json.load(open('$O/full.json'));print(j['value'],j['setup_s'])"
echo "== thermal 256"; MPPGPU_TIMING=1 timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --out $O/thermal_256.json > $O/thermal_256.log 2>&1; echo "rc=$?"; grep "mppgpu\]" $O/thermal_256.log; python3 -c "import json;j=json.load(open('$O/thermal_256.json'));print('build',j['build_s'],'setup',j['setup_s'],'step',j['step_s'],'its',j['linear_iterations'])"
echo "== thermal 512"; MPPGPU_TIMING=1 timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol-sweep 1e-5,1e-8 --out $O/thermal_512.json > $O/thermal_512.log 2>&1; echo "rc=$?"; grep "mppgpu\]" $O/thermal_512.log; python3 -c "import json;j=json.load(open('$O/thermal_512.json'));print('build',j['build_s'],'setup',j['setup_s'],'step',j['step_s'],'its',j['linear_iterations'],j['roofline']['avg_apply_ms'],j.get('rtol_sweep'))"
