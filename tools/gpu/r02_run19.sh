#!/bin/bash
# round 2, GPU run 19: box ILU solves chained by flags (one persistent launch per triangle): parity, give-up path, thermal 128^3 .. 512^3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run19; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest box tests"; timeout -k 10 300 python3 -m pytest tests/test_gpu_r02.py -q -m gpu -k "box" > $O/pytest_box.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $O/pytest_box.log; [ $rc -eq 0 ] || exit 1
echo "== pytest thermal, r02, more"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py tests/test_gpu_more.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log
for n in 128 256; do for ch in 1 0; do echo "== thermal $n ilu_chain=$ch"; timeout -k 10 600 python3 tools/thermal_bench.py --n $n --cpu-n 0 --opt ilu_chain=$ch --out $O/therm_${n}_chain$ch.json > $O/therm_${n}_chain$ch.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_${n}_chain$ch.json'));k=j['kernels']['pc_apply'];print(j['step_s'],j['linear_iterations'],k['total_ms']/k['launches'],j['roofline']['kernel'])"; done; done
echo "== thermal 512"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --out $O/therm_512_chain1.json > $O/therm_512_chain1.log 2>&1; echo "rc=$?"; tail -2 $O/therm_512_chain1.log | cut -c1-400
