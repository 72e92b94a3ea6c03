#!/bin/bash
# round 2, GPU run 15: box-wavefront ILU(0) solves with 16-byte loads (row pairs) and faces in LDS: parity, thermal 128^3 .. 512^3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run15; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest thermal, r02, more"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py tests/test_gpu_more.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log
for n in 128 256; do echo "== thermal $n"; timeout -k 10 600 python3 tools/thermal_bench.py --n $n --cpu-n 0 --out $O/therm_${n}_box.json > $O/therm_${n}_box.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_${n}_box.json'));k=j['kernels']['pc_apply'];print(j['step_s'],j['linear_iterations'],k['total_ms']/k['launches'])"; done
echo "== thermal 256 len 16"; timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --opt ilu_box_len=16 --out $O/therm_256_len16.json > $O/therm_256_len16.log 2>&1; python3 -c "import json;j=json.load(open('$O/therm_256_len16.json'));k=j['kernels']['pc_apply'];print(j['step_s'],j['linear_iterations'],k['total_ms']/k['launches'])"
echo "== thermal 256 len 64"; timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --opt ilu_box_len=64 --out $O/therm_256_len64.json > $O/therm_256_len64.log 2>&1; python3 -c "import json;j=json.load(open('$O/therm_256_len64.json'));k=j['kernels']['pc_apply'];print(j['step_s'],j['linear_iterations'],k['total_ms']/k['launches'])"
echo "== thermal 512"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --out $O/therm_512_box.json > $O/therm_512_box.log 2>&1; echo "rc=$?"; tail -2 $O/therm_512_box.log | cut -c1-400
