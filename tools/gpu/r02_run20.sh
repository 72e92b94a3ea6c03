#!/bin/bash
# round 2, GPU run 20: numeric ILU(0) factorisation by boxes (no level analysis, no elimination program): parity, set-up time, thermal
# 128^3 .. 512^3, 3-D VSFM with GMRES + ILU(0)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run20; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest box tests"; timeout -k 10 300 python3 -m pytest tests/test_gpu_r02.py -q -m gpu -k "box" > $O/pytest_box.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $O/pytest_box.log; [ $rc -eq 0 ] || exit 1
echo "== pytest thermal, r02, more, parity"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py tests/test_gpu_more.py tests/test_gpu_parity.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
for n in 128 256; do echo "== thermal $n"; timeout -k 10 600 python3 tools/thermal_bench.py --n $n --cpu-n 0 --out $O/therm_${n}.json > $O/therm_${n}.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_${n}.json'));k=j['kernels'];print(j['step_s'],j['setup_s'],j['linear_iterations'],{m:(v['launches'],round(v['total_ms']/v['launches'],3)) for m,v in k.items()})"; done
echo "== vsfm3d gmres ilu"; timeout -k 10 600 python3 tools/vsfm3d_bench.py --nx 128 --ny 128 --nz 64 --ksp gmres --pc ilu --out $O/vsfm3d_128x128x64_gmres_ilu.json > $O/v3d_ilu.log 2>&1; echo "rc=$?"; tail -1 $O/v3d_ilu.log | cut -c1-400
echo "== thermal 512"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --out $O/therm_512.json > $O/therm_512.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_512.json'));k=j['kernels'];print(j['step_s'],j['setup_s'],j['linear_iterations'],{m:(v['launches'],round(v['total_ms']/v['launches'],3)) for m,v in k.items()})"
