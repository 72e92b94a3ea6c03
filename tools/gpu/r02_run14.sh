#!/bin/bash
# round 2, GPU run 14: box-wavefront ILU(0) parity after ordering the ELL rows by the caller's columns; thermal 512^3 with it
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run14; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest thermal, r02, more"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py tests/test_gpu_more.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log
echo "== thermal 512"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --out $O/therm_512_box.json > $O/therm_512_box.log 2>&1; echo "rc=$?"; tail -2 $O/therm_512_box.log | cut -c1-400
