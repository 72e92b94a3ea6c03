#!/bin/bash
# round 2, GPU run 27: the right-hand side into the box layout through an LDS image of the box (coalesced reads): box tests, thermal timings
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run27; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest box/thermal"; timeout -k 10 600 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
for n in 128 256 512; do echo "== thermal $n"; timeout -k 10 900 python3 tools/thermal_bench.py --n $n --cpu-n 0 --out $O/therm_${n}.json > $O/therm_${n}.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_${n}.json'));k=j['kernels'];print(j['step_s'],j['linear_iterations'],{m:(v['launches'],round(v['total_ms']/v['launches'],3)) for m,v in k.items()})"; done
