#!/bin/bash
# round 3, run 17: the fused SpMV + column solve kernel after the top-row slot fix: its own test first, then the suite, then A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run17; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== the new kernel's test"; timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -q -m gpu -x > $O/pytest_fused.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest_fused.log
if [ $rc -ne 0 ] || grep -q "core dump\|Memory access fault" $O/pytest_fused.log; then echo "STOP: the fused kernel's test failed"; exit 1; fi
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 $O/pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
B="--no-cpu-baseline --no-verify"
echo "== full fused"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 > $O/full_1.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/full_1.json'));print(j['value'],j['newton_iterations'],j['roofline']['frac'],j['roofline'].get('plain_spmv_kernel'),{k:v['avg_ms'] for k,v in j['kernels'].items()})"
echo "== s8 fused"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --steps 8 --warmup 2 --profile spmv > $O/s8_1.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/s8_1.json'));print(j['value'],j['newton_iterations'])"
echo "== c2 fused"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --steps 8 --warmup 2 --profile spmv > $O/c2_1.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/c2_1.json'));print(j['value'],j['newton_iterations'])"
echo "== full unfused (same box)"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 --opt fused_spmv_pc=0 > $O/full_0.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/full_0.json'));print(j['value'],j['newton_iterations'])"
tail -3 $O/err.log
