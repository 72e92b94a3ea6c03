#!/bin/bash
# round 3, run 20: the start of a BiCGStab solve as one kernel (back substitution + SpMV + column solve): its test, then A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run20; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== the kernel's test"; timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -q -m gpu -x > $O/pytest_fused.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $O/pytest_fused.log
if [ $rc -ne 0 ] || grep -q "core dump\|Memory access fault" $O/pytest_fused.log; then echo "STOP"; exit 1; fi
echo "== vsfm tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_more.py tests/test_gpu_r02.py -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -4 $O/pytest.log
B="--no-cpu-baseline --no-verify"
for f in 1 0 1 0; do
echo "== full fused_first=$f"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 --opt fused_first=$f > $O/full_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/full_$f.json'));print(j['value'],j['newton_iterations'],{k:v['avg_ms'] for k,v in j['kernels'].items()})"
done
for f in 1 0; do
echo "== s8 fused_first=$f"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --steps 8 --warmup 2 --profile spmv --opt fused_first=$f > $O/s8_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/s8_$f.json'));print(j['value'],j['newton_iterations'])"
echo "== c2 fused_first=$f"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --steps 8 --warmup 2 --profile spmv --opt fused_first=$f > $O/c2_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/c2_$f.json'));print(j['value'],j['newton_iterations'])"
done
tail -3 $O/err.log
