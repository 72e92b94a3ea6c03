#!/bin/bash
# round 3, run 22: s = r - alpha v inside the second product's kernel; the discarded-speculation test; A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run22; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== fused tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -q -m gpu > $O/pytest_fused.log 2>&1; rc=$?; echo "rc=$rc"; tail -25 $O/pytest_fused.log | cut -c1-220
if grep -q "core dump\|Memory access fault" $O/pytest_fused.log; then echo "STOP"; exit 1; fi
B="--no-cpu-baseline --no-verify"
for f in 1 0 1 0; do
echo "== full fused_waxpy=$f"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 --opt fused_waxpy=$f > $O/full_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/full_$f.json'));print(j['value'],j['newton_iterations'],{k:v['avg_ms'] for k,v in j['kernels'].items()})"
done
for f in 1 0; do
echo "== s8 fused_waxpy=$f"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --steps 8 --warmup 2 --profile spmv --opt fused_waxpy=$f > $O/s8_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/s8_$f.json'));print(j['value'],j['newton_iterations'])"
done
tail -3 $O/err.log
