#!/bin/bash
# round 3, run 19: the fused SpMV + column solve kernel with coalesced matrix segments transposed through LDS: its test, then the bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run19; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== the kernel's test"; timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py -q -m gpu -x > $O/pytest_fused.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest_fused.log
if [ $rc -ne 0 ] || grep -q "core dump\|Memory access fault" $O/pytest_fused.log; then echo "STOP"; exit 1; fi
B="--no-cpu-baseline --no-verify"
for i in 1 2; do
echo "== full"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 > $O/full_$i.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/full_$i.json'));print(j['value'],j['newton_iterations'],{k:v['avg_ms'] for k,v in j['kernels'].items()})"
done
echo "== s8"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --steps 8 --warmup 2 --profile spmv > $O/s8.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/s8.json'));print(j['value'],j['newton_iterations'])"
echo "== c2"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --steps 8 --warmup 2 --profile spmv > $O/c2.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/c2.json'));print(j['value'],j['newton_iterations'])"
tail -3 $O/err.log
