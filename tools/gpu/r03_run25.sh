#!/bin/bash
# round 3, run 25: boundary fluxes evaluated in the march's prologue (bc_inline): the VSFM tests (fused vs separate routes bit for bit),
# then A/B on the three grids
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run25; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== vsfm tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_more.py tests/test_gpu_r02.py -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -12 $O/pytest.log | cut -c1-200
if grep -q "core dump\|Memory access fault" $O/pytest.log; then echo "STOP"; exit 1; fi
B="--no-cpu-baseline --no-verify"
for f in 1 0 1 0; do
echo "== full bc_inline=$f"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 --opt bc_inline=$f > $O/full_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/full_$f.json'));print(j['value'],j['newton_iterations'],{k:v['avg_ms'] for k,v in j['kernels'].items()})"
done
for f in 1 0; do
echo "== s8 bc_inline=$f"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --steps 8 --warmup 2 --profile spmv --opt bc_inline=$f > $O/s8_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/s8_$f.json'));print(j['value'],j['newton_iterations'])"
echo "== c2 bc_inline=$f"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --steps 8 --warmup 2 --profile spmv --opt bc_inline=$f > $O/c2_$f.json 2>> $O/err.log || exit 1; python3 -c "import json;j=json.load(open('$O/c2_$f.json'));print(j['value'],j['newton_iterations'])"
done
tail -3 $O/err.log
