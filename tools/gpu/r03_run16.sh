#!/bin/bash
# round 3, run 16: M^{-1} A v as one kernel (CSR SpMV fused with the LDS column solve): the whole GPU suite, then A/B benches
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run16; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -12 $O/pytest.log
B="--no-cpu-baseline --no-verify"
for f in 0 1; do
echo "== full fused_spmv_pc=$f"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 --opt fused_spmv_pc=$f > $O/full_$f.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/full_$f.json'));print(j['value'],j['newton_iterations'],j['roofline']['frac'],{k:v['avg_ms'] for k,v in j['kernels'].items()})"
echo "== s8 fused_spmv_pc=$f"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --steps 8 --warmup 2 --profile spmv --opt fused_spmv_pc=$f > $O/s8_$f.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/s8_$f.json'));print(j['value'],j['newton_iterations'])"
echo "== c2 fused_spmv_pc=$f"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --steps 8 --warmup 2 --profile spmv --opt fused_spmv_pc=$f > $O/c2_$f.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/c2_$f.json'));print(j['value'],j['newton_iterations'])"
done
tail -3 $O/err.log
