#!/bin/bash
# round 4, run 24: k_th_spmv_tri with the matrix rows requested two levels ahead: the bit-identity test, then config 5's size twice
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 600 python3 -m pytest tests/test_gpu_th.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/pytest.log; [ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 1; }
for r in 1 2; do
timeout -k 10 400 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_$r.json > $O/th_$r.log 2>&1; echo "rc=$?"
python3 -c "import json; d=json.load(open('$O/th_$r.json')); print(d['newton_its_per_s'], {k:(v.get('avg_ms'),v.get('launches')) for k,v in d['kernels'].items()})"
done
exit 0
