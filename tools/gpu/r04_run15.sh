#!/bin/bash
# round 4, run 15: the TH index-free SpMV with its matrix pieces staged through LDS (16-byte loads in address order instead of a
# 48-byte stride per lane): the TH tests, then config 5's size
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 900 python3 -m pytest tests/test_gpu_th.py tests/test_gpu_fuzz.py -x -q > $O/pytest_th.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_th.log; [ $rc -ne 0 ] && { keep_failure $O/pytest_th.log; exit 1; }
timeout -k 10 400 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"
python3 -c "import json; d=json.load(open('$O/th_512x512x80.json')); print(d['newton_its_per_s'], {k:(v.get('avg_ms'),v.get('launches')) for k,v in d['kernels'].items()})"
timeout -k 10 400 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt th_spmv_tri=0 --out $O/th_512x512x80_csr.json > $O/th_csr.log 2>&1; echo "rc=$?"
python3 -c "import json; d=json.load(open('$O/th_512x512x80_csr.json')); print(d['newton_its_per_s'], {k:(v.get('avg_ms'),v.get('launches')) for k,v in d['kernels'].items()})"
