#!/bin/bash
# round 4, run 16: same-box A/B of the LDS staging in the TH factorisation and SpMV (option th_stage)
source "$(dirname "$0")/_common.sh"
build_all
for st in 1 0 1 0; do
  timeout -k 10 400 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt th_stage=$st --out $O/th_stage$st.json > $O/th_stage$st.log 2>&1; echo "rc=$?"
  python3 -c "import json; d=json.load(open('$O/th_stage$st.json')); print('th_stage=$st', d['newton_its_per_s'], {k:(v.get('avg_ms'),v.get('launches')) for k,v in d['kernels'].items()})"
done
