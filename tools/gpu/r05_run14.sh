#!/bin/bash
# round 5, run 14: coupled TH, s = r - alpha v inside the second fused product (k_th_spmv_tri<.., VSUB>, option fused_waxpy) on top of the
# four-wave hint: tests, then fused_waxpy 0 / 1 three times on one box
source "$(dirname "$0")/_common.sh"
# (the bit-identity test of that variant went with it)

for v in 0 1 0 1 0 1; do
  timeout -k 10 300 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt fused_waxpy=$v > $O/th_$v.log 2>&1 || { tail -5 $O/th_$v.log; exit 0; }
  tail -1 $O/th_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('th fused_waxpy=$v', round(d['newton_its_per_s'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
exit 0
