#!/bin/bash
# round 4, run 22: the streamed box solves against one launch per box level at sizes the test suite does not reach (384^3, 320^3):
# same iteration count and the same error against the manufactured solution to the last digit = the same bits
source "$(dirname "$0")/_common.sh"
build_all
for n in 384 320; do
  timeout -k 10 500 python3 tools/thermal_bench.py --n $n --cpu-n 0 --then ilu_box_stream=0 --then ilu_box_img_out=0 --out $O/thermal_${n}_ab.json > $O/thermal_${n}_ab.log 2>&1; echo "rc=$?"
  python3 -c "
import json; d=json.load(open('$O/thermal_${n}_ab.json'))
print('n $n streams', d['roofline']['avg_apply_ms'], d['linear_iterations'], repr(d['max_error_vs_manufactured_solution']), d['step_s'])
for t in d.get('then', []): print(t['options'], t['pc_apply_avg_ms'], t['linear_iterations'], repr(t['max_error_vs_manufactured_solution']), t['step_s'], t['chain_fallbacks'])
"
done
