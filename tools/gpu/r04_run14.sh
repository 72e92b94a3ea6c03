#!/bin/bash
# round 4, run 14: box length of the streamed solves at 256^3 and 512^3 (padding of the skewed layout against the per-box overheads)
source "$(dirname "$0")/_common.sh"
build_all
for n in 256 512; do
  for bl in 32 64 128; do
    [ $n -eq 512 ] && [ $bl -eq 32 ] && continue
    timeout -k 10 500 python3 tools/thermal_bench.py --n $n --cpu-n 0 --opt ilu_box_len=$bl --out $O/thermal_${n}_len$bl.json > $O/thermal_${n}_len$bl.log 2>&1 || { echo "FAILED"; keep_failure $O/thermal_${n}_len$bl.log; }
    python3 -c "import json; d=json.load(open('$O/thermal_${n}_len$bl.json')); print('n $n len $bl', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'], d['kernels']['pc_setup'])"
  done
done
