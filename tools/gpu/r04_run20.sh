#!/bin/bash
# round 4, run 20: LDS bank swizzle in the box image (k_box_gather_in) and the way back through the image (k_box_gather_out,
# option ilu_box_img_out): box tests, then A/B at 512^3 and 256^3 on one box with kernel stats
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; [ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 1; }
for n in 512 256; do
  timeout -k 10 500 python3 tools/thermal_bench.py --n $n --cpu-n 0 --rtol 1e-3 --then ilu_box_img_out=1 --then ilu_box_img_out=0 --then ilu_box_img_out=1 --out $O/thermal_${n}_img_ab.json > $O/thermal_${n}_img_ab.log 2>&1; echo "rc=$?"
  python3 -c "
import json; d=json.load(open('$O/thermal_${n}_img_ab.json'))
print('n $n first', d['roofline']['avg_apply_ms'], d['linear_iterations'], d['step_s'])
for t in d.get('then', []): print(t)
"
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python3 tools/thermal_bench.py --n 512 --cpu-n 0 --no-events --rtol 1e-2 --opt ilu_box_img_out=1 > $O/prof.log 2>&1; echo "rc=$?"
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
grep "gather" $O/prof/t_kernel_stats.csv | cut -c1-200
