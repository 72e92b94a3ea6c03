#!/bin/bash
# round 4, run 21: Gram-Schmidt kernels with 16-byte accesses (two rows per thread): the whole GPU suite, then thermal 512^3 kernel
# stats of the short command (comparable with profiles/r04_evidence/rocprofv3_kernel_stats_thermal_512_short.csv) and a full solve
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; [ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python3 tools/thermal_bench.py --n 512 --cpu-n 0 --no-events --rtol 1e-2 > $O/prof.log 2>&1; echo "rc=$?"
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
head -9 $O/prof/t_kernel_stats.csv | cut -c1-150
timeout -k 10 500 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --out $O/thermal_512.json > $O/thermal_512.log 2>&1; echo "rc=$?"
python3 -c "import json; d=json.load(open('$O/thermal_512.json')); print('apply', d['roofline']['avg_apply_ms'], 'its', d['linear_iterations'], 'step_s', d['step_s'], 'ms/it', 1e3*d['step_s']/d['linear_iterations'], {k:v['total_ms'] for k,v in d['kernels'].items()})"
