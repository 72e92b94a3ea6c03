#!/bin/bash
# round 5, run 8: cols3d with whole-row writes (the march's tridiagonal in three coalesced arrays, the lateral pass writes the CSR row):
# tests of the 3-D routes, then the 7-point grid with cols3d_rows 0 / 1 / 0 / 1 and kernel stats
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest"; timeout -k 10 900 python3 -m pytest tests/test_gpu_more.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_r02.py -q -m gpu > $O/pytest_some.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 $O/pytest_some.log
[ $rc -ne 0 ] && keep_failure $O/pytest_some.log
for c in 0 1 0 1; do
  timeout -k 10 400 python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 --opt cols3d_rows=$c > $O/vsfm3d_rows_$c.json 2> $O/vsfm3d_rows_$c.err; echo "cols3d_rows=$c rc=$?"
  python3 -c "import json; d=json.load(open('$O/vsfm3d_rows_$c.json')); print(d['value'], d['newton_iterations'], d['linear_iterations'], d['answer_check']['relative_drift'], {k:(round(v['avg_ms'],4), v['launches']) for k,v in d['kernels'].items()})" || tail -5 $O/vsfm3d_rows_$c.err
done
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -o v3d --output-format csv -- python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 --opt ilu_graph=0 > $O/prof.log 2>&1; echo "rc=$?"
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
grep "k_flux_rows\|k_assemble_cols" $O/prof/v3d_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
exit 0
