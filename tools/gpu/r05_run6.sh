#!/bin/bash
# round 5, run 6: cols3d with the compacted lateral lists (K = 4) and the coupled system's folded line-search sums: their tests, the
# 7-point grid again (cols3d 0 / 1) with kernel stats, config 5's size with ls_fold off / on, config 4 on four in-process ranks at 512^3
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest"; timeout -k 10 900 python3 -m pytest tests/test_gpu_more.py tests/test_gpu_th.py tests/test_gpu_fuzz.py tests/test_gpu_fused.py -q -m gpu > $O/pytest_some.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $O/pytest_some.log
[ $rc -ne 0 ] && keep_failure $O/pytest_some.log
for c in 0 1; do
  timeout -k 10 400 python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 --opt cols3d=$c > $O/vsfm3d_cols3d_$c.json 2> $O/vsfm3d_cols3d_$c.err; echo "cols3d=$c rc=$?"
  python3 -c "import json; d=json.load(open('$O/vsfm3d_cols3d_$c.json')); print(d['value'], d['newton_iterations'], d['linear_iterations'], d['answer_check']['relative_drift'], {k:(round(v['avg_ms'],4), v['launches']) for k,v in d['kernels'].items()})" || tail -5 $O/vsfm3d_cols3d_$c.err
done
echo "== rocprofv3 kernel stats, vsfm3d 512x512x100, cols3d"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -o v3d --output-format csv -- python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 --opt ilu_graph=0 > $O/prof.log 2>&1; echo "rc=$?"
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
grep "k_flux_rows\|k_assemble_cols\|k_auxvars" $O/prof/v3d_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
for f in 0 1 0 1; do
  timeout -k 10 400 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt ls_fold=$f > $O/th_fold_$f.log 2>&1; echo "th ls_fold=$f rc=$?"
  tail -1 $O/th_fold_$f.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['newton_its_per_s'], d['newton_iterations'], d['linear_iterations'], {k:(round(v['avg_ms'],4), v['launches']) for k,v in d['kernels'].items()})"
done
echo "== config 4 on four in-process ranks, 512^3"
timeout -k 10 900 python3 tools/thermal_ranks_bench.py --n 512 --out $O/thermal_ranks_512.json > $O/thermal_ranks_512.log 2>&1; echo "rc=$?"; tail -1 $O/thermal_ranks_512.log | cut -c1-2500
exit 0
