#!/bin/bash
# round 5, run 4: ls_fold (line-search sums out of the one-iteration BiCGStab solve's kernels) and cols3d (3-D meshes of complete soil
# columns: column march + lateral pass): the suites that touch them, the driver's bench command with the fold off / on twice, and the
# 7-point grid with the general assembly / the column march + lateral pass
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest"; timeout -k 10 900 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_more.py tests/test_gpu_fuzz.py tests/test_gpu_r02.py -q -m gpu > $O/pytest_some.log 2>&1; rc=$?; echo "rc=$rc"; tail -12 $O/pytest_some.log
[ $rc -ne 0 ] && keep_failure $O/pytest_some.log
B="--no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5"
for r in off1 on1 off2 on2; do
  case $r in off*) F=0;; *) F=1;; esac
  timeout -k 10 400 python3 bench.py $B --opt ls_fold=$F > $O/bench_$r.json 2> $O/bench_$r.err; echo "$r rc=$?"
  python3 -c "import json; d=json.load(open('$O/bench_$r.json')); print(d['value'], d['newton_iterations'], d['linear_iterations'], d['roofline']['frac'], d['roofline'].get('roofline_kernel_class'), {k:(round(v['avg_ms'],4), v['launches']) for k,v in d['kernels'].items()})" || tail -5 $O/bench_$r.err
done
for c in 0 1; do
  timeout -k 10 400 python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 --opt cols3d=$c > $O/vsfm3d_cols3d_$c.json 2> $O/vsfm3d_cols3d_$c.err; echo "cols3d=$c rc=$?"
  python3 -c "import json; d=json.load(open('$O/vsfm3d_cols3d_$c.json')); print(d['value'], d['newton_iterations'], d['linear_iterations'], d['answer_check']['relative_drift'], {k:(round(v['avg_ms'],4), v['launches']) for k,v in d['kernels'].items()})" || tail -5 $O/vsfm3d_cols3d_$c.err
done
exit 0
