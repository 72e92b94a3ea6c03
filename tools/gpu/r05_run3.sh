#!/bin/bash
# round 5, run 3: ls_fold -- the line search's F.(J Y) and ||Y|| out of the one-iteration BiCGStab solve's own kernels: tests of the fused
# routes, then the driver's bench command with the fold off / on / off / on on one box
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest fused"; timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_fused.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 $O/pytest_fused.log
[ $rc -ne 0 ] && { keep_failure $O/pytest_fused.log; exit 0; }
B="--no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5"
for r in off1 on1 off2 on2; do
  case $r in off*) F=0;; *) F=1;; esac
  timeout -k 10 400 python3 bench.py $B --opt ls_fold=$F > $O/bench_$r.json 2> $O/bench_$r.err; echo "$r rc=$?"
  python3 -c "import json; d=json.load(open('$O/bench_$r.json')); print(d['value'], d['newton_iterations'], d['linear_iterations'], d['roofline']['frac'], d['roofline'].get('roofline_kernel_class'), {k:(round(v['avg_ms'],4), v['launches']) for k,v in d['kernels'].items()})" || tail -5 $O/bench_$r.err
done
exit 0
