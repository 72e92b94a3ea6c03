#!/bin/bash
# round 5, run 11: the column LU's back substitution in product form (plane c = up / pivot; the fused products park d / pivot and read g
# alone; the march leaves its forward-eliminated residual scaled) -- the whole GPU suite, then the headline A/B on one box against the
# library built from the commit before (mpp_amd/_build/libmppgpu_old.so), twice each
source "$(dirname "$0")/_common.sh"
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 0; }
cp mpp_amd/_build/libmppgpu.so mpp_amd/_build/libmppgpu_new.so
for v in old new old new; do
  cp mpp_amd/_build/libmppgpu_$v.so mpp_amd/_build/libmppgpu.so
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5 > $O/bench_$v.json 2>> $O/err.log || { tail -5 $O/err.log; cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so; exit 0; }
  python3 -c "import json; d=json.load(open('$O/bench_$v.json')); print('headline $v', round(d['value'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so
exit 0
