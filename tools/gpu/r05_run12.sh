#!/bin/bash
# round 5, run 12: the march writes neither l nor 1 / pivot where the fused products are the solves (plane c = g alone); depth of the
# fused products' back-sweep rings (option tri_ring) now that they carry one stream less -- the fused / parity tests, then the headline
# on one box: library of the commit before (libmppgpu_old.so), this one with tri_ring 0, 1, 2; twice
source "$(dirname "$0")/_common.sh"
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_more.py -x -q -m gpu > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 0; }
cp mpp_amd/_build/libmppgpu.so mpp_amd/_build/libmppgpu_new.so
for v in old new0 new1 new2 old new0 new1 new2; do
  cp mpp_amd/_build/libmppgpu_${v:0:3}.so mpp_amd/_build/libmppgpu.so
  X=""; [ ${v:0:3} = new ] && X="--opt tri_ring=${v:3:1}"
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5 $X > $O/bench_$v.json 2>> $O/err.log || { tail -5 $O/err.log; cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so; exit 0; }
  python3 -c "import json; d=json.load(open('$O/bench_$v.json')); print('headline $v', round(d['value'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so
exit 0
