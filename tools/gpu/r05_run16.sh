#!/bin/bash
# round 5, run 16: the headline's Jacobian march asked for four waves per SIMD (128 registers, 108 B of scratch): 16384 waves are four whole
# rounds at four waves, 5.33 at three; library without / with the hint, twice each on one box
source "$(dirname "$0")/_common.sh"
cp mpp_amd/_build/libmppgpu.so mpp_amd/_build/libmppgpu_new.so
for v in old j4 old j4; do
  cp mpp_amd/_build/libmppgpu_$v.so mpp_amd/_build/libmppgpu.so
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5 > $O/bench_$v.json 2>> $O/err.log || { tail -5 $O/err.log; cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so; exit 0; }
  python3 -c "import json; d=json.load(open('$O/bench_$v.json')); print('headline $v', round(d['value'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so
exit 0
