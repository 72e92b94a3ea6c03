#!/bin/bash
# round 4, run 10: the whole GPU suite on the new defaults (streamed box solves, folded TH march), TH at config-5 size with and without the fold,
# the default bench line (extra configurations, CPU legs), and a line-search trace of the driver's configuration
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest -m gpu"; timeout -k 10 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && keep_failure $O/pytest_gpu.log
echo "== th 512x512x80, fold on / off"
for f in 1 0; do timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt th_fused_factor=$f --out $O/th_fold$f.json > $O/th_fold$f.log 2>&1; echo "rc=$?"; python3 -c "import json; d=json.load(open('$O/th_fold$f.json')); print('fold $f', d['newton_its_per_s'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"; done
echo "== bench (default line)"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench.json; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(json.dumps(d.get('extra_configs'))[:1500]); print(d['cpu_baseline'].get('extrapolated_all_physical_cores'))"
echo "== line-search trace, driver configuration"; timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-extra-configs --opt ls_trace=1 > $O/bench_trace.json 2> $O/bench_trace.err; echo "rc=$?"; cut -c1-160 $O/bench_trace.json
