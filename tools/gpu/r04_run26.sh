#!/bin/bash
# round 4, run 26: the driver's own bench command (--gpus 1 --steps 20 --warmup 5) with its wall time, as a rehearsal of BENCH_r04
source "$(dirname "$0")/_common.sh"
build_all
t0=$(date +%s.%N)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "rc=$?"
t1=$(date +%s.%N)
python3 -c "
import json
d=json.load(open('$O/bench_driver_form.json'))
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'newton', d['newton_iterations'], 'wall_s', round($t1-$t0,1), 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'] is not None, 'extra', [(e['config'][:12], round(e['value'],1)) for e in d['extra_configs']], 'cpu x', d['cpu_baseline']['extrapolated_all_physical_cores']['gpu_over_cpu'])"
exit 0
