#!/bin/bash
# round 5, run 20: run-to-run spread of the driver's bench command on one box (three runs), and of the config-5 launcher line
source "$(dirname "$0")/_common.sh"
for i in 1 2 3; do
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$i.json 2> $O/bench_$i.err || { tail -3 $O/bench_$i.err; exit 0; }
  python3 -c "import json; d=json.load(open('$O/bench_$i.json')); print('headline run $i', round(d['value'],1), 'frac', d['roofline']['frac'], 'cpu', round(d['cpu_baseline']['value'],3))"
done
for i in 1 2 3; do
  timeout -k 10 600 python3 bench.py --workload th --no-cpu-baseline > $O/th_$i.json 2> $O/th_$i.err || { tail -3 $O/th_$i.err; exit 0; }
  python3 -c "import json; d=json.load(open('$O/th_$i.json')); print('th run $i', round(d['value'],1))"
done
exit 0
