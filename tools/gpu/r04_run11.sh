#!/bin/bash
# round 4, run 11: the whole GPU suite (fold variant removed, long GMRES restarts, history-based first-trial prediction), then the headline twice
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest -m gpu"; timeout -k 10 1500 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $O/pytest_gpu.log
[ $rc -ne 0 ] && keep_failure $O/pytest_gpu.log
echo "== headline"; timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-extra-configs > $O/bench_a.json 2> $O/bench_a.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_a.json')); print(d['value'], d['newton_iterations'], {k:(v['launches'],v['avg_ms']) for k,v in d['kernels'].items()})"
echo "== headline, prediction from the last line search only"; timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-extra-configs --opt fused_predict=1 > $O/bench_b.json 2> $O/bench_b.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_b.json')); print(d['value'], d['newton_iterations'], {k:(v['launches'],v['avg_ms']) for k,v in d['kernels'].items()})"
