#!/bin/bash
# round 4, run 27: fused SpMV + column solve kernels that redo the factorisation on the way instead of reading l: the fused / parity /
# regression tests, then the headline bench (kernel classes) twice
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; [ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 1; }
for r in 1 2; do
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5 > $O/bench_$r.json 2> $O/bench_$r.err; echo "rc=$?"
python3 -c "import json; d=json.load(open('$O/bench_$r.json')); print(d['value'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
exit 0
