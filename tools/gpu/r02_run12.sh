#!/bin/bash
# round 2, GPU run 12: reductions' second stage in one launch with the BiCGStab scalar step (parity, bench, shares of the grid)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run12; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest gpu (all)"; timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -4 $O/pytest.log
echo "== bench"; timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-200 $O/bench.json; python3 -c "import json;j=json.load(open('$O/bench.json'));print(j['kernels'], j['answer_check'])"
for ny in 512 256 128; do
  echo "== share ny=$ny"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-verify --ny $ny > $O/bench_ny$ny.json 2>> $O/bench.err; echo "rc=$?"; cut -c1-160 $O/bench_ny$ny.json
done
