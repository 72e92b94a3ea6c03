#!/bin/bash
# round 2, GPU run 34: the column march specialised for the structure of the benchmark's meshes (lean_march): the GPU suite, then
# A/B on one box: kbench and bench with lean_march 1 / 0
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run34; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_gpu.log
for l in 1 0 1 0; do
  echo "== kbench lean_march=$l"; timeout -k 10 300 python3 tools/kbench.py --reps 20 --opt lean_march=$l --only residual assemble_both > $O/kbench_$l.log 2>&1; echo "rc=$?"; cat $O/kbench_$l.log
done
for l in 1 0 1 0; do
  echo "== bench lean_march=$l"; timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-verify --opt lean_march=$l > $O/bench_$l.json 2> $O/bench_$l.err; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/bench_$l.json'));print(j['value'],{k:v['avg_ms'] for k,v in j['kernels'].items() if k in ('residual','jacobian')})"
done
