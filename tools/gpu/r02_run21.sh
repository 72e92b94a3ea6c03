#!/bin/bash
# round 2, GPU run 21: set-up passes on several host threads: the whole GPU suite, then where the set-up time goes (MPPGPU_TIMING)
# at thermal 256^3 and VSFM 1024x1024x100 with 1 host thread and with all
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run21; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_gpu.log
for t in 1 0; do
  if [ $t = 1 ]; then export MPPGPU_HOST_THREADS=1; else unset MPPGPU_HOST_THREADS; fi
  echo "== thermal 256, host threads: ${MPPGPU_HOST_THREADS:-all}"; MPPGPU_TIMING=1 timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --out $O/therm_256_t$t.json > $O/therm_256_t$t.log 2> $O/therm_256_t$t.err; echo "rc=$?"; grep mppgpu $O/therm_256_t$t.err; python3 -c "import json;j=json.load(open('$O/therm_256_t$t.json'));print('setup_s',j['setup_s'],'step_s',j['step_s'])"
  echo "== vsfm bench, host threads: ${MPPGPU_HOST_THREADS:-all}"; MPPGPU_TIMING=1 timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-verify --steps 2 > $O/bench_t$t.json 2> $O/bench_t$t.err; echo "rc=$?"; grep mppgpu $O/bench_t$t.err; python3 -c "import json;j=json.load(open('$O/bench_t$t.json'));print('setup_s',j['setup_s'],'value',j['value'])"
done
