#!/bin/bash
# round 2: the whole GPU suite and smoke() on the final code (log copied to profiles/r02_pytest_gpu_final.log)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_final_suite; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_gpu.log
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?"; tail -2 $O/smoke.log
echo "== bench"; timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/bench.json'));print(j['value'],j['roofline']['frac'],j['answer_check']['max_rel_err'])"
