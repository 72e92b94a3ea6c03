#!/bin/bash
# round 2, GPU run 25: the column march held to 128 registers (4 waves per SIMD, 9 doubles spilled) against 151 registers (3 waves)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run25; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== kbench"; timeout -k 10 600 python3 tools/kbench.py --reps 20 --out $O/kbench.json --only residual jacobian assemble_both residual jacobian assemble_both > $O/kbench.log 2>&1; echo "rc=$?"; cat $O/kbench.log
echo "== bench"; timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-verify > $O/bench.json 2> $O/bench.err; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/bench.json'));print(j['value'],{k:(v['launches'],v['avg_ms']) for k,v in j['kernels'].items()})"
