#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run7; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== kbench spmv tiles per wave"; timeout -k 10 600 python3 tools/kbench.py --reps 40 --out $O/kbench_spmv.json --only spmv:20 spmv:124 spmv:224 spmv:324 spmv:424 spmv:524 spmv:20 spmv:324 spmv:224 > $O/kbench_spmv.log 2>&1; echo "rc=$?"; cat $O/kbench_spmv.log
