#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run9; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== kbench spmv_dots tiles per wave"; timeout -k 10 600 python3 tools/kbench.py --reps 30 --out $O/kbench_dots.json --only spmv_dots:1 spmv_dots:2 spmv_dots:3 spmv_dots:4 spmv_dots:6 spmv_dots:2 spmv_dots:4 bcgs_xr > $O/kbench_dots.log 2>&1; echo "rc=$?"; cat $O/kbench_dots.log
