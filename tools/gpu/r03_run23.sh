#!/bin/bash
# round 3, run 23: the 8-rank decomposition of the headline grid in-process (one handle vs 8 ranks of 131072 columns) on the round-3
# code (fused Krylov kernels, speculation, merged all-reduces under the communicator), and the fused tests
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run23; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== fused tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_fused.py tests/test_gpu_more.py -q -m gpu > $O/pytest_fused.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 $O/pytest_fused.log | cut -c1-220
echo "== sharded large"; timeout -k 10 900 python3 tools/check_sharded_large.py 1024 8 2 > $O/check_sharded_large_8ranks.log 2>&1; echo "rc=$?"; tail -12 $O/check_sharded_large_8ranks.log
