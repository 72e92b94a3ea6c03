#!/bin/bash
# round 2, GPU run 26: box ILU(0) on ranks with ghost planes (in-process groups): parity with the general route, then the multi-rank tests
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run26; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest r02 + more + thermal"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py -q -m gpu -k "ghost or box" > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log
