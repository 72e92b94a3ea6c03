#!/bin/bash
# round 2, GPU run 32: the 128^3 / 256^3 second-order test of the thermal path (time and memory of the test itself)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run32; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
t0=$(date +%s); timeout -k 10 600 python3 -m pytest tests/test_gpu_thermal.py -q -m gpu -k "second_order" > $O/pytest.log 2>&1; echo "rc=$? in $(( $(date +%s) - t0 )) s"; tail -5 $O/pytest.log
