#!/bin/bash
# round 3, run 10: fpow (table-driven x^y) in every kernel, wide soil classes reverted: GPU suite, the march per law, the TH bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run10; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -12 $O/pytest.log
echo "== th bench"; timeout -k 10 600 python3 tools/th_bench.py --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"; tail -3 $O/th.log | cut -c1-600
