#!/bin/bash
# round 5: the whole GPU suite and smoke() on the final tree (as the driver runs them)
source "$(dirname "$0")/_common.sh"
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
exit 0
