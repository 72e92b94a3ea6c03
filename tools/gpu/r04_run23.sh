#!/bin/bash
# round 4, run 23: tests/test_gpu_setup.py with the ghost-cell case added
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 600 python3 -m pytest tests/test_gpu_setup.py -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log; [ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 1; }
exit 0
