#!/bin/bash
# round 4, run 1: what hipMemcpyAsync from pageable memory promises on this runtime (tools/upload_probe.hip), then the GPU suite of
# the round-3 code as the round's starting point.
source "$(dirname "$0")/_common.sh"
build_all
echo "== upload probe"; timeout -k 10 300 tools/_build/upload_probe > $O/upload_probe.json 2> $O/upload_probe.err; echo "rc=$?"; cat $O/upload_probe.json
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "rc=$rc"; tail -4 $O/pytest_gpu.log
[ $rc -ne 0 ] && keep_failure $O/pytest_gpu.log
exit 0
