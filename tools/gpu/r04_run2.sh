#!/bin/bash
# round 4, run 2: the WHOLE GPU suite with MPPGPU_POISON=1 -- every fresh device allocation filled with 0xFF bytes on the handle's own
# stream -- including the tests that drive several handles (sequentially in one thread, and one host thread per handle), once.
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest -m gpu, poisoned"; MPPGPU_POISON=1 timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu_poison.log 2>&1; rc=$?; echo "rc=$rc"; tail -6 $O/pytest_gpu_poison.log
[ $rc -ne 0 ] && keep_failure $O/pytest_gpu_poison.log
exit 0
