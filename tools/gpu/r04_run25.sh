#!/bin/bash
# round 4, run 25: the whole GPU suite with every device allocation poisoned (MPPGPU_POISON=1: 0xFF bytes = NaNs / negative indices
# in anything read before it is written) on the end-of-round sources -- the new set-up kernels, streamed solves and fused TH kernels
source "$(dirname "$0")/_common.sh"
build_all
MPPGPU_POISON=1 timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu_poison.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu_poison.log; [ $rc -ne 0 ] && keep_failure $O/pytest_gpu_poison.log
exit 0
