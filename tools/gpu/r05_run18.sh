#!/bin/bash
# round 5, run 18: soak of the remaining randomised tests (3-D VSFM, thermal, 3-D TH, split meshes) on the draw family of offset 1,
# 120 s per test (a draw on which the CPU oracle grinds is reported as a timeout)
source "$(dirname "$0")/_common.sh"
MPP_FUZZ_SEED_OFFSET=1 timeout -k 10 1100 python3 -u -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "not random_vsfm_problem and not th_columns" --timeout 120 -p no:cacheprovider > $O/fuzz_1.log 2>&1
echo "rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/fuzz_1.log | cut -c1-220 | tail -20
exit 0
