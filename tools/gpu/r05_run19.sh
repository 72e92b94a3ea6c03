#!/bin/bash
# round 5, run 19: the one draw of the soak that failed outright (3-D VSFM family, offset 1, seed 4): its assertion
source "$(dirname "$0")/_common.sh"
MPP_FUZZ_SEED_OFFSET=1 timeout -k 10 300 python3 -u -m pytest "tests/test_gpu_fuzz.py::test_random_3d_vsfm_problem_matches_the_oracle[4]" -q -m gpu -p no:cacheprovider > $O/seed4.log 2>&1
echo "rc=$?"; grep -E "^E |Error|assert" $O/seed4.log | cut -c1-300 | head -30
exit 0
