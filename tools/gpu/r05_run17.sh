#!/bin/bash
# round 5, run 17: soak -- the randomised differential tests of the soil-column paths (scalar and coupled: the kernels this round's second
# half changed) on other families of draws (MPP_FUZZ_SEED_OFFSET = 1, 2, 3), final sources; 120 s per test
# (first attempt, all tests: the 3-D family of offset 1 holds a draw -- GMRES + point Jacobi, smoothed Brooks-Corey, no boundary -- on
#  which the CPU ORACLE grinds through its time-step cuts for more than 15 minutes; the run was cut for silence)
source "$(dirname "$0")/_common.sh"
for k in 1 2 3; do
  MPP_FUZZ_SEED_OFFSET=$k timeout -k 10 1000 python3 -u -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "random_vsfm_problem or th_columns" --timeout 120 -p no:cacheprovider > $O/fuzz_$k.log 2>&1
  echo "offset $k: rc=$?"; tail -3 $O/fuzz_$k.log
done
exit 0
