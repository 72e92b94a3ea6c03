#!/bin/bash
# round 4, run 5: kernel trace of the per-level box solves at 512^3 (box length 64 and 32): duration of every launch against its
# number of boxes, and the gaps between launches; plus the new GPU tests of this round so far
source "$(dirname "$0")/_common.sh"
build_all
echo "== new tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_more.py -q -m gpu -x -k "four_in_process or same_pair" > $O/pytest_new.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $O/pytest_new.log
[ $rc -ne 0 ] && keep_failure $O/pytest_new.log
for bl in 64 32; do
  echo "== trace, box length $bl"
  timeout -k 10 400 rocprofv3 --kernel-trace -d $O/tr_$bl -o t --output-format csv -- python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol 3e-1 --no-events --opt ilu_graph=0 --opt ilu_box_len=$bl > $O/tr_$bl.log 2>&1; echo "rc=$?"; tail -1 $O/tr_$bl.log | cut -c1-300
  python3 tools/box_trace.py $O/tr_$bl > $O/box_trace_$bl.json 2> $O/box_trace_$bl.err; echo "rc=$?"
  rm -rf $O/tr_$bl
done
ls -la $O
