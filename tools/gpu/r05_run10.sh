#!/bin/bash
# round 5, run 10: the coupled system's block Thomas stored as (lo, inv, inv*up) with the factorisation inside BiCGStab's first solve
# (k_th_tri_factor<true, .>, option th_lazy_factor) -- parity tests, then the A/B at 512 x 512 x 80 on one box (lazy off / on, twice),
# the three-pass assembly (flat equation-of-state kernel that stores its 13 derived values + row gather, option assembly=0) as the
# measured form of the bytes-for-instructions trade, kernel stats of the new default; the Fortran-hosted vsfm_mms driver
source "$(dirname "$0")/_common.sh"
timeout -k 10 900 python3 -m pytest tests/test_gpu_th.py tests/test_gpu_more.py -x -q -m gpu -k "th or fortran or TH" > $O/pytest.log 2>&1
rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && { keep_failure $O/pytest.log; exit 0; }
for v in 00 10 11 00 10 11; do
  timeout -k 10 300 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt th_lazy_factor=${v:0:1} --opt th_inv_fly=${v:1:1} > $O/th_lazy$v.log 2>&1 || { tail -5 $O/th_lazy$v.log; exit 0; }
  tail -1 $O/th_lazy$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('th lazy,inv_fly=$v', round(d['newton_its_per_s'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
timeout -k 10 300 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt assembly=general > $O/th_3pass.log 2>&1 || { tail -5 $O/th_3pass.log; exit 0; }
tail -1 $O/th_3pass.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('th 3-pass', round(d['newton_its_per_s'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_3pass -o th3 -- python3 $GRAFT_REPO_ROOT/tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --no-events --opt assembly=general > $GRAFT_REPO_ROOT/$O/prof_3pass.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_new -o thn -- python3 $GRAFT_REPO_ROOT/tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --no-events > $GRAFT_REPO_ROOT/$O/prof_new.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_3pass prof_new; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d"; [ -n "$f" ] && python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:9]: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e6,3), r['Percentage'])"
  find $O/$d -name "*.db" -delete; find $O/$d -name "*kernel_trace.csv" -delete
done
exit 0
