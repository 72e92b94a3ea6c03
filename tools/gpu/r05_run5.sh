#!/bin/bash
# round 5, run 5: whole GPU suite (default PC = PETSc's ILU(0) semantics, cols3d, ls_fold), kernel stats of the 7-point grid with the
# column march + lateral pass, config 4 on four in-process ranks at 256^3
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $O/pytest_gpu.log
[ $rc -ne 0 ] && keep_failure $O/pytest_gpu.log
echo "== rocprofv3 kernel stats, vsfm3d 512x512x100, cols3d"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -o v3d --output-format csv -- python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 --opt ilu_graph=0 > $O/prof.log 2>&1; echo "rc=$?"
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s %6s %9.3f ms avg %8.1f us %6s%%"%(r['Name'].replace('void ','').replace('mpp::(anonymous namespace)::','')[:70],r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,r['Percentage']))
PY
echo "== config 4 on four in-process ranks, 256^3"
timeout -k 10 600 python3 tools/thermal_ranks_bench.py --n 256 --out $O/thermal_ranks_256.json > $O/thermal_ranks_256.log 2>&1; echo "rc=$?"; tail -3 $O/thermal_ranks_256.log | cut -c1-1500
exit 0
