#!/bin/bash
# round 4, run 13: kernel stats of the streamed ILU(0) application at 512^3 (which kernel takes what)
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol 1e-2 --no-events > $O/prof.log 2>&1; echo "rc=$?"; tail -1 $O/prof.log | cut -c1-200
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_thermal_512.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_thermal_512.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print("%-70s calls %6s avg %9.1f us  %5.1f%%"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
