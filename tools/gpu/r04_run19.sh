#!/bin/bash
# round 4, run 19: kernel stats of the SAME short thermal command the PMC passes of r04_evidence.sh c ran (--rtol 1e-2: 20 iterations):
# the Gram-Schmidt kernels' work per launch grows with the basis, so their averages must come from runs of the same length
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_thermal_short -o t --output-format csv -- python3 tools/thermal_bench.py --n 512 --cpu-n 0 --no-events --rtol 1e-2 > $O/prof_thermal_short.log 2>&1; echo "rc=$?"
rm -f $O/prof_thermal_short/*kernel_trace.csv $O/prof_thermal_short/*/*kernel_trace.csv
head -12 $O/prof_thermal_short/t_kernel_stats.csv | cut -c1-160
