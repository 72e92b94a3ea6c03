#!/bin/bash
# round 4, run 28: the first kernel of a BiCGStab solve reads c and 1 / pivot with plain (cacheable) loads in its first sweep -- its last
# sweep reads them again: kernel stats of the bench, twice
source "$(dirname "$0")/_common.sh"
build_all
for r in 1 2; do
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5 > $O/bench_$r.json 2> $O/bench_$r.err; echo "rc=$?"
python3 -c "import json; d=json.load(open('$O/bench_$r.json')); print(d['value'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs > $O/prof.log 2>&1; echo "rc=$?"
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
grep "k_tri_solve_lds\|k_spmv_cols\|k_assemble_cols<2" $O/prof/b_kernel_stats.csv | cut -c1-60,200-330
exit 0
