#!/bin/bash
# round 2, GPU run 10: evidence for the end-of-round code: bench line, rocprofv3 kernel stats of the same command, PMC traffic
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run10; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== kbench dots"; timeout -k 10 600 python3 tools/kbench.py --reps 30 --only spmv_dots:6 spmv_dots:8 spmv_dots:12 spmv:20 > $O/kbench_dots.log 2>&1; echo "rc=$?"; cat $O/kbench_dots.log
echo "== bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-2200 $O/bench.json
echo "== bench again"; timeout -k 10 900 python3 bench.py --no-cpu-baseline --no-verify > $O/bench2.json 2> $O/bench2.err; echo "rc=$?"; cut -c1-300 $O/bench2.json
echo "== kernel trace of the bench"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify > $O/prof_bench.log 2>&1; echo "rc=$?"; tail -2 $O/prof_bench.log | cut -c1-300
rm -f $O/prof_bench/*kernel_trace.csv
echo "== pmc fetch"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1; echo "rc=$?"
echo "== pmc write"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_write.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "rc=$?"
rm -rf $O/pmc_fetch $O/pmc_write
ls $O
