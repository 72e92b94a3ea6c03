#!/bin/bash
# round 2, evidence A for the end-of-round code: the whole GPU suite, smoke(), the bench line (with the CPU legs), the rocprofv3
# kernel stats of the same command, the PMC traffic passes.  Output under gpurun_out/r02_evidence_a/ (copied to profiles/ by hand).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_evidence_a; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_gpu.log
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?"; tail -2 $O/smoke.log
echo "== bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-1500 $O/bench.json
echo "== bench again"; timeout -k 10 900 python3 bench.py --no-cpu-baseline --no-verify > $O/bench2.json 2> $O/bench2.err; echo "rc=$?"; cut -c1-200 $O/bench2.json
echo "== config 2"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 --no-cpu-baseline > $O/bench_256x256x50.json 2>> $O/bench2.err; echo "rc=$?"; cut -c1-200 $O/bench_256x256x50.json
echo "== kernel trace of the bench"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify > $O/prof_bench.log 2>&1; echo "rc=$?"; tail -2 $O/prof_bench.log | cut -c1-300
rm -f $O/prof_bench/*kernel_trace.csv $O/prof_bench/*/*kernel_trace.csv
echo "== pmc fetch"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1; echo "rc=$?"
echo "== pmc write"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_write.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "rc=$?"
rm -rf $O/pmc_fetch $O/pmc_write
ls $O
