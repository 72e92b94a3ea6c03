#!/bin/bash
# round 3, run 1: where the small-grid regime stands on the round-2 code -- config 2 (256x256x50) and the per-GPU share of the
# 8-GPU run (1024x128 columns x 100): bench lines with and without the per-class HIP events, kernel traces and their gaps.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run1; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2"
echo "== config 2, all classes timed"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B > $O/c2_all.json 2> $O/err.log; echo "rc=$?"; cut -c1-160 $O/c2_all.json
echo "== config 2, spmv only"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --profile spmv > $O/c2_spmv.json 2>> $O/err.log; echo "rc=$?"; cut -c1-160 $O/c2_spmv.json
echo "== 1024x128, spmv only"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --profile spmv > $O/share8_spmv.json 2>> $O/err.log; echo "rc=$?"; cut -c1-160 $O/share8_spmv.json
echo "== full, spmv only"; timeout -k 10 400 python3 bench.py $B --steps 4 --warmup 1 --profile spmv > $O/full_spmv.json 2>> $O/err.log; echo "rc=$?"; cut -c1-160 $O/full_spmv.json
echo "== trace config 2"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/tr_c2 -o t --output-format csv -- python3 bench.py --nx 256 --ny 256 --nz 50 $B --profile spmv > $O/tr_c2.log 2>&1; echo "rc=$?"
python3 tools/trace_gaps.py $O/tr_c2 > $O/gaps_c2.json; echo "rc=$?"
echo "== trace 1024x128"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/tr_s8 -o t --output-format csv -- python3 bench.py --nx 1024 --ny 128 $B --profile spmv > $O/tr_s8.log 2>&1; echo "rc=$?"
python3 tools/trace_gaps.py $O/tr_s8 > $O/gaps_s8.json; echo "rc=$?"
find $O -name "*kernel_trace.csv" -size +20M -delete
ls -la $O
