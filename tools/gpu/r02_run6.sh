#!/bin/bash
# round 2, GPU run 6: SpMV with two tiles of loads ahead (experiment) vs the current kernel; 8-rank decomposition of the
# headline grid, in-process; config-2 size bench; shares of the grid one rank of an N-GPU run would own
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run6; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== spmv variants test"; timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k spmv_variants > $O/pytest_spmv.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_spmv.log
echo "== kbench spmv"; timeout -k 10 600 python3 tools/kbench.py --reps 30 --out $O/kbench_spmv.json --only spmv:20 spmv:324 spmv:624 spmv:1224 spmv:2424 spmv:20 spmv:624 > $O/kbench_spmv.log 2>&1; echo "rc=$?"; cat $O/kbench_spmv.log
echo "== sharded large (8 in-process ranks)"; timeout -k 10 900 python3 tools/check_sharded_large.py 1024 8 2 > $O/sharded8.log 2>&1; echo "rc=$?"; tail -6 $O/sharded8.log
echo "== bench 256x256x50"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err; echo "rc=$?"; cut -c1-700 $O/bench_256.json
for ny in 512 256 128; do echo "== share of the grid 1024x$ny"; timeout -k 10 300 python3 bench.py --nx 1024 --ny $ny --profile spmv --no-cpu-baseline --no-verify > $O/bench_share_$ny.json 2> $O/bench_share_$ny.err; echo "rc=$?"; cut -c1-330 $O/bench_share_$ny.json; done
ls $O
