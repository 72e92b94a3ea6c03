#!/bin/bash
# round 2, GPU run 4: full GPU suite; bench.py --gpus 2 on a one-GPU box (must fail cleanly); PMC traffic + kernel-trace of the
# bench; thermal time-to-accuracy sweeps
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run4; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -12 $O/pytest_gpu.log
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?"; tail -3 $O/smoke.log
echo "== bench --gpus 2 on one GPU"; timeout -k 10 300 python3 bench.py --gpus 2 --nx 256 --ny 256 --steps 1 --warmup 0 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "rc=$?"; cat $O/bench_gpus2.out; tail -12 $O/bench_gpus2.err
echo "== kernel trace of the bench"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 4 --warmup 1 > $O/prof_bench.log 2>&1; echo "rc=$?"; tail -2 $O/prof_bench.log | cut -c1-400
echo "== pmc fetch"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1; echo "rc=$?"
echo "== pmc write"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_write.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "rc=$?"; head -c 1500 $O/pmc_traffic.json
rm -rf $O/pmc_fetch/*/*.db $O/pmc_write/*/*.db
echo "== thermal 256^3 sweep"; timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --rtol-sweep 1e-5,1e-7,1e-9 --out $O/thermal_256_sweep.json > $O/thermal_256.log 2>&1; echo "rc=$?"; grep rtol $O/thermal_256.log
echo "== thermal 512^3 sweep"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol-sweep 1e-5,1e-8 --out $O/thermal_512_sweep.json > $O/thermal_512.log 2>&1; echo "rc=$?"; grep rtol $O/thermal_512.log
ls $O; du -sh $O
