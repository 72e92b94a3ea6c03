#!/bin/bash
# round 3, evidence for the end-of-round code: the whole GPU suite, smoke(), the bench line (with the CPU legs), the rocprofv3 kernel
# stats of the same command, the PMC traffic passes -> profiles/kernel_traffic.json (with the hash of the kernel sources), then the
# bench line again (now quoting the moved bytes), config 2, one eighth of the grid, TH at config-5 size.
# Output under gpurun_out/r03_evidence/ (copied to profiles/ by hand).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_evidence; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1500 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_gpu.log
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?"; tail -2 $O/smoke.log
echo "== kernel trace of the bench"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify > $O/prof_bench.log 2>&1; echo "rc=$?"; tail -2 $O/prof_bench.log | cut -c1-300
python3 tools/trace_gaps.py $O/prof_bench > $O/gaps_full.json 2>/dev/null
rm -f $O/prof_bench/*kernel_trace.csv $O/prof_bench/*/*kernel_trace.csv
echo "== pmc fetch"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1; echo "rc=$?"
echo "== pmc write"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 > $O/pmc_write.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "rc=$?"
python3 tools/make_traffic.py $O/pmc_traffic.json 104857600 "tools/gpu/r03_evidence.sh" > $O/kernel_traffic.json 2>> $O/pmc_traffic.err; echo "rc=$?"
cp $O/kernel_traffic.json profiles/kernel_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
echo "== bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-1200 $O/bench.json
echo "== bench again"; timeout -k 10 900 python3 bench.py --no-cpu-baseline --no-verify > $O/bench2.json 2> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench2.json
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2"
echo "== config 2"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B > $O/bench_256x256x50.json 2>> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench_256x256x50.json
echo "== config 2, spmv events only"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --profile spmv > $O/bench_256x256x50_spmv.json 2>> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench_256x256x50_spmv.json
echo "== one eighth"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --profile spmv > $O/bench_1024x128x100.json 2>> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench_1024x128x100.json
echo "== kernel trace of one eighth"; timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_s8 -o t --output-format csv -- python3 bench.py --nx 1024 --ny 128 $B --profile spmv > $O/tr_s8.log 2>&1; echo "rc=$?"
python3 tools/trace_gaps.py $O/tr_s8 > $O/gaps_1024x128x100.json; rm -rf $O/tr_s8
echo "== kernel stats of the TH and thermal runs"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_th -o t --output-format csv -- python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --no-events > $O/prof_th.log 2>&1; echo "rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_thermal -o t --output-format csv -- python3 tools/thermal_bench.py --n 256 --cpu-n 0 --no-events --opt ilu_graph=0 > $O/prof_thermal.log 2>&1; echo "rc=$?"
rm -f $O/prof_th/*kernel_trace.csv $O/prof_th/*/*kernel_trace.csv $O/prof_thermal/*kernel_trace.csv $O/prof_thermal/*/*kernel_trace.csv
echo "== th 512x512x80"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"; tail -1 $O/th.log | cut -c1-300
echo "== the driver's launch form at N = 1, and the two-ranks-on-one-GPU path up to RCCL's refusal"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err; echo "rc=$?"; cut -c1-200 $O/bench_torchrun_n1.json
timeout -k 10 300 python3 bench.py --gpus 2 --share-gpus --nx 256 --ny 256 --nz 50 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --init-timeout 120 > $O/bench_share_gpus.out 2> $O/bench_share_gpus.err; echo "rc=$? (non-zero expected: RCCL refuses two ranks on one device)"; tail -5 $O/bench_share_gpus.err | cut -c1-300
ls $O
