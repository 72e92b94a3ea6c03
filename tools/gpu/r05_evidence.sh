#!/bin/bash
# round 5, evidence for the end-of-round code: the whole GPU suite, smoke(), the bench line (with the CPU legs and extra_configs), the
# rocprofv3 kernel stats of the same command, the PMC traffic passes -> profiles/kernel_traffic.json (with the hash of the kernel
# sources), the bench line again (now quoting the moved bytes), config 2, one eighth of the grid; then configs 4 and 5 through the
# same launcher (bench.py --workload thermal / th), their kernel stats and their PMC passes.
# Output under gpurun_out/r05_evidence_<time>_<pid>/ (copied to profiles/r05_evidence/ by hand).
source "$(dirname "$0")/_common.sh"
build_all
PART="${1:-a}"     # a: suite, smoke, headline bench + traces + PMC; b: configs 4 and 5 (benches, kernel stats); c: their PMC passes; d: the 7-point grid, four in-process ranks (gpurun's limit is 20 min a call)
step() { echo "== $1"; }
if [ "$PART" = a ]; then
step "pytest -m gpu"; timeout -k 10 1500 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "rc=$rc"; tail -4 $O/pytest_gpu.log; [ $rc -ne 0 ] && keep_failure $O/pytest_gpu.log
step "smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; echo "rc=$rc"; tail -2 $O/smoke.log; [ $rc -ne 0 ] && keep_failure $O/smoke.log
step "kernel trace of the bench"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs > $O/prof_bench.log 2>&1; echo "rc=$?"; tail -2 $O/prof_bench.log | cut -c1-300
python3 tools/trace_gaps.py $O/prof_bench > $O/gaps_full.json 2>/dev/null
rm -f $O/prof_bench/*kernel_trace.csv $O/prof_bench/*/*kernel_trace.csv
step "pmc fetch"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1; echo "rc=$?"
step "pmc write"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 2 --warmup 1 > $O/pmc_write.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "rc=$?"
python3 tools/make_traffic.py $O/pmc_traffic.json 104857600 "tools/gpu/r05_evidence.sh" > $O/kernel_traffic.json 2>> $O/pmc_traffic.err; echo "rc=$?"
cp $O/kernel_traffic.json profiles/kernel_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
step "bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-1200 $O/bench.json
step "bench again"; timeout -k 10 900 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs > $O/bench2.json 2> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench2.json
B="--no-cpu-baseline --no-verify --no-extra-configs --steps 8 --warmup 2"
step "config 2"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B > $O/bench_256x256x50.json 2>> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench_256x256x50.json
step "config 2, spmv events only"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --profile spmv > $O/bench_256x256x50_spmv.json 2>> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench_256x256x50_spmv.json
step "one eighth"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --profile spmv > $O/bench_1024x128x100.json 2>> $O/bench2.err; echo "rc=$?"; cut -c1-160 $O/bench_1024x128x100.json
step "the driver's launch form at N = 1, and the two-ranks-on-one-GPU path up to RCCL's refusal"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-extra-configs > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err; echo "rc=$?"; cut -c1-200 $O/bench_torchrun_n1.json
timeout -k 10 300 python3 bench.py --gpus 2 --share-gpus --nx 256 --ny 256 --nz 50 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-extra-configs --init-timeout 120 > $O/bench_share_gpus.out 2> $O/bench_share_gpus.err; echo "rc=$? (non-zero expected: RCCL refuses two ranks on one device)"; tail -5 $O/bench_share_gpus.err | cut -c1-300
fi
if [ "$PART" = b ]; then
step "config 5 through the launcher"; timeout -k 10 600 python3 bench.py --workload th > $O/bench_th.json 2> $O/bench_th.err; echo "rc=$?"; cut -c1-400 $O/bench_th.json
step "config 4 through the launcher"; timeout -k 10 900 python3 bench.py --workload thermal > $O/bench_thermal.json 2> $O/bench_thermal.err; echo "rc=$?"; cut -c1-400 $O/bench_thermal.json
step "th 512x512x80, every class timed"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"; tail -1 $O/th.log | cut -c1-300
step "thermal 512^3 with the rtol sweep"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol-sweep 1e-8 --out $O/thermal_512.json > $O/thermal_512.log 2>&1; echo "rc=$?"; tail -1 $O/thermal_512.log | cut -c1-300
step "kernel stats of the TH and thermal runs"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_th -o t --output-format csv -- python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --no-events > $O/prof_th.log 2>&1; echo "rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_thermal -o t --output-format csv -- python3 tools/thermal_bench.py --n 512 --cpu-n 0 --no-events > $O/prof_thermal.log 2>&1; echo "rc=$?"
rm -f $O/prof_th/*kernel_trace.csv $O/prof_th/*/*kernel_trace.csv $O/prof_thermal/*kernel_trace.csv $O/prof_thermal/*/*kernel_trace.csv
fi
if [ "$PART" = d ]; then
# the 7-point VSFM grid (SURVEY 8d's 3-D variant): bench line, kernel stats and PMC passes of the same command (cols3d: column march +
# lateral pass, windowed CSR SpMV inside BiCGStab), the analytic check vsfm_mms, config 4 on four in-process ranks
V3="bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1"
step "vsfm3d 512x512x100"; timeout -k 10 400 python3 $V3 > $O/vsfm3d_512x512x100.json 2> $O/vsfm3d.err; echo "rc=$?"; cut -c1-400 $O/vsfm3d_512x512x100.json
step "vsfm3d 1024x1024x100 (BASELINE config 3's size, 7-point)"; timeout -k 10 900 python3 bench.py --workload vsfm3d --nx 1024 --ny 1024 --nz 100 --steps 2 --warmup 1 > $O/vsfm3d_1024x1024x100.json 2> $O/vsfm3d_1024.err; echo "rc=$?"; cut -c1-400 $O/vsfm3d_1024x1024x100.json
step "vsfm3d kernel stats"; timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_v3d -o v --output-format csv -- python3 $V3 --opt ilu_graph=0 > $O/prof_v3d.log 2>&1; echo "rc=$?"
rm -f $O/prof_v3d/*kernel_trace.csv $O/prof_v3d/*/*kernel_trace.csv
step "vsfm3d pmc"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/f_v3d -o f --output-format csv -- python3 $V3 --profile spmv > $O/v3d_f.log 2>&1; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/w_v3d -o w --output-format csv -- python3 $V3 --profile spmv > $O/v3d_w.log 2>&1; echo "write rc=$?"
python3 tools/pmc_summary.py $O/f_v3d $O/w_v3d k_spmv k_tri k_flux k_assemble k_auxvars k_bcgs k_waxpy k_copy > $O/pmc_vsfm3d_512x512x100.json 2> $O/pmc_v3d.err; echo "rc=$?"
rm -rf $O/f_v3d $O/w_v3d
step "the driver's bench command"; timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form_steps20_warmup5.json 2> $O/bench_driver_form.err; echo "rc=$?"; cut -c1-200 $O/bench_driver_form_steps20_warmup5.json
step "config 4 on four in-process ranks"; timeout -k 10 900 python3 tools/thermal_ranks_bench.py --n 512 --out $O/thermal_ranks_512.json > $O/thermal_ranks_512.log 2>&1; echo "rc=$?"; tail -1 $O/thermal_ranks_512.log | cut -c1-600
timeout -k 10 600 python3 tools/thermal_ranks_bench.py --n 256 --out $O/thermal_ranks_256.json > $O/thermal_ranks_256.log 2>&1; echo "rc=$?"
fi
if [ "$PART" = c ]; then
step "PMC passes of the TH (config-5 size) and thermal (512^3) kernels"
TH="tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --no-events --steps 1"
TM="tools/thermal_bench.py --n 512 --cpu-n 0 --no-events --rtol 1e-2"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/f_th -o f --output-format csv -- python3 $TH > $O/th_f.log 2>&1; echo "th fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/w_th -o w --output-format csv -- python3 $TH > $O/th_w.log 2>&1; echo "th write rc=$?"
python3 tools/pmc_summary.py $O/f_th $O/w_th k_th k_spmv k_copy k_bcgs k_finalize > $O/pmc_th_512x512x80.json 2> $O/pmc_th.err; echo "rc=$?"
rm -rf $O/f_th $O/w_th
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/f_tm -o f --output-format csv -- python3 $TM > $O/tm_f.log 2>&1; echo "thermal fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/w_tm -o w --output-format csv -- python3 $TM > $O/tm_w.log 2>&1; echo "thermal write rc=$?"
python3 tools/pmc_summary.py $O/f_tm $O/w_tm k_box k_thermal k_spmv k_mdot k_maxpy k_gather k_copy k_scale > $O/pmc_thermal_512.json 2> $O/pmc_tm.err; echo "rc=$?"
rm -rf $O/f_tm $O/w_tm
# kernel stats of the same short command (the Gram-Schmidt kernels' work per launch grows with the basis: averages of runs of the same length)
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/prof_thermal_short -o t --output-format csv -- python3 $TM > $O/prof_thermal_short.log 2>&1; echo "rc=$?"
rm -f $O/prof_thermal_short/*kernel_trace.csv $O/prof_thermal_short/*/*kernel_trace.csv
fi
ls $O
