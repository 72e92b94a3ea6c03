#!/bin/bash
# round 3: HBM bytes per launch (PMC, separate FETCH_SIZE / WRITE_SIZE passes) of the TH kernels at config-5 size and of the thermal
# (box ILU(0), GMRES) kernels at 256^3, on the end-of-round code
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_pmc2; mkdir -p $O
TH="tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --no-events --steps 1"
TM="tools/thermal_bench.py --n 256 --cpu-n 0 --no-events --opt ilu_graph=0"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/f_th -o f --output-format csv -- python3 $TH > $O/th_f.log 2>&1; echo "th fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/w_th -o w --output-format csv -- python3 $TH > $O/th_w.log 2>&1; echo "th write rc=$?"
python3 tools/pmc_summary.py $O/f_th $O/w_th k_th k_spmv k_copy k_bcgs k_finalize > $O/pmc_th_512x512x80.json 2> $O/pmc_th.err; echo "rc=$?"
rm -rf $O/f_th $O/w_th
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/f_tm -o f --output-format csv -- python3 $TM > $O/tm_f.log 2>&1; echo "thermal fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/w_tm -o w --output-format csv -- python3 $TM > $O/tm_w.log 2>&1; echo "thermal write rc=$?"
python3 tools/pmc_summary.py $O/f_tm $O/w_tm k_box k_thermal k_spmv k_mdot k_maxpy k_gather k_copy > $O/pmc_thermal_256.json 2> $O/pmc_tm.err; echo "rc=$?"
rm -rf $O/f_tm $O/w_tm
ls -la $O | head
