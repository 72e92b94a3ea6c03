#!/bin/bash
# round 2, GPU run 24: HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the thermal step at 256^3 with the box ILU(0) solves
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run24; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pmc fetch"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 tools/thermal_bench.py --n 256 --cpu-n 0 > $O/pmc_fetch.log 2>&1; echo "rc=$?"
echo "== pmc write"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 tools/thermal_bench.py --n 256 --cpu-n 0 > $O/pmc_write.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_thermal_256.json 2> $O/pmc.err; echo "rc=$?"
rm -rf $O/pmc_fetch $O/pmc_write
python3 -c "
import json
j=json.load(open('$O/pmc_thermal_256.json'))
for k,v in j['kernels'].items():
    if any(s in k for s in ('k_box','k_gather','k_spmv','k_mdot','k_maxpy','k_copy')): print(k[:80], round(v['hbm_bytes_per_launch']/1e6,3),'MB/launch')
"
