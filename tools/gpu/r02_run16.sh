#!/bin/bash
# round 2, GPU run 16: where the time of a box level goes: kernel trace (direct launches under the profiler) and SQ counters at 256^3
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run16; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cd /tmp
echo "== kernel trace"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/thermal_bench.py --n 256 --cpu-n 0 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT; find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200'
cd /tmp
echo "== pmc 1"; timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -o p -- python3 $GRAFT_REPO_ROOT/tools/thermal_bench.py --n 128 --cpu-n 0 > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1; echo "rc=$?"
echo "== pmc 2"; timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -o p -- python3 $GRAFT_REPO_ROOT/tools/thermal_bench.py --n 128 --cpu-n 0 > $GRAFT_REPO_ROOT/$O/pmc2.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT; python3 tools/pmc_sq_summary.py $O/pmc1 $O/pmc2 --match k_box > $O/pmc_box.md 2>&1; cat $O/pmc_box.md
