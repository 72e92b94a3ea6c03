#!/bin/bash
# round 5, run 2: whole GPU suite on the round's code so far (whole-column partition, refactored v-channel builders); HBM traffic (PMC
# passes) of the 7-point VSFM kernels at 512 x 512 x 100 -- the general assembly and the windowed SpMV have no PMC evidence yet;
# graph replays under rocprofv3 --kernel-trace with the library's events on (level-scheduled box solves)
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "rc=$rc"; tail -4 $O/pytest_gpu.log
[ $rc -ne 0 ] && keep_failure $O/pytest_gpu.log
V3="bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 1 --warmup 1 --profile spmv"
echo "== pmc fetch"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $V3 > $O/pmc_fetch.log 2>&1; echo "rc=$?"
echo "== pmc write"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $V3 > $O/pmc_write.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_vsfm3d_512x512x100.json 2> $O/pmc.err; echo "rc=$?"
rm -rf $O/pmc_fetch $O/pmc_write
python3 - <<PY
import json
d=json.load(open("$O/pmc_vsfm3d_512x512x100.json"))
print("read factor", d["read_factor"])
for k,v in d["kernels"].items():
    if v["hbm_bytes_per_launch"]>5e7: print("%-70s %8.1f B/cell  (%.3f GB)"%(k.split("(")[0][-70:], v["hbm_bytes_per_launch"]/26214400, v["hbm_bytes_per_launch"]/1e9))
PY
echo "== ADVICE r04: graph replays (per-level box solves) under rocprofv3 --kernel-trace, library events on"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_graph -o g --output-format csv -- python3 tools/thermal_bench.py --n 128 --cpu-n 0 --opt ilu_box_stream=0 > $O/prof_graph_events_on.log 2>&1; echo "events on: rc=$?"; tail -2 $O/prof_graph_events_on.log | cut -c1-200
rm -f $O/prof_graph/*kernel_trace.csv $O/prof_graph/*/*kernel_trace.csv
exit 0
