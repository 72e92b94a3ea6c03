#!/bin/bash
# round 3, run 8: where rocprofv3 --kernel-trace dies in a graph replay (thermal, ILU(0) by boxes): process maps to attribute the frames,
# and which ingredient matters -- HIP events around the launches, nodes per graph, number of replays, grid size
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run8; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
try() { name=$1; shift; echo "== $name"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_$name -o t --output-format csv -- python3 tools/thermal_bench.py --cpu-n 0 --dump-maps $O/maps_$name.txt "$@" > $O/$name.log 2>&1; echo "rc=$?"; rm -rf $O/p_$name; }
try n128_default --n 128
try n128_noevents --n 128 --no-events
try n128_nodes16 --n 128 --opt ilu_graph_nodes=16
try n128_few_replays --n 128 --rtol 1e-1
try n48_default --n 48
try n128_nograph --n 128 --opt ilu_graph=0
for f in $O/n*.log; do echo "-- $f"; grep -c "SIGSEGV" $f; done
