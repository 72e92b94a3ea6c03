#!/bin/bash
# round 3, run 24: where config 2 and the one-eighth grid spend their time now (kernel traces, no HIP events in the queue)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run24; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2 --profile spmv --opt profile_stride=1000000"
echo "== c2"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B > $O/c2.json 2>> $O/err.log; python3 -c "import json;j=json.load(open('$O/c2.json'));print(j['value'],j['newton_iterations'],j['host_readbacks_per_newton_iteration'])"
echo "== s8"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B > $O/s8.json 2>> $O/err.log; python3 -c "import json;j=json.load(open('$O/s8.json'));print(j['value'],j['newton_iterations'],j['host_readbacks_per_newton_iteration'])"
echo "== trace c2"; timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_c2 -o t --output-format csv -- python3 bench.py --nx 256 --ny 256 --nz 50 $B > $O/tr_c2.log 2>&1; echo "rc=$?"
python3 tools/trace_gaps.py $O/tr_c2 > $O/gaps_c2.json
echo "== trace s8"; timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_s8 -o t --output-format csv -- python3 bench.py --nx 1024 --ny 128 $B > $O/tr_s8.log 2>&1; echo "rc=$?"
python3 tools/trace_gaps.py $O/tr_s8 > $O/gaps_s8.json
