#!/bin/bash
# round 3, run 3: scalar read-backs published by the reducing kernel and polled by the host (host_poll) against the copy + stream
# synchronise of round 2; GPU tests of the solver paths first
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run3; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu (vsfm files)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_more.py tests/test_gpu_r02.py -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2 --profile spmv"
for o in "host_poll=0" "host_poll=1"; do
echo "== c2 $o"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --opt $o > $O/c2_$o.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/c2_$o.json'));print(j['value'],j['newton_iterations'])"
echo "== s8 $o"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --opt $o > $O/s8_$o.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/s8_$o.json'));print(j['value'],j['newton_iterations'])"
done
echo "== trace s8"; timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_s8 -o t --output-format csv -- python3 bench.py --nx 1024 --ny 128 $B > $O/tr_s8.log 2>&1; echo "rc=$?"
python3 tools/trace_gaps.py $O/tr_s8 > $O/gaps_s8.json; echo "rc=$?"
tail -3 $O/err.log
