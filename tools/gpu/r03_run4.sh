#!/bin/bash
# round 3, run 4: the line search's first trial point evaluated behind the Krylov solve (ls_spec), boundary values staged
# without a host loop, sampled profiler events -- GPU tests, then A/B at config 2, 1024x128 and the full grid
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run4; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu (vsfm files)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_more.py tests/test_gpu_r02.py -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2 --profile spmv"
for o in "ls_spec=0" "ls_spec=1"; do
echo "== c2 $o"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --opt $o > $O/c2_$o.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/c2_$o.json'));print(j['value'],j['newton_iterations'])"
echo "== s8 $o"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --opt $o > $O/s8_$o.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/s8_$o.json'));print(j['value'],j['newton_iterations'])"
done
echo "== s8 stride 4"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --opt profile_stride=4 > $O/s8_stride4.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/s8_stride4.json'));print(j['value'],j['newton_iterations'],j['roofline']['avg_launch_ms'])"
echo "== c2 stride 4"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --opt profile_stride=4 > $O/c2_stride4.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/c2_stride4.json'));print(j['value'],j['newton_iterations'],j['roofline']['avg_launch_ms'])"
echo "== full all"; timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/full_all.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/full_all.json'));print(j['value'],j['newton_iterations'],j['roofline']['avg_launch_ms'],j['answer_check'])"
echo "== full all stride 4"; timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --steps 4 --warmup 1 --opt profile_stride=4 > $O/full_all4.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/full_all4.json'));print(j['value'],j['newton_iterations'],j['roofline']['avg_launch_ms'])"
tail -3 $O/err.log
