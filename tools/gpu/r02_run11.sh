#!/bin/bash
# round 2, GPU run 11: ring back substitution (parity + timing); two ranks sharing the one GPU up to ncclCommInitRank
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run11; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest gpu (fused, r02, parity)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_more.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -4 $O/pytest.log
echo "== kbench tri_back"; timeout -k 10 600 python3 tools/kbench.py --reps 30 --out $O/kbench_back.json --only tri_back:0 tri_back:1:tri_group=8 tri_back:1:tri_group=16 tri_back:1:tri_group=24 tri_back:0 > $O/kbench_back.log 2>&1; echo "rc=$?"; cat $O/kbench_back.log
echo "== bench --gpus 2 --share-gpus (one GPU: RCCL must refuse)"; timeout -k 10 300 python3 bench.py --gpus 2 --share-gpus --nx 256 --ny 256 --steps 1 --warmup 0 --init-timeout 120 > $O/bench_share.out 2> $O/bench_share.err; echo "rc=$?"; cat $O/bench_share.out; grep -v "^  File\|^    " $O/bench_share.err | tail -15
echo "== bench"; timeout -k 10 900 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-300 $O/bench.json; python3 -c "import json;j=json.load(open('$O/bench.json'));print(j['kernels']['pc_apply'], j['answer_check'])"
