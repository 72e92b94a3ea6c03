#!/bin/bash
# round 3: A/B of fold_finalize (reductions of the fused SpMV + column-solve kernels finished by the kernels themselves)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_fold; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_fused.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
grep -q "core dump\|Aborted\|Segmentation" $O/pytest.log && exit 1
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2 --profile spmv"
for rep in 1 2; do
for f in 0 1; do
  timeout -k 10 200 python3 bench.py --nx 256 --ny 256 --nz 50 $B --opt fold_finalize=$f > $O/c2_f${f}_$rep.json 2>> $O/err.log; echo "c2 fold=$f: $(python3 -c "import json;print(round(json.load(open('$O/c2_f${f}_$rep.json'))['value'],1))")"
  timeout -k 10 200 python3 bench.py --nx 1024 --ny 128 $B --opt fold_finalize=$f > $O/s8_f${f}_$rep.json 2>> $O/err.log; echo "s8 fold=$f: $(python3 -c "import json;print(round(json.load(open('$O/s8_f${f}_$rep.json'))['value'],1))")"
done
done
