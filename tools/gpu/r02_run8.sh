#!/bin/bash
# round 2, GPU run 8: the unified three-set compact SpMV (plain, split, dots, windowed): parity + every bench that uses it
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run8; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -6 $O/pytest_gpu.log
echo "== bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-2600 $O/bench.json; tail -3 $O/bench.err
echo "== th bench"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 > $O/th_bench.log 2>&1; echo "rc=$?"; tail -1 $O/th_bench.log | cut -c1-900
echo "== vsfm3d bench"; timeout -k 10 600 python3 tools/vsfm3d_bench.py --nx 256 --ny 256 --nz 64 > $O/vsfm3d.log 2>&1; echo "rc=$?"; tail -1 $O/vsfm3d.log | cut -c1-700
echo "== thermal 256"; timeout -k 10 300 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --out $O/thermal_256.json > $O/t256.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/thermal_256.json'));print(j['step_s'],j['linear_iterations'],j['kernels']['spmv'],j['kernels']['pc_apply'])"
echo "== kbench"; timeout -k 10 600 python3 tools/kbench.py --reps 20 --out $O/kbench.json > $O/kbench.log 2>&1; echo "rc=$?"; cat $O/kbench.log
