#!/bin/bash
# round 2, GPU run 5: full GPU suite after the fixes; chained ILU(0) with and without tickets; bench + kbench with the
# per-level porosity x volume; TH and 3-D VSFM benches (regression check of the other systems)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run5; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -6 $O/pytest_gpu.log
echo "== thermal 256, chain in dispatch order"; timeout -k 10 300 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --out $O/thermal_256_chain.json > $O/t256a.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/thermal_256_chain.json'));print(j['step_s'],j['linear_iterations'],j['kernels']['pc_apply'])"
echo "== thermal 256, chain with tickets"; timeout -k 10 300 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --opt ilu_chain_tickets=1 --out $O/thermal_256_tickets.json > $O/t256b.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/thermal_256_tickets.json'));print(j['step_s'],j['linear_iterations'],j['kernels']['pc_apply'])"
echo "== thermal 256, per-level graph"; timeout -k 10 300 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --opt ilu_chain=0 --out $O/thermal_256_graph.json > $O/t256c.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/thermal_256_graph.json'));print(j['step_s'],j['linear_iterations'],j['kernels']['pc_apply'])"
echo "== thermal 512 sweep, chain in dispatch order"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol-sweep 1e-5,1e-8 --out $O/thermal_512_sweep.json > $O/thermal_512.log 2>&1; echo "rc=$?"; grep rtol $O/thermal_512.log; python3 -c "import json;j=json.load(open('$O/thermal_512_sweep.json'));print(j['step_s'],j['linear_iterations'],j['kernels']['pc_apply'])"
echo "== kbench"; timeout -k 10 600 python3 tools/kbench.py --reps 20 --out $O/kbench.json > $O/kbench.log 2>&1; echo "rc=$?"; cat $O/kbench.log
echo "== bench"; timeout -k 10 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cat $O/bench.json | cut -c1-1500; tail -3 $O/bench.err
echo "== th bench"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 > $O/th_bench.log 2>&1; echo "rc=$?"; tail -2 $O/th_bench.log | cut -c1-1200
echo "== vsfm3d bench"; timeout -k 10 600 python3 tools/vsfm3d_bench.py --nx 256 --ny 256 --nz 64 > $O/vsfm3d.log 2>&1; echo "rc=$?"; tail -2 $O/vsfm3d.log | cut -c1-800
ls $O
