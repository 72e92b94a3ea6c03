#!/bin/bash
# round 2, GPU run 22: coupled TH with the Krylov dot products inside the block column solve: TH tests, config-5-size bench on / off
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run22; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest th + fused + parity"; timeout -k 10 900 python3 -m pytest tests/test_gpu_th.py tests/test_gpu_fused.py tests/test_gpu_parity.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
for f in 1 0; do echo "== th 512x512x80 th_fused_dots=$f"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --opt th_fused_dots=$f --out $O/th_fused$f.json > $O/th_$f.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/th_fused$f.json'));print(j['newton_its_per_s'],j['newton_iterations'],j['linear_iterations'],{k:(v['launches'],round(v['avg_ms'],3)) for k,v in j['kernels'].items()})"; done
